"""A/B of attention builds / env switches on one box: interleaved rounds of scripts/prof_attn2.py in child processes.
usage: ab_attn.py ROUNDS "Tq:Tk Tq:Tk" name:ENV=VAL,ENV=VAL ...   (LTXK_LIB=path selects a build)"""
import os, subprocess, sys, re, collections
rounds, shapes, variants = int(sys.argv[1]), sys.argv[2].split(), sys.argv[3:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = collections.defaultdict(list)
for r in range(rounds):
    for v in variants:
        name, _, envs = v.partition(":")
        env = dict(os.environ)
        for e in filter(None, envs.split(",")):
            k, _, val = e.partition("=")
            env[k] = val
        out = subprocess.run([sys.executable, os.path.join(root, "scripts", "prof_attn2.py"), "20"] + shapes, env=env,
                             capture_output=True, text=True).stdout
        for m in re.finditer(r"Tq=(\d+) Tk=(\d+):\s+([\d.]+) us", out):
            res[(m.group(1), m.group(2), name)].append(float(m.group(3)))
for (tq, tk, name), v in sorted(res.items()):
    v = sorted(v)
    fl = 4.0 * 2 * 32 * int(tq) * int(tk) * 128
    print(f"Tq={tq} Tk={tk} {name:10s} median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f} us  {fl / v[len(v)//2] / 1e6:7.1f} TF/s", flush=True)

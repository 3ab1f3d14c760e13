"""A/B of GEMM builds on one box: interleaved rounds of scripts/prof_gemm3.py in child processes.
usage: ab_gemm.py ROUNDS "M:N:K:epi ..." name:ENV=VAL,... (LTXK_LIB=path selects a build)"""
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
        out = subprocess.run([sys.executable, os.path.join(root, "scripts", "prof_gemm3.py"), "20"] + shapes, env=env,
                             capture_output=True, text=True)
        for m in re.finditer(r"(\S+): +([\d.]+) us", out.stdout):
            res[(m.group(1), name)].append(float(m.group(2)))
        if out.returncode:
            print(out.stderr[-2000:])
for (shape, name), v in sorted(res.items()):
    v = sorted(v)
    M, N, K = map(int, shape.split(":")[:3])
    print(f"{shape:24s} {name:8s} median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f} us  {2.0*M*N*K / v[len(v)//2] / 1e6:7.1f} TF/s", flush=True)

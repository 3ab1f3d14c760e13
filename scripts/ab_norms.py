"""Interleaved A/B of library builds on scripts/prof_norms.py: ab_norms.py ROUNDS name=lib.so ..."""
import os, subprocess, sys, re, collections
rounds = int(sys.argv[1]); root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = collections.defaultdict(list)
for r in range(rounds):
    for spec in sys.argv[2:]:
        name, _, path = spec.partition("=")
        env = dict(os.environ)
        if path: env["LTXK_LIB"] = os.path.abspath(path)
        out = subprocess.run([sys.executable, os.path.join(root, "scripts", "prof_norms.py"), "100"], env=env, capture_output=True, text=True)
        for m in re.finditer(r"(\S+): +([\d.]+) us", out.stdout): res[(m.group(1), name)].append(float(m.group(2)))
        if out.returncode: print(out.stderr[-800:])
for (k, name), v in sorted(res.items()):
    v = sorted(v); print(f"{k:24s} {name:8s} median {v[len(v)//2]:7.2f} us  min {v[0]:7.2f} us", flush=True)

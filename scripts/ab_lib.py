"""A/B of two library builds on the whole DiT forward (scripts/ab_step.py, LTXK_FUSE=15), interleaved child processes.
usage: ab_lib.py ROUNDS LAYERS name=path/to/lib.so ..."""
import json, os, subprocess, sys, collections
rounds, layers = int(sys.argv[1]), sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = collections.defaultdict(list)
for r in range(rounds):
    for spec in sys.argv[3:]:
        name, _, path = spec.partition("=")
        env = dict(os.environ, AB_VARIANTS=os.environ.get("AB_VARIANTS", "15"))
        if path:
            env["LTXK_LIB"] = os.path.abspath(path)
        out = subprocess.run([sys.executable, os.path.join(root, "scripts", "ab_step.py"), layers, "5"], env=env, capture_output=True, text=True)
        for line in out.stdout.splitlines():
            if line.startswith("{"):
                res[name].append(json.loads(line)["ms_per_layer"])
        if out.returncode:
            print(out.stderr[-1500:])
for name, v in res.items():
    v = sorted(v)
    print(f"{name:10s} ms/layer median {v[len(v)//2]:.4f}  min {v[0]:.4f}  (n={len(v)})", flush=True)

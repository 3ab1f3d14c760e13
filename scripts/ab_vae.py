"""A/B of library builds on the VAE decode (33x512x512 from a 5x16x16 latent), interleaved child processes.
usage: ab_vae.py ROUNDS name=lib.so[,ENV=VAL...] ...   (empty path = the default library)"""
import json, os, subprocess, sys, collections
rounds = int(sys.argv[1])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = "import sys,json,torch; sys.path.insert(0, %r); from mlx_video_amd import video_vae; print(json.dumps(video_vae.bench_decode(torch.device('cuda:0'), 5, 16, 16, iters=5)))" % root
res = collections.defaultdict(list)
for r in range(rounds):
    for spec in sys.argv[2:]:
        name, _, rest = spec.partition("=")
        path, *envs = rest.split(",")
        env = dict(os.environ)
        if path:
            env["LTXK_LIB"] = os.path.abspath(path)
        for e in envs:
            k, _, v = e.partition("=")
            env[k] = v
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        for line in out.stdout.splitlines():
            if line.startswith("{"):
                d = json.loads(line)
                res[name].append((d["vae_decode_ms"], d["vae_kernel_breakdown_ms"].get("conv3d_k3"), d["vae_kernel_breakdown_ms"].get("pixelnorm_act")))
        if out.returncode:
            print(out.stderr[-1500:])
for name, v in res.items():
    ms = sorted(x[0] for x in v); cv = sorted(x[1] for x in v)
    pn = sorted(x[2] or 0.0 for x in v)
    print(f"{name:8s} decode ms median {ms[len(ms)//2]:.3f} min {ms[0]:.3f}   conv3d ms median {cv[len(cv)//2]:.3f}   pixelnorm ms median {pn[len(pn)//2]:.3f}", flush=True)

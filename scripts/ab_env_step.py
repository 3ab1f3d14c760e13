"""A/B of library switches that are read per call (e.g. LTXK_GEMM_BIG) on the whole DiT forward, ONE device, ONE process:
one captured graph per variant, replayed in interleaved rounds.
  python scripts/ab_env_step.py LAYERS ROUNDS name:ENV=VAL,ENV=VAL name2:...      (AB_N = tokens per sample, default 1280)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the launch-form switches exist only in the -DLTXK_AB build (csrc/common.h, `make ab`)
os.environ.setdefault("LTXK_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mlx-video_amd", "libltxk_ab.so"))
import torch
from mlx_video_amd.ltx_model import LTXModel, LTXModelConfig, TimestepPlan, precompute_freqs_cis
from mlx_video_amd.schedulers import create_position_grid

L, R = int(sys.argv[1]), int(sys.argv[2])
variants = []
for v in sys.argv[3:]:
    name, _, envs = v.partition(":")
    variants.append((name, dict(e.partition("=")[::2] for e in envs.split(",") if e)))
dev = torch.device("cuda:0")
model = LTXModel.random_init(LTXModelConfig(num_layers=L), dev, seed=1234)
N = int(os.environ.get("AB_N", "1280"))
g = torch.Generator(device=dev).manual_seed(1)
lat = torch.randn((2, N, 128), generator=g, device=dev).to(torch.bfloat16)
ctx = torch.randn((2, 1024, 3840), generator=g, device=dev).to(torch.bfloat16)
pos = create_position_grid(1, N // 256, 16, 16).to(dev)
pe = precompute_freqs_cis(pos, 4096, 10000.0, (20, 2048, 2048), 32)
plan = TimestepPlan(torch.tensor([0.7], device=dev).to(torch.bfloat16), torch.zeros(2 * N, dtype=torch.int32, device=dev))
graphs, outs = {}, {}
for name, env in variants:
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    model.fuse = int(env.get("LTXK_FUSE", os.environ.get("LTXK_FUSE_DEFAULT", "15")))       # a model attribute, spelled as a variable here
    model.attn_tail_split = env.get("LTXK_FA_SPLIT", "1") != "0"
    for _ in range(2):
        o = model.forward_tokens(lat, plan, ctx, pe)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        o = model.forward_tokens(lat, plan, ctx, pe)
    graphs[name], outs[name] = gr, o
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
times = {name: [] for name, _ in variants}
for r in range(R):
    for name, _ in variants:
        graphs[name].replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            graphs[name].replay()
        torch.cuda.synchronize()
        times[name].append((time.perf_counter() - t0) / 4)
base = variants[0][0]
for name, _ in variants:
    t = sorted(times[name])
    same = torch.equal(outs[name], outs[base])
    print(f"{name:12s} median {t[len(t)//2]*1e3:8.3f} ms/forward ({L} blocks)  min {t[0]*1e3:8.3f}  per block {t[len(t)//2]*1e6/L:7.1f} us  same bits as {base}: {same}", flush=True)

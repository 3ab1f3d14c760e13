"""A/B of the DiT forward's launch structure on ONE device in ONE process (interleaved rounds, graph replay):
LTXK_FUSE bit 1 = fused q|k|v + text k|v launches, 2 = row statistics from GEMM epilogues, 4 = q prep inside attention.
  python scripts/ab_step.py [layers] [rounds]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd.ltx_model import LTXModel, LTXModelConfig, TimestepPlan, precompute_freqs_cis
from mlx_video_amd.schedulers import create_position_grid

L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
VARIANTS = [int(v) for v in os.environ.get("AB_VARIANTS", "0,1,3,7").split(",")]
dev = torch.device("cuda:0")
model = LTXModel.random_init(LTXModelConfig(num_layers=L), dev, seed=1234)
N = int(os.environ.get("AB_N", "1280"))
Fl = N // 256
g = torch.Generator(device=dev).manual_seed(1)
lat = torch.randn((2, N, 128), generator=g, device=dev).to(torch.bfloat16)
ctx = torch.randn((2, 1024, 3840), generator=g, device=dev).to(torch.bfloat16)
pos = create_position_grid(1, Fl, 16, 16).to(dev)
pe = precompute_freqs_cis(pos, 4096, 10000.0, (20, 2048, 2048), 32)
plan = TimestepPlan(torch.tensor([0.7], device=dev).to(torch.bfloat16), torch.zeros(2 * N, dtype=torch.int32, device=dev))
graphs, outs = {}, {}
for v in VARIANTS:
    model.fuse = v
    for _ in range(2):
        o = model.forward_tokens(lat, plan, ctx, pe)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        o = model.forward_tokens(lat, plan, ctx, pe)
    graphs[v], outs[v] = gr, o
ref = None
times = {v: [] for v in VARIANTS}
for r in range(R):
    for v in VARIANTS:
        graphs[v].replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            graphs[v].replay()
        torch.cuda.synchronize()
        times[v].append((time.perf_counter() - t0) / 4 * 1e3)
if os.environ.get("AB_EAGER") == "1":
    for v in VARIANTS:
        model.fuse = v
        ts = []
        for r in range(R):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                model.forward_tokens(lat, plan, ctx, pe)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 4 * 1e3)
        ts.sort()
        print(json.dumps({"fuse": v, "eager_ms_forward_median": ts[len(ts) // 2], "eager_ms_min": ts[0]}), flush=True)
base = outs[VARIANTS[0]].float()
for v in VARIANTS:
    t = sorted(times[v])
    d = float((outs[v].float() - base).norm() / base.norm())
    print(json.dumps({"fuse": v, "layers": L, "ms_forward_median": t[len(t) // 2], "ms_min": t[0], "ms_per_layer": t[len(t) // 2] / L,
                      "rel_diff_vs_first": d}), flush=True)

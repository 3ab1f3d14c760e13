import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from mlx_video_amd import ops
from mlx_video_amd.ltx_model import LTXModel, LTXModelConfig, TimestepPlan, precompute_freqs_cis
from mlx_video_amd.schedulers import create_position_grid
dev = torch.device("cuda:0")
model = LTXModel.random_init(LTXModelConfig(num_layers=48), dev, seed=1234)
import mlx_video_amd.ltx_model as lm
for (B, F, H, W) in ((2, 9, 12, 12), (2, 13, 16, 16), (1, 9, 24, 24), (2, 5, 16, 16), (1, 5, 24, 16)):
    N = F * H * W
    g = torch.Generator(device=dev).manual_seed(7)
    lat = torch.randn((B, N, 128), generator=g, device=dev).to(torch.bfloat16)
    ctx = torch.randn((B, 1024, 3840), generator=g, device=dev).to(torch.bfloat16)
    pos = create_position_grid(1, F, H, W).to(dev)
    pe = precompute_freqs_cis(pos, 4096, 10000.0, (20, 2048, 2048), 32)
    plan = TimestepPlan(torch.tensor([0.7], device=dev).to(torch.bfloat16), torch.zeros(B * N, dtype=torch.int32, device=dev))
    res = {}
    graphs = {}
    for name, thr in (("split_qk_v", 640), ("one_qkv", 100000)):
        ops.SPLITK_MAX_M = thr      # (ops.gemm offers the workspace up to this M too, harmless: the library decides by its own M limit)
        for _ in range(2): model.forward_tokens(lat, plan, ctx, pe)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr): model.forward_tokens(lat, plan, ctx, pe)
        graphs[name] = gr
    for r in range(5):
        for name, gr in graphs.items():
            gr.replay(); torch.cuda.synchronize()
            t0 = time.perf_counter(); gr.replay(); torch.cuda.synchronize()
            res.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
    print(f"B={B} N={N}: " + "  ".join(f"{n} {sorted(v)[len(v)//2]:.2f} ms" for n, v in res.items()), flush=True)

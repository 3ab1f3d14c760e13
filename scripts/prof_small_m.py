"""DiT forward at small token counts (graph replay, L=48): how far the small-M regime is from the weight-streaming floor
(26 GB of bf16 weights per forward / ~5 TB/s = 5.2 ms)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd.ltx_model import LTXModel, LTXModelConfig, TimestepPlan, precompute_freqs_cis
from mlx_video_amd.schedulers import create_position_grid
dev = torch.device("cuda:0")
model = LTXModel.random_init(LTXModelConfig(num_layers=48), dev, seed=1234)
for (B, F, H, W) in ((2, 2, 4, 4), (1, 5, 8, 8), (2, 5, 8, 8), (1, 9, 12, 12), (1, 5, 16, 16)):
    N = F * H * W
    g = torch.Generator(device=dev).manual_seed(7)
    lat = torch.randn((B, N, 128), generator=g, device=dev).to(torch.bfloat16)
    ctx = torch.randn((B, 1024, 3840), generator=g, device=dev).to(torch.bfloat16)
    pos = create_position_grid(1, F, H, W).to(dev)
    pe = precompute_freqs_cis(pos, 4096, 10000.0, (20, 2048, 2048), 32)
    plan = TimestepPlan(torch.tensor([0.7], device=dev).to(torch.bfloat16), torch.zeros(B * N, dtype=torch.int32, device=dev))
    for _ in range(2): model.forward_tokens(lat, plan, ctx, pe)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr): model.forward_tokens(lat, plan, ctx, pe)
    gr.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); gr.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"B={B} N={N:5d} (M={B*N:5d}): forward {sorted(ts)[1]:7.2f} ms", flush=True)
    del gr

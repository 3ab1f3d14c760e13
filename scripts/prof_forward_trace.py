"""A few eager DiT forwards at one shape, for rocprofv3 --kernel-trace --stats (per-kernel time at that shape):
  python3 scripts/prof_forward_trace.py B F H W [layers] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd.ltx_model import LTXModel, LTXModelConfig, TimestepPlan, precompute_freqs_cis
from mlx_video_amd.schedulers import create_position_grid
B, F, H, W = (int(v) for v in sys.argv[1:5])
L = int(sys.argv[5]) if len(sys.argv) > 5 else 8
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 4
dev = torch.device("cuda:0")
model = LTXModel.random_init(LTXModelConfig(num_layers=L), dev, seed=1234)
N = F * H * W
g = torch.Generator(device=dev).manual_seed(7)
lat = torch.randn((B, N, 128), generator=g, device=dev).to(torch.bfloat16)
ctx = torch.randn((B, 1024, 3840), generator=g, device=dev).to(torch.bfloat16)
pos = create_position_grid(1, F, H, W).to(dev)
pe = precompute_freqs_cis(pos, 4096, 10000.0, (20, 2048, 2048), 32)
plan = TimestepPlan(torch.tensor([0.7], device=dev).to(torch.bfloat16), torch.zeros(B * N, dtype=torch.int32, device=dev))
for _ in range(iters):
    model.forward_tokens(lat, plan, ctx, pe)
torch.cuda.synchronize()

"""One attention shape on random data, few launches (for rocprofv3 --pmc / --kernel-trace): argv = Tq Tk iters."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import ops
dev = torch.device("cuda:0")
Tq, Tk, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B, H, D = 2, 32, 4096
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn((B * Tq, D), generator=g, device=dev).to(torch.bfloat16)
k = torch.randn((B * Tk, D), generator=g, device=dev).to(torch.bfloat16)
vt = torch.randn((B, D, (Tk + 63) // 64 * 64), generator=g, device=dev).to(torch.bfloat16)
out = torch.empty((B * Tq, D), dtype=torch.bfloat16, device=dev)
for _ in range(iters):
    ops.flash_attn(q, k, vt, out, B, H, Tq, Tk, 1 / math.sqrt(128))
torch.cuda.synchronize()

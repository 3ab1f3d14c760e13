"""One attention shape, few launches (for rocprofv3 --pmc): argv = Tq Tk iters."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import ops
dev = torch.device("cuda:0")
Tq, Tk, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B, H, D = 2, 32, 4096
q = torch.ones((B * Tq, D), device=dev, dtype=torch.bfloat16) * 0.1
k = torch.ones((B * Tk, D), device=dev, dtype=torch.bfloat16) * 0.1
vt = torch.ones((B, D, (Tk + 63) // 64 * 64), device=dev, dtype=torch.bfloat16)
out = torch.empty((B * Tq, D), dtype=torch.bfloat16, device=dev)
q += torch.arange(B * Tq, device=dev, dtype=torch.float32)[:, None].to(torch.bfloat16) * 1e-3
for _ in range(iters):
    ops.flash_attn(q, k, vt, out, B, H, Tq, Tk, 1 / math.sqrt(128))
torch.cuda.synchronize()

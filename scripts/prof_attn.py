"""Micro-driver: self / cross attention shapes of the bench (B=2, H=32, N=1280, S=1024) for each kernel variant."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import ops
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator(device=dev).manual_seed(0)
B, H, D = 2, 32, 4096
for name, Tq, Tk in (("self", 1280, 1280), ("cross", 1280, 1024), ("self5184", 5184, 5184)):
    q = torch.randn((B * Tq, D), generator=g, device=dev).to(torch.bfloat16)
    k = torch.randn((B * Tk, D), generator=g, device=dev).to(torch.bfloat16)
    vt = torch.randn((B, D, (Tk + 63) // 64 * 64), generator=g, device=dev).to(torch.bfloat16)
    out = torch.empty((B * Tq, D), dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        ops.flash_attn(q, k, vt, out, B, H, Tq, Tk, 1 / math.sqrt(128))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        ops.flash_attn(q, k, vt, out, B, H, Tq, Tk, 1 / math.sqrt(128))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"{name:9s} Tq={Tq} Tk={Tk}: {dt*1e6:8.1f} us  {4.0*B*H*Tq*Tk*128/dt/1e12:7.1f} TFLOP/s", flush=True)

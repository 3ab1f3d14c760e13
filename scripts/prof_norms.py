"""The norm kernels of one DiT block at the bench shape (M=2560, D=4096, H=32, T=1280) via graph replay, inputs rotated so
rows are not simply L2-resident: rmsnorm with carried row statistics (modulated / plain) and k's q_norm + RoPE."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import ops
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
M, D, H, T = 2560, 4096, 32, 1280
g0 = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn((M, D), generator=g0, device=dev).to(torch.bfloat16) for _ in range(8)]
ss = [(x.float() ** 2).reshape(M, D // 64, 64).sum(-1).contiguous() for x in xs]
ada = torch.randn((1, 6, D), generator=g0, device=dev).to(torch.bfloat16)
w = torch.randn((D,), generator=g0, device=dev).to(torch.bfloat16)
cos = torch.randn((H, T, 64), generator=g0, device=dev)
sin = torch.randn((H, T, 64), generator=g0, device=dev)
y = torch.empty_like(xs[0])
def bench(name, fn, nbytes):
    for i in range(8): fn(i)
    torch.cuda.synchronize()
    st = torch.cuda.Stream(); gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st):
            for i in range(iters): fn(i)
    gr.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); gr.replay(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"{name}: {dt*1e6:8.2f} us  {nbytes/dt/1e9:7.0f} GB/s", flush=True)
bench("rmsnorm_ss_modulated", lambda i: ops.rmsnorm_modulate(xs[i % 8], 1e-6, ada[:, 1], ada[:, 0], 6 * D, None, out=y, sumsq=ss[i % 8], scale_is_one_plus=True), 4.0 * M * D)
bench("rmsnorm_ss_plain", lambda i: ops.rmsnorm_modulate(xs[i % 8], 1e-6, out=y, sumsq=ss[i % 8]), 4.0 * M * D)
bench("qknorm_rope_ss_k", lambda i: ops.qknorm_rope(xs[i % 8], 1, D, w, cos, sin, T, H, 1e-6, sumsq=ss[i % 8]), 4.0 * M * D + 8.0 * M * D // 2)

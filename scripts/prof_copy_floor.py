"""What a 21 MB -> 21 MB pass costs on this chip at all: torch's bf16 copy and a bf16 multiply by a scalar (one read, one
write, no reduction, no tables) against ltxk's rmsnorm_modulate with carried row statistics, same shape, graph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import ops
dev = torch.device("cuda:0")
iters = 200
M, D = 2560, 4096
xs = [torch.randn((M, D), device=dev).to(torch.bfloat16) for _ in range(8)]
y = torch.empty_like(xs[0])
ada = torch.randn((2, 6, D), device=dev).to(torch.bfloat16)
row = (torch.arange(M, device=dev) % 2).to(torch.int32)
ss = torch.rand((M, D // 64), device=dev) * 64


def timeit(name, fn):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for i in range(iters):
                fn(i)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"{name:44s} {dt*1e6:6.1f} us  {2*M*D*2/dt/1e9:6.0f} GB/s", flush=True)


timeit("torch copy_ (bf16)", lambda i: y.copy_(xs[i % 8]))
timeit("torch mul scalar (bf16)", lambda i: torch.mul(xs[i % 8], 1.5, out=y))
timeit("ltxk rmsnorm_modulate (self-reducing)", lambda i: ops.rmsnorm_modulate(xs[i % 8], 1e-6, ada[:, 1], ada[:, 0], 6 * D, row, out=y))
timeit("ltxk rmsnorm_modulate (carried statistics)", lambda i: ops.rmsnorm_modulate(xs[i % 8], 1e-6, ada[:, 1], ada[:, 0], 6 * D, row, out=y, sumsq=ss, scale_is_one_plus=True))
timeit("ltxk rmsnorm_modulate (carried, one mod row)", lambda i: ops.rmsnorm_modulate(xs[i % 8], 1e-6, ada[:, 1], ada[:, 0], 6 * D, None, out=y, sumsq=ss, scale_is_one_plus=True))
timeit("ltxk rmsnorm (no modulation, carried)", lambda i: ops.rmsnorm_modulate(xs[i % 8], 1e-6, out=y, sumsq=ss))

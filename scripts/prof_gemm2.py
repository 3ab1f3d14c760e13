"""GEMM timing at chosen M:N:K triples via graph replay: argv = iters M:N:K ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import ops
dev = torch.device("cuda:0")
iters = int(sys.argv[1])
for spec in sys.argv[2:]:
    M, N, K = map(int, spec.split(":"))
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randn((M, K), generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device=dev) * 0.02).to(torch.bfloat16)
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, w, b, out=out)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st):
            for _ in range(iters):
                ops.gemm(a, w, b, out=out)
    gr.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    gr.replay(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"M={M} N={N} K={K}: {dt*1e6:8.1f} us  {2.0*M*N*K/dt/1e12:7.1f} TFLOP/s", flush=True)

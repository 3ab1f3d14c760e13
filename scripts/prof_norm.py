"""Micro-driver: rmsnorm_modulate at the bench shape (M=2560, D=4096, 2 distinct timestep rows)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import ops
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
M, D = 2560, 4096
x = torch.randn((M, D), device=dev).to(torch.bfloat16)
ada = torch.randn((2, 6, D), device=dev).to(torch.bfloat16)
row = (torch.arange(M, device=dev) % 2).to(torch.int32)
y = torch.empty_like(x)
xs = [x.clone() for _ in range(8)]    # rotate inputs so the row is not simply L2-resident
for i in range(10):
    ops.rmsnorm_modulate(xs[i % 8], 1e-6, ada[:, 1], ada[:, 0], 6 * D, row, out=y)
torch.cuda.synchronize()
st = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(st):
    with torch.cuda.graph(g, stream=st):                 # host launch rate out of the measurement
        for i in range(iters):
            ops.rmsnorm_modulate(xs[i % 8], 1e-6, ada[:, 1], ada[:, 0], 6 * D, row, out=y)
g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
g.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"rmsnorm_modulate M={M} D={D}: {dt*1e6:.1f} us  {2*M*D*2/dt/1e9:.0f} GB/s", flush=True)

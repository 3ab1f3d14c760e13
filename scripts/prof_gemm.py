"""Micro-driver for rocprofv3: runs the DiT's GEMM shapes (M=2560) through ltxk_gemm_bf16.
usage: python scripts/prof_gemm.py [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import ops

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shapes = {"ff1_gelu": (2560, 16384, 4096, ops.EPI_BIAS_GELU), "ff2_gate": (2560, 4096, 16384, ops.EPI_BIAS_GATE_RES),
          "qk": (2560, 8192, 4096, ops.EPI_BIAS), "out_gate": (2560, 4096, 4096, ops.EPI_BIAS_GATE_RES),
          "ctx_k": (2048, 4096, 4096, ops.EPI_BIAS)}
g = torch.Generator(device=dev).manual_seed(0)
for name, (M, N, K, epi) in shapes.items():
    a = torch.randn((M, K), generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device=dev) * 0.02).to(torch.bfloat16)
    b = (torch.randn((N,), generator=g, device=dev) * 0.01).to(torch.bfloat16)
    res = torch.randn((M, N), generator=g, device=dev).to(torch.bfloat16)
    gate = torch.randn((1, N), generator=g, device=dev).to(torch.bfloat16)
    rows = torch.zeros((M,), dtype=torch.int32, device=dev)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    kw = dict(epilogue=epi, out=out)
    if epi == ops.EPI_BIAS_GATE_RES:
        kw.update(resid=res, gate=gate, gate_row=rows, gate_stride=N)
    for _ in range(3):
        ops.gemm(a, w, b, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        ops.gemm(a, w, b, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"{name:10s} M={M} N={N} K={K}: {dt*1e6:8.1f} us  {2.0*M*N*K/dt/1e12:7.1f} TFLOP/s", flush=True)

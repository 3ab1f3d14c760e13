"""Micro-driver for rocprofv3: the DiT's GEMM launches (B=2 CFG pair, N=1280: M=2560; text M=2048) through ltxk_gemm_bf16.
usage: python scripts/prof_gemm.py [iters] [shape-name ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import ops

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = set(sys.argv[2:])
D = 4096
SHAPES = {"ff1_gelu": (2560, 4 * D, D, ops.EPI_BIAS_GELU, 0), "ff2_gate": (2560, D, 4 * D, ops.EPI_BIAS_GATE_RES, 0),
          "qk_sumsq": (2560, 2 * D, D, ops.EPI_BIAS, -1), "v_transposed": (2560, D, D, ops.EPI_BIAS, -2),
          "qkv_split": (2560, 3 * D, D, ops.EPI_BIAS, 2 * D), "out_gate": (2560, D, D, ops.EPI_BIAS_GATE_RES, 0),
          "o2_res": (2560, D, D, ops.EPI_BIAS_RES, 0), "q2": (2560, D, D, ops.EPI_BIAS, 0), "ctx_kv_split": (2048, 2 * D, D, ops.EPI_BIAS, D)}
g = torch.Generator(device=dev).manual_seed(0)
for name, (M, N, K, epi, split) in SHAPES.items():
    if only and name not in only:
        continue
    a = torch.randn((M, K), generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device=dev) * 0.02).to(torch.bfloat16)
    b = (torch.randn((N,), generator=g, device=dev) * 0.01).to(torch.bfloat16)
    kw = dict(epilogue=epi)
    if split == -1:            # q|k: row-major with row statistics (the 320x256-tile kernel at M=2560)
        kw.update(out=torch.empty((M, N), dtype=torch.bfloat16, device=dev), sumsq=torch.empty((M, N // 64), dtype=torch.float32, device=dev))
    elif split == -2:          # v: transposed per batch
        kw.update(out=torch.empty((2, N, M // 2), dtype=torch.bfloat16, device=dev), out_tokens_per_batch=M // 2)
    elif split:
        T = M // 2
        kw.update(out=torch.empty((M, split), dtype=torch.bfloat16, device=dev), out2=torch.empty((2, N - split, T), dtype=torch.bfloat16, device=dev),
                  n_split=split, out_tokens_per_batch=T, sumsq=torch.empty((M, split // 64), dtype=torch.float32, device=dev))
    else:
        kw.update(out=torch.empty((M, N), dtype=torch.bfloat16, device=dev))
        if epi in (ops.EPI_BIAS_GATE_RES, ops.EPI_BIAS_RES):
            kw.update(resid=torch.randn((M, N), generator=g, device=dev).to(torch.bfloat16), sumsq=torch.empty((M, N // 64), dtype=torch.float32, device=dev))
        if epi == ops.EPI_BIAS_GATE_RES:
            kw.update(gate=torch.randn((1, N), generator=g, device=dev).to(torch.bfloat16), gate_row=torch.zeros((M,), dtype=torch.int32, device=dev), gate_stride=N)
    for _ in range(3):
        ops.gemm(a, w, b, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        ops.gemm(a, w, b, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"{name:12s} M={M} N={N} K={K}: {dt*1e6:8.1f} us  {2.0*M*N*K/dt/1e12:7.1f} TFLOP/s", flush=True)

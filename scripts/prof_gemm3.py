"""GEMM timing with epilogues via graph replay: argv = iters M:N:K:epi[:split] ...  (epi 0 bias, 1 gelu, 3 gate+res, 4 res;
split=1: q|k|v-style split output with the last third transposed, sumsq on)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import ops
dev = torch.device("cuda:0")
iters = int(sys.argv[1])
for spec in sys.argv[2:]:
    f = spec.split(":")
    M, N, K = map(int, f[:3]); epi = int(f[3]) if len(f) > 3 else 0; split = len(f) > 4 and f[4] in ("1", "2"); half = len(f) > 4 and f[4] == "2"    # 2: k | V^T halves (text K/V)
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randn((M, K), generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device=dev) * 0.02).to(torch.bfloat16)
    b = (torch.randn(N, generator=g, device=dev) * 0.01).to(torch.bfloat16)
    if os.environ.get("GEMM_ZERO_DATA") == "1":          # DVFS probe: all-zero operands draw less power per MFMA
        a.zero_(); w.zero_()
    kw = dict(epilogue=epi)
    if split:
        T = M // 2
        ns = N // 2 if half else N * 2 // 3
        kw.update(out=torch.empty((M, ns), device=dev, dtype=torch.bfloat16), out2=torch.empty((2, N - ns, T), device=dev, dtype=torch.bfloat16),
                  n_split=ns, out_tokens_per_batch=T, sumsq=torch.empty((M, ns // 64), device=dev, dtype=torch.float32))
    else:
        kw.update(out=torch.empty((M, N), device=dev, dtype=torch.bfloat16))
        if epi in (3, 4):
            kw.update(resid=torch.randn((M, N), generator=g, device=dev).to(torch.bfloat16), sumsq=torch.empty((M, N // 64), device=dev, dtype=torch.float32))
        if epi == 3:
            kw.update(gate=torch.randn((1, N), generator=g, device=dev).to(torch.bfloat16), gate_row=torch.zeros(M, dtype=torch.int32, device=dev), gate_stride=N)
    for _ in range(3):
        ops.gemm(a, w, b, **kw)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st):
            for _ in range(iters):
                ops.gemm(a, w, b, **kw)
    gr.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    gr.replay(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"{spec}: {dt*1e6:8.1f} us  {2.0*M*N*K/dt/1e12:7.1f} TFLOP/s", flush=True)

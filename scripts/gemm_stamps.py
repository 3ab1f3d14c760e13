"""Per-workgroup phase stamps of the GEMM kernel (diagnostic build libltxk_diag.so: make -C mlx-video_amd/csrc diag).
LTXK_LIB=mlx-video_amd/libltxk_diag.so python scripts/gemm_stamps.py M:N:K[:epi] ..."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
lib.ltxk_diag_set_gemm_stamps.argtypes = [ctypes.c_void_p]
for spec in sys.argv[1:]:
    f = spec.split(":")
    M, N, K = map(int, f[:3]); epi = int(f[3]) if len(f) > 3 else 0
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randn((M, K), generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device=dev) * 0.02).to(torch.bfloat16)
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    res = torch.randn((M, N), generator=g, device=dev).to(torch.bfloat16) if epi in (3, 4) else None
    gate = torch.randn((1, N), generator=g, device=dev).to(torch.bfloat16) if epi == 3 else None
    nwg = ((M + 159) // 160) * ((N + 255) // 256)
    st = torch.zeros((nwg, 8), dtype=torch.int64, device=dev)
    kw = dict(epilogue=epi, out=out, resid=res, gate=gate, gate_stride=0)
    for _ in range(5):
        ops.gemm(a, w, b, **kw)
    torch.cuda.synchronize()
    lib.ltxk_diag_set_gemm_stamps(st.data_ptr())
    ops.gemm(a, w, b, **kw)
    torch.cuda.synchronize()
    lib.ltxk_diag_set_gemm_stamps(None)
    s = st.cpu().numpy().astype("float64")
    t0 = s[:, 4].min()
    tick = 0.01   # us per s_memrealtime tick (100 MHz)
    import numpy as np
    start, first, k8, loop_end, done, setup, bar0 = [(s[:, i] - t0) * tick for i in (4, 1, 2, 3, 5, 0, 7)]
    order = np.argsort(start)
    print(f"M={M} N={N} K={K} epi={epi}: {nwg} workgroups, kernel span {done.max():.1f} us")
    rounds = (nwg + 255) // 256
    for r in range(rounds):
        idx = order[r * 256:(r + 1) * 256]
        print(f"  round {r}: start {np.median(start[idx]):7.1f} (spread {start[idx].max()-start[idx].min():5.1f})  setup {np.median(setup[idx]-start[idx]):5.2f} to-barrier0 {np.median(bar0[idx]-setup[idx]):5.2f} fill {np.median(first[idx]-start[idx]):5.2f}"
              f"  8 K-steps {np.median(k8[idx]-first[idx]):5.2f}  loop {np.median(loop_end[idx]-first[idx]):7.2f}"
              f"  epilogue {np.median(done[idx]-loop_end[idx]):5.2f} (max {np.max(done[idx]-loop_end[idx]):5.2f})  end {np.median(done[idx]):7.1f} (spread {done[idx].max()-done[idx].min():5.1f})")

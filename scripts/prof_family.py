"""One launch shape of one non-GEMM kernel family on random data, a few eager launches (for rocprofv3 --pmc / --kernel-trace;
the GEMM shapes have scripts/prof_gemm.py):  python3 scripts/prof_family.py CASE [iters]
  attn_TqxTk[_b1]   flash_attn16_kernel, B=2 (or 1), H=32
  conv128           the 128-channel 3x3x3 convolution of the decoder's last stage (33x128x128 voxels, reflect pad; kw-reuse kernel)
  conv256           the 256-channel one (17x64x64 voxels)
  norm_mod          rms_norm with carried row statistics + (1+scale), shift       (M=2560, D=4096)
  norm_plain        the same without modulation (the cross-attention pre-norm)
  qknorm_k          k's q_norm + SPLIT RoPE in place, carried statistics
  pixelnorm128      PixelNorm + SiLU over a 33x128x128x128 volume (decoder.py:136-180)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import ops
dev = torch.device("cuda:0")
case = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
g = torch.Generator(device=dev).manual_seed(0)
BF = torch.bfloat16
rnd = lambda *s: torch.randn(s, generator=g, device=dev)
if case.startswith("attn_"):
    parts = case.split("_")
    Tq, Tk = (int(v) for v in parts[1].split("x"))
    B, H, D = (1 if "b1" in parts[2:] else 2), 32, 4096
    q, k = rnd(B * Tq, D).to(BF), rnd(B * Tk, D).to(BF)
    vt = rnd(B, D, (Tk + 63) // 64 * 64).to(BF)
    out = torch.empty((B * Tq, D), dtype=BF, device=dev)
    fn = lambda i: ops.flash_attn(q, k, vt, out, B, H, Tq, Tk, 1 / math.sqrt(128))
elif case in ("conv128", "conv256"):
    from mlx_video_amd import video_vae as V
    C, vol = (128, (33, 128, 128)) if case == "conv128" else (256, (17, 64, 64))
    x = rnd(1, *vol, C).to(BF)
    w = (rnd(C, 3, 3, 3, C) * 0.02).to(BF)
    b = (rnd(C) * 0.01).to(BF)
    fn = lambda i: V.conv3d(x, w, b, False, V.PAD_REFLECT)
elif case == "pixelnorm128":
    from mlx_video_amd import video_vae as V
    xs = [rnd(1, 33, 128, 128, 128).to(BF) for _ in range(4)]
    fn = lambda i: V.pixelnorm_act(xs[i % 4], 1e-8, True)
else:
    M, D, H, T = 2560, 4096, 32, 1280
    xs = [rnd(M, D).to(BF) for _ in range(8)]                      # rotated: 8 x 21 MB, rows are not simply L2-resident
    ss = [(x.float() ** 2).reshape(M, D // 64, 64).sum(-1).contiguous() for x in xs]
    ada = rnd(1, 6, D).to(BF)
    w = rnd(D).to(BF)
    cos, sin = rnd(H, T, 64), rnd(H, T, 64)
    y = torch.empty_like(xs[0])
    if case == "norm_mod":
        fn = lambda i: ops.rmsnorm_modulate(xs[i % 8], 1e-6, ada[:, 1], ada[:, 0], 6 * D, None, out=y, sumsq=ss[i % 8], scale_is_one_plus=True)
    elif case == "norm_plain":
        fn = lambda i: ops.rmsnorm_modulate(xs[i % 8], 1e-6, out=y, sumsq=ss[i % 8])
    elif case == "qknorm_k":
        fn = lambda i: ops.qknorm_rope(xs[i % 8], 1, D, w, cos, sin, T, H, 1e-6, sumsq=ss[i % 8])
    else:
        raise SystemExit(f"unknown case {case}")
for i in range(iters):
    fn(i)
torch.cuda.synchronize()

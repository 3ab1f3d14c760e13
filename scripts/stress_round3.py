"""Determinism stress of the kernels that are new in round 3 (counted-vmcnt LDS-DMA pipelines race silently when a wait is
miscounted): the kw-reuse convolution, attention on 16x16x32 MFMAs (plain, fused query prep, no tail split), the 256x256 GEMM
tile (text k|v shape), the norm kernels - every launch repeated under memory pressure from a concurrent copy stream and
compared bit for bit with its first result.   python scripts/stress_round3.py [seconds]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import ops, video_vae as V
dev = torch.device("cuda:0"); BF = torch.bfloat16
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(s, generator=g, device=dev) * sc).to(BF)
cases = {}
x = rn(1, 9, 128, 128, 128); w = rn(128, 3, 3, 3, 128, sc=0.02); b = rn(128, sc=0.1); r = rn(1, 9, 128, 128, 128)
cases["conv_kw_res"] = lambda: V.conv3d(x, w, b, False, V.PAD_REFLECT, resid=r)
x2 = rn(2, 3, 77, 101, 64); w2 = rn(48, 3, 3, 3, 64, sc=0.03); b2 = rn(48, sc=0.1)
cases["conv_kw_zero_pad_odd"] = lambda: V.conv3d(x2, w2, b2, True, V.PAD_ZEROS)
B, H, T, D = 2, 32, 1280, 4096
q = rn(B * T, D); k = rn(B * T, D); vt = rn(B, D, T); wq = rn(D, sc=0.1) + 1
ss = (q.float() ** 2).reshape(B * T, D // 64, 64).sum(-1).contiguous()
cos = torch.randn((H, T, 64), generator=g, device=dev); sin = torch.randn((H, T, 64), generator=g, device=dev)
def attn(**kw):
    o = torch.empty((B * T, D), dtype=BF, device=dev)
    ops.flash_attn(q, k, vt, o, B, H, T, T, 1 / math.sqrt(128), **kw)
    return o
cases["attn16"] = lambda: attn()
cases["attn16_qprep"] = lambda: attn(q_sumsq=ss, q_norm_weight=wq, cos=cos, sin=sin, eps=1e-6)
cases["attn16_no_split"] = lambda: attn(tail_split=False)
a = rn(2048, 4096); wk = rn(8192, 4096, sc=0.02); bk = rn(8192, sc=0.01)
def kv():
    k2 = torch.empty((2048, 4096), dtype=BF, device=dev); v2 = torch.empty((2, 4096, 1024), dtype=BF, device=dev)
    s2 = torch.empty((2048, 64), dtype=torch.float32, device=dev)
    ops.gemm(a, wk, bk, out=k2, out2=v2, n_split=4096, out_tokens_per_batch=1024, sumsq=s2)
    return torch.cat([k2.reshape(-1).float(), v2.reshape(-1).float(), s2.reshape(-1)])
cases["gemm_256tile_text_kv"] = kv
xm = rn(2560, 4096); ada = rn(1, 6, 4096); xs = (xm.float() ** 2).reshape(2560, 64, 64).sum(-1).contiguous()
cases["rmsnorm_ss_mod"] = lambda: ops.rmsnorm_modulate(xm, 1e-6, ada[:, 1], ada[:, 0], 6 * 4096, None, sumsq=xs, scale_is_one_plus=True)
cases["qknorm_rope_ss"] = lambda: ops.qknorm_rope(xm.clone(), 1, 4096, wq, cos, sin, T, H, 1e-6, sumsq=xs)
ref = {n: f().clone() for n, f in cases.items()}
torch.cuda.synchronize()
noise_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev); noise_b = torch.empty_like(noise_a)
side = torch.cuda.Stream()
t0, it, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    with torch.cuda.stream(side):                      # HBM / fabric contention beside the kernels under test
        noise_b.copy_(noise_a)
    for n, f in cases.items():
        out = f()
        if not torch.equal(out, ref[n]):
            bad += 1
            print(f"MISMATCH {n} at iteration {it}: {int((out != ref[n]).sum())} elements", flush=True)
    it += 1
    if it % 20 == 0:
        print(f"[progress] {it} iterations, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
torch.cuda.synchronize()
print(f"done: {it} iterations x {len(cases)} kernels, {bad} mismatches")
sys.exit(1 if bad else 0)

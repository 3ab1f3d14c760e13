"""Determinism stress of the launch forms that are new in round 4 (counted-vmcnt LDS-DMA pipelines and LDS hand-overs race
silently when a wait or a barrier is misplaced): the 128-column GEMM tile with the row statistic handed from the even to the odd
wave (gate + residual epilogue, M=1280), the split-K weight-streaming form + its epilogue launch (M=64: 6-deep ring; M=320),
one q|k|v launch with split output through split-K, and the mixed 192 / 128-row attention grid (plain, fused query prep) - every
launch repeated under memory pressure from a concurrent copy stream and compared bit for bit with its first result.
  python scripts/stress_round4.py [seconds]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(s, generator=g, device=dev) * sc).to(BF)
cases = {}
D = 4096
w4 = rn(D, D, sc=0.02); b4 = rn(D, sc=0.01); gate = rn(1, D)
def gemm_case(M, K=D, w=None, epi=ops.EPI_BIAS_GATE_RES):
    a = rn(M, K); res = rn(M, D); ww = w4 if w is None else w
    grow = torch.zeros(M, dtype=torch.int32, device=dev)
    def f():
        ss = torch.empty((M, D // 64), dtype=torch.float32, device=dev)
        y = ops.gemm(a, ww, b4, epilogue=epi, resid=res, gate=gate, gate_row=grow, gate_stride=D, sumsq=ss)
        return torch.cat([y.reshape(-1).float(), ss.reshape(-1)])
    return f
cases["gemm_128col_gate_res_sumsq_M1280"] = gemm_case(1280)
cases["gemm_splitk_M64"] = gemm_case(64)
cases["gemm_splitk_M320"] = gemm_case(320)
w16 = rn(D, 4 * D, sc=0.01)
cases["gemm_splitk_ff2_M64"] = gemm_case(64, 4 * D, w16)
wqkv = rn(3 * D, D, sc=0.02); bqkv = rn(3 * D, sc=0.01); a64 = rn(64, D)
def qkv():
    qk = torch.empty((64, 2 * D), dtype=BF, device=dev); vt = torch.zeros((2, D, 64), dtype=BF, device=dev)
    ss = torch.empty((64, 2 * D // 64), dtype=torch.float32, device=dev)
    ops.gemm(a64, wqkv, bqkv, out=qk, out2=vt, n_split=2 * D, out_tokens_per_batch=32, sumsq=ss)
    return torch.cat([qk.reshape(-1).float(), vt.reshape(-1).float(), ss.reshape(-1)])
cases["gemm_splitk_qkv_split_output_M64"] = qkv
B, H, T = 2, 32, 1280
q = rn(B * T, D); k = rn(B * T, D); vt_ = rn(B, D, T); wq = rn(D, sc=0.1) + 1
ssq = (q.float() ** 2).reshape(B * T, D // 64, 64).sum(-1).contiguous()
cos = torch.randn((H, T, 64), generator=g, device=dev); sin = torch.randn((H, T, 64), generator=g, device=dev)
def attn(**kw):
    o = torch.empty((B * T, D), dtype=BF, device=dev)
    ops.flash_attn(q, k, vt_, o, B, H, T, T, 1 / math.sqrt(128), **kw)
    return o
cases["attn_mixed_grid"] = lambda: attn()
cases["attn_mixed_grid_qprep"] = lambda: attn(q_sumsq=ssq, q_norm_weight=wq, cos=cos, sin=sin, eps=1e-6)
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
    if it % 200 == 0:
        print(f"[progress] {it} iterations, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
torch.cuda.synchronize()
print(f"done: {it} iterations x {len(cases)} launches, {bad} mismatches")
sys.exit(1 if bad else 0)

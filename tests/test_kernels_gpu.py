"""Per-kernel parity: each libltxk entry point (through the C ABI) against the CPU oracle
(oracle/dit.py, bf16-storage policy) on the same seeded inputs.

Tolerances: kernels reproduce every bf16 rounding point of the reference's op chain, so the
only differences are fp32 accumulation order (GEMM/row reductions: off-by-one-bf16-ulp on a
small fraction of elements) and, for attention, P rounded to bf16 before P.V.  bf16 ulp =
2^-8 relative; bounds are written per test."""
import math

import numpy as np
import parity
import pytest
import torch

from oracle import dit as O

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _ops():
    from mlx_video_amd import ops
    return ops


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def ulp_diff_frac(a, b, ulps=1, mag=None):
    """fraction of elements further than `ulps` bf16 ulps apart; `mag` = magnitude the ulp is
    taken at (defaults to |b|; pass the operand magnitude when the last op is a cancelling add)."""
    a, b = a.float().cpu(), b.float().cpu()
    m = b.abs() if mag is None else torch.maximum(b.abs(), mag.float().cpu())
    tol = ulps * (2.0 ** -7) * m.clamp_min(1e-6)
    return float(((a - b).abs() > tol).float().mean())


@pytest.mark.parametrize("M,N,K", [(160, 256, 128), (1280, 4096, 4096), (333, 640, 256), (6, 24576, 4096),
                                   (2048, 8192, 3840 // 64 * 64), (40, 128, 4096)])
def test_gemm_bias(dev, M, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g).to(BF)
    w = (torch.randn(N, K, generator=g) * 0.02).to(BF)
    b = (torch.randn(N, generator=g) * 0.01).to(BF)
    ref = O.linear(a.float(), w, b, O.BF16)
    out = ops.gemm(a.to(dev), w.to(dev), b.to(dev))
    torch.cuda.synchronize()
    parity.auto(rel_l2(out, ref), 2e-3)
    assert ulp_diff_frac(out, ref, 1) < 1e-3


@pytest.mark.parametrize("epi", ["gelu", "silu", "gate_res", "res"])
def test_gemm_epilogues(dev, epi):
    ops = _ops()
    M, N, K, U = 417, 768, 512, 3
    g = torch.Generator().manual_seed(5)
    a = torch.randn(M, K, generator=g).to(BF)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF)
    b = (torch.randn(N, generator=g) * 0.1).to(BF)
    res = torch.randn(M, N, generator=g).to(BF)
    gate = torch.randn(U, 2 * N, generator=g).to(BF)     # stride 2N, use columns [N:2N)
    rows = torch.randint(0, U, (M,), generator=g, dtype=torch.int32)
    p = O.BF16
    y = O.linear(a.float(), w, b, p)
    if epi == "gelu":
        ref = O.gelu_tanh(y, p)
        out = ops.gemm(a.to(dev), w.to(dev), b.to(dev), epilogue=ops.EPI_BIAS_GELU)
    elif epi == "silu":
        ref = O.silu(y, p)
        out = ops.gemm(a.to(dev), w.to(dev), b.to(dev), epilogue=ops.EPI_BIAS_SILU)
    elif epi == "gate_res":
        gv = gate[:, N:].float()[rows.long()]
        ref = p.r(res.float() + p.r(y * gv))
        gd = gate.to(dev)
        out = ops.gemm(a.to(dev), w.to(dev), b.to(dev), epilogue=ops.EPI_BIAS_GATE_RES, resid=res.to(dev),
                       gate=gd[:, N:], gate_row=rows.to(dev), gate_stride=2 * N)
    else:
        ref = p.r(res.float() + y)
        out = ops.gemm(a.to(dev), w.to(dev), b.to(dev), epilogue=ops.EPI_BIAS_RES, resid=res.to(dev))
    torch.cuda.synchronize()
    parity.auto(rel_l2(out, ref), 3e-3)
    assert ulp_diff_frac(out, ref, 2) < 2e-3


def test_gemm_transposed_out(dev):
    ops = _ops()
    B, T, N, K = 2, 1296, 512, 256
    g = torch.Generator().manual_seed(9)
    a = torch.randn(B * T, K, generator=g).to(BF)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF)
    b = (torch.randn(N, generator=g) * 0.1).to(BF)
    ref = O.linear(a.float(), w, b, O.BF16).reshape(B, T, N).transpose(1, 2)
    Tp = (T + 63) // 64 * 64
    out = torch.zeros(B, N, Tp, dtype=BF, device=dev)
    ops.gemm(a.to(dev), w.to(dev), b.to(dev), out=out, out_tokens_per_batch=T)
    torch.cuda.synchronize()
    parity.auto(rel_l2(out[:, :, :T], ref), 2e-3)
    assert float(out[:, :, T:].abs().max()) == 0.0


def test_gemm_bad_args(dev):
    ops = _ops()
    from mlx_video_amd._lib import LtxkError
    a = torch.zeros(8, 100, dtype=BF, device=dev)
    w = torch.zeros(16, 100, dtype=BF, device=dev)
    with pytest.raises(LtxkError, match="multiple of 64"):
        ops.gemm(a, w, None)


@pytest.mark.parametrize("B,H,Tq,Tk", [(1, 2, 128, 64), (2, 4, 320, 1024), (1, 3, 1296, 1296), (1, 32, 1280, 1280), (1, 2, 77, 200)])
def test_flash_attn(dev, B, H, Tq, Tk):
    ops = _ops()
    D = H * 128
    g = torch.Generator().manual_seed(B * 100 + Tq)
    q = torch.randn(B, Tq, D, generator=g).to(BF)
    k = torch.randn(B, Tk, D, generator=g).to(BF)
    v = torch.randn(B, Tk, D, generator=g).to(BF)
    ref = O.sdpa(q.float(), k.float(), v.float(), H, O.BF16)
    Tp = (Tk + 63) // 64 * 64
    vt = torch.zeros(B, D, Tp, dtype=BF)
    vt[:, :, :Tk] = v.transpose(1, 2)
    out = torch.empty(B * Tq, D, dtype=BF, device=dev)
    ops.flash_attn(q.reshape(B * Tq, D).to(dev), k.reshape(B * Tk, D).to(dev), vt.to(dev), out, B, H, Tq, Tk,
                   1.0 / math.sqrt(128))
    torch.cuda.synchronize()
    # P is rounded to bf16 before P.V (flash form); the oracle keeps P in fp32: stated tolerance 1e-2 rel-L2
    parity.auto(rel_l2(out.reshape(B, Tq, D), ref), 1e-2)
    # ... and against the oracle's "flash" policy, which rounds P where the kernel does (integer exp2-domain offset, so
    # the rounded P does not depend on the tiling): what is left is fp32 summation order + final bf16 rounding flips
    ref_f = O.sdpa(q.float(), k.float(), v.float(), H, O.BF16_FLASH)
    parity.auto(rel_l2(out.reshape(B, Tq, D), ref_f), 5e-4, tag="vs_flash_policy")
    # batch-invariant launch form (LTXK_ATTN_NO_TAIL_SPLIT): the same rounding points, another summation order
    out2 = torch.empty_like(out)
    ops.flash_attn(q.reshape(B * Tq, D).to(dev), k.reshape(B * Tk, D).to(dev), vt.to(dev), out2, B, H, Tq, Tk,
                   1.0 / math.sqrt(128), tail_split=False)
    torch.cuda.synchronize()
    parity.auto(rel_l2(out2.reshape(B, Tq, D), ref_f), 5e-4, tag="no_tail_split_vs_flash_policy")


def test_flash_attn_spiked_max(dev):
    """Force the online-softmax rescale: one key dominates late in the sequence."""
    ops = _ops()
    B, H, Tq, Tk, D = 1, 1, 64, 256, 128
    g = torch.Generator().manual_seed(3)
    q = torch.randn(B, Tq, D, generator=g).to(BF)
    k = torch.randn(B, Tk, D, generator=g).to(BF)
    v = torch.randn(B, Tk, D, generator=g).to(BF)
    k[0, 200] = q[0, 5] * 4.0
    ref = O.sdpa(q.float(), k.float(), v.float(), H, O.BF16)
    vt = v.transpose(1, 2).contiguous()
    out = torch.empty(B * Tq, D, dtype=BF, device=dev)
    ops.flash_attn(q.reshape(-1, D).to(dev), k.reshape(-1, D).to(dev), vt.to(dev), out, B, H, Tq, Tk, 1.0 / math.sqrt(128))
    torch.cuda.synchronize()
    parity.auto(rel_l2(out.reshape(B, Tq, D), ref), 1e-2)
    # the rescale multiplies O and l by an exact power of two (integer offsets): still only summation order vs the flash policy
    parity.auto(rel_l2(out.reshape(B, Tq, D), O.sdpa(q.float(), k.float(), v.float(), H, O.BF16_FLASH)), 5e-4, tag="vs_flash_policy")


@pytest.mark.parametrize("mod", [False, True])
def test_rmsnorm_modulate(dev, mod):
    ops = _ops()
    M, D, U = 70, 4096, 3
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(M, D, generator=g) * 3).to(BF)
    p = O.BF16
    n = O.rms_norm(x.float(), p, 1e-6)
    if mod:
        tab = torch.randn(U, 6 * D, generator=g).to(BF)
        rows = torch.randint(0, U, (M,), generator=g, dtype=torch.int32)
        sc, sh = tab[:, D:2 * D].float()[rows.long()], tab[:, :D].float()[rows.long()]
        ref = O.modulate(n, sc, sh, p)
        mag = (n * (1 + sc)).abs() + sh.abs()      # the final add can cancel
        td = tab.to(dev)
        out = ops.rmsnorm_modulate(x.to(dev), 1e-6, td[:, D:2 * D], td[:, :D], 6 * D, rows.to(dev))
    else:
        ref, mag = n, None
        out = ops.rmsnorm_modulate(x.to(dev), 1e-6)
    torch.cuda.synchronize()
    assert ulp_diff_frac(out, ref, 1, mag) < 1e-3
    parity.auto(rel_l2(out, ref), 1e-3)


def test_layernorm_modulate(dev):
    ops = _ops()
    M, D = 33, 4096
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(M, D, generator=g) * 2 + 0.3).to(BF)
    tab = torch.randn(1, 2 * D, generator=g).to(BF)
    p = O.BF16
    ref = O.modulate(O.layer_norm_noaffine(x.float(), p, 1e-6), tab[:, D:].float(), tab[:, :D].float(), p)
    td = tab.to(dev)
    out = ops.layernorm_modulate(x.to(dev), 1e-6, td[:, D:], td[:, :D], 2 * D, None)
    torch.cuda.synchronize()
    mag = (O.layer_norm_noaffine(x.float(), p, 1e-6) * (1 + tab[:, D:].float())).abs() + tab[:, :D].float().abs()
    parity.auto(rel_l2(out, ref), 2e-3)
    assert ulp_diff_frac(out, ref, 1, mag) < 2e-3


@pytest.mark.parametrize("rope", [True, False])
def test_qknorm_rope(dev, rope):
    ops = _ops()
    B, F, Hh, Ww, H = 2, 2, 3, 4, 32
    T, D = F * Hh * Ww, 4096
    g = torch.Generator().manual_seed(4)
    qk = torch.randn(B * T, 2 * D, generator=g).to(BF)
    w = (1 + 0.1 * torch.randn(2, D, generator=g)).to(BF)
    p = O.BF16
    pos = torch.from_numpy(O.create_position_grid(1, F, Hh, Ww))
    cos, sin = O.precompute_freqs_cis(pos, D, heads=H)
    outs = []
    for s in range(2):
        x = O.rms_norm(qk[:, s * D:(s + 1) * D].float().reshape(B, T, D), p, 1e-6, w[s])
        if rope:
            x = O.apply_split_rotary_emb(x, cos.expand(B, -1, -1, -1), sin.expand(B, -1, -1, -1), p)
        outs.append(x.reshape(B * T, D))
    ref = torch.cat(outs, dim=1)
    buf = qk.to(dev).clone()
    ops.qknorm_rope(buf, 2, D, w.to(dev), cos[0].contiguous().to(dev) if rope else None,
                    sin[0].contiguous().to(dev) if rope else None, T, H, 1e-6)
    torch.cuda.synchronize()
    parity.auto(rel_l2(buf, ref), 1e-3)
    assert ulp_diff_frac(buf, ref, 1) < 2e-3


def test_timestep_embed_ada_silu(dev):
    ops = _ops()
    t = torch.tensor([1000.0, 912.0, 0.0, 50.0]).to(BF)
    ref = O.BF16.r(O.get_timestep_embedding(t.float()))
    out = ops.timestep_embed(t.to(dev))
    torch.cuda.synchronize()
    # args up to 1000 rad in fp32: |d cos| <= ulp(1000)=6e-5, below bf16 resolution except at roundings
    assert float((out.float().cpu() - ref).abs().max()) < 8e-3
    g = torch.Generator().manual_seed(8)
    tab = torch.randn(3, 6, 512, generator=g).to(BF)
    ada = torch.randn(2, 6 * 512, generator=g).to(BF)
    ref2 = O.BF16.r(tab.float()[:, None] + ada.float().reshape(2, 6, 512)[None])
    out2 = ops.ada_combine(tab.to(dev), ada.to(dev), 3, 2, 6, 512)
    x = torch.randn(4096, generator=g).to(BF)
    out3 = ops.silu(x.to(dev))
    torch.cuda.synchronize()
    assert torch.equal(out2.float().cpu(), ref2)
    assert ulp_diff_frac(out3, O.silu(x.float(), O.BF16), 1) == 0.0


def test_latent_tokens_and_euler(dev):
    ops = _ops()
    B, C, F, Hh, Ww = 2, 128, 3, 4, 5
    S = F * Hh * Ww
    g = torch.Generator().manual_seed(6)
    lat = torch.randn(B, C, F, Hh, Ww, generator=g).to(BF)
    tok = ops.latent_to_tokens(lat.to(dev), rep=2)
    torch.cuda.synchronize()
    ref_tok = O.latent_to_tokens(lat)
    assert torch.equal(tok[:B].cpu(), ref_tok) and torch.equal(tok[B:].cpu(), ref_tok)   # bit-exact index map
    vp = torch.randn(B, S, C, generator=g).to(BF)
    vn = torch.randn(B, S, C, generator=g).to(BF)
    clean = torch.randn(B, C, F, Hh, Ww, generator=g).to(BF)
    mask = torch.ones(B, 1, F, 1, 1)
    mask[:, :, 0] = 0.0
    mask[:, :, 1] = 0.75
    p = O.BF16
    sig, sig_n = O.bf16_round_scalar(0.909375), O.bf16_round_scalar(0.725)
    for use_mask, sn in [(False, sig_n), (True, sig_n), (True, 0.0)]:
        v = O.cfg_combine(vp.float(), vn.float(), 4.0, p)
        x0 = O.to_denoised(lat.float(), O.tokens_to_latent(v, lat.shape), sig, p)
        if use_mask:
            x0 = O.apply_denoise_mask(x0, clean.float(), mask, p)
        ref = p.r(x0 + sn * (lat.float() - x0) / sig) if sn > 0 else x0
        mt = mask.expand(B, 1, F, Hh, Ww).reshape(B, S).contiguous().to(dev) if use_mask else None
        out = ops.cfg_euler_step(vp.to(dev), vn.to(dev), lat.to(dev), 4.0, sig, sn,
                                 clean.to(dev) if use_mask else None, mt)
        torch.cuda.synchronize()
        assert ulp_diff_frac(out, ref, 1) < 1e-4, (use_mask, sn)


# ---------------------------------------------------------------------------------------------------- round 2: fused forms
@pytest.mark.parametrize("M,T,D", [(2 * 1280, 1280, 4096), (2 * 77, 77, 512), (433, 433, 1024)])
def test_gemm_split_output_and_sumsq(dev, M, T, D):
    """One launch over the packed q|k|v panel: columns [0,2D) row-major + their per-row sums of squares in 64-column
    partials, columns [2D,3D) transposed per batch (V^T).  Each part must equal the separate launches bit for bit.
    (Row counts above the split-K range: whether a small-M launch sums K in slices depends on its column count, so launches
    of different widths only agree bit for bit when they are single-pass; tests/test_gemm_epilogues_gpu.py covers split-K.)"""
    ops = _ops()
    g = torch.Generator().manual_seed(M + D)
    a = torch.randn(M, D, generator=g).to(BF).to(dev)
    w = (torch.randn(3 * D, D, generator=g) * 0.02).to(BF).to(dev)
    b = (torch.randn(3 * D, generator=g) * 0.01).to(BF).to(dev)
    B = M // T
    Tp = (T + 63) // 64 * 64
    qk = torch.empty(M, 2 * D, dtype=BF, device=dev)
    vt = torch.zeros(B, D, Tp, dtype=BF, device=dev)
    ss = torch.zeros(M, 2 * D // 64, dtype=torch.float32, device=dev)
    ops.gemm(a, w, b, out=qk, out2=vt, n_split=2 * D, out_tokens_per_batch=T, sumsq=ss)
    qk_ref = ops.gemm(a, w[:2 * D].contiguous(), b[:2 * D].contiguous())
    vt_ref = torch.zeros(B, D, Tp, dtype=BF, device=dev)
    ops.gemm(a, w[2 * D:].contiguous(), b[2 * D:].contiguous(), out=vt_ref, out_tokens_per_batch=T)
    torch.cuda.synchronize()
    assert torch.equal(qk, qk_ref) and torch.equal(vt, vt_ref)
    want = (qk.double() ** 2).reshape(M, -1, 64).sum(-1)
    assert float(((ss.double() - want).abs() / want.clamp_min(1e-30)).max()) < 1e-5       # fp32 sums of exact products
    # and against the oracle's Linear
    ref = O.linear(a.float().cpu(), w.cpu(), b.cpu(), O.BF16)
    parity.auto(rel_l2(qk, ref[:, :2 * D]), 2e-3)


@pytest.mark.parametrize("mod", [False, True])
@pytest.mark.parametrize("D", [4096, 512, 1536])
def test_rmsnorm_modulate_with_row_stats(dev, mod, D):
    """rms_norm + modulation with the sums of squares supplied (gemm's sumsq form) and the (1+scale) factor
    precomputed by ada_combine's one_plus_mask: same oracle, same tolerance as the self-reducing kernel."""
    ops = _ops()
    M, U = 70, 3
    g = torch.Generator().manual_seed(11 + D)
    x = (torch.randn(M, D, generator=g) * 3).to(BF)
    p = O.BF16
    n = O.rms_norm(x.float(), p, 1e-6)
    ss = (x.float() ** 2).reshape(M, D // 64, 64).sum(-1).to(dev)
    if mod:
        tab = torch.randn(1, 6, D, generator=g).to(BF)
        ada = torch.randn(U, 6 * D, generator=g).to(BF)
        rows = torch.randint(0, U, (M,), generator=g, dtype=torch.int32)
        comb = p.r(tab.float() + ada.float().reshape(U, 6, D))                       # (U,6,D)
        sc, sh = comb[:, 1][rows.long()], comb[:, 0][rows.long()]
        ref = O.modulate(n, sc, sh, p)
        mag = (n * (1 + sc)).abs() + sh.abs()
        mods = ops.ada_combine(tab.to(dev), ada.to(dev), 1, U, 6, D, one_plus_mask=0b010010)[0]
        torch.cuda.synchronize()
        assert torch.equal(mods[:, 1].float().cpu(), p.r(1.0 + comb[:, 1])) and torch.equal(mods[:, 0].float().cpu(), comb[:, 0])
        out = ops.rmsnorm_modulate(x.to(dev), 1e-6, mods[:, 1], mods[:, 0], 6 * D, rows.to(dev), sumsq=ss, scale_is_one_plus=True)
    else:
        ref, mag = n, None
        out = ops.rmsnorm_modulate(x.to(dev), 1e-6, sumsq=ss)
    torch.cuda.synchronize()
    assert ulp_diff_frac(out, ref, 1, mag) < 1e-3
    parity.auto(rel_l2(out, ref), 1e-3)


@pytest.mark.parametrize("rope", [True, False])
@pytest.mark.parametrize("H", [32, 4])
def test_attention_with_fused_query_prep(dev, rope, H):
    """q_norm (+ SPLIT RoPE) applied to the Q fragments inside the attention kernel from the raw projection and its
    row statistics, against (i) the standalone qknorm_rope kernel followed by plain attention and (ii) the oracle's
    attention.py:129-136 + sdpa chain."""
    ops = _ops()
    B, F, Hh, Ww = 2, 3, 4, 5
    T, D = F * Hh * Ww, H * 128
    Tk = 96
    g = torch.Generator().manual_seed(21 + H)
    q = torch.randn(B * T, D, generator=g).to(BF)
    k = torch.randn(B * Tk, D, generator=g).to(BF)
    v = torch.randn(B, Tk, D, generator=g).to(BF)
    w = (1 + 0.1 * torch.randn(1, D, generator=g)).to(BF)
    pos = torch.from_numpy(O.create_position_grid(1, F, Hh, Ww))
    cos, sin = O.precompute_freqs_cis(pos, D, heads=H)
    cd, sd = (cos[0].contiguous().to(dev), sin[0].contiguous().to(dev)) if rope else (None, None)
    vt = torch.zeros(B, D, 128, dtype=BF)
    vt[:, :, :Tk] = v.transpose(1, 2)
    sc = 1.0 / math.sqrt(128)
    ss = (q.float() ** 2).reshape(B * T, D // 64, 64).sum(-1).to(dev)
    fused = torch.empty(B * T, D, dtype=BF, device=dev)
    ops.flash_attn(q.to(dev), k.to(dev), vt.to(dev), fused, B, H, T, Tk, sc, q_sumsq=ss, q_norm_weight=w.to(dev), cos=cd, sin=sd, eps=1e-6)
    qn = q.to(dev).clone()
    ops.qknorm_rope(qn, 1, D, w.to(dev), cd, sd, T, H, 1e-6)
    two = torch.empty(B * T, D, dtype=BF, device=dev)
    ops.flash_attn(qn, k.to(dev), vt.to(dev), two, B, H, T, Tk, sc)
    qs = q.to(dev).clone()
    ops.qknorm_rope(qs, 1, D, w.to(dev), cd, sd, T, H, 1e-6, sumsq=ss)                 # the stats-driven standalone form
    torch.cuda.synchronize()
    # same op order and rounding points; only the fp32 order of the row's sum of squares differs (1-ulp flips of rstd)
    assert ulp_diff_frac(qs, qn, 1) < 2e-3 and rel_l2(qs, qn) < 1e-3
    assert rel_l2(fused, two) < 2e-3
    p = O.BF16
    xq = O.rms_norm(q.float().reshape(B, T, D), p, 1e-6, w[0])
    if rope:
        xq = O.apply_split_rotary_emb(xq, cos.expand(B, -1, -1, -1), sin.expand(B, -1, -1, -1), p)
    ref = O.sdpa(xq, k.float().reshape(B, Tk, D), v.float(), H, p)
    parity.auto(rel_l2(fused.reshape(B, T, D), ref), 1e-2)


@pytest.mark.parametrize("B,H,Tq,Tk", [(2, 32, 1280, 1280), (1, 4, 200, 333), (2, 8, 1280, 1024)])
def test_flash_attn_mfma_shapes_agree(dev, B, H, Tq, Tk, monkeypatch, ab_lib):
    """The shipped kernel computes on v_mfma_f32_16x16x32_bf16 (fa_body16); the 32x32x16 form it replaced stays in the A/B
    build (LTXK_FA_MFMA=32).  Same rounding points (integer softmax offset, P rounded to bf16, l in fp32), another operand
    map and summation order: fp32-order differences only, with and without the fused query preparation and the tail split."""
    ops = _ops()
    D = H * 128
    g = torch.Generator(device=dev).manual_seed(B + H + Tq + Tk)
    q = torch.randn((B * Tq, D), generator=g, device=dev).to(BF)
    k = torch.randn((B * Tk, D), generator=g, device=dev).to(BF)
    vt = torch.randn((B, D, (Tk + 63) // 64 * 64), generator=g, device=dev).to(BF)
    ss = (q.float() ** 2).reshape(B * Tq, D // 64, 64).sum(-1).contiguous()
    w = (1 + 0.1 * torch.randn(D, generator=g, device=dev)).to(BF)
    outs = {}
    for shape in ("16", "32"):
        monkeypatch.setenv("LTXK_FA_MFMA", shape)
        for name, kw in (("plain", {}), ("no_split", dict(tail_split=False)), ("qprep", dict(q_sumsq=ss, q_norm_weight=w, eps=1e-6))):
            o = torch.empty((B * Tq, D), dtype=BF, device=dev)
            ops.flash_attn(q, k, vt, o, B, H, Tq, Tk, 1.0 / math.sqrt(128), **kw)
            outs[(shape, name)] = o
    torch.cuda.synchronize()
    for name in ("plain", "no_split", "qprep"):
        parity.auto(rel_l2(outs[("16", name)], outs[("32", name)]), 3e-4, tag=name)


@pytest.mark.parametrize("B,H,Tq,Tk", [(2, 32, 1280, 1280), (2, 32, 1296, 1024), (1, 8, 200, 333), (3, 16, 777, 130), (4, 32, 640, 640), (2, 8, 2100, 64)])
def test_flash_attn_query_blocks_per_wave_agree(dev, B, H, Tq, Tk, monkeypatch, ab_lib):
    """Round 4: a launch is a mixed grid of 192-row tiles (three 16-row MFMA blocks per wave) and 128-row tiles (two), chosen per
    shape (attention.hip, fa_pick_mix).  More rows per wave changes which K / V^T fragment feeds how many MFMAs, not the order in
    which a row meets its keys: the mixed grid, 192-row tiles only (LTXK_FA_QB=3 in the A/B build) and the 128-row kernel without
    its key-split tail (LTXK_FA_QB=2 + LTXK_ATTN_NO_TAIL_SPLIT) must agree BIT FOR BIT - plain and with the fused query
    preparation, ragged query tiles (1296 = 6 x 192 + 144, 777, 200) and ragged key tiles."""
    ops = _ops()
    D = H * 128
    g = torch.Generator(device=dev).manual_seed(B * 7 + H + Tq + Tk)
    q = torch.randn((B * Tq, D), generator=g, device=dev).to(BF)
    k = torch.randn((B * Tk, D), generator=g, device=dev).to(BF)
    vt = torch.randn((B, D, (Tk + 63) // 64 * 64), generator=g, device=dev).to(BF)
    ss = (q.float() ** 2).reshape(B * Tq, D // 64, 64).sum(-1).contiguous()
    w = (1 + 0.1 * torch.randn(D, generator=g, device=dev)).to(BF)
    cos, sin = torch.randn((H, Tq, 64), generator=g, device=dev), torch.randn((H, Tq, 64), generator=g, device=dev)
    outs = {}
    # (tail_split=False throughout: below 1.25 rounds of 128-row tiles the default launch is the 128-row kernel, whose key-split
    # tail is the one form with another summation order)
    for form, env, ts in (("two_blocks", "2", False), ("mixed", "0", False), ("three_blocks", "3", False)):
        monkeypatch.setenv("LTXK_FA_QB", env)
        for name, kw in (("plain", {}), ("qprep", dict(q_sumsq=ss, q_norm_weight=w, cos=cos, sin=sin, eps=1e-6))):
            o = torch.full((B * Tq + 1, D), 7.0, dtype=BF, device=dev)
            ops.flash_attn(q, k, vt, o[:B * Tq], B, H, Tq, Tk, 1.0 / math.sqrt(128), tail_split=ts, **kw)
            torch.cuda.synchronize()
            assert bool((o[B * Tq:] == 7.0).all()), "wrote past the last query row"
            outs[(form, name)] = o[:B * Tq]
    for name in ("plain", "qprep"):
        assert torch.equal(outs[("mixed", name)], outs[("two_blocks", name)]), f"mixed grid vs 128-row tiles ({name})"
        assert torch.equal(outs[("three_blocks", name)], outs[("two_blocks", name)]), f"192-row tiles vs 128-row tiles ({name})"

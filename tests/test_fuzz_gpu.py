"""Seeded shape fuzzing of the three MFMA kernels through the C ABI against the CPU oracle.

The fixed-shape tests (test_kernels_gpu.py, test_vae_gpu.py) cover the model's own sizes; these sweep the
edges the kernels' tilings care about: rows/columns that are not multiples of the 160x256 tile, outputs whose
row stride rules out the 16-byte store path, attention grids with and without a short last round (the
tail-split path of attention.hip), ragged key tiles, odd convolution volumes with every padding mode.
One process, fixed seeds, oracle sizes that finish in seconds."""
import math
import random

import parity
import pytest
import torch

from oracle import dit as O
from oracle import vae as OV

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _gemm_cases(n=29):
    rnd = random.Random(20261)
    cases = [(1, 8, 64, "bias", 0), (161, 264, 64, "res", 4), (159, 248, 128, "gate_res", 0), (320, 512, 192, "gelu", 12),
             # small M x long K: the split-K weight-streaming form (gemm.hip; config 1's geometry, M = B*32 tokens)
             (64, 1024, 4096, "gate_res", 0), (64, 2048, 2048, "gelu", 0), (32, 512, 1024, "bias", 4), (320, 1024, 4096, "res", 0),
             (200, 776, 1536, "silu", 12)]
    while len(cases) < n:
        M = rnd.choice([rnd.randint(1, 40), rnd.randint(100, 700)])
        N = 8 * rnd.randint(1, 140)
        K = 64 * rnd.randint(1, 8)
        epi = rnd.choice(["bias", "gelu", "silu", "gate_res", "res"])
        pad = rnd.choice([0, 0, 4, 8, 12])          # extra output row stride: 4 and 12 force the 8-byte store path
        cases.append((M, N, K, epi, pad))
    return cases


@pytest.mark.parametrize("M,N,K,epi,pad", _gemm_cases())
def test_gemm_fuzz(dev, M, N, K, epi, pad):
    from mlx_video_amd import ops
    g = torch.Generator().manual_seed(M * 131 + N * 7 + K)
    a = torch.randn(M, K, generator=g).to(BF)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF)
    b = (torch.randn(N, generator=g) * 0.1).to(BF)
    res = torch.randn(M, N, generator=g).to(BF)
    U = 3
    gate = torch.randn(U, N, generator=g).to(BF)
    rows = torch.randint(0, U, (M,), generator=g, dtype=torch.int32)
    p = O.BF16
    y = O.linear(a.float(), w, b, p)
    kw = {}
    if epi == "bias":
        ref, e = y, ops.EPI_BIAS
    elif epi == "gelu":
        ref, e = O.gelu_tanh(y, p), ops.EPI_BIAS_GELU
    elif epi == "silu":
        ref, e = O.silu(y, p), ops.EPI_BIAS_SILU
    elif epi == "gate_res":
        ref, e = p.r(res.float() + p.r(y * gate.float()[rows.long()])), ops.EPI_BIAS_GATE_RES
        kw = dict(resid=res.to(dev), gate=gate.to(dev), gate_row=rows.to(dev), gate_stride=N)
    else:
        ref, e = p.r(res.float() + y), ops.EPI_BIAS_RES
        kw = dict(resid=res.to(dev))
    buf = torch.full((M, N + pad), 7.0, dtype=BF, device=dev)       # sentinel in the padding columns
    out = buf[:, :N]
    ops.gemm(a.to(dev), w.to(dev), b.to(dev), epilogue=e, out=out, **kw)
    torch.cuda.synchronize()
    parity.auto(rel_l2(out, ref), 4e-3)
    if pad:
        assert bool((buf[:, N:] == 7.0).all())                     # nothing written past N


def _attn_cases():
    # (B,H,Tq,Tk): ragged q and k, k shorter than one tile, grids below / exactly / above the 512-slot round
    # (8*32*2=512 tiles -> no tail; 8*32*3=768 -> tail of 256 tiles split over 512 half workgroups; 5*8*13 = 520 ->
    # tail of 8), H not a multiple of 8 (plain tile order instead of the XCD map)
    return [(1, 1, 1, 1), (1, 2, 33, 63), (2, 3, 129, 65), (1, 8, 200, 700), (8, 32, 256, 128), (8, 32, 384, 192),
            (5, 8, 1664, 96), (1, 5, 640, 320), (3, 7, 100, 1000), (2, 16, 2100, 64)]


@pytest.mark.parametrize("B,H,Tq,Tk", _attn_cases())
def test_flash_attn_fuzz(dev, B, H, Tq, Tk):
    from mlx_video_amd import ops
    D = H * 128
    g = torch.Generator(device=dev).manual_seed(B * 1000 + H * 100 + Tq + Tk)
    q = torch.randn((B, Tq, D), generator=g, device=dev).to(BF)
    k = torch.randn((B, Tk, D), generator=g, device=dev).to(BF)
    v = torch.randn((B, Tk, D), generator=g, device=dev).to(BF)
    Tp = (Tk + 63) // 64 * 64
    vt = torch.zeros(B, D, Tp, dtype=BF, device=dev)
    vt[:, :, :Tk] = v.transpose(1, 2)
    out = torch.full((B * Tq, D), 9.0, dtype=BF, device=dev)
    ops.flash_attn(q.reshape(B * Tq, D), k.reshape(B * Tk, D), vt, out, B, H, Tq, Tk, 1.0 / math.sqrt(128))
    torch.cuda.synchronize()
    # fp32 reference of the same op (the big grids would take the CPU oracle minutes); same math as oracle.dit.sdpa
    qh = q.float().reshape(B, Tq, H, 128).transpose(1, 2)
    kh = k.float().reshape(B, Tk, H, 128).transpose(1, 2)
    vh = v.float().reshape(B, Tk, H, 128).transpose(1, 2)
    ref = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(128), -1) @ vh
    ref = ref.transpose(1, 2).reshape(B, Tq, D)
    parity.auto(rel_l2(out.reshape(B, Tq, D), ref), 1e-2)
    if B * H * Tq * Tk <= 2_000_000:                                  # small cases also against the oracle itself
        oref = O.sdpa(q.float().cpu(), k.float().cpu(), v.float().cpu(), H, O.BF16)
        parity.auto(rel_l2(out.reshape(B, Tq, D), oref), 1e-2)


def _conv_cases(n=14):
    rnd = random.Random(77)
    # the kernel's contract: Cin a multiple of 64 (callers zero-pad, e.g. the encoder's 48-channel input), Cout of 8
    cases = [(64, 8, 1, False, (1, 1, 2, 2)), (64, 128, 1, False, (1, 1, 3, 3)), (128, 48, 0, True, (1, 3, 2, 9))]
    while len(cases) < n:
        cin = 64 * rnd.randint(1, 6)
        cout = 8 * rnd.randint(1, 40)
        causal = rnd.choice([0, 1])
        reflect = rnd.choice([False, True])
        shape = (rnd.randint(1, 2), rnd.randint(1, 5), rnd.randint(2, 9), rnd.randint(2, 9))
        cases.append((cin, cout, causal, reflect, shape))
    return cases


@pytest.mark.parametrize("cin,cout,causal,reflect,shape", _conv_cases())
def test_conv3d_fuzz(dev, cin, cout, causal, reflect, shape):
    from mlx_video_amd import video_vae as V
    b, d, h, w = shape
    g = torch.Generator().manual_seed(cin * 13 + cout + d * h * w)
    x = torch.randn(b, cin, d, h, w, generator=g).to(BF)
    wt = (torch.randn(cout, 3, 3, 3, cin, generator=g) / (27 * cin) ** 0.5).to(BF)
    bias = (torch.randn(cout, generator=g) * 0.1).to(BF)
    res = torch.randn(b, cout, d, h, w, generator=g).to(BF)
    ref = OV.causal_conv3d(x.float(), wt, bias, O.BF16, bool(causal), reflect)
    cl = lambda t: t.permute(0, 2, 3, 4, 1).contiguous()
    cf = lambda t: t.permute(0, 4, 1, 2, 3).contiguous()
    mode = V.PAD_REFLECT if reflect else V.PAD_ZEROS
    out = V.conv3d(cl(x).to(dev), wt.to(dev), bias.to(dev), bool(causal), mode)
    out_r = V.conv3d(cl(x).to(dev), wt.to(dev), bias.to(dev), bool(causal), mode, resid=cl(res).to(dev))
    torch.cuda.synchronize()
    parity.auto(rel_l2(cf(out), ref), 4e-3)
    parity.auto(rel_l2(cf(out_r), O.BF16.r(ref + res.float())), 4e-3)


def _kw_conv_cases(n=10):
    """Seeded shapes that take the kw-reuse convolution kernel (conv3d.hip: Cout <= 128, W >= 64, more than 128 tiles of 256
    rows): odd widths and heights, 1-2 batch items, every channel-block count the ABI allows."""
    rnd = random.Random(303)
    cases = []
    while len(cases) < n:
        b, d = rnd.randint(1, 2), rnd.randint(1, 4)
        w = rnd.randint(64, 210)
        h = max(2, -(-33500 // (b * d * w)) + rnd.randint(0, 6))          # B*D*H*W > 128 * 256 rows
        cases.append((64 * rnd.randint(1, 3), rnd.choice([8, 48, 64, 104, 128]), rnd.choice([0, 1]), rnd.choice([False, True]), (b, d, h, w)))
    return cases


@pytest.mark.parametrize("cin,cout,causal,reflect,shape", _kw_conv_cases())
def test_conv3d_kw_form_fuzz_vs_oracle(dev, cin, cout, causal, reflect, shape):
    from mlx_video_amd import video_vae as V
    b, d, h, w = shape
    assert b * d * h * w > 128 * 256
    g = torch.Generator().manual_seed(cin * 7 + cout * 3 + d * h + w)
    x = torch.randn(b, cin, d, h, w, generator=g).to(BF)
    wt = (torch.randn(cout, 3, 3, 3, cin, generator=g) / (27 * cin) ** 0.5).to(BF)
    bias = (torch.randn(cout, generator=g) * 0.1).to(BF)
    res = torch.randn(b, cout, d, h, w, generator=g).to(BF)
    ref = OV.causal_conv3d(x.float(), wt, bias, O.BF16, bool(causal), reflect)
    cl = lambda t: t.permute(0, 2, 3, 4, 1).contiguous()
    cf = lambda t: t.permute(0, 4, 1, 2, 3).contiguous()
    mode = V.PAD_REFLECT if reflect else V.PAD_ZEROS
    out = V.conv3d(cl(x).to(dev), wt.to(dev), bias.to(dev), bool(causal), mode)
    out_r = V.conv3d(cl(x).to(dev), wt.to(dev), bias.to(dev), bool(causal), mode, resid=cl(res).to(dev))
    torch.cuda.synchronize()
    parity.auto(rel_l2(cf(out), ref), 4e-3)
    parity.auto(rel_l2(cf(out_r), O.BF16.r(ref + res.float())), 4e-3)

"""The VAE decoder's error budget, split the way tests/test_block_stages_gpu.py splits the DiT's: one
FULL-SIZE residual block of the 128-channel stage (decoder.py:94-180; 33 x 128 x 128 voxels, the launches that hold 42 % of
the decoder's FLOPs) and one full-size DepthToSpaceUpsample (sampling.py:143-197; 256 -> 128 channels, 17 x 64 x 64 ->
33 x 128 x 128), each run two ways against oracle/vae.py's taps:

* TEACHER-FORCED: every HIP stage is fed the ORACLE's input of that stage, so its error is the kernel's own (fp32 summation
  order, plus the bf16 flips that causes in its own output);
* CHAINED as in the product: each stage eats the previous HIP output, which is how the 1.4e-2 of the full decode accumulates.

Stated tolerance per stage, teacher-forced: PixelNorm + SiLU (HBM kernel, op-by-op bf16 roundings reproduced) <= 2e-3 - one
bf16 flip of the rounded mean moves a whole voxel row by an ulp; convolutions (K = 27 * 128 = 3456 or 6912 terms) <= 1e-3;
chained block <= 6e-3."""
import math

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


def cl(x):
    return x.permute(0, 2, 3, 4, 1).contiguous()


def cf(x):
    return x.permute(0, 4, 1, 2, 3).contiguous()


def test_res_block_128_full_size_stage_budget(dev):
    from mlx_video_amd import video_vae as V
    p = O.BF16
    C, vol = 128, (33, 128, 128)
    g = torch.Generator().manual_seed(404)
    x = torch.randn(1, C, *vol, generator=g).to(BF)
    W = {}
    for n in ("conv1", "conv2"):
        W[f"b.{n}.conv.weight"] = (torch.randn(C, 3, 3, 3, C, generator=g) / math.sqrt(27 * C)).to(BF)
        W[f"b.{n}.conv.bias"] = (torch.randn(C, generator=g) * 0.01).to(BF)
    # ---- the oracle's taps (the formulas of OV.resnet_block_simple, stage by stage)
    t0 = O.silu(OV.pixel_norm(x.float(), p, 1e-8), p)
    t1 = OV.causal_conv3d(t0, W["b.conv1.conv.weight"], W["b.conv1.conv.bias"], p, False, True)
    t2 = O.silu(OV.pixel_norm(t1, p, 1e-8), p)
    t3 = p.r(OV.causal_conv3d(t2, W["b.conv2.conv.weight"], W["b.conv2.conv.bias"], p, False, True) + x.float())
    Wd = {k: v.to(dev) for k, v in W.items()}
    xd = cl(x).to(dev)
    # ---- teacher-forced
    f0 = V.pixelnorm_act(xd, 1e-8, True)
    f1 = V.conv3d(cl(t0.to(BF)).to(dev), Wd["b.conv1.conv.weight"], Wd["b.conv1.conv.bias"], False, V.PAD_REFLECT)
    f2 = V.pixelnorm_act(cl(t1.to(BF)).to(dev), 1e-8, True)
    f3 = V.conv3d(cl(t2.to(BF)).to(dev), Wd["b.conv2.conv.weight"], Wd["b.conv2.conv.bias"], False, V.PAD_REFLECT, resid=xd)
    # ---- chained, as LTX2VideoDecoder runs the block
    c1 = V.conv3d(f0, Wd["b.conv1.conv.weight"], Wd["b.conv1.conv.bias"], False, V.PAD_REFLECT)
    c2 = V.pixelnorm_act(c1, 1e-8, True)
    c3 = V.conv3d(c2, Wd["b.conv2.conv.weight"], Wd["b.conv2.conv.bias"], False, V.PAD_REFLECT, resid=xd)
    torch.cuda.synchronize()
    parity.check("vae.stage.res128.pixelnorm_silu_1.teacher_forced", rel_l2(cf(f0), t0), 2e-3)
    parity.check("vae.stage.res128.conv1.teacher_forced", rel_l2(cf(f1), t1), 1e-3)
    parity.check("vae.stage.res128.pixelnorm_silu_2.teacher_forced", rel_l2(cf(f2), t2), 2e-3)
    parity.check("vae.stage.res128.conv2_residual.teacher_forced", rel_l2(cf(f3), t3), 1e-3)
    parity.check("vae.stage.res128.conv1.chained", rel_l2(cf(c1), t1), 4e-3)
    parity.check("vae.stage.res128.pixelnorm_silu_2.chained", rel_l2(cf(c2), t2), 6e-3)
    parity.check("vae.stage.res128.block_out.chained", rel_l2(cf(c3), t3), 6e-3)
    # the taps ARE the oracle's block: its own composition of the same formulas gives the same bits
    assert torch.equal(OV.resnet_block_simple(x.float(), W, "b", p, False, None), t3)


def test_d2s_block_256_to_128_full_size_stage_budget(dev):
    from mlx_video_amd import video_vae as V
    p = O.BF16
    Ci, vol = 256, (17, 64, 64)
    g = torch.Generator().manual_seed(405)
    x = torch.randn(1, Ci, *vol, generator=g).to(BF)
    W = {"u.conv.weight": (torch.randn(4 * Ci, 3, 3, 3, Ci, generator=g) / math.sqrt(27 * Ci)).to(BF),
         "u.conv.bias": (torch.randn(4 * Ci, generator=g) * 0.01).to(BF)}
    t_conv = OV.causal_conv3d(x.float(), W["u.conv.weight"], W["u.conv.bias"], p, False, True)
    xr = OV.depth_to_space(x.float(), 2, 2, 2).repeat(1, 4, 1, 1, 1)[:, :, 1:]
    t_out = p.r(OV.depth_to_space(t_conv, 2, 2, 2)[:, :, 1:] + xr)
    assert torch.equal(OV.d2s_upsample(x.float(), W, "u", p, False), t_out)
    xd = cl(x).to(dev)
    f_conv = V.conv3d(xd, W["u.conv.weight"].to(dev), W["u.conv.bias"].to(dev), False, V.PAD_REFLECT)
    f_out_tf = V.d2s_add(cl(t_conv.to(BF)).to(dev), xd)            # the rearrangement + residual fed the oracle's conv output
    f_out = V.d2s_add(f_conv, xd)
    torch.cuda.synchronize()
    assert f_out.shape == (1, 33, 128, 128, 128)
    parity.check("vae.stage.d2s256.conv.teacher_forced", rel_l2(cf(f_conv), t_conv), 1e-3)
    assert torch.equal(cf(f_out_tf).float().cpu(), t_out), "depth-to-space + residual is an index map plus one rounding: bit-exact"
    parity.check("vae.stage.d2s256.block_out.chained", rel_l2(cf(f_out), t_out), 2e-3)

"""CPU restatement (torch) of the LTX-2 causal-3D-conv video VAE: decoder
(video_vae/decoder.py), encoder (video_vae/video_vae.py:220-372), their building blocks and the
tiled decode (video_vae/tiling.py).  TEST INFRASTRUCTURE — see oracle/__init__.py.

Tensors are channels-first (B,C,D,H,W) as in the reference's module interfaces.  Conv weights
use the reference's MLX layout (O,kD,kH,kW,I) (decoder.py:708-710)."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from .dit import Prec, linear, silu

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# CausalConv3d (convolution.py:78-222)
# --------------------------------------------------------------------------------------
def causal_conv3d(x: Tensor, w: Tensor, b: Tensor, p: Prec, causal: bool, reflect: bool) -> Tensor:
    """3x3x3 stride-1 conv; temporal halo by frame replication (causal: 2x first; else first+last,
    convolution.py:122-137); spatial halo reflect or zeros (143-157).  w: (O,3,3,3,I)."""
    x = x.to(p.dtype)
    if causal:
        x = torch.cat([x[:, :, :1].repeat(1, 1, 2, 1, 1), x], dim=2)
    else:
        x = torch.cat([x[:, :, :1], x, x[:, :, -1:]], dim=2)
    if reflect:
        x = torch.cat([x[:, :, :, 1:2], x, x[:, :, :, -2:-1]], dim=3)
        x = torch.cat([x[:, :, :, :, 1:2], x, x[:, :, :, :, -2:-1]], dim=4)
    else:
        x = F.pad(x, (1, 1, 1, 1))
    wt = w.to(p.dtype).permute(0, 4, 1, 2, 3).contiguous()          # (O,I,kD,kH,kW)
    return p.r(F.conv3d(x, wt, b.to(p.dtype)))


def pixel_norm(x: Tensor, p: Prec, eps: float) -> Tensor:
    """decoder.py:136-138 (eps 1e-8) / utils.py:477-483 (eps 1e-6): x/sqrt(mean_c(x^2)+eps), op by op."""
    xf = x.to(p.dtype)
    m = p.r(p.r(xf * xf).mean(dim=1, keepdim=True))
    return p.r(xf / p.r(torch.sqrt(p.r(m + eps))))


def depth_to_space(x: Tensor, st: int, sh: int, sw: int) -> Tensor:
    """sampling.py:143-157: channel = (c, st, sh, sw)."""
    b, cp, d, h, w = x.shape
    c = cp // (st * sh * sw)
    x = x.reshape(b, c, st, sh, sw, d, h, w).permute(0, 1, 5, 2, 6, 3, 7, 4)
    return x.reshape(b, c, d * st, h * sh, w * sw)


def space_to_depth(x: Tensor, st: int, sh: int, sw: int) -> Tensor:
    """sampling.py:53-72."""
    b, c, d, h, w = x.shape
    x = x.reshape(b, c, d // st, st, h // sh, sh, w // sw, sw).permute(0, 1, 3, 5, 7, 2, 4, 6)
    return x.reshape(b, c * st * sh * sw, d // st, h // sh, w // sw)


def patchify(x: Tensor, ps: int = 4) -> Tensor:
    """ops.py:9-44: channel order (c, p_w, p_h) — width before height."""
    b, c, f, h, w = x.shape
    x = x.reshape(b, c, f, 1, h // ps, ps, w // ps, ps).permute(0, 1, 3, 7, 5, 2, 4, 6)
    return x.reshape(b, c * ps * ps, f, h // ps, w // ps)


def unpatchify(x: Tensor, ps: int = 4) -> Tensor:
    """ops.py:47-80."""
    b, cp, f, h, w = x.shape
    c = cp // (ps * ps)
    x = x.reshape(b, c, 1, ps, ps, f, h, w).permute(0, 1, 5, 2, 6, 4, 7, 3)
    return x.reshape(b, c, f, h * ps, w * ps)


# --------------------------------------------------------------------------------------
# decoder (decoder.py:94-450)
# --------------------------------------------------------------------------------------
DEC_CH = (1024, 512, 256, 128)


def vae_timestep_embedding(t: Tensor, dim: int = 256) -> Tensor:
    """decoder.py:30-55 (sin|cos then flipped => cos|sin)."""
    half = dim // 2
    e = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
    a = t.reshape(-1, 1).float() * e[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


def time_embedder(t: Tensor, W: Dict[str, Tensor], prefix: str, p: Prec) -> Tensor:
    """PixArtAlphaTimestepEmbedder (decoder.py:75-91)."""
    proj = p.r(vae_timestep_embedding(t))
    h = linear(proj, W[f"{prefix}.timestep_embedder.linear_1.weight"], W[f"{prefix}.timestep_embedder.linear_1.bias"], p)
    h = silu(h, p)
    return linear(h, W[f"{prefix}.timestep_embedder.linear_2.weight"], W[f"{prefix}.timestep_embedder.linear_2.bias"], p)


def _mod(x: Tensor, scale: Tensor, shift: Tensor, p: Prec) -> Tensor:
    return p.r(p.r(x * p.r(1.0 + scale)) + shift)


def resnet_block_simple(x: Tensor, W: Dict[str, Tensor], pre: str, p: Prec, causal: bool,
                        ts_embed: Optional[Tensor]) -> Tensor:
    """ResnetBlock3DSimple (decoder.py:94-180)."""
    res = x
    b = x.shape[0]
    h = pixel_norm(x, p, 1e-8)
    if ts_embed is not None:
        c = x.shape[1]
        ada = p.r(W[f"{pre}.scale_shift_table"].to(p.dtype)[None] + ts_embed.reshape(b, 4, c))   # (B,4,C)
        sh1, sc1, sh2, sc2 = [ada[:, i].reshape(b, c, 1, 1, 1) for i in range(4)]
        h = _mod(h, sc1, sh1, p)
    h = silu(h, p)
    h = causal_conv3d(h, W[f"{pre}.conv1.conv.weight"], W[f"{pre}.conv1.conv.bias"], p, causal, True)
    h = pixel_norm(h, p, 1e-8)
    if ts_embed is not None:
        h = _mod(h, sc2, sh2, p)
    h = silu(h, p)
    h = causal_conv3d(h, W[f"{pre}.conv2.conv.weight"], W[f"{pre}.conv2.conv.bias"], p, causal, True)
    return p.r(h + res)


def d2s_upsample(x: Tensor, W: Dict[str, Tensor], pre: str, p: Prec, causal: bool) -> Tensor:
    """DepthToSpaceUpsample stride (2,2,2), residual, reduction 2 (sampling.py:159-197)."""
    xr = depth_to_space(x, 2, 2, 2).repeat(1, 4, 1, 1, 1)[:, :, 1:]          # tile x4, drop first frame
    h = causal_conv3d(x, W[f"{pre}.conv.weight"], W[f"{pre}.conv.bias"], p, causal, True)
    h = depth_to_space(h, 2, 2, 2)[:, :, 1:]
    return p.r(h + xr)


def vae_decode(latent: Tensor, W: Dict[str, Tensor], p: Prec, causal: bool = False,
               timestep: Optional[float] = None, noise: Optional[Tensor] = None,
               layers_per_block: int = 5) -> Tensor:
    """LTX2VideoDecoder.__call__ (decoder.py:361-450).  latent (B,128,F',H',W') ->
    (B,3,8(F'-1)+1,32H',32W').  ``timestep`` not None => timestep-conditioned variant
    (noise must then be given: the reference draws mx.random.normal, decoder.py:381-385)."""
    x = p.r(latent)
    b = x.shape[0]
    tc = timestep is not None
    if tc:
        x = p.r(p.r(noise * 0.025) + p.r((1.0 - 0.025) * x))
    mean = W["latents_mean"].float().reshape(1, -1, 1, 1, 1)
    std = W["latents_std"].float().reshape(1, -1, 1, 1, 1)
    x = p.r(x * std + mean)
    st = None
    if tc:
        st = p.r(torch.full((b,), float(timestep)) * 1000.0)
    x = causal_conv3d(x, W["conv_in.conv.weight"], W["conv_in.conv.bias"], p, causal, True)
    for bi in range(7):
        pre = f"up_blocks.{bi}"
        if bi % 2 == 0:
            emb = time_embedder(st, W, f"{pre}.time_embedder", p) if tc else None
            for li in range(layers_per_block):
                x = resnet_block_simple(x, W, f"{pre}.res_blocks.{li}", p, causal, emb)
        else:
            x = d2s_upsample(x, W, pre, p, causal)
    x = pixel_norm(x, p, 1e-8)
    if tc:
        emb = time_embedder(st, W, "last_time_embedder", p)
        ada = p.r(W["last_scale_shift_table"].to(p.dtype)[None] + emb.reshape(b, 2, 128))
        x = _mod(x, ada[:, 1].reshape(b, 128, 1, 1, 1), ada[:, 0].reshape(b, 128, 1, 1, 1), p)
    x = silu(x, p)
    x = causal_conv3d(x, W["conv_out.conv.weight"], W["conv_out.conv.bias"], p, causal, True)
    return unpatchify(x, 4)


def make_decoder_weights(seed: int = 1234, dtype=torch.bfloat16, timestep_conditioning: bool = False,
                         layers_per_block: int = 5, scale: float = 1.0) -> Dict[str, Tensor]:
    """Synthetic decoder checkpoint, key names as after decoder.py:544-591 (conv weights (O,3,3,3,I)).
    Conv weights N(0, 1/(27 I)) so activations stay O(1) through 40+ layers."""
    g = torch.Generator().manual_seed(seed)
    W: Dict[str, Tensor] = {}

    def conv(name, o, i):
        W[f"{name}.weight"] = (torch.randn(o, 3, 3, 3, i, generator=g) * (scale / math.sqrt(27 * i))).to(dtype)
        W[f"{name}.bias"] = (torch.randn(o, generator=g) * 0.01).to(dtype)

    def lin(name, o, i):
        W[f"{name}.weight"] = (torch.randn(o, i, generator=g) * 0.02).to(dtype)
        W[f"{name}.bias"] = (torch.randn(o, generator=g) * 0.01).to(dtype)

    W["latents_mean"] = (torch.randn(128, generator=g) * 0.1).to(dtype)
    W["latents_std"] = (1.0 + 0.1 * torch.randn(128, generator=g)).abs().to(dtype)
    conv("conv_in.conv", 1024, 128)
    for bi in range(7):
        pre = f"up_blocks.{bi}"
        if bi % 2 == 0:
            c = DEC_CH[bi // 2]
            if timestep_conditioning:
                lin(f"{pre}.time_embedder.timestep_embedder.linear_1", 4 * c, 256)
                lin(f"{pre}.time_embedder.timestep_embedder.linear_2", 4 * c, 4 * c)
            for li in range(layers_per_block):
                conv(f"{pre}.res_blocks.{li}.conv1.conv", c, c)
                conv(f"{pre}.res_blocks.{li}.conv2.conv", c, c)
                if timestep_conditioning:
                    W[f"{pre}.res_blocks.{li}.scale_shift_table"] = (torch.randn(4, c, generator=g) * 0.05).to(dtype)
        else:
            c = DEC_CH[bi // 2]
            conv(f"{pre}.conv", c * 4, c)           # out = (c/2)*8
    conv("conv_out.conv", 48, 128)
    if timestep_conditioning:
        lin("last_time_embedder.timestep_embedder.linear_1", 256, 256)
        lin("last_time_embedder.timestep_embedder.linear_2", 256, 256)
        W["last_scale_shift_table"] = (torch.randn(2, 128, generator=g) * 0.05).to(dtype)
    return W


# --------------------------------------------------------------------------------------
# encoder (video_vae.py:220-372, sampling.py:11-103, resnet.py:33-121, encoder.py:95-105)
# --------------------------------------------------------------------------------------
ENC_BLOCKS = [("res_x", 4), ("compress_space_res", (1, 2, 2)), ("res_x", 6), ("compress_time_res", (2, 1, 1)),
              ("res_x", 6), ("compress_all_res", (2, 2, 2)), ("res_x", 2), ("compress_all_res", (2, 2, 2)),
              ("res_x", 2)]


def resnet_block_enc(x: Tensor, W: Dict[str, Tensor], pre: str, p: Prec) -> Tensor:
    """ResnetBlock3D with PixelNorm(eps 1e-6), causal, zero spatial pad (resnet.py:95-121)."""
    h = silu(pixel_norm(x, p, 1e-6), p)
    h = causal_conv3d(h, W[f"{pre}.conv1.weight"], W[f"{pre}.conv1.bias"], p, True, False)
    h = silu(pixel_norm(h, p, 1e-6), p)
    h = causal_conv3d(h, W[f"{pre}.conv2.weight"], W[f"{pre}.conv2.bias"], p, True, False)
    return p.r(h + x)


def s2d_downsample(x: Tensor, W: Dict[str, Tensor], pre: str, p: Prec, stride, out_channels: int) -> Tensor:
    """SpaceToDepthDownsample (sampling.py:74-103)."""
    st, sh, sw = stride
    if st == 2:
        x = torch.cat([x[:, :, :1], x], dim=2)
    xin = space_to_depth(x.to(p.dtype), st, sh, sw)
    b, c2, d2, h2, w2 = xin.shape
    g = c2 // out_channels
    xin = p.r(xin.reshape(b, out_channels, g, d2, h2, w2).mean(dim=2))
    h = causal_conv3d(x, W[f"{pre}.conv.weight"], W[f"{pre}.conv.bias"], p, True, False)
    return p.r(space_to_depth(h, st, sh, sw) + xin)


def vae_encode(video: Tensor, W: Dict[str, Tensor], p: Prec, blocks=None) -> Tensor:
    """VideoEncoder.__call__ (video_vae.py:321-372): (B,3,F,H,W) in [-1,1] -> normalised means
    (B,128,F',H/32,W/32).  F must be 1+8k."""
    blocks = ENC_BLOCKS if blocks is None else blocks
    if (video.shape[2] - 1) % 8 != 0:
        raise ValueError("Invalid number of frames: Encode input must have 1 + 8 * x frames")
    x = patchify(p.r(video), 4)
    x = causal_conv3d(x, W["conv_in.weight"], W["conv_in.bias"], p, True, False)
    ch = 128
    for bi, (kind, arg) in enumerate(blocks):
        pre = f"down_blocks.{bi}"
        if kind == "res_x":
            for li in range(arg):
                x = resnet_block_enc(x, W, f"{pre}.res_blocks.{li}", p)
        else:
            ch *= 2
            x = s2d_downsample(x, W, pre, p, arg, ch)
    x = silu(pixel_norm(x, p, 1e-6), p)
    x = causal_conv3d(x, W["conv_out.weight"], W["conv_out.bias"], p, True, False)
    means = x[:, :128]
    mean = W["per_channel_statistics.mean"].float().reshape(1, -1, 1, 1, 1)
    std = W["per_channel_statistics.std"].float().reshape(1, -1, 1, 1, 1)
    return p.r((means - mean) / std)


def make_encoder_weights(seed: int = 4321, dtype=torch.bfloat16, blocks=None) -> Dict[str, Tensor]:
    blocks = ENC_BLOCKS if blocks is None else blocks
    g = torch.Generator().manual_seed(seed)
    W: Dict[str, Tensor] = {}

    def conv(name, o, i):
        W[f"{name}.weight"] = (torch.randn(o, 3, 3, 3, i, generator=g) / math.sqrt(27 * i)).to(dtype)
        W[f"{name}.bias"] = (torch.randn(o, generator=g) * 0.01).to(dtype)

    conv("conv_in", 128, 48)
    ch = 128
    for bi, (kind, arg) in enumerate(blocks):
        pre = f"down_blocks.{bi}"
        if kind == "res_x":
            for li in range(arg):
                conv(f"{pre}.res_blocks.{li}.conv1", ch, ch)
                conv(f"{pre}.res_blocks.{li}.conv2", ch, ch)
        else:
            mult = arg[0] * arg[1] * arg[2]
            conv(f"{pre}.conv", ch * 2 // mult, ch)
            ch *= 2
    conv("conv_out", 129, ch)
    W["per_channel_statistics.mean"] = (torch.randn(128, generator=g) * 0.1).to(dtype)
    W["per_channel_statistics.std"] = (1.0 + 0.1 * torch.randn(128, generator=g)).abs().to(dtype)
    return W


def to_uint8(video: Tensor, p: Prec) -> Tensor:
    """generate.py:3894-3898: (C,F,H,W)->(F,H,W,C), clip((x+1)/2,0,1)*255 -> uint8 (truncation)."""
    v = video.permute(1, 2, 3, 0)
    v = torch.clamp(p.r(p.r(v + 1.0) / 2.0), 0.0, 1.0)
    return p.r(v * 255).to(torch.uint8)


# --------------------------------------------------------------------------------------
# tiled decode (tiling.py:17-62,223-509)
# --------------------------------------------------------------------------------------
def trapezoid_mask(length: int, ramp_left: int, ramp_right: int, left_starts_from_0: bool = False) -> Tensor:
    """tiling.py:17-62."""
    if length <= 0:
        raise ValueError("Mask length must be positive.")
    ramp_left = max(0, min(ramp_left, length))
    ramp_right = max(0, min(ramp_right, length))
    mask = [1.0] * length
    if ramp_left > 0:
        n = ramp_left + 1 if left_starts_from_0 else ramp_left + 2
        full = [i / (n - 1) for i in range(n)]
        fade = full[:-1]
        if not left_starts_from_0:
            fade = fade[1:]
        for i in range(min(ramp_left, len(fade))):
            mask[i] *= fade[i]
    if ramp_right > 0:
        fade_out = [(ramp_right + 1 - i) / (ramp_right + 1) for i in range(1, ramp_right + 1)]
        for i in range(ramp_right):
            mask[length - ramp_right + i] *= fade_out[i]
    return torch.tensor(mask, dtype=torch.float32).clamp(0, 1)


def split_spatial(size: int, overlap: int, dim: int):
    """tiling.py:223-235 -> (starts, ends, left_ramps, right_ramps)."""
    if dim <= size:
        return [0], [dim], [0], [0]
    n = (dim + size - 2 * overlap - 1) // (size - overlap)
    starts = [i * (size - overlap) for i in range(n)]
    ends = [s + size for s in starts]
    ends[-1] = dim
    return starts, ends, [0] + [overlap] * (n - 1), [overlap] * (n - 1) + [0]


def split_temporal(size: int, overlap: int, dim: int):
    """tiling.py:238-254."""
    if dim <= size:
        return [0], [dim], [0], [0]
    starts, ends, lefts, rights = split_spatial(size, overlap, dim)
    starts = [s - 1 if i > 0 else s for i, s in enumerate(starts)]
    lefts = [l + 1 if i > 0 else l for i, l in enumerate(lefts)]
    return starts, ends, lefts, rights


def decode_with_tiling(decode_fn: Callable[[Tensor], Tensor], latents: Tensor, spatial_tile_px: Optional[int],
                       spatial_overlap_px: int, temporal_tile_f: Optional[int], temporal_overlap_f: int, p: Prec) -> Tensor:
    """tiling.py:279-509 with spatial_scale 32, temporal_scale 8."""
    b, c, fl, hl, wl = latents.shape
    of, oh, ow = 1 + (fl - 1) * 8, hl * 32, wl * 32
    st, so = (spatial_tile_px // 32, spatial_overlap_px // 32) if spatial_tile_px else (max(hl, wl), 0)
    tt, to = (temporal_tile_f // 8, temporal_overlap_f // 8) if temporal_tile_f else (fl, 0)
    T, Hs, Ws = split_temporal(tt, to, fl), split_spatial(st, so, hl), split_spatial(st, so, wl)
    out = torch.zeros(b, 3, of, oh, ow, dtype=torch.float32)
    wsum = torch.zeros(b, 1, of, oh, ow, dtype=torch.float32)
    for ti in range(len(T[0])):
        t0, t1 = T[0][ti] * 8, 1 + (T[1][ti] - 1) * 8
        lr = 1 + (T[2][ti] - 1) * 8 if T[2][ti] > 0 else 0
        tm = trapezoid_mask(t1 - t0, lr, T[3][ti] * 8, True)
        for hi in range(len(Hs[0])):
            h0, h1 = Hs[0][hi] * 32, Hs[1][hi] * 32
            hm = trapezoid_mask(h1 - h0, Hs[2][hi] * 32, Hs[3][hi] * 32, False)
            for wi in range(len(Ws[0])):
                w0, w1 = Ws[0][wi] * 32, Ws[1][wi] * 32
                wm = trapezoid_mask(w1 - w0, Ws[2][wi] * 32, Ws[3][wi] * 32, False)
                tile = decode_fn(latents[:, :, T[0][ti]:T[1][ti], Hs[0][hi]:Hs[1][hi], Ws[0][wi]:Ws[1][wi]]).float()
                m = tm.reshape(1, 1, -1, 1, 1) * hm.reshape(1, 1, 1, -1, 1) * wm.reshape(1, 1, 1, 1, -1)
                out[:, :, t0:t1, h0:h1, w0:w1] += tile * m
                wsum[:, :, t0:t1, h0:h1, w0:w1] += m
    return p.r(out / wsum.clamp_min(1e-8))


# --------------------------------------------------------------------------------------
# latent upsampler (models/ltx/upsampler.py:6-316)
# --------------------------------------------------------------------------------------
def conv3d_zero(x: Tensor, w: Tensor, b: Tensor, p: Prec) -> Tensor:
    """upsampler.Conv3d: padding=1 zeros in all three dims (upsampler.py:6-62)."""
    wt = w.to(p.dtype).permute(0, 4, 1, 2, 3).contiguous()
    return p.r(F.conv3d(x.to(p.dtype), wt, b.to(p.dtype), padding=1))


def group_norm3d(x: Tensor, gamma: Tensor, beta: Tensor, p: Prec, groups: int = 32, eps: float = 1e-5) -> Tensor:
    """GroupNorm3d (upsampler.py:65-98): fp32 stats over (voxels, C/G); x channels-first here."""
    b, c = x.shape[:2]
    xf = x.float().reshape(b, groups, c // groups, -1)
    mean = xf.mean(dim=(2, 3), keepdim=True)
    var = ((xf - mean) ** 2).mean(dim=(2, 3), keepdim=True)
    y = ((xf - mean) / torch.sqrt(var + eps)).reshape(x.shape)
    return p.r(y * gamma.float().reshape(1, -1, 1, 1, 1) + beta.float().reshape(1, -1, 1, 1, 1))


def upsampler_forward(latent: Tensor, W: Dict[str, Tensor], p: Prec, nb: int = 4) -> Tensor:
    """LatentUpsampler.__call__ (upsampler.py:211-294), channels-first throughout."""
    def res(x, pre):
        h = conv3d_zero(x, W[f"{pre}.conv1.weight"], W[f"{pre}.conv1.bias"], p)
        h = silu(group_norm3d(h, W[f"{pre}.norm1.weight"], W[f"{pre}.norm1.bias"], p), p)
        h = conv3d_zero(h, W[f"{pre}.conv2.weight"], W[f"{pre}.conv2.bias"], p)
        h = group_norm3d(h, W[f"{pre}.norm2.weight"], W[f"{pre}.norm2.bias"], p)
        return silu(p.r(h + x), p)
    x = conv3d_zero(p.r(latent), W["initial_conv.weight"], W["initial_conv.bias"], p)
    x = silu(group_norm3d(x, W["initial_norm.weight"], W["initial_norm.bias"], p), p)
    for i in range(nb):
        x = res(x, f"res_blocks.{i}")
    b, c, d, h, w = x.shape
    x2 = x.permute(0, 2, 1, 3, 4).reshape(b * d, c, h, w)
    w2 = W["upsampler.conv.weight"].to(p.dtype).permute(0, 3, 1, 2).contiguous()            # (O,3,3,I)->(O,I,3,3)
    y = p.r(F.conv2d(x2.to(p.dtype), w2, W["upsampler.conv.bias"].to(p.dtype), padding=1))
    y = F.pixel_shuffle(y, 2)                                                                 # channel (oc,ry,rx)
    y = y.reshape(b, d, c, 2 * h, 2 * w).permute(0, 2, 1, 3, 4)
    for i in range(nb):
        y = res(y, f"post_upsample_res_blocks.{i}")
    return conv3d_zero(y, W["final_conv.weight"], W["final_conv.bias"], p)


def upsample_latents(latent: Tensor, W: Dict[str, Tensor], mean: Tensor, std: Tensor, p: Prec, nb: int = 4) -> Tensor:
    """upsampler.py:297-316."""
    m, s = mean.float().reshape(1, -1, 1, 1, 1), std.float().reshape(1, -1, 1, 1, 1)
    x = p.r(p.r(latent) * s + m)
    x = upsampler_forward(x, W, p, nb)
    return p.r((x - m) / s)


def make_upsampler_weights(mid: int = 128, seed: int = 99, dtype=torch.bfloat16, nb: int = 4) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    W: Dict[str, Tensor] = {}

    def conv(name, o, i):
        W[f"{name}.weight"] = (torch.randn(o, 3, 3, 3, i, generator=g) / math.sqrt(27 * i)).to(dtype)
        W[f"{name}.bias"] = (torch.randn(o, generator=g) * 0.01).to(dtype)

    def norm(name, c):
        W[f"{name}.weight"] = (1.0 + 0.1 * torch.randn(c, generator=g)).to(dtype)
        W[f"{name}.bias"] = (0.1 * torch.randn(c, generator=g)).to(dtype)

    conv("initial_conv", mid, 128)
    norm("initial_norm", mid)
    for stage in ("res_blocks", "post_upsample_res_blocks"):
        for i in range(nb):
            conv(f"{stage}.{i}.conv1", mid, mid); norm(f"{stage}.{i}.norm1", mid)
            conv(f"{stage}.{i}.conv2", mid, mid); norm(f"{stage}.{i}.norm2", mid)
    W["upsampler.conv.weight"] = (torch.randn(4 * mid, 3, 3, mid, generator=g) / math.sqrt(9 * mid)).to(dtype)
    W["upsampler.conv.bias"] = (torch.randn(4 * mid, generator=g) * 0.01).to(dtype)
    conv("final_conv", 128, mid)
    return W

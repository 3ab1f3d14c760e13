"""Latent upsampler (mlx_video/models/ltx/upsampler.py:6-373): Conv3d(zero pad) + GroupNorm(32) + SiLU
residual stack, per-frame Conv2d + PixelShuffle(2), and the un-normalise / re-normalise wrapper
(upsampler.py:297-316).  Needed between the two stages of the distilled / keyframe / ic_lora pipelines
(generate.py:3196).  Convolutions reuse the VAE's implicit-GEMM kernel with zero temporal padding
(``causal=2``); the per-frame 3x3 Conv2d runs as that kernel's 9-tap mode (``taps_d=1``: K = 9*Cin, only the centre temporal tap)."""
from __future__ import annotations

from typing import Dict

import torch

from . import _lib
from ._lib import check
from .video_vae import BF16, PAD_ZEROS, _p, _stream, conv3d

ZERO_T = 2      # ltxk_conv3d_args.causal: zeros on both temporal sides


def groupnorm_act(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, resid=None, silu: bool = True,
                  groups: int = 32, eps: float = 1e-5) -> torch.Tensor:
    B, C = x.shape[0], x.shape[-1]
    V = x.numel() // (B * C)
    out = torch.empty_like(x)
    check(_lib.load().ltxk_groupnorm_act(_p(x), _p(out), _p(gamma), _p(beta), _p(resid), B, V, C, groups, eps, int(silu),
                                         _stream()), "ltxk_groupnorm_act")
    return out


class LatentUpsampler:
    """upsampler.py:178-294.  ``weights``: bf16 device tensors; Conv3d weights (O,3,3,3,I), the Conv2d
    weight (O,3,3,I) (MLX layouts, upsampler.py:352-358)."""

    def __init__(self, weights: Dict[str, torch.Tensor], num_blocks_per_stage: int = 4):
        W = {k: v.contiguous() for k, v in weights.items()}
        if "initial_conv.weight" not in W or "upsampler.conv.weight" not in W:
            raise ValueError("Missing latent-upsampler parameters (initial_conv / upsampler.conv)")
        self.mid_channels = W["initial_conv.weight"].shape[0]       # detected from the weights (upsampler.py:333-337)
        self.nb = num_blocks_per_stage
        self.W = W

    def _res(self, x: torch.Tensor, pre: str) -> torch.Tensor:
        W = self.W
        h = conv3d(x, W[f"{pre}.conv1.weight"], W[f"{pre}.conv1.bias"], ZERO_T, PAD_ZEROS)
        h = groupnorm_act(h, W[f"{pre}.norm1.weight"], W[f"{pre}.norm1.bias"], None, True)
        h = conv3d(h, W[f"{pre}.conv2.weight"], W[f"{pre}.conv2.bias"], ZERO_T, PAD_ZEROS)
        return groupnorm_act(h, W[f"{pre}.norm2.weight"], W[f"{pre}.norm2.bias"], x, True)   # silu(norm2 + residual)

    def __call__(self, latent: torch.Tensor, debug: bool = False) -> torch.Tensor:
        """(B,C,F,H,W) -> (B,C,F,2H,2W)."""
        W = self.W
        x = latent.to(BF16).permute(0, 2, 3, 4, 1).contiguous()
        x = conv3d(x, W["initial_conv.weight"], W["initial_conv.bias"], ZERO_T, PAD_ZEROS)
        x = groupnorm_act(x, W["initial_norm.weight"], W["initial_norm.bias"], None, True)
        for i in range(self.nb):
            x = self._res(x, f"res_blocks.{i}")
        B, D, H, Wd, C = x.shape
        # per-frame 3x3 Conv2d (upsampler.py:64-99): the conv kernel's 9-tap mode (no zero-filled temporal taps)
        y = conv3d(x, W["upsampler.conv.weight"], W["upsampler.conv.bias"], ZERO_T, PAD_ZEROS)
        # PixelShuffle(2) (upsampler.py:101-122): channel (oc, ry, rx) -> pixel (2h+ry, 2w+rx); pure index map
        y = y.reshape(B * D, H, Wd, C, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(B, D, 2 * H, 2 * Wd, C).contiguous()
        for i in range(self.nb):
            y = self._res(y, f"post_upsample_res_blocks.{i}")
        y = conv3d(y, W["final_conv.weight"], W["final_conv.bias"], ZERO_T, PAD_ZEROS)
        return y.permute(0, 4, 1, 2, 3).contiguous()


def upsample_latents(latent: torch.Tensor, upsampler: LatentUpsampler, latent_mean: torch.Tensor,
                     latent_std: torch.Tensor, debug: bool = False) -> torch.Tensor:
    """upsampler.py:297-316: un-normalise (x*std+mean), upsample, re-normalise ((x-mean)/std), each an
    elementwise bf16 op in the reference; done here by the VAE's (de)normalise kernels."""
    lib = _lib.load()
    B, C, F, H, W = latent.shape
    S = F * H * W
    lat = latent.to(BF16).contiguous()
    cl = torch.empty((B, S, C), dtype=BF16, device=lat.device)
    check(lib.ltxk_latent_denorm_cl(_p(lat), None, 0.0, _p(latent_mean), _p(latent_std), _p(cl), B, C, S, _stream()), "ltxk_latent_denorm_cl")
    up = upsampler(cl.reshape(B, F, H, W, C).permute(0, 4, 1, 2, 3))
    B2, C2, F2, H2, W2 = up.shape
    S2 = F2 * H2 * W2
    ucl = up.permute(0, 2, 3, 4, 1).contiguous()
    out = torch.empty((B2, C2, F2, H2, W2), dtype=BF16, device=lat.device)
    check(lib.ltxk_latent_norm_cf(_p(ucl), C2, _p(latent_mean), _p(latent_std), _p(out), B2, C2, S2, _stream()), "ltxk_latent_norm_cf")
    return out

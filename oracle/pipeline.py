"""CPU restatement of the two-stage (distilled / keyframe / ic_lora) generate flow, mlx_video/generate.py:3050-3372,
composed from the oracle's stage functions.  TEST INFRASTRUCTURE — see oracle/__init__.py.

Flow restated (file:line of the reference):
  conditioning: every image is VAE-encoded at the half- and full-resolution stage and applied with the chosen mode;
    video conditionings (IC-LoRA) are encoded at half resolution and guide stage 1 only, as keyframes (3064-3113)
  stage 1: zeros -> apply_conditioning -> masked noise blend n*(m*s0) + x*(1-m*s0) (3143-3160), or plain noise
    (3162-3164); denoise_distilled over the subsampled STAGE_1 schedule (3170-3181)
  upsample_latents (3196)
  stage 2: optional LoRA-merged transformer (3229-3237); with conditionings the upsampled latent is the state's
    latent, conditioning applied, masked noise blend with sigma2[0] (3290-3311); otherwise
    noise*bf16(s0) + latents*bf16(1-s0) (3317-3321); denoise_distilled over the refinement schedule (3359-3369).
Random draws are explicit inputs (``noise_fn(shape)`` is called in the reference's order: stage-1 noise, stage-2 noise).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import dit as O
from . import sched as S
from . import vae as OV

Tensor = torch.Tensor


def merge_lora(W: Dict[str, Tensor], pairs: Dict[str, Tuple[Tensor, Tensor]], strength: float, p: O.Prec) -> Dict[str, Tensor]:
    """lora.py:94-127 in merge mode: W' = bf16(W + bf16(strength * (B @ A))) with the product in fp32; ``pairs`` maps a
    sanitised weight key to (A (r,in), B (out,r))."""
    out = dict(W)
    for key, (A, B) in pairs.items():
        delta = p.r(float(strength) * (p.r(B).to(p.dtype) @ p.r(A).to(p.dtype)))
        out[key] = p.r(W[key].to(p.dtype) + delta).to(W[key].dtype)
    return out


def masked_noise_blend(noise: Tensor, latent: Tensor, mask: Tensor, sigma0: float, p: O.Prec) -> Tensor:
    """generate.py:3153-3160 / 3300-3307 with bf16 arrays: scaled = mask*bf16(s0); n*scaled + x*(1 - scaled), per-op rounding."""
    sm = p.r(p.r(mask) * (O.bf16_round_scalar(float(sigma0)) if p.emulate_bf16 else float(sigma0)))
    return p.r(p.r(p.r(noise) * sm) + p.r(p.r(latent) * p.r(1.0 - sm)))


def plain_renoise(noise: Tensor, latent: Tensor, sigma0: float, p: O.Prec) -> Tensor:
    """generate.py:3317-3321: noise*bf16(s0) + latents*bf16(1 - s0) (1 - s0 evaluated in Python floats first)."""
    ns = O.bf16_round_scalar(float(sigma0)) if p.emulate_bf16 else float(sigma0)
    oms = O.bf16_round_scalar(1.0 - float(sigma0)) if p.emulate_bf16 else 1.0 - float(sigma0)
    return p.r(p.r(p.r(noise) * ns) + p.r(p.r(latent) * oms))


def two_stage(noise_fn: Callable, ctx: Tensor, W1: Dict[str, Tensor], W2: Dict[str, Tensor], cfg: O.DiTConfig,
              Wu: Dict[str, Tensor], nb_up: int, lat_mean: Tensor, lat_std: Tensor, latent_frames: int,
              s1hw: Tuple[int, int], s2hw: Tuple[int, int], sig1: Sequence[float], sig2: Sequence[float],
              conds1: List[tuple], conds2: List[tuple], p: O.Prec, compiled: bool = False, fp32_euler: bool = True,
              fps: float = 24.0) -> Tuple[Tensor, Tensor]:
    """Returns (stage-1 latents, final stage-2 latents).  conds*: (mode, latent (1,128,f,h,w), frame_idx, strength)."""
    f = latent_frames
    shape1 = (1, 128, f, s1hw[0], s1hw[1])
    pos1 = O.create_position_grid(1, f, s1hw[0], s1hw[1], fps=fps)
    clean1 = mask1 = None
    if conds1:
        l0, clean1, mask1 = S.apply_conditioning(torch.zeros(shape1), torch.zeros(shape1), torch.ones(1, 1, f, 1, 1), conds1)
        lat = masked_noise_blend(noise_fn(shape1).float(), l0, mask1, sig1[0], p)
    else:
        lat = p.r(noise_fn(shape1).float())
    lat1 = O.denoise_dev(lat, pos1, ctx, ctx, W1, cfg, list(sig1), p, 1.0, clean1, mask1, compiled=compiled,
                         bf16_euler=not fp32_euler)
    up = OV.upsample_latents(lat1, Wu, lat_mean, lat_std, p, nb_up)
    pos2 = O.create_position_grid(1, f, s2hw[0], s2hw[1], fps=fps)
    clean2 = mask2 = None
    if conds2:
        l2, clean2, mask2 = S.apply_conditioning(up, torch.zeros_like(up), torch.ones(1, 1, f, 1, 1), conds2)
        lat = masked_noise_blend(noise_fn(tuple(up.shape)).float(), l2, mask2, sig2[0], p)
    else:
        lat = plain_renoise(noise_fn(tuple(up.shape)).float(), up, sig2[0], p)
    lat2 = O.denoise_dev(lat, pos2, ctx, ctx, W2, cfg, list(sig2), p, 1.0, clean2, mask2, compiled=compiled,
                         bf16_euler=not fp32_euler)
    return lat1, lat2

"""Latent conditioning hooks (mlx_video/conditioning/latent.py:13-196): frame-indexed replace /
guide conditioning, the denoise mask and the masked initial-noise blend.  Setup-time tensor
plumbing (slicing / concatenation of device tensors); the per-step mask blend itself runs in
ltxk_cfg_euler_step."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple, Union

import torch


@dataclass
class VideoConditionByLatentIndex:
    """Replace the latent at ``frame_idx`` with ``latent`` (B,C,f,H,W); mask = 1-strength."""
    latent: torch.Tensor
    frame_idx: int = 0
    strength: float = 1.0

    def get_num_latent_frames(self) -> int:
        return self.latent.shape[2]


@dataclass
class VideoConditionByKeyframeIndex:
    """Guide with ``keyframes``: noisy latent untouched, clean reference + mask set."""
    keyframes: torch.Tensor
    frame_idx: int = 0
    strength: float = 1.0

    def get_num_latent_frames(self) -> int:
        return self.keyframes.shape[2]


@dataclass
class LatentState:
    latent: torch.Tensor          # (B,C,F,H,W)
    clean_latent: torch.Tensor    # (B,C,F,H,W)
    denoise_mask: torch.Tensor    # (B,1,F,1,1): 1 = full denoise, 0 = keep clean

    def clone(self) -> "LatentState":
        return LatentState(self.latent, self.clean_latent, self.denoise_mask)


def create_initial_state(shape: Tuple[int, ...], noise: torch.Tensor, noise_scale: float = 1.0) -> LatentState:
    """latent.py:79-101.  The reference draws mx.random.normal(seed); MLX's threefry stream is
    not reproducible here, so the noise tensor is an explicit input."""
    return LatentState(latent=noise * noise_scale, clean_latent=torch.zeros_like(noise),
                       denoise_mask=torch.ones((shape[0], 1, shape[2], 1, 1), dtype=noise.dtype, device=noise.device))


def apply_conditioning(state: LatentState,
                       conditionings: List[Union[VideoConditionByLatentIndex, VideoConditionByKeyframeIndex]]) -> LatentState:
    """latent.py:104-177."""
    latent, clean, mask = state.latent.clone(), state.clean_latent.clone(), state.denoise_mask.clone()
    b, c, f, h, w = latent.shape
    for cond in conditionings:
        guide = isinstance(cond, VideoConditionByKeyframeIndex)
        src = cond.keyframes if guide else cond.latent
        _, cc, cf, ch, cw = src.shape
        if (cc, ch, cw) != (c, h, w):
            raise ValueError(f"Conditioning latent spatial shape ({cc}, {ch}, {cw}) does not match target shape ({c}, {h}, {w})")
        if cond.frame_idx >= f:
            raise ValueError(f"Frame index {cond.frame_idx} is out of bounds for latent with {f} frames")
        end = min(cond.frame_idx + cf, f)
        span = end - cond.frame_idx
        if not guide:
            latent[:, :, cond.frame_idx:end] = src[:, :, :span].to(latent.dtype)
        clean[:, :, cond.frame_idx:end] = src[:, :, :span].to(clean.dtype)
        mask[:, :, cond.frame_idx:end] = 1.0 - cond.strength
    return LatentState(latent, clean, mask)


def apply_denoise_mask(denoised: torch.Tensor, clean: torch.Tensor, denoise_mask: torch.Tensor) -> torch.Tensor:
    """latent.py:180-196 (hook contract for host tensors; the denoise loops use the fused kernel)."""
    return denoised * denoise_mask + clean * (1.0 - denoise_mask)


def noise_blend(state: LatentState, noise: torch.Tensor, sigma0: float) -> LatentState:
    """Masked initial noising n*(m*s0) + x*(1-m*s0) (generate.py:3153-3160,3442-3449)."""
    sm = state.denoise_mask * torch.tensor(sigma0, dtype=state.denoise_mask.dtype, device=state.latent.device)
    return LatentState(noise * sm + state.latent * (1.0 - sm), state.clean_latent, state.denoise_mask)

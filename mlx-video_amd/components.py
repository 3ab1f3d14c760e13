"""Hook contracts of ``ltx_core/components`` (protocols.py:8-33): diffusion step, guiders, noiser,
schedulers, patchifier.  In the reference these are only reached from `ltx_pipelines/utils/helpers.py`,
which no entry point executes, so they are mirrored as small host-side
torch-tensor helpers with the *intended* contracts (diffusion_steps.py:9-13, guiders.py:23-108,
noisers.py, schedulers.py:16-107, patchifiers.py:12-60).  The fused production path is
``denoise.denoise_dev`` / ``ltxk_cfg_euler_step``; tests check that the two agree."""
from __future__ import annotations

import math
from dataclasses import dataclass
from functools import lru_cache
from typing import Optional, Tuple

import numpy as np
import torch

from .schedulers import LTX2Scheduler  # noqa: F401  (SchedulerProtocol: execute(steps, latent=...))


def to_velocity(sample: torch.Tensor, denoised: torch.Tensor, sigma) -> torch.Tensor:
    """ltx_core/utils.py:38-52: (x - x0)/sigma in fp32."""
    s = torch.as_tensor(sigma, dtype=torch.float32, device=sample.device)
    return ((sample.float() - denoised.float()) / s).to(sample.dtype)


class EulerDiffusionStep:
    """diffusion_steps.py:9-13: x + v*dt with v = to_velocity(x, x0, sigma_i), dt = sigma_{i+1} - sigma_i."""

    def execute(self, sample: torch.Tensor, denoised_sample: torch.Tensor, sigmas: torch.Tensor, step_index: int) -> torch.Tensor:
        v = to_velocity(sample, denoised_sample, sigmas[step_index])
        dt = float(sigmas[step_index + 1]) - float(sigmas[step_index])
        return (sample.float() + v.float() * dt).to(sample.dtype)


def _l2_norm(x, dims, keepdim=False):
    return torch.sqrt((x * x).sum(dim=dims, keepdim=keepdim) + 1e-8)


def projection_coef(to_project: torch.Tensor, project_onto: torch.Tensor) -> torch.Tensor:
    b = to_project.shape[0]
    p, n = to_project.reshape(b, -1), project_onto.reshape(b, -1)
    return (p * n).sum(dim=1, keepdim=True) / ((n * n).sum(dim=1, keepdim=True) + 1e-8)


@dataclass(frozen=True)
class CFGGuider:
    scale: float

    def delta(self, cond, uncond):
        return (self.scale - 1) * (cond - uncond)

    def enabled(self) -> bool:
        return self.scale != 1.0


@dataclass(frozen=True)
class CFGStarRescalingGuider:
    scale: float

    def delta(self, cond, uncond):
        coef = projection_coef(cond, uncond).reshape(-1, *([1] * (cond.dim() - 1)))
        return (self.scale - 1) * (cond - coef * uncond)

    def enabled(self) -> bool:
        return self.scale != 1.0


@dataclass(frozen=True)
class STGGuider:
    scale: float

    def delta(self, pos_denoised, perturbed_denoised):
        return self.scale * (pos_denoised - perturbed_denoised)

    def enabled(self) -> bool:
        return self.scale != 0.0


@dataclass(frozen=True)
class LtxAPGGuider:
    scale: float
    eta: float = 1.0
    norm_threshold: float = 0.0

    def delta(self, cond, uncond):
        g = cond - uncond
        if self.norm_threshold > 0:
            nrm = _l2_norm(g, (-1, -2, -3), keepdim=True)
            g = g * torch.minimum(torch.ones_like(g), self.norm_threshold / nrm)
        coef = projection_coef(g, cond).reshape(-1, *([1] * (cond.dim() - 1)))
        par = coef * cond
        return (par * self.eta + (g - par)) * (self.scale - 1)

    def enabled(self) -> bool:
        return self.scale != 1.0


@dataclass
class GaussianNoiser:
    """noisers.py: N(0,1) of the latent's shape/dtype (torch generator; MLX seeds are not reproducible)."""
    seed: Optional[int] = None

    def noise(self, latents: torch.Tensor) -> torch.Tensor:
        g = None
        if self.seed is not None:
            g = torch.Generator(device=latents.device).manual_seed(self.seed)
        return torch.randn(latents.shape, generator=g, device=latents.device, dtype=torch.float32).to(latents.dtype)


class LinearQuadraticScheduler:
    """schedulers.py:58-79."""

    def execute(self, steps: int, threshold_noise: float = 0.025, linear_steps: Optional[int] = None, **_kw) -> torch.Tensor:
        if steps == 1:
            return torch.tensor([1.0, 0.0], dtype=torch.float32)
        if linear_steps is None:
            linear_steps = steps // 2
        lin = [i * threshold_noise / linear_steps for i in range(linear_steps)]
        diff = linear_steps - threshold_noise * steps
        qsteps = steps - linear_steps
        quad = []
        if qsteps > 0:
            qc = diff / (linear_steps * qsteps ** 2)
            lc = threshold_noise / linear_steps - 2 * diff / (qsteps ** 2)
            const = qc * (linear_steps ** 2)
            quad = [qc * (i ** 2) + lc * i + const for i in range(linear_steps, steps)]
        return torch.tensor([1.0 - x for x in lin + quad + [1.0]], dtype=torch.float32)


def flux_time_shift(mu: float, sigma: float, t: float) -> float:
    return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)


@lru_cache(maxsize=5)
def _model_sampling_sigmas(shift: float, n: int) -> np.ndarray:
    return np.array([flux_time_shift(shift, 1.0, t) for t in np.arange(1, n + 1) / n])


class BetaScheduler:
    """schedulers.py:82-97."""
    shift = 2.37
    timesteps_length = 10000

    def execute(self, steps: int, alpha: float = 0.6, beta: float = 0.6) -> torch.Tensor:
        import scipy.stats
        ms = _model_sampling_sigmas(self.shift, self.timesteps_length)
        total = len(ms) - 1
        ts = 1 - np.linspace(0, 1, steps, endpoint=False)
        ts = list(dict.fromkeys(np.rint(scipy.stats.beta.ppf(ts, alpha, beta) * total).tolist()))
        return torch.tensor([float(ms[int(t)]) for t in ts] + [0.0], dtype=torch.float32)


class VideoLatentPatchifier:
    """patchifiers.py:12-60 (pure index maps)."""

    def __init__(self, patch_size: int = 1):
        self._patch_size = (1, patch_size, patch_size)

    @property
    def patch_size(self) -> Tuple[int, int, int]:
        return self._patch_size

    def get_token_count(self, shape) -> int:
        return math.prod(tuple(shape)[2:]) // math.prod(self._patch_size)

    def patchify(self, latents: torch.Tensor) -> torch.Tensor:
        b, c, f, h, w = latents.shape
        p1, p2, p3 = self._patch_size
        if f % p1 or h % p2 or w % p3:
            raise ValueError("Latents not divisible by patch size")
        x = latents.reshape(b, c, f // p1, p1, h // p2, p2, w // p3, p3).permute(0, 2, 4, 6, 1, 3, 5, 7)
        return x.reshape(b, (f // p1) * (h // p2) * (w // p3), c * p1 * p2 * p3)

    def unpatchify(self, tokens: torch.Tensor, output_shape) -> torch.Tensor:
        b, c, fr, hh, ww = output_shape
        p1, p2, p3 = self._patch_size
        if p1 != 1:
            raise ValueError("Temporal patch size must be 1 for symmetric patchifier")
        x = tokens.reshape(b, fr // p1, hh // p2, ww // p3, c, p1, p2, p3).permute(0, 4, 1, 5, 2, 6, 3, 7)
        return x.reshape(b, c, fr, hh, ww)

"""CPU restatement of the sigma schedules and latent conditioning (host-side math).
TEST INFRASTRUCTURE — see oracle/__init__.py."""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

# generate.py:339-340
STAGE_1_SIGMAS = [1.0, 0.99375, 0.9875, 0.98125, 0.975, 0.909375, 0.725, 0.421875, 0.0]
STAGE_2_SIGMAS = [0.909375, 0.725, 0.421875, 0.0]
BASE_SHIFT_ANCHOR = 1024   # generate.py:343
MAX_SHIFT_ANCHOR = 4096    # generate.py:344


def ltx2_scheduler(steps: int, num_tokens: Optional[int] = None, max_shift: float = 2.05,
                   base_shift: float = 0.95, stretch: bool = True, terminal: float = 0.1) -> np.ndarray:
    """generate.py:410-467 (float64 numpy, cast to float32 at the end)."""
    tokens = MAX_SHIFT_ANCHOR if num_tokens is None else min(num_tokens, MAX_SHIFT_ANCHOR)
    sig = np.linspace(1.0, 0.0, steps + 1)
    mm = (max_shift - base_shift) / (MAX_SHIFT_ANCHOR - BASE_SHIFT_ANCHOR)
    b = base_shift - mm * BASE_SHIFT_ANCHOR
    shift = tokens * mm + b
    out = np.zeros_like(sig)
    nz = sig != 0
    out[nz] = math.exp(shift) / (math.exp(shift) + (1.0 / sig[nz] - 1.0))
    sig = out
    if stretch:
        nzm = sig != 0
        omz = 1.0 - sig[nzm]
        sf = omz[-1] / (1.0 - terminal)
        if np.isfinite(sf) and sf != 0:
            sig[nzm] = 1.0 - omz / sf
    return sig.astype(np.float32)


def subsample_sigmas_farthest(sigmas: List[float], steps: int) -> List[float]:
    """generate.py:182-224: farthest-point sampling in log-sigma."""
    if steps < 1:
        raise ValueError("steps must be >= 1")
    if steps >= len(sigmas) - 1:
        return sigmas
    if steps == 1:
        return [sigmas[0], sigmas[-1]]
    pool = sigmas[:-1]
    xs = [math.log(max(s, 1e-6)) for s in pool]
    chosen = {0, len(pool) - 1}
    while len(chosen) < steps:
        best_i, best = None, -1.0
        for i in range(len(pool)):
            if i in chosen:
                continue
            sc = min(abs(xs[i] - xs[j]) for j in chosen)
            if sc > best:
                best, best_i = sc, i
        chosen.add(best_i)
    return [sigmas[i] for i in sorted(chosen)] + [sigmas[-1]]


def subsample_sigmas_uniform(sigmas: List[float], steps: int) -> List[float]:
    """generate.py:227-258."""
    if steps < 1:
        raise ValueError("steps must be >= 1")
    if steps >= len(sigmas) - 1:
        return sigmas
    if steps == 1:
        return [sigmas[0], sigmas[-1]]
    pool = sigmas[:-1]
    last = len(pool) - 1
    idxs = [0] + [int(round(i * last / (steps - 1))) for i in range(1, steps - 1)] + [last]
    uniq = sorted(set(idxs))
    if len(uniq) < steps:
        for i in range(last + 1):
            if i in uniq:
                continue
            uniq.append(i)
            if len(uniq) == steps:
                break
        uniq = sorted(uniq)
    return [pool[i] for i in uniq] + [sigmas[-1]]


def subsample_refinement_sigmas(sigmas: List[float], steps: int, method: str) -> List[float]:
    """generate.py:268-277."""
    if steps == 1 and method == "farthest" and len(sigmas) >= 3:
        return [sigmas[-2], sigmas[-1]]
    return subsample_sigmas_farthest(sigmas, steps) if method == "farthest" else subsample_sigmas_uniform(sigmas, steps)


def apply_conditioning(latent: torch.Tensor, clean: torch.Tensor, mask: torch.Tensor,
                       items: Sequence[Tuple[str, torch.Tensor, int, float]]):
    """conditioning/latent.py:104-177.  items: (mode in {"replace","guide"}, cond (B,C,f,H,W),
    frame_idx, strength).  Returns (latent, clean, mask)."""
    latent, clean, mask = latent.clone(), clean.clone(), mask.clone()
    b, c, f, h, w = latent.shape
    for mode, cond, frame_idx, strength in items:
        _, cc, cf, ch, cw = cond.shape
        if (cc, ch, cw) != (c, h, w):
            raise ValueError("conditioning latent shape mismatch")
        if frame_idx >= f:
            raise ValueError("frame index out of bounds")
        end = min(frame_idx + cf, f)
        for i in range(frame_idx, end):
            if mode == "replace":
                latent[:, :, i] = cond[:, :, i - frame_idx]
            clean[:, :, i] = cond[:, :, i - frame_idx]
            mask[:, :, i] = 1.0 - strength
    return latent, clean, mask


def noise_blend(noise: torch.Tensor, latent: torch.Tensor, mask: torch.Tensor, sigma0: float) -> torch.Tensor:
    """generate.py:3153-3160,3442-3449: n*(m*s0) + x*(1 - m*s0)."""
    eff = sigma0 * mask
    return noise * eff + latent * (1.0 - eff)

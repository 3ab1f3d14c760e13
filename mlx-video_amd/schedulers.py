"""Sigma schedules and the RoPE position grid: host-side integer / float64 math that the
reference also runs on the host (numpy).  mlx_video/generate.py:182-277,339-344,410-525;
ltx_core/components/schedulers.py:16-55."""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np
import torch

STAGE_1_SIGMAS = [1.0, 0.99375, 0.9875, 0.98125, 0.975, 0.909375, 0.725, 0.421875, 0.0]
STAGE_2_SIGMAS = [0.909375, 0.725, 0.421875, 0.0]
BASE_SHIFT_ANCHOR = 1024
MAX_SHIFT_ANCHOR = 4096


def ltx2_scheduler(steps: int, num_tokens: Optional[int] = None, max_shift: float = 2.05,
                   base_shift: float = 0.95, stretch: bool = True, terminal: float = 0.1) -> torch.Tensor:
    """Token-count-shifted, terminal-stretched schedule (generate.py:410-467). (steps+1,) fp32."""
    tokens = MAX_SHIFT_ANCHOR if num_tokens is None else min(int(num_tokens), MAX_SHIFT_ANCHOR)
    lin = np.linspace(1.0, 0.0, steps + 1)
    slope = (max_shift - base_shift) / (MAX_SHIFT_ANCHOR - BASE_SHIFT_ANCHOR)
    shift = tokens * slope + (base_shift - slope * BASE_SHIFT_ANCHOR)
    e = math.exp(shift)
    sig = np.where(lin != 0, e / (e + (1.0 / np.where(lin != 0, lin, 1.0) - 1.0)), 0.0)
    if stretch:
        live = sig != 0
        gap = 1.0 - sig[live]
        factor = gap[-1] / (1.0 - terminal)
        if np.isfinite(factor) and factor != 0:
            sig[live] = 1.0 - gap / factor
    return torch.from_numpy(sig.astype(np.float32))


class LTX2Scheduler:
    """SchedulerProtocol twin (ltx_core/components/schedulers.py:16-55): tokens from latent.shape[2:]."""

    def execute(self, steps: int, latent: Optional[torch.Tensor] = None, max_shift: float = 2.05,
                base_shift: float = 0.95, stretch: bool = True, terminal: float = 0.1, **_kw) -> torch.Tensor:
        tokens = int(np.prod(latent.shape[2:])) if latent is not None else MAX_SHIFT_ANCHOR
        return ltx2_scheduler(steps, tokens, max_shift, base_shift, stretch, terminal)


def _check_steps(steps: int) -> None:
    if steps < 1:
        raise ValueError("steps must be >= 1")


def _subsample_sigmas_farthest(sigmas: List[float], steps: int) -> List[float]:
    """Farthest-point subset in log-sigma, endpoints kept (generate.py:182-224)."""
    _check_steps(steps)
    if steps >= len(sigmas) - 1:
        return sigmas
    if steps == 1:
        return [sigmas[0], sigmas[-1]]
    body = sigmas[:-1]
    logs = [math.log(max(s, 1e-6)) for s in body]
    picked = {0, len(body) - 1}
    while len(picked) < steps:
        cand = [(min(abs(logs[i] - logs[j]) for j in picked), -i) for i in range(len(body)) if i not in picked]
        picked.add(-max(cand)[1])       # largest distance, lowest index on ties
    return [sigmas[i] for i in sorted(picked)] + [sigmas[-1]]


def _subsample_sigmas_uniform(sigmas: List[float], steps: int) -> List[float]:
    """generate.py:227-258."""
    _check_steps(steps)
    if steps >= len(sigmas) - 1:
        return sigmas
    if steps == 1:
        return [sigmas[0], sigmas[-1]]
    body = sigmas[:-1]
    last = len(body) - 1
    want = sorted({0, last, *[int(round(i * last / (steps - 1))) for i in range(1, steps - 1)]})
    for i in range(last + 1):
        if len(want) >= steps:
            break
        if i not in want:
            want.append(i)
    want = sorted(want)
    return [body[i] for i in want] + [sigmas[-1]]


def _subsample_sigmas(sigmas: List[float], steps: int, method: str) -> List[float]:
    if method == "uniform":
        return _subsample_sigmas_uniform(sigmas, steps)
    if method == "farthest":
        return _subsample_sigmas_farthest(sigmas, steps)
    raise ValueError(f"Unknown sigma subsample method: {method}")


def _subsample_refinement_sigmas(sigmas: List[float], steps: int, method: str) -> List[float]:
    """generate.py:261-277: a single refinement step starts from the last non-zero sigma."""
    if steps == 1 and method == "farthest" and len(sigmas) >= 3:
        return [sigmas[-2], sigmas[-1]]
    return _subsample_sigmas(sigmas, steps, method)


def create_position_grid(batch_size: int, num_frames: int, height: int, width: int, temporal_scale: int = 8,
                         spatial_scale: int = 32, fps: float = 24.0, causal_fix: bool = True) -> torch.Tensor:
    """Per-token [start,end) pixel coordinates, (B,3,N,2) fp32, token n=(f*H+h)*W+w
    (generate.py:470-525).  Integer index math + one fp32 divide; bit-exact contract."""
    n = num_frames * height * width
    idx = np.arange(n, dtype=np.int64)
    f, rem = np.divmod(idx, height * width)
    h, w = np.divmod(rem, width)
    start = np.stack([f * temporal_scale, h * spatial_scale, w * spatial_scale], 0)       # (3,N)
    end = np.stack([(f + 1) * temporal_scale, (h + 1) * spatial_scale, (w + 1) * spatial_scale], 0)
    grid = np.stack([start, end], -1).astype(np.float32)                                  # (3,N,2)
    if causal_fix:
        grid[0] = np.maximum(grid[0] + np.float32(1 - temporal_scale), np.float32(0))
    grid[0] = grid[0] / np.float32(fps)
    return torch.from_numpy(np.broadcast_to(grid[None], (batch_size, 3, n, 2)).copy())


def cfg_delta(cond: torch.Tensor, uncond: torch.Tensor, scale: float) -> torch.Tensor:
    """generate.py:382-393 / guiders.py CFGGuider.delta (hook contract; host tensors)."""
    return (scale - 1.0) * (cond - uncond)

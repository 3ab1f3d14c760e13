"""Hypothesis fuzzing of the host-side index maps against the oracle, plus the size-independent properties the
reference's own tests state (test_generate_dev.py:21-142, test_vae_streaming.py:159-263): grid bounds and causal
fix, schedule monotonicity / terminal value, tile coverage and blend-weight positivity, subsampling endpoints."""
import math

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from oracle import dit as O
from oracle import sched as OS
from oracle import vae as OV

from mlx_video_amd import schedulers as S
from mlx_video_amd import video_vae as V

SET = dict(max_examples=60, deadline=None)


@settings(**SET)
@given(b=st.integers(1, 3), f=st.integers(1, 13), h=st.integers(1, 24), w=st.integers(1, 24),
       fps=st.sampled_from([8.0, 24.0, 25.0, 30.0, 23.976]))
def test_position_grid_matches_oracle_and_bounds(b, f, h, w, fps):
    got = S.create_position_grid(b, f, h, w, fps=fps)
    ref = torch.from_numpy(O.create_position_grid(b, f, h, w, fps=fps))
    assert got.dtype == torch.float32 and tuple(got.shape) == (b, 3, f * h * w, 2)
    assert torch.equal(got, ref)                                        # bit-exact index math
    # token order n = (f*H + h)*W + w; spatial cells are 32 px; first frame starts at 0 and never goes negative
    n = (f - 1) * h * w + (h - 1) * w + (w - 1)
    assert float(got[0, 1, n, 0]) == (h - 1) * 32 and float(got[0, 2, n, 1]) == w * 32
    assert float(got[:, 0].min()) == 0.0
    assert bool((got[..., 1] > got[..., 0]).all() or f == 1)            # [start,end) non-empty (frame 0 is [0, 1/fps))
    t_end = got[0, 0, n, 1] * fps
    assert abs(float(t_end) - max(0, f * 8 + 1 - 8)) < 1e-3             # causal fix on the end bound


@settings(**SET)
@given(steps=st.integers(1, 60), tokens=st.one_of(st.none(), st.integers(1, 9000)), stretch=st.booleans())
def test_scheduler_matches_oracle_and_invariants(steps, tokens, stretch):
    got = S.ltx2_scheduler(steps, tokens, stretch=stretch)
    ref = OS.ltx2_scheduler(steps, tokens, stretch=stretch)
    g = np.asarray(got, dtype=np.float64)
    assert g.shape == (steps + 1,)
    np.testing.assert_allclose(g, np.asarray(ref, dtype=np.float64), rtol=0, atol=1e-6)
    assert abs(g[0] - 1.0) < 1e-6 and g[-1] == 0.0                      # test_generate_dev.py:21-40, 59-64 (no stretch)
    assert np.all(np.diff(g) < 0)                                       # strictly decreasing
    if steps > 1 and stretch:
        assert abs(g[-2] - 0.1) < 1e-5                                  # stretched so the last non-zero sigma is 0.1


@settings(**SET)
@given(n=st.integers(3, 40), steps=st.integers(1, 12), method=st.sampled_from(["farthest", "uniform"]))
def test_sigma_subsampling(n, steps, method):
    sig = [float(x) for x in np.linspace(1.0, 0.0, n + 1)[:-1] ** 2 + 1e-3] + [0.0]
    if steps > len(sig) - 1:
        steps = len(sig) - 1
    fn_h = S._subsample_sigmas_farthest if method == "farthest" else S._subsample_sigmas_uniform
    fn_o = OS.subsample_sigmas_farthest if method == "farthest" else OS.subsample_sigmas_uniform
    got, ref = fn_h(sig, steps), fn_o(sig, steps)
    assert got == ref
    assert len(got) == steps + 1 and got[0] == sig[0] and got[-1] == 0.0
    assert all(a > b for a, b in zip(got, got[1:]))
    assert set(got) <= set(sig)


@settings(**SET)
@given(size=st.integers(2, 40), overlap=st.integers(0, 19), dim=st.integers(1, 200), temporal=st.booleans())
def test_tile_intervals_cover_and_match_oracle(size, overlap, dim, temporal):
    if overlap * 2 >= size:
        overlap = (size - 1) // 2
    iv = (V.split_in_temporal if temporal else V.split_in_spatial)(size, overlap, dim)
    ref = (OV.split_temporal if temporal else OV.split_spatial)(size, overlap, dim)
    assert (iv.starts, iv.ends, iv.left_ramps, iv.right_ramps) == tuple(ref)
    assert iv.starts[0] == 0 and iv.ends[-1] == dim                     # test_vae_streaming.py:159-197 coverage
    covered = np.zeros(dim, bool)
    for s, e in zip(iv.starts, iv.ends):
        assert 0 <= s < e <= dim
        covered[s:e] = True
    assert covered.all()
    assert iv.left_ramps[0] == 0 and iv.right_ramps[-1] == 0


@settings(**SET)
@given(length=st.integers(1, 80), rl=st.integers(0, 90), rr=st.integers(0, 90), from0=st.booleans())
def test_trapezoid_mask(length, rl, rr, from0):
    got = V.compute_trapezoidal_mask_1d(length, rl, rr, from0)
    ref = OV.trapezoid_mask(length, rl, rr, from0)
    assert torch.equal(got.float().cpu(), ref)
    assert float(got.min()) >= 0.0 and float(got.max()) <= 1.0          # test_vae_streaming.py:223-263
    if rl == 0 and rr == 0:
        assert bool((got == 1).all())
    if not from0 and rl + rr < length:
        assert float(got.min()) > 0.0                                   # interior tiles never weight a sample by 0


@settings(max_examples=25, deadline=None)
@given(size=st.integers(4, 12), overlap=st.integers(1, 3), dim=st.integers(5, 40), scale=st.sampled_from([8, 32]))
def test_blend_weights_sum_positive(size, overlap, dim, scale):
    """Every output sample of a tiled decode has a positive total blend weight (so the normalisation in
    ltxk_tile_blend_finalize never divides by ~0) for the spatial mapping."""
    iv = V.split_in_spatial(size, overlap, dim)
    total = torch.zeros(dim * scale)
    for s, e, l, r in zip(iv.starts, iv.ends, iv.left_ramps, iv.right_ramps):
        sl, m = V.map_spatial_slice(s, e, l, r, scale)
        total[sl] += m.float().cpu()
    assert float(total.min()) > 0.0


@settings(max_examples=25, deadline=None)
@given(size=st.integers(4, 12), overlap=st.integers(1, 3), dim=st.integers(5, 40))
def test_temporal_blend_weights_sum_positive(size, overlap, dim):
    iv = V.split_in_temporal(size, overlap, dim)
    out_f = 1 + (dim - 1) * 8
    total = torch.zeros(out_f)
    for s, e, l, r in zip(iv.starts, iv.ends, iv.left_ramps, iv.right_ramps):
        sl, m = V.map_temporal_slice(s, e, l, r, 8)
        assert sl.stop <= out_f and m.numel() == sl.stop - sl.start
        total[sl] += m.float().cpu()
    assert float(total[1:].min()) > 0.0 and float(total[0]) > 0.0


@settings(**SET)
@given(b=st.integers(1, 2), c=st.sampled_from([3, 8]), f=st.integers(1, 4), h=st.integers(1, 5), w=st.integers(1, 5))
def test_patchify_roundtrip_oracle(b, c, f, h, w):
    x = torch.arange(b * c * f * h * 4 * w * 4, dtype=torch.float32).reshape(b, c, f, h * 4, w * 4)
    y = OV.patchify(x, 4)
    assert tuple(y.shape) == (b, c * 16, f, h, w)
    assert torch.equal(OV.unpatchify(y, 4), x)
    # channel order (c, p_w, p_h): ops.py:9-44
    assert float(y[0, 1, 0, 0, 0]) == float(x[0, 0, 0, 1, 0]) and float(y[0, 4, 0, 0, 0]) == float(x[0, 0, 0, 0, 1])

"""Pin the CPU oracle and the product's host logic to the closed-form known answers that the
reference's own tests hold: tests/test_generate_dev.py:21-70 (scheduler),
76-142 (position grid), 148-190 (cfg), 293-332 (latent dims); tests/test_rope.py:246-276
(SPLIT rope shape); tests/test_lora.py:8-29.  CPU only."""
import math

import numpy as np
import pytest
import torch

from oracle import dit as O
from oracle import sched as S
from mlx_video_amd import schedulers as P


@pytest.mark.parametrize("impl", [S.ltx2_scheduler, lambda **k: P.ltx2_scheduler(**k).numpy()])
def test_scheduler_invariants(impl):
    # test_generate_dev.py:21-70
    for steps in [5, 10, 20, 40, 50]:
        s = impl(steps=steps)
        assert s.shape == (steps + 1,) and s.dtype == np.float32
        assert abs(float(s[0]) - 1.0) < 1e-5 and abs(float(s[-1])) < 1e-5
        assert all(s[i] >= s[i + 1] for i in range(steps))
    assert abs(float(impl(steps=20)[-2]) - 0.1) < 1e-6            # stretched to terminal 0.1
    ns = impl(steps=20, stretch=False)
    assert ns.shape == (21,) and ns[0] > 0 and ns[-1] == 0.0
    assert impl(steps=20, num_tokens=1920).shape == (21,)


def test_scheduler_product_equals_oracle():
    for steps, tok in [(40, 1280), (8, 32), (20, None), (30, 5184), (40, 3328)]:
        assert np.array_equal(S.ltx2_scheduler(steps, tok), P.ltx2_scheduler(steps, tok).numpy())


def test_scheduler_shift_formula():
    # generate.py:410-467: sigma' = e^s/(e^s + 1/sigma - 1), s = tokens*m + b, before the stretch
    s = S.ltx2_scheduler(4, 1024, stretch=False)
    sh = 0.95
    exp = [math.exp(sh) / (math.exp(sh) + (1 / x - 1)) for x in (1.0, 0.75, 0.5, 0.25)]
    assert np.allclose(s[:4], np.array(exp, dtype=np.float32), atol=1e-7)


@pytest.mark.parametrize("mod", [S, P])
def test_subsample_sigmas(mod):
    far = getattr(mod, "subsample_sigmas_farthest", None) or mod._subsample_sigmas_farthest
    uni = getattr(mod, "subsample_sigmas_uniform", None) or mod._subsample_sigmas_uniform
    ref = getattr(mod, "subsample_refinement_sigmas", None) or mod._subsample_refinement_sigmas
    s1 = list(S.STAGE_1_SIGMAS)
    assert far(s1, 8) == s1 and uni(s1, 20) == s1
    assert far(s1, 1) == [1.0, 0.0]
    for k in range(2, 8):
        for fn in (far, uni):
            out = fn(s1, k)
            assert len(out) == k + 1 and out[0] == 1.0 and out[-1] == 0.0
            assert all(out[i] > out[i + 1] for i in range(k))
    # farthest-point in log-sigma keeps the low-sigma points (generate.py:186-192)
    assert far(s1, 3) == [1.0, 0.725, 0.421875, 0.0]
    assert ref(list(S.STAGE_2_SIGMAS), 1, "farthest") == [0.421875, 0.0]
    with pytest.raises(ValueError):
        far(s1, 0)


def test_subsample_product_equals_oracle():
    s1 = list(S.STAGE_1_SIGMAS)
    for k in range(1, 10):
        assert S.subsample_sigmas_farthest(s1, k) == P._subsample_sigmas_farthest(s1, k)
        assert S.subsample_sigmas_uniform(s1, k) == P._subsample_sigmas_uniform(s1, k)


@pytest.mark.parametrize("impl", [O.create_position_grid, lambda *a, **k: P.create_position_grid(*a, **k).numpy()])
def test_position_grid(impl):
    # test_generate_dev.py:76-142
    g = impl(1, 5, 16, 24)
    assert g.shape == (1, 3, 5 * 16 * 24, 2) and g.dtype == np.float32
    for b in (1, 2, 4):
        assert impl(b, 5, 16, 24).shape[0] == b
    assert g[0, 0].max() < 10 and g[0, 1].max() <= 512 and g[0, 2].max() <= 768
    assert np.abs(g - impl(1, 5, 16, 24, causal_fix=False)).max() > 0
    assert np.isfinite(g).all()
    # closed form: token n=(f*H+h)*W+w; t bounds max(0, 8f+1-8)/24 .. max(0, 8(f+1)+1-8)/24
    n = (2 * 16 + 3) * 24 + 5
    assert g[0, 0, n, 0] == np.float32(9.0) / np.float32(24.0) and g[0, 0, n, 1] == np.float32(17.0) / np.float32(24.0)
    assert tuple(g[0, 1, n]) == (96.0, 128.0) and tuple(g[0, 2, n]) == (160.0, 192.0)
    assert g[0, 0, 0, 0] == 0.0 and g[0, 0, 0, 1] == np.float32(1.0) / np.float32(24.0)


def test_position_grid_product_equals_oracle():
    for a in [(1, 2, 4, 4), (2, 5, 16, 16), (1, 13, 16, 16), (1, 9, 24, 24)]:
        assert np.array_equal(O.create_position_grid(*a), P.create_position_grid(*a).numpy())


def test_cfg_formula():
    # test_generate_dev.py:148-190, literal case
    cond, uncond = torch.tensor([[[1.0, 2.0, 3.0]]]), torch.tensor([[[0.5, 1.0, 1.5]]])
    for fn in (O.cfg_delta, P.cfg_delta):
        assert torch.equal(fn(cond, uncond, 4.0), torch.tensor([[[1.5, 3.0, 4.5]]]))
        assert float(fn(cond, uncond, 1.0).abs().max()) == 0.0
        assert fn(cond.bfloat16(), uncond.bfloat16(), 4.0).dtype == torch.bfloat16
    v = O.cfg_combine(cond, uncond, 4.0, O.F32)
    assert torch.equal(v, cond + O.cfg_delta(cond, uncond, 4.0))


def test_latent_dims_and_tokens():
    # test_generate_dev.py:293-332
    for frames, lf in [(1, 1), (9, 2), (17, 3), (33, 5), (65, 9)]:
        assert 1 + (frames - 1) // 8 == lf
    assert (1 + (33 - 1) // 8) * (512 // 32) * (768 // 32) == 1920
    lat = torch.arange(2 * 128 * 5 * 2 * 3, dtype=torch.float32).reshape(2, 128, 5, 2, 3)
    tok = O.latent_to_tokens(lat)
    assert tok.shape == (2, 30, 128)
    assert tok[1, (3 * 2 + 1) * 3 + 2, 77] == lat[1, 77, 3, 1, 2]
    assert torch.equal(O.tokens_to_latent(tok, lat.shape), lat)


def test_split_rope_shape_and_pad():
    # test_rope.py:246-276: dim=128, H=32 -> (B,H,T,2); front pad cos=1/sin=0 (rope.py:504-509)
    pos = torch.from_numpy(O.create_position_grid(1, 4, 4, 4))
    cos, sin = O.precompute_freqs_cis(pos, 128, heads=32)
    assert cos.shape == (1, 32, 64, 2) and sin.shape == (1, 32, 64, 2) and cos.dtype == torch.float32
    assert cos.abs().max() <= 1.0 and sin.abs().max() <= 1.0 and torch.isfinite(cos).all()
    cos, sin = O.precompute_freqs_cis(pos, 4096)
    assert cos.shape == (1, 32, 64, 64)
    assert torch.equal(cos[0, 0, :, :2], torch.ones(64, 2)) and torch.equal(sin[0, 0, :, :2], torch.zeros(64, 2))
    # frequency f=2 is index 0, dim 0 (time): angle = (mid_t/20*2-1)*pi/2
    mid_t = (pos[0, 0, :, 0] + pos[0, 0, :, 1]) / 2
    assert torch.allclose(cos[0, 0, :, 2], torch.cos((mid_t / 20 * 2 - 1) * (math.pi / 2)), atol=1e-6)


def test_rope_is_rotation():
    g = torch.Generator().manual_seed(0)
    pos = torch.from_numpy(O.create_position_grid(1, 2, 3, 4))
    cos, sin = O.precompute_freqs_cis(pos, 512, heads=4)
    x = torch.randn(1, 24, 512, generator=g)
    y = O.apply_split_rotary_emb(x, cos, sin, O.F64)
    assert torch.allclose(y.norm(dim=-1), x.double().norm(dim=-1), rtol=1e-6)       # norm preserving
    back = O.apply_split_rotary_emb(y, cos, -sin, O.F64)
    assert torch.allclose(back, x.double(), atol=1e-6)


def test_timestep_embedding_known_values():
    e = O.get_timestep_embedding(torch.tensor([0.0, 1000.0]))
    assert e.shape == (2, 256)
    assert torch.equal(e[0, :128], torch.ones(128)) and torch.equal(e[0, 128:], torch.zeros(128))   # cos first
    assert abs(float(e[1, 0]) - math.cos(1000.0)) < 1e-4 and abs(float(e[1, 128]) - math.sin(1000.0)) < 1e-4


def test_euler_and_denoise_algebra():
    x, v = torch.tensor([1.0, -2.0]), torch.tensor([0.5, 0.25])
    x0 = O.to_denoised(x, v, 0.5, O.F32)
    assert torch.equal(x0, torch.tensor([0.75, -2.125]))
    assert torch.equal(O.euler_step(x, x0, 0.5, 0.25, O.F32), x0 + 0.25 * (x - x0) / 0.5)
    assert torch.equal(O.euler_step(x, x0, 0.5, 0.0, O.F32), x0)
    m = torch.tensor([1.0, 0.0])
    assert torch.equal(O.apply_denoise_mask(x0, torch.tensor([9.0, 9.0]), m, O.F32), torch.tensor([0.75, 9.0]))


def test_lora_known_answer():
    # test_lora.py:8-29: I + 0.5*(1_{4x2} . 1_{2x4}) = I + 1
    w = torch.eye(4)
    merged = w + 0.5 * (torch.ones(4, 2) @ torch.ones(2, 4))
    assert torch.equal(merged, torch.eye(4) + 1.0)


def test_bf16_policy_close_to_fp32_on_tiny_model():
    cfg = O.DiTConfig(num_layers=1, heads=4, caption_channels=64)
    W = O.make_weights(cfg, seed=3)
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(1, 8, 128, generator=g)
    ctx = torch.randn(1, 16, 64, generator=g)
    ts = torch.full((1, 8), 0.5)
    pos = torch.from_numpy(O.create_position_grid(1, 2, 2, 2))
    pe = O.precompute_freqs_cis(pos, cfg.dim, heads=4)
    a = O.ltx_forward(lat, ts, ctx, pe, W, cfg, O.F32)
    b = O.ltx_forward(lat, ts, ctx, pe, W, cfg, O.BF16)
    assert a.shape == (1, 8, 128)
    assert float((a - b).norm() / a.norm()) < 3e-2


def test_area_resize_restatement_known_answers():
    """oracle/media.py (OpenCV INTER_AREA, utils.py:699-705): weights of every destination pixel sum to 1; integer factors
    are the plain block mean; a constant image stays constant; the 3 -> 2 table is the textbook [1, .5 | .5, 1] / 1.5."""
    import numpy as np
    from oracle import media as OM
    for ss, ds in ((12, 8), (768, 384), (50, 33), (7, 7), (100, 1)):
        A = OM.area_weights(ss, ds)
        assert np.allclose(A.sum(1), 1.0, atol=1e-6) and (A >= 0).all()
    assert np.allclose(OM.area_weights(3, 2), np.array([[2 / 3, 1 / 3, 0], [0, 1 / 3, 2 / 3]], np.float32), atol=1e-7)
    x = np.random.default_rng(0).random((2, 3, 8, 12), dtype=np.float32)
    assert np.allclose(OM.resize_area(x, 4, 6), x.reshape(2, 3, 4, 2, 6, 2).mean((3, 5)), atol=1e-6)
    assert np.allclose(OM.resize_area(np.full((5, 9), 0.37, np.float32), 3, 4), 0.37, atol=1e-6)

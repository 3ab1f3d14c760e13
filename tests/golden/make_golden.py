#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  The reference (pure Python on Apple MLX) cannot be imported in the
build container (ModuleNotFoundError: mlx), so these vectors come from the repository's
CPU oracle (oracle/, fp32 policy "F32" and bf16-storage policy "BF16") on seeded inputs — they pin the
oracle against silent drift and give the GPU tests a fixture that does not depend on re-running the
oracle.  Closed-form known answers from the reference's own tests are in tests/test_oracle_kat.py.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import dit as O      # noqa: E402
from oracle import sched as S    # noqa: E402
from oracle import vae as OV     # noqa: E402


def f32(t):
    return t.detach().to(torch.float32).numpy()


def main():
    # ---- schedules + position grid + rope table ----
    np.savez_compressed(os.path.join(HERE, "schedules.npz"),
                        dev40_n1280=S.ltx2_scheduler(40, 1280), dev8_n32=S.ltx2_scheduler(8, 32),
                        dev30_n5184=S.ltx2_scheduler(30, 5184), nostretch20=S.ltx2_scheduler(20, None, stretch=False),
                        far5=np.array(S.subsample_sigmas_farthest(list(S.STAGE_1_SIGMAS), 5)),
                        uni5=np.array(S.subsample_sigmas_uniform(list(S.STAGE_1_SIGMAS), 5)),
                        grid_2x4x4=O.create_position_grid(1, 2, 4, 4), grid_5x16x16_tail=O.create_position_grid(1, 5, 16, 16)[0, :, -8:, :])
    pos = torch.from_numpy(O.create_position_grid(1, 2, 3, 4))
    cos, sin = O.precompute_freqs_cis(pos, 4096)
    np.savez_compressed(os.path.join(HERE, "rope_2x3x4.npz"), cos=f32(cos[0, ::8, :, ::8]), sin=f32(sin[0, ::8, :, ::8]))

    # ---- DiT: one full-width-free small model forward + one CFG step ----
    cfg = O.DiTConfig(num_layers=2, heads=4, caption_channels=256)
    W = O.make_weights(cfg, seed=2024)
    g = torch.Generator().manual_seed(5)
    B, F, H, Wd, Sx = 1, 2, 3, 4, 48
    N = F * H * Wd
    lat = torch.randn(B, N, 128, generator=g).to(torch.bfloat16)
    ctx = torch.randn(B, Sx, 256, generator=g).to(torch.bfloat16)
    ts = torch.full((B, N), 0.725).to(torch.bfloat16)
    ts[:, :H * Wd] = 0.0
    posd = torch.from_numpy(O.create_position_grid(B, F, H, Wd))
    pe = O.precompute_freqs_cis(posd, cfg.dim, heads=cfg.heads)
    v_bf = O.ltx_forward(lat.float(), ts.float(), ctx.float(), pe, W, cfg, O.BF16)
    v_f32 = O.ltx_forward(lat.float(), ts.float(), ctx.float(), pe, W, cfg, O.F32)
    np.savez_compressed(os.path.join(HERE, "dit_forward_small.npz"), seed_weights=2024, latent=f32(lat), context=f32(ctx),
                        timesteps=f32(ts), velocity_bf16_policy=f32(v_bf), velocity_fp32_policy=f32(v_f32))

    # ---- step algebra ----
    vp = torch.randn(1, N, 128, generator=g).to(torch.bfloat16)
    vn = torch.randn(1, N, 128, generator=g).to(torch.bfloat16)
    x = torch.randn(1, 128, F, H, Wd, generator=g).to(torch.bfloat16)
    p = O.BF16
    s, sn = O.bf16_round_scalar(0.909375), O.bf16_round_scalar(0.725)
    v = O.cfg_combine(vp.float(), vn.float(), 4.0, p)
    x0 = O.to_denoised(x.float(), O.tokens_to_latent(v, x.shape), s, p)
    nxt = p.r(x0 + sn * (x.float() - x0) / s)
    np.savez_compressed(os.path.join(HERE, "step_algebra.npz"), v_pos=f32(vp), v_neg=f32(vn), latent=f32(x), sigma=s, sigma_next=sn,
                        cfg=4.0, x0=f32(x0), next_latent=f32(nxt))

    # ---- VAE: one conv per halo mode, tiny decode ----
    gx = torch.Generator().manual_seed(6)
    xin = torch.randn(1, 64, 3, 4, 5, generator=gx).to(torch.bfloat16)
    wt = (torch.randn(64, 3, 3, 3, 64, generator=gx) / (27 * 64) ** 0.5).to(torch.bfloat16)
    bs = (torch.randn(64, generator=gx) * 0.1).to(torch.bfloat16)
    out = {f"conv_causal{int(c)}_reflect{int(r)}": f32(OV.causal_conv3d(xin.float(), wt, bs, O.BF16, c, r))
           for c in (False, True) for r in (False, True)}
    np.savez_compressed(os.path.join(HERE, "vae_conv3d.npz"), x=f32(xin), w=f32(wt), b=f32(bs), **out)
    Wd_ = OV.make_decoder_weights(seed=77, layers_per_block=1)
    z = torch.randn(1, 128, 2, 2, 2, generator=gx).to(torch.bfloat16)
    vid = OV.vae_decode(z.float(), Wd_, O.BF16, layers_per_block=1)
    np.savez_compressed(os.path.join(HERE, "vae_decode_tiny.npz"), seed_weights=77, latent=f32(z), video=f32(vid).astype(np.float16),
                        uint8=OV.to_uint8(vid[0], O.BF16).numpy())

    # ---- round 3: attention under both rounding policies (oracle/dit.py::sdpa), incl. a ragged key count and a spiked row;
    # the area-resize restatement (oracle/media.py) at an integer and a fractional factor ----
    ga = torch.Generator().manual_seed(9)
    q = torch.randn(1, 70, 256, generator=ga).to(torch.bfloat16)
    k = torch.randn(1, 150, 256, generator=ga).to(torch.bfloat16)
    v = torch.randn(1, 150, 256, generator=ga).to(torch.bfloat16)
    k[0, 99] = q[0, 7] * 3.0
    np.savez_compressed(os.path.join(HERE, "sdpa_policies.npz"), q=f32(q), k=f32(k), v=f32(v), heads=2,
                        out_fp32P=f32(O.sdpa(q.float(), k.float(), v.float(), 2, O.BF16)),
                        out_flash=f32(O.sdpa(q.float(), k.float(), v.float(), 2, O.BF16_FLASH)))
    from oracle import media as OM
    fr = np.random.default_rng(11).random((2, 3, 24, 36), dtype=np.float32) * 2 - 1
    np.savez_compressed(os.path.join(HERE, "area_resize.npz"), frames=fr, half=OM.resize_area(fr, 12, 18), frac=OM.resize_area(fr, 16, 20))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()

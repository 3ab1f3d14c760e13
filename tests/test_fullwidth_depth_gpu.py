"""Full WIDTH and DEPTH together: the default launch structure - row statistics
carried FF2 -> next block's norm, raw q normalised + rotated inside flash_attn, q|k on the 320x256 tile, (1+scale) folded
into ada_combine - inside a multi-layer comparison at the bench shape (D=4096, 32 heads, FF 16384, N=1280 tokens, S=1024,
B=2 CFG pair), with the per-layer hidden error against the oracle under BOTH attention policies:

  * ``O.BF16``        P stays fp32 until after P.V (MLX's documented "fp32 inside, one rounding");
  * ``O.BF16_FLASH``  P rounded to bf16 where the kernel rounds it (oracle/dit.py::sdpa) - what is left against this one is
                      fp32 summation order and the bf16 rounding flips it causes downstream.

Plus: the same model with every launch-structure fusion off (fuse=0) agrees with the default (fuse=15) to <= 2 bf16 ulps
on >= 99 % (<= 1 ulp on >= 92 %) of the residual stream after every layer; and the full-size 33x512x512 VAE decode (the real 5 res blocks per
stage) against the oracle."""
import pytest
import torch

import parity
from oracle import dit as O
from parity import rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
L = 4


def _ulp_far_frac(a, b, ulps=1):
    """fraction of elements more than `ulps` bf16 ulps apart.  The ulp is taken at max(|a|, |b|, rms of the row): an entry
    of the residual stream is a sum res + y whose error is absolute at the scale of its operands, so for the few entries
    that cancel to |x| << rms the element's own ulp is not the unit of that error."""
    a, b = a.float().cpu(), b.float().cpu()
    rms = b.pow(2).mean(dim=-1, keepdim=True).sqrt()
    tol = ulps * (2.0 ** -7) * torch.maximum(torch.maximum(a.abs(), b.abs()), rms)
    return float(((a - b).abs() > tol).float().mean())


def test_forward_fullwidth_4_layers_vs_oracle_per_layer(dev):
    from mlx_video_amd.ltx_model import LTXModel, LTXModelConfig, TimestepPlan, precompute_freqs_cis
    cfg = O.DiTConfig(num_layers=L)
    W = O.make_weights(cfg, seed=31)
    Wd = {k: v.to(dev) for k, v in W.items()}
    model = LTXModel(LTXModelConfig(num_layers=L), Wd)                      # default launch structure (fuse=15)
    plain = LTXModel(LTXModelConfig(num_layers=L), Wd, fuse=0)              # every fusion off
    B, F, Hh, Ww, S = 2, 5, 16, 16, 1024
    N = F * Hh * Ww
    g = torch.Generator().manual_seed(45)
    lat = torch.randn(1, N, 128, generator=g).to(BF).expand(B, N, 128).contiguous()
    ctx = torch.randn(B, S, 3840, generator=g).to(BF)
    ts = torch.full((B, N), 0.909375).to(BF)
    pos = torch.from_numpy(O.create_position_grid(1, F, Hh, Ww))
    pe = O.precompute_freqs_cis(pos, cfg.dim)
    pe_d = precompute_freqs_cis(pos.to(dev), cfg.dim)
    plan = TimestepPlan.from_timesteps(ts.to(dev))
    hid, hid0 = [], []
    v = model.forward_tokens(lat.to(dev), plan, ctx.to(dev), pe_d, hidden=hid)
    v0 = plain.forward_tokens(lat.to(dev), plan, ctx.to(dev), pe_d, hidden=hid0)
    torch.cuda.synchronize()
    assert len(hid) == L and v.shape == (B, N, 128)
    # ---- launch structure: fuse=15 vs fuse=0, layer by layer.  The two differ by fp32 summation order only (row
    # statistics reduced in the GEMM epilogue vs by the norm kernel), which chained bf16 rounding turns into single-ulp
    # flips of ~10 % of the stream per residual add (tests/test_block_stages_gpu.py explains the growth law); an entry
    # that flips in two of a block's three residual adds is 2 ulps off.  The difference grows layer by layer towards the
    # saturation level of the law (a few 1e-3, the bf16 ulp): stated bounds <= 1 ulp on >= 92 %, <= 2 ulps on >= 99 %,
    # rel-L2 <= 1e-2 after every layer (measured on MI355X: 0.4 % / 0.003 % / 1.6e-3 after layer 0), pinned at 2x measured. ----
    for i in range(L):
        parity.check(f"dit.fullwidth_L{L}.fuse15_vs_fuse0.layer{i}.frac_beyond_1ulp", _ulp_far_frac(hid[i], hid0[i], 1), 8e-2)
        parity.check(f"dit.fullwidth_L{L}.fuse15_vs_fuse0.layer{i}.frac_beyond_2ulp", _ulp_far_frac(hid[i], hid0[i], 2), 1e-2)
        parity.check(f"dit.fullwidth_L{L}.fuse15_vs_fuse0.layer{i}.rel_l2", rel_l2(hid[i], hid0[i]), 1e-2)
    parity.check(f"dit.fullwidth_L{L}.fuse15_vs_fuse0.velocity_rel_l2", rel_l2(v, v0), 1e-2)
    # ---- against the oracle, both attention policies, per layer ----
    for name, pol, tol_h, tol_v in (("fp32P", O.BF16, 1e-2, 1e-2), ("flash", O.BF16_FLASH, 1e-2, 1e-2)):
        ref, rh = O.ltx_forward(lat.float(), ts.float(), ctx.float(), pe, W, cfg, pol, return_hidden=True)
        for i in range(L):
            parity.check(f"dit.fullwidth_L{L}.hidden{i}_vs_bf16_oracle_{name}", rel_l2(hid[i].float().cpu(), rh[i]), tol_h)
        parity.check(f"dit.fullwidth_L{L}.velocity_vs_bf16_oracle_{name}", rel_l2(v.float().cpu(), ref), tol_v)
    assert not torch.equal(v[0], v[1])


def test_vae_decode_fullsize_vs_oracle(dev):
    """33x512x512 (latent 5x16x16), the decoder's real depth (5 res blocks per stage), same weights: decoder.py:361-450."""
    from oracle import vae as OV
    from mlx_video_amd.video_vae import LTX2VideoDecoder, to_uint8_frames
    Wv = OV.make_decoder_weights(seed=9, layers_per_block=5)
    dec = LTX2VideoDecoder({k: v.to(dev) for k, v in Wv.items()}, num_layers_per_block=5)
    g = torch.Generator().manual_seed(10)
    zl = torch.randn(1, 128, 5, 16, 16, generator=g).to(BF)
    vid = dec(zl.to(dev))
    u8 = to_uint8_frames(vid)
    torch.cuda.synchronize()
    ref = OV.vae_decode(zl.float(), Wv, O.BF16, layers_per_block=5)
    assert vid.shape == ref.shape == (1, 3, 33, 512, 512)
    parity.check("vae.decode_fullsize_33x512x512_5blocks.rel_l2_vs_bf16_oracle", rel_l2(vid.float().cpu(), ref), 2e-2)
    # generate.py:3894-3898 on the oracle's frames
    ru8 = ((ref.to(BF).float() + 1) / 2).clamp(0, 1).to(BF).float().mul(255).to(BF).to(torch.uint8)[0].permute(1, 2, 3, 0)
    d = (u8[0].cpu().int() - ru8.int()).abs().float()
    parity.check("vae.decode_fullsize_33x512x512_5blocks.uint8_mean_abs_diff", float(d.mean()), 1.0)
    parity.check("vae.decode_fullsize_33x512x512_5blocks.uint8_p99_abs_diff", float(torch.quantile(d.flatten()[::97], 0.99)), 4.0)


@pytest.mark.skipif(__import__("os").environ.get("LTXK_DEEP_PARITY") != "1",
                    reason="full width AND full depth (D=4096, L=48: 13 B parameters, ~4 min of CPU oracle): LTXK_DEEP_PARITY=1; "
                           "the measured ledger of the last run is profiles/r04_parity_fulldepth.json")
def test_forward_fullwidth_48_layers_vs_oracle(dev):
    """The whole video DiT at its real size - 48 blocks of D=4096 / FF 16384, B=2 CFG pair, N=1280, S=1024 (one forward of the
    bench step, 34.9 TFLOP per batch row; ltx.py:459-506) - against the oracle's flash policy, hidden state after every block.
    Closes the last "never compared" of the error budget: the per-layer error keeps to the bf16 saturation level
    the growth law predicts (a few 1e-3) all the way to layer 47."""
    import json
    import os
    from mlx_video_amd.ltx_model import LTXModel, LTXModelConfig, TimestepPlan, precompute_freqs_cis
    LL = 48
    cfg = O.DiTConfig(num_layers=LL)
    W = O.make_weights(cfg, seed=31)                                        # 26 GB of bf16 on the host
    model = LTXModel(LTXModelConfig(num_layers=LL), {k: v.to(dev) for k, v in W.items()})
    B, F, Hh, Ww, S = 2, 5, 16, 16, 1024
    N = F * Hh * Ww
    g = torch.Generator().manual_seed(45)
    lat = torch.randn(1, N, 128, generator=g).to(BF).expand(B, N, 128).contiguous()
    ctx = torch.randn(B, S, 3840, generator=g).to(BF)
    ts = torch.full((B, N), 0.909375).to(BF)
    pos = torch.from_numpy(O.create_position_grid(1, F, Hh, Ww))
    pe = O.precompute_freqs_cis(pos, cfg.dim)
    plan = TimestepPlan.from_timesteps(ts.to(dev))
    hid = []
    v = model.forward_tokens(lat.to(dev), plan, ctx.to(dev), precompute_freqs_cis(pos.to(dev), cfg.dim), hidden=hid)
    torch.cuda.synchronize()
    hid = [h.float().cpu() for h in hid]
    v = v.float().cpu()
    del model
    torch.cuda.empty_cache()
    ref, rh = O.ltx_forward(lat.float(), ts.float(), ctx.float(), pe, W, cfg, O.BF16_FLASH, return_hidden=True)
    errs = [rel_l2(hid[i], rh[i]) for i in range(LL)]
    for i in (0, 3, 11, 23, 35, 47):
        parity.check(f"dit.fullwidth_L48.hidden{i}_vs_bf16_oracle_flash", errs[i], 1.5e-2)
    ev = parity.check("dit.fullwidth_L48.velocity_vs_bf16_oracle_flash", rel_l2(v, ref), 1.5e-2)
    root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    json.dump({"shape": "B=2 N=1280 S=1024 D=4096 L=48, oracle policy BF16_FLASH", "hidden_rel_l2_per_layer": errs, "velocity_rel_l2": ev},
              open(os.path.join(root, "gpurun_out", "r04_parity_fulldepth.json"), "w"), indent=1)

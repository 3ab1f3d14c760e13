"""CPU-side checks of the product: the C-ABI library loads and exports every symbol that
include/ltxk.h declares (no compute calls), conditioning hooks, error behaviour."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            src = open(os.path.join(ROOT, "include", fn)).read()
            names |= set(re.findall(r"\b(ltxk_[a-z0-9_]+)\s*\(", src))
    return names


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    from mlx_video_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(lib, name), f"libltxk.so does not export {name}"
    assert set(_lib.SIGNATURES) == declared, set(_lib.SIGNATURES) ^ declared
    hdr = open(os.path.join(ROOT, "include", "ltxk.h")).read()
    assert lib.ltxk_version() == int(re.search(r"#define LTXK_VERSION (\d+)", hdr).group(1))
    import ctypes
    for which, st in enumerate((_lib.GemmArgs, _lib.Conv3dArgs, _lib.AttnArgs)):       # binding layout == compiled layout
        assert lib.ltxk_abi_sizeof(which) == ctypes.sizeof(st)


def test_product_refuses_cpu_tensors():
    from mlx_video_amd import ops
    from mlx_video_amd._lib import LtxkError
    a = torch.zeros(8, 64, dtype=torch.bfloat16)
    with pytest.raises(LtxkError, match="no CPU fallback"):
        ops.gemm(a, a, None)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "mlx-video_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f"{f} imports the oracle"


def test_conditioning_hooks():
    from mlx_video_amd.conditioning import (LatentState, VideoConditionByKeyframeIndex, VideoConditionByLatentIndex,
                                            apply_conditioning, apply_denoise_mask, noise_blend)
    from oracle import sched as S
    g = torch.Generator().manual_seed(0)
    shape = (1, 8, 4, 2, 2)
    st = LatentState(torch.zeros(shape), torch.zeros(shape), torch.ones(1, 1, 4, 1, 1))
    c1 = torch.randn(1, 8, 1, 2, 2, generator=g)
    c2 = torch.randn(1, 8, 2, 2, 2, generator=g)
    out = apply_conditioning(st, [VideoConditionByLatentIndex(c1, 0, 1.0), VideoConditionByKeyframeIndex(c2, 3, 0.5)])
    rl, rc, rm = S.apply_conditioning(st.latent, st.clean_latent, st.denoise_mask,
                                      [("replace", c1, 0, 1.0), ("guide", c2, 3, 0.5)])
    assert torch.equal(out.latent, rl) and torch.equal(out.clean_latent, rc) and torch.equal(out.denoise_mask, rm)
    assert out.denoise_mask.flatten().tolist() == [0.0, 1.0, 1.0, 0.5]
    assert torch.equal(out.latent[:, :, 0], c1[:, :, 0]) and float(out.latent[:, :, 3].abs().max()) == 0.0   # guide keeps latent
    assert torch.equal(out.clean_latent[:, :, 3], c2[:, :, 0])                                              # clipped at F
    with pytest.raises(ValueError, match="out of bounds"):
        apply_conditioning(st, [VideoConditionByLatentIndex(c1, 4)])
    with pytest.raises(ValueError, match="does not match"):
        apply_conditioning(st, [VideoConditionByLatentIndex(torch.zeros(1, 8, 1, 3, 2), 0)])
    noise = torch.randn(shape, generator=g)
    nb = noise_blend(out, noise, 0.9)
    assert torch.allclose(nb.latent, S.noise_blend(noise, out.latent, out.denoise_mask, 0.9))
    assert torch.equal(nb.latent[:, :, 0], c1[:, :, 0])           # mask 0 => untouched
    x0 = torch.randn(shape, generator=g)
    assert torch.equal(apply_denoise_mask(x0, out.clean_latent, out.denoise_mask)[:, :, 0], c1[:, :, 0])


def test_model_strict_load_and_sanitize():
    from mlx_video_amd.ltx_model import LTXModel, LTXModelConfig
    cfg = LTXModelConfig(num_layers=1)
    with pytest.raises(ValueError, match="Missing"):
        LTXModel(cfg, {})
    raw = {"model.diffusion_model.transformer_blocks.0.attn1.to_out.0.weight": 1,
           "model.diffusion_model.transformer_blocks.0.ff.net.0.proj.bias": 2,
           "model.diffusion_model.transformer_blocks.0.ff.net.2.weight": 3,
           "model.diffusion_model.adaln_single.emb.timestep_embedder.linear_1.weight": 4,
           "model.diffusion_model.audio_embeddings_connector.x": 5, "vae.decoder.conv_in.weight": 6}
    s = LTXModel.sanitize(raw)
    assert set(s) == {"transformer_blocks.0.attn1.to_out.weight", "transformer_blocks.0.ff.proj_in.bias",
                      "transformer_blocks.0.ff.proj_out.weight", "adaln_single.emb.timestep_embedder.linear1.weight"}
    assert len(LTXModel.expected_keys(LTXModelConfig())) == 15 + 48 * 25


def test_component_hook_contracts():
    from mlx_video_amd import components as C
    from oracle import dit as O
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 8, 3, 4, 5, generator=g)
    x0 = torch.randn(2, 8, 3, 4, 5, generator=g)
    sig = torch.tensor([0.9, 0.6, 0.0])
    # Euler via velocity == the denoise loop's form x0 + s'(x-x0)/s
    out = C.EulerDiffusionStep().execute(x, x0, sig, 0)
    assert torch.allclose(out, O.euler_step(x, x0, 0.9, 0.6, O.F32), atol=1e-5)
    assert torch.allclose(C.EulerDiffusionStep().execute(x, x0, sig, 1), x0, atol=1e-5)     # last step lands on x0
    c, u = torch.tensor([[[1.0, 2.0, 3.0]]]), torch.tensor([[[0.5, 1.0, 1.5]]])
    assert torch.equal(C.CFGGuider(4.0).delta(c, u), torch.tensor([[[1.5, 3.0, 4.5]]])) and not C.CFGGuider(1.0).enabled()
    assert C.STGGuider(0.0).enabled() is False and C.LtxAPGGuider(3.0).enabled()
    # CFG* with uncond parallel to cond: rescaled negative equals cond -> zero delta
    assert float(C.CFGStarRescalingGuider(4.0).delta(c, 0.5 * c).abs().max()) < 1e-6
    pz = C.VideoLatentPatchifier(1)
    tok = pz.patchify(x)
    assert tok.shape == (2, 60, 8) and pz.get_token_count(x.shape) == 60
    assert torch.equal(tok, O.latent_to_tokens(x)) and torch.equal(pz.unpatchify(tok, x.shape), x)
    n = C.GaussianNoiser(seed=3)
    assert torch.equal(n.noise(x), n.noise(x)) and n.noise(x.bfloat16()).dtype == torch.bfloat16
    lq = C.LinearQuadraticScheduler().execute(8)
    assert lq.shape == (9,) and lq[0] == 1.0 and abs(float(lq[-1])) < 1e-6 and all(lq[i] > lq[i + 1] for i in range(8))
    bs = C.BetaScheduler().execute(8)
    assert bs[-1] == 0.0 and all(bs[i] >= bs[i + 1] for i in range(len(bs) - 1))
    from mlx_video_amd.schedulers import LTX2Scheduler, ltx2_scheduler
    assert torch.equal(LTX2Scheduler().execute(10, latent=torch.zeros(1, 128, 5, 16, 16)), ltx2_scheduler(10, 1280))


def test_tiling_host_logic_known_answers():
    # reference tests/test_vae_streaming.py:223-297
    from mlx_video_amd.video_vae import (TilingConfig, compute_trapezoidal_mask_1d, split_in_spatial, split_in_temporal)
    from oracle import vae as OV
    for length in (16, 32, 64, 128):
        for ramp in (0, 4, 8, 16):
            if ramp < length:
                mk = compute_trapezoidal_mask_1d(length, ramp, ramp, False)
                assert float(mk.min()) >= 0 and float(mk.max()) <= 1
                assert torch.equal(mk, OV.trapezoid_mask(length, ramp, ramp, False))
    mk = compute_trapezoidal_mask_1d(32, 8, 8, False)
    assert torch.allclose(mk[12:20], torch.ones(8)) and bool((mk[:8].diff() >= 0).all()) and bool((mk[-8:].diff() <= 0).all())
    assert float(compute_trapezoidal_mask_1d(32, 8, 0, True)[0]) == 0.0 and float(compute_trapezoidal_mask_1d(32, 8, 0, False)[0]) > 0.0
    d = TilingConfig.default()
    assert d.spatial_config.tile_size_in_pixels == 512 and d.temporal_config.tile_size_in_frames == 64
    assert TilingConfig.aggressive().spatial_config.tile_size_in_pixels == 256 and TilingConfig.aggressive().temporal_config.tile_size_in_frames == 32
    assert TilingConfig.conservative().spatial_config.tile_size_in_pixels == 768 and TilingConfig.conservative().temporal_config.tile_size_in_frames == 96
    assert TilingConfig.auto(256, 256, 33) is None and TilingConfig.auto(512, 512, 33) is None
    assert TilingConfig.auto(1024, 768, 145) is not None
    a = TilingConfig.auto(512, 512, 97)          # config 4: temporal 64f / 24f overlap only
    assert a.spatial_config is None and (a.temporal_config.tile_size_in_frames, a.temporal_config.tile_overlap_in_frames) == (64, 24)
    a = TilingConfig.auto(768, 768, 65)          # configs 3/5: spatial 384px / 64px overlap only
    assert a.temporal_config is None and (a.spatial_config.tile_size_in_pixels, a.spatial_config.tile_overlap_in_pixels) == (384, 64)
    with pytest.raises(ValueError):
        TilingConfig.spatial_only(48, 0)
    iv = split_in_temporal(4, 1, 12)             # 12 latent frames, 32f tiles / 8f overlap (the 89-frame case)
    assert iv.starts == OV.split_temporal(4, 1, 12)[0] and iv.ends[-1] == 12 and iv.starts[0] == 0
    assert split_in_spatial(16, 2, 24).starts == OV.split_spatial(16, 2, 24)[0]


def test_media_and_checkpoint_keymaps(tmp_path):
    import numpy as np
    from PIL import Image
    from safetensors.torch import save_file
    from mlx_video_amd import media, weights
    rng = np.random.default_rng(0)
    Image.fromarray(rng.integers(0, 255, (70, 100, 3), dtype=np.uint8)).save(tmp_path / "a.png")
    assert media.load_image(tmp_path / "a.png").shape == (64, 96, 3)                    # rounded down to /32
    im = media.load_image(tmp_path / "a.png", 128, 160)
    assert im.shape == (128, 160, 3) and 0.0 <= im.min() and im.max() <= 1.0
    c = media.image_to_conditioning(im)
    assert c.shape == (1, 3, 1, 128, 160) and -1.0 <= float(c.min()) and float(c.max()) <= 1.0
    fr = media.load_frames(rng.integers(0, 255, (12, 32, 48, 3), dtype=np.uint8), 64, 96)
    assert fr.shape == (12, 64, 96, 3) and media.frames_to_conditioning(fr).shape == (1, 3, 9, 64, 96)   # 12 -> 9 = 1+8k
    # checkpoint key maps + conv layout (ltx.py:508-533, decoder.py:544-591,708-710, encoder.py:135-179)
    sd = {"model.diffusion_model.transformer_blocks.0.attn1.to_out.0.weight": torch.zeros(4, 4),
          "model.diffusion_model.transformer_blocks.0.ff.net.0.proj.weight": torch.zeros(8, 4),
          "model.diffusion_model.audio_patchify_proj.weight": torch.zeros(2, 2),
          "vae.decoder.mid_block.resnets.1.conv1.conv.weight": torch.arange(2 * 3 * 27, dtype=torch.float32).reshape(2, 3, 3, 3, 3),
          "vae.decoder.up_blocks.1.upsamplers.0.conv.bias": torch.zeros(2),
          "vae.decoder.up_blocks.2.resnets.4.conv2.conv.bias": torch.zeros(2),
          "vae.per_channel_statistics.mean-of-means": torch.zeros(128),
          "vae.per_channel_statistics.std-of-means": torch.ones(128),
          "vae.encoder.down_blocks.0.res_blocks.1.conv1.conv.weight": torch.zeros(2, 3, 3, 3, 3)}
    save_file(sd, str(tmp_path / "m.safetensors"), metadata={"config": '{"vae": {"timestep_conditioning": true}}'})
    raw = weights.read_safetensors([tmp_path / "m.safetensors"])
    tw = weights.transformer_weights(raw, "cpu")
    assert set(tw) == {"transformer_blocks.0.attn1.to_out.weight", "transformer_blocks.0.ff.proj_in.weight"}
    dw = weights.vae_decoder_weights(raw, "cpu")
    assert {"up_blocks.0.res_blocks.1.conv1.conv.weight", "up_blocks.3.conv.bias", "up_blocks.6.res_blocks.4.conv2.conv.bias",
            "latents_mean", "latents_std"} <= set(dw)
    w = dw["up_blocks.0.res_blocks.1.conv1.conv.weight"]
    assert w.shape == (2, 3, 3, 3, 3) and float(w[1, 2, 1, 0, 2]) == float(sd["vae.decoder.mid_block.resnets.1.conv1.conv.weight"][1, 2, 2, 1, 0])
    ew = weights.vae_encoder_weights(raw, "cpu")
    assert "down_blocks.0.res_blocks.1.conv1.weight" in ew and "per_channel_statistics.mean" in ew
    # "std-of-means" contains the word "mean": it must still land on .std (it once overwrote .mean and left .std missing)
    assert float(ew["per_channel_statistics.std"][0]) == 1.0 and float(ew["per_channel_statistics.mean"][0]) == 0.0
    assert weights.sniff_timestep_conditioning(tmp_path / "m.safetensors") is True


def test_ffmpeg_writer_contract(tmp_path, monkeypatch):
    """generate.py:1833-1893: raw RGB24 pipe command; FileNotFoundError without ffmpeg; RuntimeError on failure."""
    import shutil, stat
    import numpy as np
    from mlx_video_amd import media
    cmd = media.ffmpeg_command("ffmpeg", 64, 32, 24.0, "o.mp4")
    assert cmd[:12] == ["ffmpeg", "-y", "-f", "rawvideo", "-pix_fmt", "rgb24", "-s", "64x32", "-r", "24.0", "-i", "-"]
    assert cmd[-3:] == ["-pix_fmt", "yuv420p", "o.mp4"] and "libx264" in cmd and "18" in cmd
    frames = np.zeros((3, 32, 64, 3), np.uint8)
    monkeypatch.setattr(shutil, "which", lambda name: None)
    with pytest.raises(FileNotFoundError):
        media.write_video_ffmpeg(frames, tmp_path / "o.mp4", 24.0)
    with pytest.raises(ValueError):
        media.write_video_ffmpeg(frames.astype(np.float32), tmp_path / "o.mp4", 24.0)
    # a stand-in encoder that counts stdin bytes proves frames are streamed whole and a failure surfaces stderr
    fake = tmp_path / "ffmpeg"
    fake.write_text("#!/bin/sh\nn=$(wc -c)\nfor a; do last=$a; done\necho $n > \"$last\"\n")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setattr(shutil, "which", lambda name: str(fake))
    media.write_video_ffmpeg(frames, tmp_path / "o.txt", 24.0)
    assert int((tmp_path / "o.txt").read_text()) == frames.size
    bad = tmp_path / "ffmpeg_bad"
    bad.write_text("#!/bin/sh\ncat > /dev/null\necho boom >&2\nexit 3\n")
    bad.chmod(bad.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setattr(shutil, "which", lambda name: str(bad))
    with pytest.raises(RuntimeError, match="boom"):
        media.write_video_ffmpeg(frames, tmp_path / "o2.mp4", 24.0)


def test_rope_bf16_positions_warn():
    """test_rope.py:207-221: a bfloat16 position grid triggers the reference's UserWarning (and is then used as
    float32); float32 grids do not warn.  (The table itself is a HIP kernel: the call fails loudly on CPU tensors.)"""
    import warnings
    from mlx_video_amd.ltx_model import precompute_freqs_cis
    from mlx_video_amd.schedulers import create_position_grid
    pos = create_position_grid(1, 4, 4, 4)
    with pytest.warns(UserWarning, match="Position grid has dtype bfloat16"):
        with pytest.raises(Exception):
            precompute_freqs_cis(pos.to(torch.bfloat16), 128, 10000.0, [20, 2048, 2048], 32)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        with pytest.raises(Exception) as ei:
            precompute_freqs_cis(pos, 128, 10000.0, [20, 2048, 2048], 32)
        assert not isinstance(ei.value, UserWarning)


def test_trapezoid_mask_known_properties():
    """test_vae_streaming.py:223-263 on the host implementation and the oracle: values in [0,1], centre exactly 1,
    monotonic ramps, temporal masks start from 0, spatial masks start above 0."""
    from oracle import vae as OV
    from mlx_video_amd.video_vae import compute_trapezoidal_mask_1d
    for fn in (compute_trapezoidal_mask_1d, OV.trapezoid_mask):
        for length in (16, 32, 64, 128):
            for ramp in (0, 4, 8, 16):
                if ramp < length:
                    m = fn(length, ramp, ramp, False).float()
                    assert float(m.min()) >= 0.0 and float(m.max()) <= 1.0
        m = fn(32, 8, 8, False).float()
        assert torch.allclose(m[12:20], torch.ones(8), rtol=1e-5)
        assert bool((m[1:9] >= m[:8]).all()) and bool((m[-8:] <= m[-9:-1]).all())        # ramps monotonic
        assert float(fn(32, 8, 0, True)[0]) == 0.0                                          # temporal: starts from 0
        assert float(fn(32, 8, 0, False)[0]) > 0.0                                          # spatial: starts above 0


def test_bench_flop_accounting_matches_survey_table():
    """bench.dit_forward_flops restates SURVEY.md §8d's formula: the per-forward totals of its table must come out
    (TFLOP/forward at N = 32, 1280, 1296, 5184, 3328), and the "executed" variant only drops the per-token AdaLN MLPs."""
    import bench
    for n, tflop in ((32, 4.12), (1280, 34.85), (1296, 35.26), (5184, 146.81), (3328, 90.59)):
        assert abs(bench.dit_forward_flops(n) / 1e12 - tflop) < 0.006 * tflop, n
    full, executed = bench.dit_forward_flops(1280, B=2), bench.dit_forward_flops(1280, B=2, U=1)
    D = 4096
    assert abs((full - executed) - (2 * 2 * 1280 - 2) * (256 * D + D * D + 6 * D * D)) < 1e6
    assert 69.5e12 < full < 69.9e12                       # the 69.7 TFLOP/step the bench line quotes
    assert len(bench.source_sha()) == 16


def test_gemm_tile_order_is_a_bijection(tmp_path):
    """map_tile (csrc/gemm_core.h) permutes workgroup ids onto tiles for L2 locality; whatever the shape it must hit every
    tile exactly once.  The function is plain integer C++: its text is compiled for the host with g++ and swept over the
    tile grids the model produces and a set of awkward ones."""
    import re
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    src = open(os.path.join(ROOT, "mlx-video_amd", "csrc", "gemm_core.h")).read()
    m = re.search(r"__device__ __forceinline__ void map_tile\(.*?\n}\n", src, re.S)
    assert m, "map_tile not found"
    body = m.group(0).replace("__device__ __forceinline__", "static")
    prog = body + r"""
#include <cstdio>
#include <vector>
int main() {
  const int cfgs[][2] = {{16,16},{16,64},{16,48},{16,32},{8,16},{24,48},{16,80},{13,16},{16,112},{32,16},{5,3},{16,24},{1,1},{7,128},
                         {33,16},{41,32},{16,96},{8,48},{2,16},{40,2},{11,64},
                         /* 320-row tile grids: FF1, q|k, text k|v at M=2560 / 2048, FF1 at M=5184 / 6656 */ {8,64},{8,32},{7,32},{17,64},{21,64},{3,4}};
  for (auto& c : cfgs) {
    const int RT = c[0], CT = c[1];
    std::vector<int> seen(RT * CT, 0);
    for (int b = 0; b < RT * CT; ++b) {
      int rt = -1, ct = -1;
      map_tile(b, RT, CT, rt, ct);
      if (rt < 0 || rt >= RT || ct < 0 || ct >= CT || seen[rt * CT + ct]++) { printf("FAIL RT=%d CT=%d bid=%d -> (%d,%d)\n", RT, CT, b, rt, ct); return 1; }
    }
  }
  printf("OK\n");
  return 0;
}
"""
    (tmp_path / "m.cpp").write_text(prog)
    subprocess.run(["g++", "-O1", "-o", str(tmp_path / "m"), str(tmp_path / "m.cpp")], check=True)
    out = subprocess.run([str(tmp_path / "m")], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "OK", out.stdout


def test_vae_statistics_keys_match_exactly(tmp_path):
    """A real PyTorch LTX-2 checkpoint carries five per-channel statistics tensors plus the audio VAE's pair of the same
    shape (reference convert.py:277-286 skips the extras; decoder.py:667-690 / encoder.py:141-157 read only
    vae.per_channel_statistics.{mean-of-means,std-of-means} and the plain mean/std names).  In adversarial file order
    (the impostors LAST, so a last-key-wins map would take them) mean and std must still be the right tensors."""
    import collections
    from mlx_video_amd import weights
    f = lambda v: torch.full((128,), float(v))
    raw = collections.OrderedDict()
    raw["vae.per_channel_statistics.mean-of-means"] = f(1)
    raw["vae.per_channel_statistics.std-of-means"] = f(2)
    raw["vae.per_channel_statistics.channel"] = f(30)
    raw["vae.per_channel_statistics.mean-of-stds"] = f(40)
    raw["vae.per_channel_statistics.mean-of-stds_over_std-of-means"] = f(50)
    raw["audio_vae.per_channel_statistics.mean-of-means"] = f(60)
    raw["audio_vae.per_channel_statistics.std-of-means"] = f(70)
    raw["vae.decoder.conv_in.conv.bias"] = torch.zeros(4)
    raw["vae.encoder.conv_in.conv.bias"] = torch.zeros(4)
    dw = weights.vae_decoder_weights(raw, "cpu")
    ew = weights.vae_encoder_weights(raw, "cpu")
    assert float(dw["latents_mean"][0]) == 1.0 and float(dw["latents_std"][0]) == 2.0
    assert float(ew["per_channel_statistics.mean"][0]) == 1.0 and float(ew["per_channel_statistics.std"][0]) == 2.0
    assert not any("audio" in k or "of-stds" in k or k.endswith("channel") for k in list(dw) + list(ew))
    for k in ("vae.per_channel_statistics.channel", "vae.per_channel_statistics.mean-of-stds", "audio_vae.per_channel_statistics.mean-of-means",
              "audio_vae.per_channel_statistics.std-of-means", "vae.per_channel_statistics.mean-of-stds_over_std-of-means"):
        assert weights._vae_encoder_key(k) is None and weights._vae_decoder_key(k) is None, k
    # plain names override *-of-means in either file order (decoder.py:676-689, encoder.py:152-157) ...
    for order in (("per_channel_statistics.mean", "vae.per_channel_statistics.mean-of-means"),
                  ("vae.per_channel_statistics.mean-of-means", "per_channel_statistics.mean")):
        r2 = collections.OrderedDict((k, f(9 if k == "per_channel_statistics.mean" else 1)) for k in order)
        r2["vae.per_channel_statistics.std-of-means"] = f(2)
        assert float(weights.vae_decoder_weights(r2, "cpu")["latents_mean"][0]) == 9.0
        assert float(weights.vae_encoder_weights(r2, "cpu")["per_channel_statistics.mean"][0]) == 9.0
    # ... and an explicit latents_mean overrides both (decoder.py:690-695)
    r3 = collections.OrderedDict([("latents_mean", f(5)), ("per_channel_statistics.mean", f(9)), ("latents_std", f(6))])
    d3 = weights.vae_decoder_weights(r3, "cpu")
    assert float(d3["latents_mean"][0]) == 5.0 and float(d3["latents_std"][0]) == 6.0


def test_cli_heuristics_table():
    """generate.py:4545-4560,4629-4644: which flavour of the hot loop the CLI picks (compiled vs eager step, batched vs
    sequential CFG, sync cadence, step counts) from flags and LTX_COMPILE / LTX_CFG_BATCH / LTX_EVAL_INTERVAL."""
    from mlx_video_amd.generate import build_parser, resolve_cli_heuristics

    def run(argv, env=None):
        return resolve_cli_heuristics(build_parser().parse_args(argv), env or {})

    a = run(["--pipeline", "dev"])                                   # dev, 40 steps, cfg 4.0
    assert (a.compile_step, a.cfg_batch, a.eval_interval, a.stage1_steps, a.stage2_steps) == (True, True, 2, 8, 3)
    a = run(["--pipeline", "dev", "--steps", "7"])                   # short dev loop: no auto compile
    assert (a.compile_step, a.cfg_batch) == (False, True)
    a = run(["--pipeline", "dev", "--cfg-scale", "1.0"])             # no CFG -> nothing to batch
    assert (a.compile_step, a.cfg_batch) == (True, False)
    a = run(["--pipeline", "dev", "--no-compile", "--no-cfg-batch"], {"LTX_COMPILE": "1", "LTX_CFG_BATCH": "1"})
    assert (a.compile_step, a.cfg_batch) == (False, False)           # the --no-* flags beat the environment
    a = run(["--pipeline", "dev", "--steps", "4", "--cfg-scale", "1.0"], {"LTX_COMPILE": "true", "LTX_CFG_BATCH": "yes"})
    assert (a.compile_step, a.cfg_batch) == (True, True)             # the environment beats the auto rules
    a = run(["--pipeline", "dev", "--model-repo", "org/LTX-2-dev-4bit"])
    assert a.cfg_batch is False                                      # quantised repos never auto-batch
    a = run(["--pipeline", "distilled"])
    assert (a.compile_step, a.cfg_batch, a.eval_interval, a.stage1_steps, a.stage2_steps) == (False, False, 4, 5, 1)
    a = run(["--pipeline", "distilled", "--num-frames", "97", "--stage1-steps", "4"])      # 4 + 1 >= 5 and a long clip
    assert a.compile_step is True
    a = run(["--pipeline", "distilled", "--num-frames", "97", "--stage1-steps", "3"])
    assert a.compile_step is False
    a = run(["--pipeline", "keyframe"]) if "keyframe" in [c for c in build_parser()._option_string_actions["--pipeline"].choices] else None
    if a is not None:
        assert (a.stage1_steps, a.stage2_steps, a.eval_interval) == (8, 3, 4)
    assert run(["--pipeline", "dev"], {"LTX_EVAL_INTERVAL": "7"}).eval_interval == 7
    assert run(["--pipeline", "dev", "--eval-interval", "0"]).eval_interval == 1
    assert run(["--pipeline", "dev", "--eval-interval", "3"], {"LTX_EVAL_INTERVAL": "7"}).eval_interval == 3


def test_video_file_decode_through_ffmpeg(tmp_path, monkeypatch):
    """utils.py:578-613: video FILE conditioning.  No decoder library exists in this image; media.decode_video_ffmpeg drives
    ffprobe + ffmpeg child processes.  Stand-in binaries prove the protocol (size from ffprobe's JSON, raw rgb24 frames from
    ffmpeg's stdout, frame_cap as -frames:v) and the reference's error behaviour."""
    import shutil, stat
    import numpy as np
    from mlx_video_amd import media
    monkeypatch.setattr(shutil, "which", lambda name: None)
    with pytest.raises(FileNotFoundError, match="ffmpeg"):
        media.decode_video_ffmpeg(tmp_path / "v.mp4")
    frames = np.random.default_rng(1).integers(0, 255, (5, 6, 8, 3), dtype=np.uint8)
    (tmp_path / "raw.bin").write_bytes(frames.tobytes())
    probe = tmp_path / "ffprobe"
    probe.write_text('#!/bin/sh\necho \'{"streams": [{"width": 8, "height": 6}]}\'\n')
    ff = tmp_path / "ffmpeg"
    ff.write_text(f'#!/bin/sh\nn=5\nprev=""\nfor a in "$@"; do if [ "$prev" = "-frames:v" ]; then n=$a; fi; prev=$a; done\n'
                  f'head -c $((n*6*8*3)) {tmp_path}/raw.bin\n')
    for f in (probe, ff):
        f.chmod(f.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setattr(shutil, "which", lambda name: str(tmp_path / name))
    vid = tmp_path / "v.mp4"
    vid.write_bytes(b"x")
    got = media.decode_video_ffmpeg(vid)
    assert got.shape == (5, 6, 8, 3) and np.array_equal(got, frames)
    assert np.array_equal(media.decode_video_ffmpeg(vid, frame_cap=3), frames[:3])
    fr = media.load_frames(vid, 6, 8, frame_cap=4)                      # the path generate._cond_pixels takes for --video-conditioning FILE
    assert fr.shape == (4, 6, 8, 3) and np.allclose(fr, frames[:4].astype(np.float32) / 255.0)
    assert media.load_frames(vid, 3, 4).shape == (5, 3, 4, 3)           # area-filter resize of the decoded uint8 frames
    probe.write_text("#!/bin/sh\necho '{}'\n")
    with pytest.raises(ValueError, match="Unable to open video"):
        media.decode_video_ffmpeg(vid)


def test_product_library_reads_no_environment():
    """include/ltxk.h, Conventions: "no global mutable state".  The A/B switches live in libltxk_ab.so (-DLTXK_AB) only: the
    product library must not even import getenv."""
    import shutil, subprocess
    from mlx_video_amd import _lib
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    def imports(path):
        out = subprocess.run([nm, "-D", "--undefined-only", path], capture_output=True, text=True, check=True).stdout
        return {ln.split()[-1].split("@")[0] for ln in out.splitlines() if ln.strip()}
    assert "getenv" not in imports(_lib.LIB_PATH) and "secure_getenv" not in imports(_lib.LIB_PATH)
    if os.path.exists(_lib.AB_LIB_PATH):
        assert "getenv" in imports(_lib.AB_LIB_PATH)          # ... and the measurement build is the one that does


def test_bench_source_hash_ignores_comments():
    """bench.source_sha() stamps the PMC artefacts under profiles/ (bench.py quotes `roofline.traffic` only while the kernel
    sources still hash to it): comment / blank-line edits must not change it, code edits must."""
    import bench
    a = "int a = 1; // c1\n/* block\n more */ const char* s = \"// kept\"; char q = 'x'; // x\n\n  int b; /* y */ int c;\n"
    assert bench._strip_comments(a) == "int a = 1;\n const char* s = \"// kept\"; char q = 'x';\n  int b;  int c;"
    assert bench._strip_comments(a + "// more\n\n") == bench._strip_comments(a)
    assert bench._strip_comments(a + "int d;\n") != bench._strip_comments(a)
    assert len(bench.source_sha()) == 16

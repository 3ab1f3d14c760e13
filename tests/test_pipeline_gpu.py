"""End-to-end pipelines on the GPU: the dev pipeline (config 1: 128x128x9, T2V and I2V) against the
oracle composed end to end (denoise loop -> VAE decode -> uint8), and the three two-stage pipelines
(distilled / keyframe / ic_lora) for plumbing: shapes, dtype, finiteness, phase names, error behaviour.
Tolerance for uint8 frames after a full generate: mean |diff| <= 1.5 grey levels, 99 % within 6."""
import json

import numpy as np
import parity
import pytest
import torch

from oracle import dit as O
from oracle import vae as OV

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _mods(dev, enc_blocks=None):
    from mlx_video_amd.ltx_model import LTXModel, LTXModelConfig
    from mlx_video_amd.upsampler import LatentUpsampler
    from mlx_video_amd.video_vae import LTX2VideoDecoder, VideoEncoder
    cfg = O.DiTConfig(num_layers=2, heads=4, caption_channels=256)
    W = O.make_weights(cfg, seed=31)
    mc = LTXModelConfig(num_attention_heads=4, num_layers=2, caption_channels=256, cross_attention_dim=cfg.dim)
    Wd = OV.make_decoder_weights(seed=32, layers_per_block=1)
    blocks = [("res_x", 1), ("compress_space_res", (1, 2, 2)), ("res_x", 1), ("compress_time_res", (2, 1, 1)),
              ("res_x", 1), ("compress_all_res", (2, 2, 2)), ("res_x", 1), ("compress_all_res", (2, 2, 2)), ("res_x", 1)]
    We = OV.make_encoder_weights(seed=33, blocks=blocks)
    Wu = OV.make_upsampler_weights(mid=128, nb=1)
    return dict(cfg=cfg, W=W, Wd=Wd, We=We, blocks=blocks, Wu=Wu,
                transformer=LTXModel(mc, {k: v.to(dev) for k, v in W.items()}),
                vae_decoder=LTX2VideoDecoder({k: v.to(dev) for k, v in Wd.items()}, num_layers_per_block=1),
                vae_encoder=VideoEncoder({k: v.to(dev) for k, v in We.items()}, encoder_blocks=blocks),
                upsampler=LatentUpsampler({k: v.to(dev) for k, v in Wu.items()}, num_blocks_per_stage=1))


class _Noise:
    """Deterministic noise source shared by the product run and the oracle run."""

    def __init__(self, seed, dev=None):
        self.g = torch.Generator().manual_seed(seed)
        self.dev = dev
        self.log = []

    def __call__(self, shape):
        n = torch.randn(shape, generator=self.g).to(BF)
        self.log.append(n)
        return n.to(self.dev) if self.dev is not None else n


@pytest.mark.parametrize("i2v", [False, True])
def test_dev_pipeline_matches_oracle(dev, tmp_path, i2v):
    from mlx_video_amd.generate import PipelineType, generate_video
    from mlx_video_amd.schedulers import ltx2_scheduler
    m = _mods(dev)
    g = torch.Generator().manual_seed(50)
    pe_pos = torch.randn(1, 64, 256, generator=g).to(BF)
    pe_neg = torch.randn(1, 64, 256, generator=g).to(BF)
    img = (torch.rand(1, 3, 1, 128, 128, generator=g) * 2 - 1).to(BF)
    noise = _Noise(7, dev)
    pj = tmp_path / "p.json"
    frames = generate_video(prompt="x", pipeline=PipelineType.DEV, height=128, width=128, num_frames=9,
                            num_inference_steps=2, cfg_scale=4.0, transformer=m["transformer"], vae_decoder=m["vae_decoder"],
                            vae_encoder=m["vae_encoder"], prompt_embeds=pe_pos, negative_prompt_embeds=pe_neg, noise_fn=noise,
                            images=[(img, 0, 1.0)] if i2v else None, compile_step=True, cfg_batch=True, device=dev,
                            profile_json_path=str(pj), output_path=str(tmp_path / "o.npy"))
    assert frames.shape == (9, 128, 128, 3) and frames.dtype == np.uint8
    assert np.array_equal(np.load(tmp_path / "o.npy"), frames)
    prof = json.loads(pj.read_text())
    assert prof["pipeline"] == "dev" and {"dev_denoise", "vae_decode", "to_uint8_numpy"} <= set(prof["phases_s"])
    # ---- oracle, same noise ----
    p = O.BF16
    sig = ltx2_scheduler(2, 2 * 4 * 4)
    pos = O.create_position_grid(1, 2, 4, 4)
    n0 = noise.log[0].float()
    clean = mask = None
    lat0 = n0
    if i2v:
        z = OV.vae_encode(img.float(), m["We"], p, m["blocks"])
        from oracle import sched as S
        l0, clean, mask = S.apply_conditioning(torch.zeros(1, 128, 2, 4, 4), torch.zeros(1, 128, 2, 4, 4), torch.ones(1, 1, 2, 1, 1),
                                               [("replace", z, 0, 1.0)])
        sm = p.r(mask * O.bf16_round_scalar(float(sig[0])))
        lat0 = p.r(p.r(n0 * sm) + p.r(l0 * p.r(1.0 - sm)))
    lat = O.denoise_dev(lat0, pos, pe_pos.float(), pe_neg.float(), m["W"], m["cfg"], sig.tolist(), p, 4.0, clean, mask, compiled=True)
    vid = OV.vae_decode(lat, m["Wd"], p, layers_per_block=1)
    ref = OV.to_uint8(vid[0], p).numpy()
    d = np.abs(frames.astype(np.int32) - ref.astype(np.int32))
    parity.check(f"pipeline.dev_128x128x9_{'i2v' if i2v else 't2v'}.uint8_mean_abs_diff", float(d.mean()), 1.5)
    parity.check(f"pipeline.dev_128x128x9_{'i2v' if i2v else 't2v'}.uint8_p99_abs_diff", float(np.percentile(d, 99)), 6)


@pytest.mark.parametrize("pipe", ["distilled", "keyframe", "ic_lora"])
def test_two_stage_pipelines_run(dev, pipe):
    from mlx_video_amd.generate import PipelineType, generate_video
    m = _mods(dev)
    g = torch.Generator().manual_seed(51)
    emb = torch.randn(1, 64, 256, generator=g).to(BF)
    kw = {}
    if pipe == "keyframe":
        kw["images"] = [((torch.rand(1, 3, 1, 128, 128, generator=g) * 2 - 1).to(BF), 0, 1.0),
                        ((torch.rand(1, 3, 1, 128, 128, generator=g) * 2 - 1).to(BF), 8, 0.8)]
    if pipe == "ic_lora":
        kw["video_conditionings"] = [((torch.rand(1, 3, 9, 128, 128, generator=g) * 2 - 1).to(BF), 0, 1.0)]
    frames = generate_video(prompt="x", pipeline=PipelineType(pipe), height=128, width=128, num_frames=9, stage1_steps=2,
                            stage2_steps=1, transformer=m["transformer"], vae_decoder=m["vae_decoder"], vae_encoder=m["vae_encoder"],
                            upsampler=m["upsampler"], prompt_embeds=emb, device=dev, seed=3, **kw)
    assert frames.shape == (9, 128, 128, 3) and frames.dtype == np.uint8
    assert 5 < frames.mean() < 250
    # hoist_context: the text-only part of the forward computed once per denoise call - the same kernels on the same inputs
    hoisted = generate_video(prompt="x", pipeline=PipelineType(pipe), height=128, width=128, num_frames=9, stage1_steps=2,
                             stage2_steps=1, transformer=m["transformer"], vae_decoder=m["vae_decoder"], vae_encoder=m["vae_encoder"],
                             upsampler=m["upsampler"], prompt_embeds=emb, device=dev, seed=3, hoist_context=True, **kw)
    assert np.array_equal(hoisted, frames)


def test_pipeline_surface_and_errors(dev):
    from mlx_video_amd.generate import PipelineType, _pad_dims, _resolve_frame_idx, _round_frames, generate_video
    from mlx_video_amd.pipelines import TI2VidOneStagePipeline
    assert _round_frames(33) == 33 and _round_frames(30) == 33 and _round_frames(34) == 41      # round UP (generate.py:2261-2266)
    assert _pad_dims(480, 832, 64) == (512, 832, (16, 0, 480, 832)) and _pad_dims(512, 512, 32) == (512, 512, None)
    assert _resolve_frame_idx(3, 33, 5) == 3 and _resolve_frame_idx(32, 33, 5) == 4 and _resolve_frame_idx(16, 33, 5) == 2
    m = _mods(dev)
    emb = torch.zeros(1, 64, 256, dtype=BF)
    with pytest.raises(ValueError, match="IC-LoRA pipeline requires"):
        generate_video(pipeline=PipelineType.IC_LORA, transformer=m["transformer"], vae_decoder=m["vae_decoder"], prompt_embeds=emb)
    with pytest.raises(ValueError, match="only supported in ic_lora/distilled"):
        generate_video(pipeline=PipelineType.DEV, transformer=m["transformer"], vae_decoder=m["vae_decoder"], prompt_embeds=emb,
                       video_conditionings=[(torch.zeros(1, 3, 9, 64, 64), 0, 1.0)])
    with pytest.raises(ValueError, match="prompt_embeds is required"):
        generate_video(pipeline=PipelineType.DEV, transformer=m["transformer"], vae_decoder=m["vae_decoder"])
    with pytest.raises(ValueError, match="audio"):
        generate_video(pipeline=PipelineType.DEV, audio=True, transformer=m["transformer"], vae_decoder=m["vae_decoder"], prompt_embeds=emb)
    # padded dims: 100x120 -> 128x128 internally, cropped back
    pipe = TI2VidOneStagePipeline(height=100, width=120, num_frames=9, steps=1)
    fr = pipe("x", output_path=None, transformer=m["transformer"], vae_decoder=m["vae_decoder"], prompt_embeds=emb, device=dev)
    assert fr.shape == (9, 100, 120, 3)


def _write_checkpoint(root, m, with_upsampler=True):
    """A synthetic LTX-2 checkpoint DIRECTORY in the PyTorch naming / conv layout the reference converts from
    (ltx.py:508-533, decoder.py:675-721, encoder.py:135-179): `model.diffusion_model.*`, `vae.decoder.*`, `vae.encoder.*`,
    per-channel statistics under their PyTorch names, conv weights (O,I,D,H,W), `config` metadata."""
    from safetensors.torch import save_file
    sd = {}
    for k, v in m["W"].items():
        kk = k.replace(".to_out.", ".to_out.0.").replace(".ff.proj_in.", ".ff.net.0.proj.").replace(".ff.proj_out.", ".ff.net.2.")
        kk = kk.replace(".linear1.", ".linear_1.").replace(".linear2.", ".linear_2.")
        sd["model.diffusion_model." + kk] = v.contiguous()
    sd["model.diffusion_model.audio_embeddings_connector.dummy"] = torch.zeros(4, dtype=BF)      # must be skipped (ltx.py:516-518)
    for k, v in m["Wd"].items():
        if k == "latents_mean":
            sd["vae.per_channel_statistics.mean-of-means"] = v.contiguous()
        elif k == "latents_std":
            sd["vae.per_channel_statistics.std-of-means"] = v.contiguous()
        else:
            sd["vae.decoder." + k] = (v.permute(0, 4, 1, 2, 3) if v.ndim == 5 else v).contiguous()   # (O,D,H,W,I) -> (O,I,D,H,W)
    for k, v in m["We"].items():
        if k.startswith("per_channel_statistics"):
            continue                                        # shared with the decoder's entries above in real checkpoints
        sd["vae.encoder." + k] = (v.permute(0, 4, 1, 2, 3) if v.ndim == 5 else v).contiguous()
    root.mkdir(parents=True, exist_ok=True)
    save_file(sd, str(root / "ltx-2-synthetic.safetensors"), metadata={"config": json.dumps({"vae": {"timestep_conditioning": False}})})
    if with_upsampler:
        su = {}
        for k, v in m["Wu"].items():
            su[k] = (v.permute(0, 4, 1, 2, 3) if v.ndim == 5 else (v.permute(0, 3, 1, 2) if v.ndim == 4 else v)).contiguous()
        save_file(su, str(root / "ltx-2-spatial-upscaler-x2.safetensors"))


def test_model_repo_directory_end_to_end(dev, tmp_path):
    """weights.load_pipeline_modules from a checkpoint directory (header scan, config inferred from shapes, tensors streamed to
    the device one by one) must give the SAME frames as the modules built directly from the same tensors - dev I2V at
    128x128x9 (needs transformer + decoder + encoder) and the distilled two-stage pipeline (needs the upsampler file)."""
    from mlx_video_amd.generate import PipelineType, generate_video
    from mlx_video_amd.weights import load_pipeline_modules, scan_header
    m = _mods(dev)
    # the encoder's statistics are the decoder's in a real checkpoint: make the direct modules agree with that
    We = dict(m["We"])
    We["per_channel_statistics.mean"], We["per_channel_statistics.std"] = m["Wd"]["latents_mean"], m["Wd"]["latents_std"]
    from mlx_video_amd.video_vae import VideoEncoder
    enc = VideoEncoder({k: v.to(dev) for k, v in We.items()}, encoder_blocks=m["blocks"])
    repo = tmp_path / "repo"
    _write_checkpoint(repo, m)
    hdr = scan_header(repo / "ltx-2-synthetic.safetensors")
    assert "__metadata__" in hdr and len(hdr) > 100
    mods = load_pipeline_modules(str(repo), dev, need_encoder=True, need_upsampler=True)
    tc = mods["transformer_config"]
    assert (tc.num_layers, tc.num_attention_heads, tc.caption_channels, tc.in_channels) == (2, 4, 256, 128)
    assert mods["vae_decoder"].timestep_conditioning is False and "upsampler" in mods and "vae_encoder" in mods
    assert all(v.is_cuda and v.dtype == BF for v in mods["transformer_weights"].values())
    g = torch.Generator().manual_seed(51)
    pe_pos = torch.randn(1, 64, 256, generator=g).to(BF)
    pe_neg = torch.randn(1, 64, 256, generator=g).to(BF)
    img = (torch.rand(1, 3, 1, 128, 128, generator=g) * 2 - 1).to(BF)
    common = dict(prompt="x", height=128, width=128, num_frames=9, prompt_embeds=pe_pos, negative_prompt_embeds=pe_neg,
                  images=[(img, 0, 1.0)], compile_step=True, cfg_batch=True, device=dev, cfg_scale=4.0)
    for pipe, extra in ((PipelineType.DEV, dict(num_inference_steps=2)), (PipelineType.DISTILLED, dict(stage1_steps=2, stage2_steps=1))):
        a = generate_video(model_repo=str(repo), pipeline=pipe, noise_fn=_Noise(9, dev), **common, **extra)
        b = generate_video(pipeline=pipe, transformer=m["transformer"], vae_decoder=m["vae_decoder"], vae_encoder=enc,
                           upsampler=m["upsampler"], noise_fn=_Noise(9, dev), **common, **extra)
        assert a.shape == b.shape == (9, 128, 128, 3)
        assert np.array_equal(a, b), f"{pipe}: frames from the checkpoint directory differ from the directly built modules"
    with pytest.raises(FileNotFoundError):
        load_pipeline_modules(str(tmp_path / "nope"), dev)


def test_cli_ic_lora_from_paths(dev, tmp_path):
    """`python -m mlx_video_amd.generate --pipeline ic_lora` with everything given as PATHS (generate.py:4346-4415, 4667-4698):
    a checkpoint directory, an image file, a directory of conditioning frames, a LoRA safetensors file, prompt embeddings as
    .npy; the frames it writes equal generate_video() called with the loaded modules and the same path arguments."""
    from PIL import Image
    from safetensors.torch import save_file
    from mlx_video_amd import generate as G
    from mlx_video_amd.weights import load_pipeline_modules
    m = _mods(dev)
    repo = tmp_path / "repo"
    _write_checkpoint(repo, m)
    rng = np.random.default_rng(3)
    Image.fromarray(rng.integers(0, 255, (128, 128, 3), dtype=np.uint8)).save(tmp_path / "first.png")
    fdir = tmp_path / "guide"
    fdir.mkdir()
    for i in range(9):
        Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8)).save(fdir / f"f{i:03d}.png")
    g = torch.Generator().manual_seed(61)
    sd = {}
    for i in range(2):
        for raw in ("attn1.to_q", "attn1.to_out.0", "ff.net.0.proj"):
            o, n_in = m["W"][f"transformer_blocks.{i}." + raw.replace("to_out.0", "to_out").replace("ff.net.0.proj", "ff.proj_in") + ".weight"].shape
            sd[f"diffusion_model.transformer_blocks.{i}.{raw}.lora_A.weight"] = (torch.randn(8, n_in, generator=g) * 0.1).to(BF)
            sd[f"diffusion_model.transformer_blocks.{i}.{raw}.lora_B.weight"] = (torch.randn(o, 8, generator=g) * 0.1).to(BF)
    save_file(sd, str(tmp_path / "ic.safetensors"))
    np.save(tmp_path / "pe.npy", torch.randn(1, 64, 256, generator=g).numpy().astype(np.float32))
    out = tmp_path / "cli.npy"
    pj = tmp_path / "cli.json"
    G.main(["--pipeline", "ic_lora", "--model-repo", str(repo), "--height", "128", "--width", "128", "--num-frames", "9",
            "--stage1-steps", "2", "--stage2-steps", "1", "--seed", "11", "--image", str(tmp_path / "first.png"), "0", "1.0",
            "--video-conditioning", str(fdir), "0", "0.9", "--lora", str(tmp_path / "ic.safetensors"), "0.8",
            "--prompt-embeds", str(tmp_path / "pe.npy"), "--output-path", str(out), "--profile-json", str(pj), "--tiling", "none"])
    frames = np.load(out)
    assert frames.shape == (9, 128, 128, 3) and frames.dtype == np.uint8
    prof = json.loads(pj.read_text())
    assert prof["pipeline"] == "ic_lora" and {"cond_encode", "stage1_denoise", "upsample", "stage2_denoise", "vae_decode"} <= set(prof["phases_s"])
    mods = load_pipeline_modules(str(repo), dev, need_encoder=True, need_upsampler=True, build_transformer=False)
    ref = G.generate_video(prompt="", pipeline=G.PipelineType.IC_LORA, height=128, width=128, num_frames=9, stage1_steps=2, stage2_steps=1,
                           seed=11, images=[(str(tmp_path / "first.png"), 0, 1.0)], video_conditionings=[(str(fdir), 0, 0.9)],
                           loras=[(str(tmp_path / "ic.safetensors"), 0.8)], transformer_weights=mods["transformer_weights"],
                           transformer_config=mods["transformer_config"], vae_decoder=mods["vae_decoder"], vae_encoder=mods["vae_encoder"],
                           upsampler=mods["upsampler"], prompt_embeds=torch.from_numpy(np.load(tmp_path / "pe.npy")), device=dev, tiling="none")
    assert np.array_equal(frames, ref)
    # and the LoRA is really applied: without it the frames differ
    base = G.generate_video(prompt="", pipeline=G.PipelineType.IC_LORA, height=128, width=128, num_frames=9, stage1_steps=2, stage2_steps=1,
                            seed=11, images=[(str(tmp_path / "first.png"), 0, 1.0)], video_conditionings=[(str(fdir), 0, 0.9)],
                            transformer_weights=mods["transformer_weights"], transformer_config=mods["transformer_config"],
                            vae_decoder=mods["vae_decoder"], vae_encoder=mods["vae_encoder"], upsampler=mods["upsampler"],
                            prompt_embeds=torch.from_numpy(np.load(tmp_path / "pe.npy")), device=dev, tiling="none")
    assert not np.array_equal(frames, base)

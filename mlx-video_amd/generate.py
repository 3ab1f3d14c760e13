"""Pipeline driver with the surface of ``mlx_video.generate`` (generate.py:2035-4197, main() 4200-4758):
``generate_video(...)`` for the distilled / dev / keyframe / ic_lora pipelines and a CLI.

What is kept: argument names and meaning, dimension padding (/64 distilled-style, /32 dev) with crop
back, frames -> 1+8k round-up, sigma schedules and subsampling, conditioning semantics (replace /
guide, per-pipeline forcing), two-stage flow (half-res stage 1 -> latent upsample -> re-noise ->
stage 2, optional stage-2 LoRA transformer), tiling policy, the `_PhaseTimer` phase names
(`stage1_denoise`, `upsample`, `stage2_denoise`, `dev_denoise`, `vae_decode`, `to_uint8_numpy`) and
the `--profile-json` payload.

What is injected instead of loaded (nothing exists offline and they are out of scope, SURVEY.md §2a
#17/#21): text embeddings (the Gemma-3 text encoder's output tensor is an INPUT of this path), model
weights (`weights=` dicts or ready modules; `model_repo` may point at a local directory of
safetensors read by ``weights.py``), random draws (`noise_fn`: MLX's threefry stream is not
reproducible, so seeds are not cross-compatible), audio (not supported).  Frames are returned as the reference returns them (uint8 (F,H,W,3));
`output_path` accepts `.npy`, or a video file written through an ffmpeg child process when an ffmpeg binary
is installed (media.write_video_ffmpeg); audio muxing is out of scope.
"""
from __future__ import annotations

import argparse
import json
import time
from contextlib import contextmanager
from enum import Enum
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .conditioning import (LatentState, VideoConditionByKeyframeIndex, VideoConditionByLatentIndex, apply_conditioning,
                           noise_blend)
from . import ops
from .denoise import denoise_dev, denoise_distilled
from .ltx_model import LTXModel, LTXModelConfig
from .schedulers import (STAGE_1_SIGMAS, STAGE_2_SIGMAS, _subsample_refinement_sigmas, _subsample_sigmas,
                         create_position_grid, ltx2_scheduler)
from .video_vae import LTX2VideoDecoder, TilingConfig, VideoEncoder, to_uint8_frames

BF16 = torch.bfloat16
DEFAULT_NEGATIVE_PROMPT = ""


class PipelineType(Enum):
    """generate.py:299-304."""
    DISTILLED = "distilled"
    DEV = "dev"
    KEYFRAME = "keyframe"
    IC_LORA = "ic_lora"


class _PhaseTimer:
    """generate.py:64-94 (phases are closed with a device sync so they mean GPU time)."""

    def __init__(self, enabled: bool):
        self.enabled = enabled
        self.times_s: Dict[str, float] = {}

    @contextmanager
    def phase(self, name: str):
        if not self.enabled:
            yield
            return
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            yield
        finally:
            torch.cuda.synchronize()
            self.times_s[name] = self.times_s.get(name, 0.0) + (time.perf_counter() - t0)

    def render(self, elapsed_s: float) -> str:
        if not self.times_s:
            return ""
        total = sum(self.times_s.values())
        denom = total if total > 0 else (elapsed_s if elapsed_s > 0 else 1.0)
        lines = [f"{n:>18}: {dt:6.2f}s ({100.0 * dt / denom:5.1f}%)"
                 for n, dt in sorted(self.times_s.items(), key=lambda kv: kv[1], reverse=True)]
        lines += [f"{'total(phases)':>18}: {total:6.2f}s", f"{'elapsed(wall)':>18}: {elapsed_s:6.2f}s"]
        return "\n".join(lines)


def _default_noise_fn(seed: int, device) -> Callable[[Tuple[int, ...]], torch.Tensor]:
    g = torch.Generator(device=device).manual_seed(int(seed))
    return lambda shape: torch.randn(shape, generator=g, device=device, dtype=torch.float32).to(BF16)


def _pad_dims(height: int, width: int, divisor: int):
    """generate.py:2238-2259: pad to a multiple of `divisor`, remember the crop."""
    if height % divisor == 0 and width % divisor == 0:
        return height, width, None
    ph, pw = (-height) % divisor, (-width) % divisor
    top, left = ph // 2, pw // 2
    return height + ph, width + pw, (top, left, height, width)


def _round_frames(num_frames: int) -> int:
    """generate.py:2261-2266: round UP to 1 + 8k."""
    return num_frames if num_frames % 8 == 1 else ((num_frames - 1 + 7) // 8) * 8 + 1


def _resolve_frame_idx(frame_idx: int, num_frames: int, latent_frames: int) -> int:
    """generate.py:2612-2619."""
    if frame_idx < latent_frames:
        return frame_idx
    if num_frames <= 1 or latent_frames <= 1:
        return 0
    return int(max(0, min(latent_frames - 1, int((frame_idx / (num_frames - 1) * (latent_frames - 1)) + 0.5))))


def _cond_pixels(src, height: int, width: int, is_video: bool, num_frames: int, device=None) -> torch.Tensor:
    """One conditioning source at one stage's resolution -> (1,3,F,height,width) in [-1,1].
    Paths go through media.load_image / load_frames on the host (utils.py:529-613: LANCZOS for images, an area filter on
    the decoded uint8 frames for video, frame_cap=num_frames).  Pixel tensors (1,3,F,H,W) in [-1,1] of another size are
    resized the way prepare_image_for_encoding / prepare_video_for_encoding do (utils.py:643-715): images through uint8 +
    LANCZOS on the host (one frame); VIDEO frames with cv2.INTER_AREA on the float frames - here the HIP kernel
    ltxk_resize_area on the device (65 frames of 768x768 cost 1.0 s as a host loop, config 5), bf16 out."""
    from . import media
    if isinstance(src, (str, Path)):
        if is_video:
            return media.frames_to_conditioning(media.load_frames(src, height, width, frame_cap=num_frames))
        return media.image_to_conditioning(media.load_image(src, height=height, width=width))
    if isinstance(src, np.ndarray):
        src = torch.from_numpy(src)
    if not torch.is_tensor(src) or src.dim() != 5 or src.shape[1] != 3:
        raise ValueError("conditioning source must be a path, or a (1,3,F,H,W) pixel tensor in [-1,1]")
    t = src.detach()
    if is_video:
        t = t[:, :, :num_frames]
        t = t[:, :, : 1 + ((t.shape[2] - 1) // 8) * 8]            # the encoder takes 1+8k frames (video_vae.py:332-337)
    if tuple(t.shape[-2:]) == (height, width):
        return t
    if is_video and height <= t.shape[-2] and width <= t.shape[-1] and device is not None:
        td = t.to(device)
        return ops.resize_area(td if td.dtype in (torch.float32, BF16) else td.float(), height, width)
    t = t.to("cpu", torch.float32)
    frames01 = ((t[0].permute(1, 2, 3, 0) + 1.0) / 2.0).clamp(0, 1).numpy()        # (F,H,W,3) in [0,1]
    return media.resize_conditioning(frames01, height, width, is_video)


def _encode_conditionings(items, encoder: Optional[VideoEncoder], height: int, width: int, num_frames: int,
                          latent_frames: int, guide: bool, device, is_video: bool = False):
    """items: (path | pixels (1,3,F,H,W) in [-1,1] | latent (1,128,f,h,w), frame_idx, strength).  Pixels are
    brought to the stage's resolution and encoded with the VAE encoder (generate.py:3064-3113)."""
    out = []
    for src, frame_idx, strength in items:
        if torch.is_tensor(src) and src.dim() == 5 and src.shape[1] == 128:
            t = src.to(device).to(BF16)                                             # already a latent
        else:
            if encoder is None:
                raise ValueError("pixel conditioning needs a VAE encoder (pass vae_encoder= or a model_repo with VAE encoder weights)")
            t = encoder(_cond_pixels(src, height, width, is_video, num_frames, device).to(device).to(BF16))
        idx = _resolve_frame_idx(int(frame_idx), num_frames, latent_frames)
        out.append(VideoConditionByKeyframeIndex(t, idx, float(strength)) if guide
                   else VideoConditionByLatentIndex(t, idx, float(strength)))
    return out


def generate_video(model_repo: Optional[str] = None, text_encoder_repo: Optional[str] = None, prompt: str = "",
                   pipeline: PipelineType = PipelineType.DISTILLED, negative_prompt: str = DEFAULT_NEGATIVE_PROMPT,
                   height: int = 512, width: int = 512, num_frames: int = 33, num_inference_steps: int = 40,
                   cfg_scale: float = 4.0, seed: int = 42, fps: float = 24.0, output_path: Optional[str] = None,
                   save_frames: bool = False, verbose: bool = False, profile: bool = False,
                   profile_json_path: Optional[str] = None, image=None, image_strength: float = 1.0,
                   image_frame_idx: int = 0, images: Optional[list] = None, video_conditionings: Optional[list] = None,
                   distilled_loras: Optional[list] = None, conditioning_mode: str = "replace", tiling: str = "auto",
                   stream: bool = False, audio: bool = False, eval_interval: int = 1, compile_step: bool = False,
                   compile_shapeless: bool = False, cfg_batch: bool = False, fp32_euler: bool = True,
                   loras: Optional[list] = None, stage2_dev: bool = False, stage1_steps: int = 8, stage2_steps: int = 3,
                   sigma_subsample: str = "farthest",
                   # ---- injected in place of what the reference downloads / samples ----
                   transformer: Optional[LTXModel] = None, stage2_transformer: Optional[LTXModel] = None,
                   transformer_weights: Optional[Dict[str, torch.Tensor]] = None, transformer_config: Optional[LTXModelConfig] = None,
                   vae_decoder: Optional[LTX2VideoDecoder] = None, vae_encoder: Optional[VideoEncoder] = None,
                   upsampler=None, prompt_embeds: Optional[torch.Tensor] = None,
                   negative_prompt_embeds: Optional[torch.Tensor] = None, text_encoder: Optional[Callable] = None,
                   noise_fn: Optional[Callable] = None, device=None, on_frames_ready: Optional[Callable] = None,
                   return_latents: bool = False, lora_in_place: Optional[bool] = None,
                   hoist_context: bool = False) -> np.ndarray:
    """See the module docstring.  Returns uint8 frames (F,H,W,3) (generate.py:4195-4197).
    ``hoist_context`` (not in the reference, off by default): the part of the forward that depends on the text context only -
    caption projection and the 48 cross-attention K / V^T projections, 3.37 TFLOP that the reference recomputes in every forward
    (ltx.py:77-89, attention.py:123-126) - is computed once per denoise call and reused by every step: the same kernels on the
    same inputs, hence the same latents bit for bit, 8-10 % less work per dev step at 512x512x33."""
    t_start = time.perf_counter()
    if isinstance(pipeline, str):
        pipeline = PipelineType(pipeline)
    if audio:
        raise ValueError("audio generation is not supported (audio branch is out of scope, SURVEY.md §2a #21)")
    images_list = list(images or [])
    if image is not None:
        images_list.append((image, image_frame_idx, image_strength))
    video_conditionings = list(video_conditionings or [])
    if conditioning_mode not in ("replace", "guide"):
        raise ValueError(f"Unknown conditioning_mode: {conditioning_mode}")
    if pipeline == PipelineType.KEYFRAME:                         # generate.py:2223-2225
        conditioning_mode = "guide"
    if pipeline == PipelineType.IC_LORA:                          # generate.py:2226-2229
        if not video_conditionings:
            raise ValueError("IC-LoRA pipeline requires --video-conditioning PATH [FRAME_IDX] STRENGTH")
        conditioning_mode = "replace"
    is_distilled = pipeline in (PipelineType.DISTILLED, PipelineType.KEYFRAME, PipelineType.IC_LORA)
    if pipeline == PipelineType.DEV and video_conditionings:     # generate.py:2234-2235
        raise ValueError("Video conditioning is only supported in ic_lora/distilled pipelines.")
    if sigma_subsample not in ("uniform", "farthest"):
        raise ValueError(f"Unknown sigma subsample method: {sigma_subsample}")

    out_h, out_w = height, width
    height, width, crop = _pad_dims(height, width, 64 if is_distilled else 32)
    num_frames = _round_frames(num_frames)
    latent_frames = 1 + (num_frames - 1) // 8

    if is_distilled:                                               # generate.py:3054-3057
        if stage1_steps < 1 or stage1_steps > (len(STAGE_1_SIGMAS) - 1):
            raise ValueError("--stage1-steps must be between 1 and 8.")
        if stage2_steps not in (1, 2, 3):
            raise ValueError("--stage2-steps must be 1, 2, or 3.")
    if stage2_transformer is not None and distilled_loras:          # generate.py:3210-3211
        raise ValueError("--stage2-model-repo cannot be combined with --distilled-lora (stage-2 LoRA).")

    def _with_loras(lora_list, what: str) -> LTXModel:
        """generate.py:2957-3031 (_load_transformer_with_loras), merge mode: base weights + the listed LoRAs."""
        if transformer_weights is None:
            raise ValueError(f"{what} were given but the base transformer weights are not reachable: pass "
                             "transformer_weights= (the dict the transformer was built from) or a model_repo")
        from .lora import LoraSpec, apply_lora_to_weights
        merged = apply_lora_to_weights(transformer_weights, [LoraSpec(Path(pth), float(st)) for pth, st in lora_list], verbose=verbose)
        return LTXModel(transformer_config or (transformer.config if transformer is not None else LTXModelConfig()), merged)

    # LoRAs may be merged INTO the model when nobody needs its un-merged weights again (`loras` alone: one merged model serves every
    # stage; `distilled_loras` alone: the stage-1 model is dead when stage 2 starts): by default only when the weights are loaded
    # here (this call owns them); a caller that passes its own model opts in explicitly
    if lora_in_place is None:
        lora_in_place = transformer is None and transformer_weights is None
    if transformer is None and transformer_weights is None:
        if model_repo is None:
            raise FileNotFoundError("no transformer: pass transformer= or a local model_repo directory with LTX-2 safetensors")
        from .weights import load_pipeline_modules
        mods = load_pipeline_modules(model_repo, device or torch.device("cuda:0"), need_encoder=bool(images_list or video_conditionings),
                                     need_upsampler=is_distilled, build_transformer=False)
        transformer_weights, transformer_config = mods["transformer_weights"], mods["transformer_config"]
        vae_decoder = vae_decoder or mods["vae_decoder"]
        vae_encoder, upsampler = vae_encoder or mods.get("vae_encoder"), upsampler or mods.get("upsampler")
    if loras:
        if lora_in_place and not distilled_loras:
            # nobody needs the un-merged weights again (stage 2 of a two-stage pipeline runs the same merged model): merge into the
            # model's own panels instead of building a second 21-GB copy (0.5 s of first-touch allocation, 17 GB of HBM)
            if transformer_weights is None:
                raise ValueError("loras were given but the base transformer weights are not reachable: pass transformer_weights= "
                                 "(the dict the transformer was built from) or a model_repo")
            from .lora import LoraSpec, apply_lora_to_weights
            if transformer is None:
                transformer = LTXModel(transformer_config or LTXModelConfig(), transformer_weights)
            apply_lora_to_weights(transformer.weight_views(), [LoraSpec(Path(pth), float(st)) for pth, st in loras], verbose=verbose, in_place=True)
        else:
            transformer = _with_loras(loras, "loras")              # stage 1 / dev: base + loras
    elif transformer is None:
        transformer = LTXModel(transformer_config or LTXModelConfig(), transformer_weights)
    if vae_decoder is None:
        raise FileNotFoundError("no VAE decoder: pass vae_decoder= or a model_repo containing VAE weights")
    dev = device or transformer.tables.device
    if noise_fn is None:
        noise_fn = _default_noise_fn(seed, dev)

    if prompt_embeds is None:
        if text_encoder is None:
            raise ValueError("prompt_embeds is required: the Gemma-3 text encoder is outside this package "
                             "(SURVEY.md §2a #17); pass prompt_embeds=(1,1024,3840) or text_encoder=callable")
        prompt_embeds = text_encoder(prompt)
        negative_prompt_embeds = text_encoder(negative_prompt)
    ctx_pos = prompt_embeds.to(dev).to(BF16)
    ctx_neg = (negative_prompt_embeds if negative_prompt_embeds is not None else torch.zeros_like(prompt_embeds)).to(dev).to(BF16)

    timer = _PhaseTimer(profile or bool(profile_json_path))
    guide = conditioning_mode == "guide"

    def init_state(shape, conds, sigma0):
        """generate.py:3139-3160 / 3430-3449: zeros -> apply_conditioning -> masked noise blend."""
        st = LatentState(torch.zeros(shape, dtype=BF16, device=dev), torch.zeros(shape, dtype=BF16, device=dev),
                         torch.ones((1, 1, shape[2], 1, 1), dtype=BF16, device=dev))
        st = apply_conditioning(st, conds)
        return noise_blend(st, noise_fn(shape), float(sigma0))

    if is_distilled:
        if upsampler is None:
            raise FileNotFoundError("the two-stage pipelines need the latent upsampler (pass upsampler=)")
        s1h, s1w, s2h, s2w = height // 2 // 32, width // 2 // 32, height // 32, width // 32
        sig1 = _subsample_sigmas(list(STAGE_1_SIGMAS), stage1_steps, sigma_subsample)
        sig2 = _subsample_refinement_sigmas(list(STAGE_2_SIGMAS), stage2_steps, sigma_subsample)
        conds1, conds2 = [], []
        if images_list or video_conditionings:
            with timer.phase("cond_encode"):
                # every image is encoded twice, at the half- and the full-resolution stage, with the chosen mode;
                # a video conditioning (IC-LoRA) guides stage 1 only, always as keyframes (generate.py:3073-3110)
                conds1 = _encode_conditionings(images_list, vae_encoder, height // 2, width // 2, num_frames, latent_frames, guide, dev)
                conds2 = _encode_conditionings(images_list, vae_encoder, height, width, num_frames, latent_frames, guide, dev)
                conds1 += _encode_conditionings(video_conditionings, vae_encoder, height // 2, width // 2, num_frames, latent_frames,
                                                True, dev, is_video=True)
        shape1 = (1, 128, latent_frames, s1h, s1w)
        pos1 = create_position_grid(1, latent_frames, s1h, s1w, fps=fps).to(dev)
        state1 = init_state(shape1, conds1, sig1[0]) if conds1 else None
        latents = state1.latent if state1 is not None else noise_fn(shape1)
        with timer.phase("stage1_denoise"):
            latents, _ = denoise_distilled(latents, pos1, ctx_pos, transformer, sig1, state=state1,
                                           compile_step=compile_step, fp32_euler=fp32_euler, use_graph=compile_step, cache_context=hoist_context)
        with timer.phase("upsample"):
            from .upsampler import upsample_latents
            latents = upsample_latents(latents, upsampler, vae_decoder.latents_mean, vae_decoder.latents_std)
        tr2 = stage2_transformer or transformer
        if distilled_loras:                                          # generate.py:3229-3237: base + distilled LoRAs only
            with timer.phase("stage2_transformer_load"):
                if lora_in_place and not loras:
                    # the stage-1 model IS the base model and is not used again: merge into its own panels (same EPI_SCALE_RES
                    # launches with the output aliasing the residual, same bits) instead of building a second 21-GB replica
                    from .lora import LoraSpec, apply_lora_to_weights
                    apply_lora_to_weights(transformer.weight_views(), [LoraSpec(Path(pth), float(st)) for pth, st in distilled_loras],
                                          verbose=verbose, in_place=True)
                    tr2 = transformer
                else:
                    tr2 = _with_loras(distilled_loras, "distilled_loras")
        pos2 = create_position_grid(1, latent_frames, s2h, s2w, fps=fps).to(dev)
        state2 = None
        if conds2:                                                  # generate.py:3290-3311
            st = LatentState(latents, torch.zeros_like(latents), torch.ones((1, 1, latent_frames, 1, 1), dtype=BF16, device=dev))
            state2 = noise_blend(apply_conditioning(st, conds2), noise_fn(tuple(latents.shape)), float(sig2[0]))
            latents = state2.latent
        else:                                                       # generate.py:3317-3321
            s0 = torch.tensor(sig2[0], dtype=BF16, device=dev)
            latents = (noise_fn(tuple(latents.shape)) * s0 + latents * torch.tensor(1.0 - sig2[0], dtype=BF16, device=dev)).to(BF16)
        with timer.phase("stage2_denoise"):
            if stage2_dev:
                latents = denoise_dev(latents, pos2, ctx_pos, ctx_neg, tr2, torch.tensor(sig2), cfg_scale=cfg_scale,
                                      state=state2, compile_step=compile_step, cfg_batch=cfg_batch, use_graph=compile_step, cache_context=hoist_context)
            else:
                latents, _ = denoise_distilled(latents, pos2, ctx_pos, tr2, sig2, state=state2, compile_step=compile_step,
                                               fp32_euler=fp32_euler, use_graph=compile_step, cache_context=hoist_context)
    else:
        lh, lw = height // 32, width // 32
        n_tok = latent_frames * lh * lw
        sigmas = ltx2_scheduler(num_inference_steps, n_tok)          # generate.py:3410-3411
        pos = create_position_grid(1, latent_frames, lh, lw, fps=fps).to(dev)
        shape = (1, 128, latent_frames, lh, lw)
        conds = []
        if images_list:
            with timer.phase("cond_encode"):
                conds = _encode_conditionings(images_list, vae_encoder, height, width, num_frames, latent_frames, guide, dev)
        state = init_state(shape, conds, float(sigmas[0])) if conds else None
        latents = state.latent if state is not None else noise_fn(shape)
        with timer.phase("dev_denoise"):
            latents = denoise_dev(latents, pos, ctx_pos, ctx_neg, transformer, sigmas, cfg_scale=cfg_scale, state=state,
                                  compile_step=compile_step, cfg_batch=cfg_batch, use_graph=compile_step, cache_context=hoist_context)

    if return_latents:
        return latents
    # ---- VAE decode (generate.py:3794-3830) ----
    tcfg = None
    if tiling == "auto":
        tcfg = TilingConfig.auto(height, width, num_frames)
    elif tiling not in ("none", None):
        tcfg = {"default": TilingConfig.default, "aggressive": TilingConfig.aggressive,
                "conservative": TilingConfig.conservative, "spatial": TilingConfig.spatial_only,
                "temporal": TilingConfig.temporal_only}[tiling]()
    # a timestep-conditioned decoder draws fresh noise per decode call, i.e. per tile (decoder.py:381-385)
    dec_noise_fn = noise_fn if vae_decoder.timestep_conditioning else None
    with timer.phase("vae_decode"):
        if tcfg is not None and (tiling != "auto" or stream):
            video = vae_decoder.decode_tiled(latents, tiling_config=tcfg, tiling_mode=tiling, on_frames_ready=on_frames_ready,
                                             noise_fn=dec_noise_fn)
        else:
            # "auto": the reference first tries the non-tiled decode and only falls back on an OOM-looking
            # exception (generate.py:3798-3818); 288 GB of HBM never takes that fallback at these sizes.
            video = vae_decoder(latents, noise_fn=dec_noise_fn)
    with timer.phase("to_uint8_numpy"):
        frames = to_uint8_frames(video)[0]
        video_np = frames.cpu().numpy()
        if crop is not None:
            top, left, oh, ow = crop
            video_np = video_np[:, top:top + oh, left:left + ow, :]
    elapsed = time.perf_counter() - t_start
    if output_path:
        p = Path(output_path)
        p.parent.mkdir(parents=True, exist_ok=True)
        if p.suffix == ".npy":
            np.save(p, video_np)
        else:                                                          # generate.py:3900-3913 (ffmpeg encoder)
            from .media import write_video_ffmpeg
            write_video_ffmpeg(video_np, p, fps)
    if profile and verbose:
        print(timer.render(elapsed))
    if profile_json_path and timer.times_s:                          # generate.py:4158-4189 (same keys)
        outp = Path(output_path).with_suffix(".profile.json") if profile_json_path == "auto" and output_path else Path(profile_json_path)
        payload = {"elapsed_s": float(elapsed), "num_frames": int(num_frames), "fps": float(fps), "pipeline": pipeline.value,
                   "stage1_steps": int(stage1_steps) if is_distilled else None,
                   "stage2_steps": int(stage2_steps) if is_distilled else None, "eval_interval": int(eval_interval),
                   "compile_step": bool(compile_step), "compile_shapeless": bool(compile_shapeless),
                   "fp32_euler": bool(fp32_euler), "audio": False, "phases_s": {k: float(v) for k, v in timer.times_s.items()},
                   "peak_memory_gb": float(torch.cuda.max_memory_allocated() / (1024 ** 3))}
        outp.parent.mkdir(parents=True, exist_ok=True)
        outp.write_text(json.dumps(payload, indent=2, sort_keys=True), encoding="utf-8")
    return video_np


class _ImageConditionAction(argparse.Action):
    """generate.py:4201-4215: --image PATH [FRAME_IDX STRENGTH], repeatable."""

    def __call__(self, parser, namespace, values, option_string=None):
        if len(values) not in (1, 3):
            raise argparse.ArgumentError(self, f"{option_string} accepts 1 or 3 args (PATH [FRAME_IDX STRENGTH]), got {len(values)}")
        item = (values[0], int(values[1]), float(values[2])) if len(values) == 3 else (values[0], None, None)
        setattr(namespace, self.dest, (getattr(namespace, self.dest) or []) + [item])


class _VideoConditionAction(argparse.Action):
    """generate.py:4217-4231: --video-conditioning PATH STRENGTH | PATH FRAME_IDX STRENGTH, repeatable."""

    def __call__(self, parser, namespace, values, option_string=None):
        if len(values) not in (2, 3):
            raise argparse.ArgumentError(self, f"{option_string} accepts PATH STRENGTH or PATH FRAME_IDX STRENGTH")
        item = (values[0], 0, float(values[1])) if len(values) == 2 else (values[0], int(values[1]), float(values[2]))
        setattr(namespace, self.dest, (getattr(namespace, self.dest) or []) + [item])


class _LoraAction(argparse.Action):
    """generate.py:4233-4242: --lora PATH [STRENGTH], repeatable."""

    def __call__(self, parser, namespace, values, option_string=None):
        if len(values) not in (1, 2):
            raise argparse.ArgumentError(self, f"{option_string} accepts PATH or PATH STRENGTH, got {len(values)}")
        item = (values[0], float(values[1]) if len(values) == 2 else 1.0)
        setattr(namespace, self.dest, (getattr(namespace, self.dest) or []) + [item])


def build_parser() -> argparse.ArgumentParser:
    """The subset of generate.py:4244-4525 that drives this path."""
    ap = argparse.ArgumentParser(description="LTX-2 video generation on MI355X (libltxk)")
    ap.add_argument("--image", "-i", action=_ImageConditionAction, nargs="+", metavar="PATH", default=[],
                    help="Image conditioning: --image path.jpg or --image path.jpg FRAME_IDX STRENGTH (repeatable)")
    ap.add_argument("--condition-image", type=str, default=None, help="Alias for --image (frame 0, strength 1.0)")
    ap.add_argument("--image-strength", type=float, default=1.0)
    ap.add_argument("--image-frame-idx", type=int, default=0)
    ap.add_argument("--video-conditioning", action=_VideoConditionAction, nargs="+", metavar="ARG", default=[],
                    help="Video conditioning for IC-LoRA: PATH [FRAME_IDX] STRENGTH (a directory of frames or an .npy array here)")
    ap.add_argument("--reference-video", type=str, default=None, help="Alias for --video-conditioning (frame 0, strength 1.0)")
    ap.add_argument("--lora", "--lora-path", dest="lora", action=_LoraAction, nargs="+", metavar="ARG", default=[],
                    help="LoRA weights to merge (repeatable): --lora path 0.8")
    ap.add_argument("--distilled-lora", action=_LoraAction, nargs="+", metavar="ARG", default=[],
                    help="LoRA(s) for stage-2 refinement (distilled pipelines only)")
    ap.add_argument("--conditioning-mode", choices=["replace", "guide"], default="replace")
    ap.add_argument("--stream", action="store_true", help="Decode tiled and hand frames over as they are finished")
    ap.add_argument("--stage2-dev", action="store_true")
    ap.add_argument("--no-fp32-euler", dest="fp32_euler", action="store_false",
                    default=__import__("os").getenv("LTX_FP32_EULER", "").lower() not in ("0", "false", "no"))
    ap.add_argument("--prompt", type=str, default="")
    ap.add_argument("--negative-prompt", type=str, default=DEFAULT_NEGATIVE_PROMPT)
    ap.add_argument("--pipeline", choices=[p.value for p in PipelineType], default="distilled")
    ap.add_argument("--model-repo", type=str, default=None)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--num-frames", type=int, default=33)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--cfg-scale", type=float, default=4.0)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--fps", type=float, default=24.0)
    ap.add_argument("--output-path", type=str, default="output.npy")
    ap.add_argument("--tiling", type=str, default="auto")
    ap.add_argument("--stage1-steps", type=int, default=None)
    ap.add_argument("--stage2-steps", type=int, default=None)
    ap.add_argument("--sigma-subsample", choices=["uniform", "farthest"], default=__import__("os").getenv("LTX_SIGMA_SUBSAMPLE", "farthest"))
    ap.add_argument("--eval-interval", type=int, default=None, help="Host sync cadence of the denoise loop (default: auto)")
    ap.add_argument("--compile", dest="compile_step", action="store_true", default=False,
                    help="Captured step (the reference's mx.compile'd step: bf16-rounded sigmas in x0 and Euler)")
    ap.add_argument("--no-compile", action="store_true", help="Disable the compiled step even if LTX_COMPILE enables it")
    ap.add_argument("--cfg-batch", action="store_true", default=False, help="CFG pair as one B=2 forward")
    ap.add_argument("--no-cfg-batch", action="store_true", help="Two sequential forwards even if LTX_CFG_BATCH enables batching")
    ap.add_argument("--hoist-context", action="store_true",
                    help="(not in the reference CLI) compute the text-only part of the forward once per denoise call instead of in every step; same latents")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--profile-json", type=str, default=None)
    ap.add_argument("--prompt-embeds", type=str, default=None, help=".pt/.npy file with (1,1024,3840) text embeddings")
    ap.add_argument("--negative-prompt-embeds", type=str, default=None)
    ap.add_argument("--synthetic", action="store_true", help="random-init weights + random text embeddings (no checkpoints offline)")
    ap.add_argument("--layers", type=int, default=48)
    return ap


def resolve_cli_heuristics(args, env=None):
    """The auto rules of generate.py:4545-4560 and 4629-4644 that pick the hot path's flavour: step counts, the host sync
    cadence, compiled-vs-eager step (they differ in the sigma the Euler update sees, generate.py:1170-1174 vs 1293-1301)
    and batched-vs-sequential CFG.  Pure function of the parsed args and the environment (LTX_COMPILE, LTX_CFG_BATCH,
    LTX_EVAL_INTERVAL); mutates and returns ``args``."""
    env = __import__("os").environ if env is None else env
    truthy = lambda name: str(env.get(name, "")).lower() in ("1", "true", "yes")
    is_dev = args.pipeline == "dev"
    if args.stage1_steps is None:
        args.stage1_steps = 5 if args.pipeline == "distilled" else 8
    if args.stage2_steps is None:
        args.stage2_steps = 1 if args.pipeline == "distilled" else 3
    if args.eval_interval is None:
        ev = env.get("LTX_EVAL_INTERVAL")
        args.eval_interval = int(ev) if ev is not None else (2 if is_dev else 4)
    if args.eval_interval < 1:
        args.eval_interval = 1
    repo_l = str(getattr(args, "model_repo", None) or "").lower()
    is_quant_repo = any(tag in repo_l for tag in ("4bit", "8bit", "q4", "q8", "int4", "int8"))
    if getattr(args, "no_compile", False):
        args.compile_step = False
    elif truthy("LTX_COMPILE"):
        args.compile_step = True
    elif not args.compile_step:
        if is_dev and args.steps >= 8:
            args.compile_step = True
        elif (not is_dev) and args.num_frames >= 97 and (args.stage1_steps + args.stage2_steps) >= 5:
            args.compile_step = True
    if getattr(args, "no_cfg_batch", False):
        args.cfg_batch = False
    elif truthy("LTX_CFG_BATCH"):
        args.cfg_batch = True
    elif not args.cfg_batch and is_dev and args.cfg_scale > 1.0 and not is_quant_repo:
        args.cfg_batch = True
    return args


def main(argv: Optional[Sequence[str]] = None) -> None:
    args = resolve_cli_heuristics(build_parser().parse_args(argv))
    is_dev = args.pipeline == "dev"
    dev = torch.device("cuda:0")
    kw = {}
    # generate.py:4667-4694: --image items without an explicit index/strength take --image-frame-idx / --image-strength
    images = [(pth, args.image_frame_idx if fi is None else fi, args.image_strength if st is None else st) for pth, fi, st in args.image]
    if args.condition_image:
        images.append((args.condition_image, 0, 1.0))
    videos = list(args.video_conditioning)
    if args.reference_video:
        videos.append((args.reference_video, 0, 1.0))
    if args.synthetic:
        from .video_vae import random_decoder_weights
        kw["transformer"] = LTXModel.random_init(LTXModelConfig(num_layers=args.layers), dev)
        kw["vae_decoder"] = LTX2VideoDecoder(random_decoder_weights(dev))
        g = torch.Generator(device=dev).manual_seed(43)
        kw["prompt_embeds"] = torch.randn((1, 1024, 3840), generator=g, device=dev).to(BF16)
        kw["negative_prompt_embeds"] = torch.randn((1, 1024, 3840), generator=g, device=dev).to(BF16)
        if not is_dev:
            from .upsampler import LatentUpsampler
            from .weights import random_upsampler_weights
            kw["upsampler"] = LatentUpsampler(random_upsampler_weights(dev))
        if images or videos:
            from .video_vae import random_encoder_weights
            kw["vae_encoder"] = VideoEncoder(random_encoder_weights(dev))
    else:
        def _load(path):
            return torch.from_numpy(np.load(path)) if path.endswith(".npy") else torch.load(path, weights_only=True)
        if args.prompt_embeds:
            kw["prompt_embeds"] = _load(args.prompt_embeds)
        if args.negative_prompt_embeds:
            kw["negative_prompt_embeds"] = _load(args.negative_prompt_embeds)
    generate_video(model_repo=args.model_repo, prompt=args.prompt, pipeline=PipelineType(args.pipeline),
                   negative_prompt=args.negative_prompt, height=args.height, width=args.width, num_frames=args.num_frames,
                   num_inference_steps=args.steps, cfg_scale=args.cfg_scale, seed=args.seed, fps=args.fps,
                   output_path=args.output_path, tiling=args.tiling, compile_step=args.compile_step, cfg_batch=args.cfg_batch, hoist_context=args.hoist_context,
                   eval_interval=args.eval_interval,
                   profile=args.profile, profile_json_path=args.profile_json, stage1_steps=args.stage1_steps,
                   stage2_steps=args.stage2_steps, sigma_subsample=args.sigma_subsample, verbose=True, device=dev,
                   images=images, video_conditionings=videos, loras=args.lora, distilled_loras=args.distilled_lora,
                   conditioning_mode=args.conditioning_mode, stream=args.stream, stage2_dev=args.stage2_dev,
                   fp32_euler=args.fp32_euler, **kw)


if __name__ == "__main__":
    main()

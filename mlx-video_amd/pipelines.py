"""The five pipeline wrappers and the bridge of ``ltx_pipelines`` (ltx_pipelines/{distilled,ic_lora,
keyframe_interpolation,ti2vid_one_stage,ti2vid_two_stages}.py, mlx_bridge.py:15-121): same dataclass
fields and call signatures; ``run_generate`` forwards to this package's ``generate_video``.  Extra
keyword arguments (text embeddings, preloaded modules) pass through ``**inject`` because nothing can be
downloaded here."""
from __future__ import annotations

import sys
from dataclasses import dataclass, fields
from typing import Iterable, Optional

from .generate import PipelineType, generate_video


@dataclass
class MLXPipelineConfig:
    """mlx_bridge.py:15-30 (name kept for drop-in imports)."""
    model_repo: str = "Lightricks/LTX-2"
    text_encoder_repo: Optional[str] = None
    height: int = 512
    width: int = 512
    num_frames: int = 33
    steps: int = 40
    cfg_scale: float = 4.0
    seed: int = 42
    fps: float = 24.0
    audio: bool = False
    verbose: bool = False
    stream: bool = False
    tiling: str = "auto"
    conditioning_mode: str = "replace"


def _ensure_list(v):
    return [] if v is None else list(v)


def _normalize_loras(loras):
    """mlx_bridge.py:39-48."""
    out = []
    for it in _ensure_list(loras):
        if isinstance(it, (list, tuple)) and len(it) == 2:
            out.append((str(it[0]), float(it[1])))
        elif isinstance(it, (list, tuple)) and len(it) == 1:
            out.append((str(it[0]), 1.0))
        else:
            out.append((str(it), 1.0))
    return out


def _normalize_images(images):
    """mlx_bridge.py:51-60 (items may be tensors instead of paths)."""
    out = []
    for it in _ensure_list(images):
        if isinstance(it, (list, tuple)) and len(it) == 3:
            out.append((it[0], int(it[1]), float(it[2])))
        elif isinstance(it, (list, tuple)) and len(it) == 1:
            out.append((it[0], 0, 1.0))
        else:
            out.append((it, 0, 1.0))
    return out


def _normalize_video_conditions(vcs):
    """mlx_bridge.py:63-72."""
    out = []
    for it in _ensure_list(vcs):
        if isinstance(it, (list, tuple)) and len(it) == 3:
            out.append((it[0], int(it[1]), float(it[2])))
        elif isinstance(it, (list, tuple)) and len(it) == 2:
            out.append((it[0], 0, float(it[1])))
        else:
            out.append((it, 0, 1.0))
    return out


def run_generate(prompt: str, pipeline: PipelineType, cfg: MLXPipelineConfig, output_path: Optional[str],
                 images=None, video_conditionings=None, loras=None, distilled_loras=None,
                 negative_prompt: Optional[str] = None, **inject):
    """mlx_bridge.py:75-112.  Returns output_path like the reference (frames when output_path is None)."""
    frames = generate_video(model_repo=cfg.model_repo if inject.get("transformer") is None else None,
                            text_encoder_repo=cfg.text_encoder_repo, prompt=prompt, pipeline=pipeline,
                            negative_prompt=negative_prompt or "", height=cfg.height, width=cfg.width,
                            num_frames=cfg.num_frames, num_inference_steps=cfg.steps, cfg_scale=cfg.cfg_scale,
                            seed=cfg.seed, fps=cfg.fps, output_path=output_path, save_frames=False, verbose=cfg.verbose,
                            image=None, images=_normalize_images(images),
                            video_conditionings=_normalize_video_conditions(video_conditionings),
                            conditioning_mode=cfg.conditioning_mode, tiling=cfg.tiling, stream=cfg.stream, audio=cfg.audio,
                            loras=_normalize_loras(loras), distilled_loras=_normalize_loras(distilled_loras), **inject)
    return output_path if output_path is not None else frames


def run_cli(default_pipeline: str) -> None:
    """mlx_bridge.py:115-121."""
    if "--pipeline" not in sys.argv:
        sys.argv.extend(["--pipeline", default_pipeline])
    from .generate import main as _main
    _main()


@dataclass
class _Base:
    model_repo: str = "Lightricks/LTX-2"
    text_encoder_repo: Optional[str] = None
    height: int = 512
    width: int = 512
    num_frames: int = 33
    steps: int = 40
    cfg_scale: float = 4.0
    seed: int = 42
    fps: float = 24.0
    audio: bool = False
    verbose: bool = False
    stream: bool = False
    tiling: str = "auto"

    def _cfg(self, **over) -> MLXPipelineConfig:
        names = {f.name for f in fields(MLXPipelineConfig)}
        d = {k: getattr(self, k) for k in names if hasattr(self, k)}
        d.update(over)
        return MLXPipelineConfig(**d)


@dataclass
class TI2VidOneStagePipeline(_Base):
    """ti2vid_one_stage.py:13-58: dev pipeline, 40 steps, CFG 4."""

    def __call__(self, prompt: str, output_path: Optional[str] = "output.npy", images=None,
                 negative_prompt: Optional[str] = None, **inject):
        return run_generate(prompt, PipelineType.DEV, self._cfg(), output_path, images=images,
                            negative_prompt=negative_prompt, **inject)


@dataclass
class TI2VidTwoStagesPipeline(_Base):
    """ti2vid_two_stages.py: distilled two-stage with optional distilled LoRAs for stage 2."""
    steps: int = 8
    cfg_scale: float = 1.0

    def __call__(self, prompt: str, output_path: Optional[str] = "output.npy", images=None, distilled_loras=None,
                 negative_prompt: Optional[str] = None, **inject):
        return run_generate(prompt, PipelineType.DISTILLED, self._cfg(), output_path, images=images,
                            distilled_loras=distilled_loras, negative_prompt=negative_prompt, **inject)


@dataclass
class DistilledPipeline(_Base):
    """distilled.py."""
    steps: int = 8
    cfg_scale: float = 1.0

    def __call__(self, prompt: str, output_path: Optional[str] = "output.npy", images=None, loras=None,
                 negative_prompt: Optional[str] = None, **inject):
        return run_generate(prompt, PipelineType.DISTILLED, self._cfg(), output_path, images=images, loras=loras,
                            negative_prompt=negative_prompt, **inject)


@dataclass
class KeyframeInterpolationPipeline(_Base):
    """keyframe_interpolation.py: guiding keyframes (conditioning_mode forced to "guide")."""
    steps: int = 8
    cfg_scale: float = 1.0

    def __call__(self, prompt: str, output_path: Optional[str] = "output.npy", images=None,
                 negative_prompt: Optional[str] = None, **inject):
        return run_generate(prompt, PipelineType.KEYFRAME, self._cfg(conditioning_mode="guide"), output_path, images=images,
                            negative_prompt=negative_prompt, **inject)


@dataclass
class ICLoraPipeline(_Base):
    """ic_lora.py: video conditioning + merged LoRA."""
    steps: int = 8
    cfg_scale: float = 1.0

    def __call__(self, prompt: str, output_path: Optional[str] = "output.npy", video_conditionings=None, images=None,
                 loras=None, negative_prompt: Optional[str] = None, **inject):
        return run_generate(prompt, PipelineType.IC_LORA, self._cfg(), output_path, images=images,
                            video_conditionings=video_conditionings, loras=loras, negative_prompt=negative_prompt, **inject)

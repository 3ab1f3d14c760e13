"""Checkpoint ingestion: safetensors -> bf16 device tensors with the reference's key
maps (ltx.py:508-533 transformer; decoder.py:544-591,675-721 VAE decoder; encoder.py:135-179 VAE encoder;
upsampler.py:319-373), PyTorch conv layouts (O,I,D,H,W) transposed to the MLX layout (O,D,H,W,I) the
kernels consume, and VAE `timestep_conditioning` sniffed from the safetensors metadata
(decoder.py:621-635).  Files are read with safetensors only (nothing is unpickled)."""
from __future__ import annotations

import json
import math
from pathlib import Path
from typing import Dict, Iterable, List, Optional

import torch

BF16 = torch.bfloat16


def _to_dev(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=BF16).contiguous()


def _conv_to_mlx(key: str, v: torch.Tensor) -> torch.Tensor:
    if v.ndim == 5 and "weight" in key:
        return v.permute(0, 2, 3, 4, 1).contiguous()      # (O,I,D,H,W) -> (O,D,H,W,I)
    if v.ndim == 4 and "weight" in key:
        return v.permute(0, 2, 3, 1).contiguous()         # (O,I,H,W)   -> (O,H,W,I)
    return v


def read_safetensors(paths: Iterable[Path]) -> Dict[str, torch.Tensor]:
    from safetensors import safe_open
    out: Dict[str, torch.Tensor] = {}
    for p in paths:
        with safe_open(str(p), framework="pt", device="cpu") as f:
            for k in f.keys():
                out[k] = f.get_tensor(k)
    return out


def read_metadata(path: Path) -> dict:
    from safetensors import safe_open
    with safe_open(str(path), framework="pt", device="cpu") as f:
        return dict(f.metadata() or {})


def transformer_weights(raw: Dict[str, torch.Tensor], device) -> Dict[str, torch.Tensor]:
    from .ltx_model import LTXModel
    san = LTXModel.sanitize(raw) or {k: v for k, v in raw.items() if k.startswith(("transformer_blocks.", "patchify_proj.", "adaln_single.", "caption_projection.", "proj_out.", "scale_shift_table"))}
    return {k: _to_dev(v, device) for k, v in san.items() if not k.startswith(("audio_", "av_ca_")) and ".audio_" not in k
            and "audio_to_video" not in k and "video_to_audio" not in k and "a2v_ca" not in k}


def vae_decoder_weights(raw: Dict[str, torch.Tensor], device) -> Dict[str, torch.Tensor]:
    """decoder.py:675-721: strip `vae.decoder.` / `decoder.`, remap diffusers names, transpose convs,
    per-channel statistics -> latents_mean / latents_std."""
    from .video_vae import LTX2VideoDecoder
    out: Dict[str, torch.Tensor] = {}
    for k, v in raw.items():
        kk = k
        for pre in ("vae.decoder.", "decoder."):
            if kk.startswith(pre):
                kk = kk[len(pre):]
                break
        else:
            if "per_channel_statistics" in k or k in ("latents_mean", "latents_std"):
                kk = k.split("vae.")[-1]
            else:
                continue
        kk = LTX2VideoDecoder.remap_decoder_key(kk)
        out[kk] = _to_dev(_conv_to_mlx(kk, v), device)
    for src, dst in (("per_channel_statistics.mean-of-means", "latents_mean"), ("per_channel_statistics.std-of-means", "latents_std"),
                     ("per_channel_statistics.mean", "latents_mean"), ("per_channel_statistics.std", "latents_std")):
        if src in out and dst not in out:
            out[dst] = out[src]
    return out


def vae_encoder_weights(raw: Dict[str, torch.Tensor], device) -> Dict[str, torch.Tensor]:
    """encoder.py:135-179: strip `vae.encoder.` / `encoder.`, drop the `.conv.` wrapper level, transpose convs."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in raw.items():
        kk = None
        for pre in ("vae.encoder.", "encoder."):
            if k.startswith(pre):
                kk = k[len(pre):]
        if kk is None:
            if "per_channel_statistics" in k:
                kk = "per_channel_statistics." + ("mean" if "mean" in k.split(".")[-1] else "std")
            else:
                continue
        kk = kk.replace(".conv.conv.", ".conv.").replace("conv_in.conv.", "conv_in.").replace("conv_out.conv.", "conv_out.")
        kk = kk.replace(".conv1.conv.", ".conv1.").replace(".conv2.conv.", ".conv2.")
        out[kk] = _to_dev(_conv_to_mlx(kk, v), device)
    return out


def upsampler_weights(raw: Dict[str, torch.Tensor], device) -> Dict[str, torch.Tensor]:
    """upsampler.py:345-367."""
    return {k: _to_dev(_conv_to_mlx(k, v) if "conv" in k else v, device) for k, v in raw.items()}


def sniff_timestep_conditioning(path: Path) -> bool:
    """decoder.py:621-635."""
    try:
        cfg = json.loads(read_metadata(path).get("config", "{}"))
        return bool(cfg.get("vae", {}).get("timestep_conditioning", False))
    except Exception:
        return False


def load_pipeline_modules(model_repo: str, device, need_encoder: bool = False, need_upsampler: bool = False,
                          loras: Optional[list] = None) -> dict:
    """Local-directory loader (no network: repo *names* are not resolved, utils.py:78-374 is out of scope)."""
    from .ltx_model import LTXModel, LTXModelConfig
    from .video_vae import LTX2VideoDecoder, VideoEncoder
    root = Path(model_repo)
    if not root.exists():
        raise FileNotFoundError(f"model_repo {model_repo!r} is not a local directory; HF downloads are not available offline")
    files = sorted(root.glob("*.safetensors"))
    if not files:
        raise FileNotFoundError(f"no .safetensors under {root}")
    main = [f for f in files if "upscaler" not in f.name and "upsampler" not in f.name]
    raw = read_safetensors(main)
    tw = transformer_weights(raw, device)
    if loras:
        from .lora import LoraSpec, apply_lora_to_weights
        tw = apply_lora_to_weights(tw, [LoraSpec(Path(p), float(s)) for p, s in loras])
    mods = {"transformer": LTXModel(LTXModelConfig(), tw)}
    mods["vae_decoder"] = LTX2VideoDecoder(vae_decoder_weights(raw, device), timestep_conditioning=sniff_timestep_conditioning(main[0]))
    if need_encoder:
        mods["vae_encoder"] = VideoEncoder(vae_encoder_weights(raw, device))
    ups = [f for f in files if "upscaler" in f.name or "upsampler" in f.name]
    if need_upsampler and ups:
        from .upsampler import LatentUpsampler
        mods["upsampler"] = LatentUpsampler(upsampler_weights(read_safetensors(ups[:1]), device))
    return mods


def random_upsampler_weights(device, mid: int = 1024, seed: int = 99, nb: int = 4) -> Dict[str, torch.Tensor]:
    g = torch.Generator(device=device).manual_seed(seed)
    W: Dict[str, torch.Tensor] = {}

    def rn(*shape, std=1.0, mean=0.0):
        return (torch.randn(shape, generator=g, device=device) * std + mean).to(BF16)

    def conv(name, o, i):
        W[f"{name}.weight"], W[f"{name}.bias"] = rn(o, 3, 3, 3, i, std=1.0 / math.sqrt(27 * i)), rn(o, std=0.01)

    def norm(name, c):
        W[f"{name}.weight"], W[f"{name}.bias"] = rn(c, std=0.1, mean=1.0), rn(c, std=0.1)

    conv("initial_conv", mid, 128); norm("initial_norm", mid)
    for stage in ("res_blocks", "post_upsample_res_blocks"):
        for i in range(nb):
            conv(f"{stage}.{i}.conv1", mid, mid); norm(f"{stage}.{i}.norm1", mid)
            conv(f"{stage}.{i}.conv2", mid, mid); norm(f"{stage}.{i}.norm2", mid)
    W["upsampler.conv.weight"], W["upsampler.conv.bias"] = rn(4 * mid, 3, 3, mid, std=1.0 / math.sqrt(9 * mid)), rn(4 * mid, std=0.01)
    conv("final_conv", 128, mid)
    return W

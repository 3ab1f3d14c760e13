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


def scan_header(path: Path) -> Dict[str, dict]:
    """safetensors header parsed directly (8-byte length + JSON), no tensor is materialised: the key / shape scan of
    ltx.py:563-586.  Returns {key: {"dtype", "shape", "data_offsets"}}; "__metadata__" is kept under that name."""
    import struct
    with open(path, "rb") as f:
        n = struct.unpack("<Q", f.read(8))[0]
        return json.loads(f.read(n))


def iter_safetensors(paths: Iterable[Path], want=None):
    """Yield (key, host tensor) one tensor at a time (each is dropped by the caller after its device copy, so the
    host never holds the checkpoint: a 26 GB transformer streams through a few hundred MB)."""
    from safetensors import safe_open
    for p in paths:
        with safe_open(str(p), framework="pt", device="cpu") as f:
            for k in f.keys():
                if want is None or want(k):
                    yield k, f.get_tensor(k)


def read_safetensors(paths: Iterable[Path]) -> Dict[str, torch.Tensor]:
    return dict(iter_safetensors(paths))


def read_metadata(path: Path) -> dict:
    return dict(scan_header(path).get("__metadata__") or {})


_TR_PREFIXES = ("transformer_blocks.", "patchify_proj.", "adaln_single.", "caption_projection.", "proj_out.", "scale_shift_table")


def _transformer_key(raw_key: str) -> Optional[str]:
    """Checkpoint key -> module key of the VIDEO transformer, or None (ltx.py:508-533 plus the converted-MLX layout
    where keys are already sanitised, optionally under "transformer.")."""
    from .ltx_model import LTXModel
    k = raw_key
    if k.startswith("model.diffusion_model."):
        m = LTXModel.sanitize({k: None})
        if not m:
            return None
        k = next(iter(m))
    elif k.startswith("transformer."):
        k = k[len("transformer."):]
    if not k.startswith(_TR_PREFIXES):
        return None
    if k.startswith(("audio_", "av_ca_")) or ".audio_" in k or "audio_to_video" in k or "video_to_audio" in k or "a2v_ca" in k:
        return None
    return k


def transformer_weights(raw: Dict[str, torch.Tensor], device) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in raw.items():
        kk = _transformer_key(k)
        if kk is not None:
            out[kk] = _to_dev(v, device)
    return out


def infer_transformer_config(shapes: Dict[str, Iterable[int]]):
    """LTXModelConfig from the tensor shapes of a (sanitised) checkpoint: depth from the block indices, inner dim
    from patchify_proj, caption width from caption_projection.linear1, head count = inner_dim / 128 (the head dim is
    fixed by the RoPE / attention kernels, config.py:93-129)."""
    from .ltx_model import LTXModelConfig
    layers = 1 + max((int(k.split(".")[1]) for k in shapes if k.startswith("transformer_blocks.")), default=-1)
    if layers <= 0 or "patchify_proj.weight" not in shapes:
        raise ValueError("not an LTX-2 transformer checkpoint: no transformer_blocks.* / patchify_proj.weight")
    inner, in_ch = (int(x) for x in shapes["patchify_proj.weight"])
    if inner % 128:
        raise ValueError(f"inner dim {inner} is not a multiple of the head dim 128")
    cap = int(list(shapes["caption_projection.linear1.weight"])[1])
    out_ch = int(list(shapes["proj_out.weight"])[0])
    return LTXModelConfig(num_attention_heads=inner // 128, attention_head_dim=128, in_channels=in_ch, out_channels=out_ch,
                          num_layers=layers, cross_attention_dim=inner, caption_channels=cap)


def vae_decoder_weights(raw: Dict[str, torch.Tensor], device) -> Dict[str, torch.Tensor]:
    """decoder.py:675-721: strip `vae.decoder.` / `decoder.`, remap diffusers names, transpose convs,
    per-channel statistics -> latents_mean / latents_std."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in raw.items():
        kk = _vae_decoder_key(k)
        if kk is not None:
            out[kk] = _to_dev(_conv_to_mlx(kk, v), device)
    _alias_stats(out)
    return out


# Per-channel statistics tensors the reference loads, by exact name (decoder.py:667-690, encoder.py:141-157).  A PyTorch
# LTX-2 checkpoint also carries `channel`, `mean-of-stds`, `mean-of-stds_over_std-of-means` (convert.py:277-286 skips them)
# and an `audio_vae.per_channel_statistics.*` pair of the same shape: none of those may land on the video VAE's mean / std.
_STATS_PREFIXES = ("vae.per_channel_statistics.", "per_channel_statistics.")
_STATS_NAMES = {"mean-of-means": "mean", "mean": "mean", "std-of-means": "std", "std": "std"}


def _stats_kind(k: str) -> Optional[str]:
    """'mean' / 'std' for the video VAE's statistics tensors, None for every other key."""
    for pre in _STATS_PREFIXES:
        if k.startswith(pre):
            return _STATS_NAMES.get(k[len(pre):])
    return None


def _vae_decoder_key(k: str) -> Optional[str]:
    from .video_vae import LTX2VideoDecoder
    kk = k
    for pre in ("vae.decoder.", "vae_decoder.", "decoder."):
        if kk.startswith(pre):
            kk = kk[len(pre):]
            break
    else:
        if k in ("latents_mean", "latents_std"):
            return k
        kind = _stats_kind(k)
        if kind is None:
            return None
        # keep the checkpoint's own name so that _alias_stats applies the reference's precedence
        # (per_channel_statistics.mean / .std override *-of-means, decoder.py:676-689)
        return "per_channel_statistics." + k.rsplit(".", 1)[-1]
    return LTX2VideoDecoder.remap_decoder_key(kk)


def _alias_stats(out: Dict[str, torch.Tensor]) -> None:
    """The chain of `if key in weights` assignments of decoder.py:676-695: *-of-means first, plain per_channel_statistics
    .mean / .std override them, explicit latents_mean / latents_std override both."""
    explicit = {d: out[d] for d in ("latents_mean", "latents_std") if d in out}
    for src, dst in (("per_channel_statistics.mean-of-means", "latents_mean"), ("per_channel_statistics.std-of-means", "latents_std"),
                     ("per_channel_statistics.mean", "latents_mean"), ("per_channel_statistics.std", "latents_std")):
        if src in out:
            out[dst] = out[src]
    out.update(explicit)


def _resolve_encoder_stats(out: Dict[str, torch.Tensor]) -> None:
    """encoder.py:141-157: plain `mean` / `std` override the *-of-means tensors, whatever the order in the file."""
    for kind in ("mean", "std"):
        of = out.pop(f"per_channel_statistics.{kind}#of-means", None)
        if of is not None and f"per_channel_statistics.{kind}" not in out:
            out[f"per_channel_statistics.{kind}"] = of


def _vae_encoder_key(k: str) -> Optional[str]:
    kk = None
    for pre in ("vae.encoder.", "vae_encoder.", "encoder."):
        if k.startswith(pre):
            kk = k[len(pre):]
    if kk is None:
        kind = _stats_kind(k)
        if kind is None:
            return None
        # "-of-means" names first, plain "mean"/"std" override them (encoder.py:141-157): see vae_encoder_weights
        return "per_channel_statistics." + kind + ("" if k.rsplit(".", 1)[-1] in ("mean", "std") else "#of-means")
    kk = kk.replace(".conv.conv.", ".conv.").replace("conv_in.conv.", "conv_in.").replace("conv_out.conv.", "conv_out.")
    return kk.replace(".conv1.conv.", ".conv1.").replace(".conv2.conv.", ".conv2.")


def vae_encoder_weights(raw: Dict[str, torch.Tensor], device) -> Dict[str, torch.Tensor]:
    """encoder.py:135-179: strip `vae.encoder.` / `encoder.`, drop the `.conv.` wrapper level, transpose convs."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in raw.items():
        kk = _vae_encoder_key(k)
        if kk is not None:
            out[kk] = _to_dev(_conv_to_mlx(kk, v), device)
    _resolve_encoder_stats(out)
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


def infer_encoder_blocks(keys: Iterable[str]):
    """Encoder block list from the checkpoint's key set: `down_blocks.i.res_blocks.j.*` => ("res_x", n); a bare
    `down_blocks.i.conv.*` is a compress block whose stride follows the default schedule of encoder.py:95-105."""
    from .video_vae import ENC_BLOCKS
    res: Dict[int, int] = {}
    comp = set()
    for k in keys:
        p = k.split(".")
        if len(p) >= 3 and p[0] == "down_blocks" and p[1].isdigit():
            i = int(p[1])
            if p[2] == "res_blocks" and len(p) > 3 and p[3].isdigit():
                res[i] = max(res.get(i, 0), int(p[3]) + 1)
            elif p[2] == "conv":
                comp.add(i)
    n = 1 + max(list(res) + list(comp), default=-1)
    if n != len(ENC_BLOCKS):
        return None
    blocks = []
    for i in range(n):
        kind, arg = ENC_BLOCKS[i]
        if i in res:
            if kind != "res_x":
                return None
            blocks.append(("res_x", res[i]))
        elif i in comp and kind != "res_x":
            blocks.append((kind, arg))
        else:
            return None
    return blocks


def load_pipeline_modules(model_repo: str, device, need_encoder: bool = False, need_upsampler: bool = False,
                          loras: Optional[list] = None, build_transformer: bool = True) -> dict:
    """Local-directory loader (no network: repo *names* are not resolved, utils.py:78-374 is out of scope).
    Headers are scanned first (keys, shapes, `timestep_conditioning` metadata), the architecture is inferred from the
    shapes, then tensors stream one by one straight into device memory under their module keys (ltx.py:548-826,
    decoder.py:594-740, encoder.py:108-187).  Returns the modules plus ``transformer_weights`` / ``transformer_config``
    (the base dict stays reachable so LoRA-merged twins can be built, generate.py:2957-3031,3229-3237)."""
    from .ltx_model import LTXModel
    from .video_vae import LTX2VideoDecoder, VideoEncoder
    root = Path(model_repo)
    if not root.exists():
        raise FileNotFoundError(f"model_repo {model_repo!r} is not a local directory; HF downloads are not available offline")
    files = sorted(root.glob("*.safetensors"))
    if not files:
        raise FileNotFoundError(f"no .safetensors under {root}")
    main = [f for f in files if "upscaler" not in f.name and "upsampler" not in f.name]
    ups = [f for f in files if "upscaler" in f.name or "upsampler" in f.name]
    # ---- pass 1: headers only ----
    tshapes: Dict[str, list] = {}
    enc_keys = []
    has_quant = False
    for f in main:
        for k, meta in scan_header(f).items():
            if k == "__metadata__":
                continue
            has_quant = has_quant or k.endswith((".scales", ".biases"))
            tk = _transformer_key(k)
            if tk is not None:
                tshapes[tk] = meta["shape"]
            ek = _vae_encoder_key(k)
            if ek is not None:
                enc_keys.append(ek.replace("#of-means", ""))
    if has_quant:
        raise ValueError("pre-quantised MLX checkpoints (.scales/.biases) are not supported: MLX affine quantisation is out of scope "
                         "(SURVEY.md §2a); use bf16 weights")
    tcfg = infer_transformer_config(tshapes)
    missing = [k for k in LTXModel.expected_keys(tcfg) if k not in tshapes]
    if missing:   # strict load (ltx.py:874-881)
        raise ValueError(f"Missing {len(missing)} parameters in checkpoint, e.g. {missing[:4]}")
    # ---- pass 2: stream tensors to the device ----
    tw: Dict[str, torch.Tensor] = {}
    dw: Dict[str, torch.Tensor] = {}
    ew: Dict[str, torch.Tensor] = {}
    for k, v in iter_safetensors(main):
        tk = _transformer_key(k)
        if tk is not None:
            tw[tk] = _to_dev(v, device)
            continue
        dk = _vae_decoder_key(k)
        if dk is not None:
            dw[dk] = _to_dev(_conv_to_mlx(dk, v), device)
        if need_encoder:
            ek = _vae_encoder_key(k)
            if ek is not None:
                ew[ek] = _to_dev(_conv_to_mlx(ek, v), device)
    _alias_stats(dw)
    _resolve_encoder_stats(ew)
    mods = {"transformer_weights": tw, "transformer_config": tcfg}
    if build_transformer:
        w = tw
        if loras:
            from .lora import LoraSpec, apply_lora_to_weights
            w = apply_lora_to_weights(tw, [LoraSpec(Path(p), float(s)) for p, s in loras])
        mods["transformer"] = LTXModel(tcfg, w)
    nres = 1 + max((int(k.split(".")[3]) for k in dw if k.startswith("up_blocks.0.res_blocks.")), default=4)
    mods["vae_decoder"] = LTX2VideoDecoder(dw, timestep_conditioning=any(sniff_timestep_conditioning(f) for f in main),
                                           num_layers_per_block=nres)
    if need_encoder:
        mods["vae_encoder"] = VideoEncoder(ew, encoder_blocks=infer_encoder_blocks(ew.keys()))
    if need_upsampler and ups:
        from .upsampler import LatentUpsampler
        uw = {k: _to_dev(_conv_to_mlx(k, v) if "conv" in k else v, device) for k, v in iter_safetensors(ups[:1])}
        nb = 1 + max((int(k.split(".")[1]) for k in uw if k.startswith("res_blocks.")), default=3)
        mods["upsampler"] = LatentUpsampler(uw, num_blocks_per_stage=nb)
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

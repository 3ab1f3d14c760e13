"""Conditioning-input I/O (mlx_video/utils.py:529-715): image / frame-sequence loading,
LANCZOS resize to the stage resolution and [-1,1] normalisation into the (1,3,F,H,W) tensors the VAE
encoder takes.  Host-side PIL/numpy work, as in the reference; video *files* need cv2/PyAV (absent in this
image), so video conditionings are taken from a directory of frames or an .npy array instead."""
from __future__ import annotations

from pathlib import Path
from typing import Optional, Union

import numpy as np
import torch


def load_image(image_path: Union[str, Path], height: Optional[int] = None, width: Optional[int] = None) -> np.ndarray:
    """utils.py:529-575: RGB, LANCZOS resize (both dims given: exact; one: aspect-preserving, other rounded
    down to /32; none: round both down to /32).  Returns (H,W,3) float32 in [0,1]."""
    from PIL import Image
    img = Image.open(image_path).convert("RGB")
    ow, oh = img.size
    if height is not None and width is not None:
        size = (width, height)
    elif height is not None:
        size = ((int(ow * height / oh) // 32) * 32, height)
    elif width is not None:
        size = (width, (int(oh * width / ow) // 32) * 32)
    else:
        size = ((ow // 32) * 32, (oh // 32) * 32)
    if size != (ow, oh):
        img = img.resize(size, Image.Resampling.LANCZOS)
    return np.asarray(img).astype(np.float32) / 255.0


def image_to_conditioning(image: np.ndarray) -> torch.Tensor:
    """(H,W,3) in [0,1] -> (1,3,1,H,W) in [-1,1] (generate.py:3064-3076: x*2-1, channels first, one frame)."""
    t = torch.from_numpy(np.ascontiguousarray(image)).permute(2, 0, 1)[None, :, None]
    return t * 2.0 - 1.0


def load_frames(source: Union[str, Path, np.ndarray], height: int, width: int, frame_cap: Optional[int] = None) -> np.ndarray:
    """Frame sequence for video conditioning (stand-in for utils.py:578-613, which decodes with cv2):
    a directory of images (sorted) or an (F,H,W,3) uint8/float array or .npy file.  (F,H,W,3) in [0,1]."""
    from PIL import Image
    if isinstance(source, (str, Path)) and Path(source).is_dir():
        files = sorted(p for p in Path(source).iterdir() if p.suffix.lower() in (".png", ".jpg", ".jpeg", ".bmp"))
        if not files:
            raise ValueError(f"No frames decoded from video: {source}")
        return np.stack([load_image(f, height, width) for f in files[:frame_cap]], 0)
    arr = np.load(source) if isinstance(source, (str, Path)) else np.asarray(source)
    if arr.ndim != 4 or arr.shape[-1] != 3:
        raise ValueError(f"expected (F,H,W,3) frames, got {arr.shape}")
    arr = arr[:frame_cap]
    if arr.dtype == np.uint8:
        arr = arr.astype(np.float32) / 255.0
    if arr.shape[1:3] != (height, width):
        arr = np.stack([np.asarray(Image.fromarray((f * 255).round().astype(np.uint8)).resize((width, height), Image.Resampling.BOX)).astype(np.float32) / 255.0
                        for f in arr], 0)              # INTER_AREA analogue
    return arr.astype(np.float32)


def frames_to_conditioning(frames: np.ndarray) -> torch.Tensor:
    """(F,H,W,3) in [0,1] -> (1,3,F,H,W) in [-1,1]; F is trimmed to 1+8k as the encoder requires
    (video_vae.py:332-337)."""
    f = frames.shape[0]
    keep = 1 + ((f - 1) // 8) * 8
    t = torch.from_numpy(np.ascontiguousarray(frames[:keep])).permute(3, 0, 1, 2)[None]
    return t * 2.0 - 1.0

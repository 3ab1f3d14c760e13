"""Conditioning-input I/O (mlx_video/utils.py:529-715): image / frame-sequence loading,
LANCZOS resize of images, cv2.INTER_AREA resize of video frames (evaluated here from OpenCV's published coefficient
tables - cv2 itself is not in this image) and [-1,1] normalisation into the (1,3,F,H,W) tensors the VAE encoder
takes.  Host-side PIL/numpy work, as in the reference; video FILES are decoded by an ffmpeg child process."""
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


def area_taps(ssize: int, dsize: int):
    """One axis of cv2.resize(..., INTER_AREA) as a tap table: (idx, w), both (dsize, T); destination d =
    sum_t w[d,t] * src[idx[d,t]].  Shrinking (ssize >= dsize) is OpenCV's area table (imgproc/resize.cpp,
    computeResizeAreaTab): d covers the source interval [d*scale, (d+1)*scale), whole pixels inside weigh 1/cell, the two
    cut pixels their overlap/cell with cell = min(scale, ssize - d*scale), overlaps below 1e-3 dropped.  Enlarging is
    what INTER_AREA falls back to there: two taps, sx = floor(d*scale), fx = (d+1) - (sx+1)/scale (0 if negative, else
    its fractional part), the last source pixel repeated.  The same coefficients as ltxk_resize_area on the device."""
    scale = ssize / dsize
    d = np.arange(dsize, dtype=np.float64)
    if dsize > ssize:
        sx = np.floor(d * scale).astype(np.int64)
        fx = (d + 1.0) - (sx + 1.0) / scale
        fx = np.where(fx <= 0, 0.0, fx - np.floor(fx))
        edge = sx >= ssize - 1
        sx, fx = np.where(edge, ssize - 1, sx), np.where(edge, 0.0, fx)
        idx = np.stack([sx, np.minimum(sx + 1, ssize - 1)], 1)
        return idx, np.stack([1.0 - fx, fx], 1).astype(np.float32)
    f1 = d * scale
    f2 = f1 + scale
    cell = np.minimum(scale, ssize - f1)
    s2 = np.minimum(np.floor(f2), ssize - 1).astype(np.int64)
    s1 = np.minimum(np.ceil(f1).astype(np.int64), s2)
    T = int(np.max(s2 - s1)) + 2
    t = np.arange(T)[None, :]
    idx = s1[:, None] - 1 + t                                      # tap 0 = the cut pixel on the left
    w = np.where((t >= 1) & (idx < s2[:, None]), 1.0 / cell[:, None], 0.0)
    left = np.where(s1 - f1 > 1e-3, (s1 - f1) / cell, 0.0)
    right = np.where(f2 - s2 > 1e-3, np.minimum(np.minimum(f2 - s2, 1.0), cell) / cell, 0.0)
    w[:, 0] = left
    np.put_along_axis(w, (s2 - s1 + 1)[:, None], right[:, None], 1)
    return np.clip(idx, 0, ssize - 1), w.astype(np.float32)


def _apply_taps_rows(a: np.ndarray, idx: np.ndarray, w: np.ndarray) -> np.ndarray:
    """sum_t w[:,t] * a[idx[:,t]] along axis 0, taps added in table order (cv2's loop order); a tap whose indices form
    an arithmetic progression (every tap of an integer factor) is a strided view instead of a gather."""
    acc = None
    for t in range(idx.shape[1]):
        if not w[:, t].any():
            continue
        i = idx[:, t]
        step = int(i[1] - i[0]) if len(i) > 1 else 1
        src = a[i[0]: i[-1] + 1: step] if step > 0 and np.array_equal(i, i[0] + step * np.arange(len(i))) else a[i]
        term = src * w[:, t].reshape((-1,) + (1,) * (a.ndim - 1))
        acc = term if acc is None else np.add(acc, term, out=acc)
    return acc


def resize_area(frames: np.ndarray, height: int, width: int) -> np.ndarray:
    """cv2.resize(frame, (width, height), interpolation=cv2.INTER_AREA) for every frame of an (F,H,W,C) array
    (utils.py:597-598 on the decoded uint8 frames, utils.py:699-705 on float frames): horizontal pass then vertical pass
    in fp32; uint8 input gives uint8 output, rounded to nearest-even as cv2's saturate_cast does.  (One filter for every
    container a conditioning clip can arrive in: tensors take the same table through ltxk_resize_area.)  Each axis follows
    its own rule; a resize that shrinks one axis and enlarges the other goes through cv2's linear path for BOTH axes there -
    not reproduced (conditioning resizes keep the aspect; parity unpinned like every fractional factor)."""
    F, H, W, C = frames.shape
    tx = area_taps(W, width) if W != width else None
    ty = area_taps(H, height) if H != height else None
    out = np.empty((F, height, width, C), frames.dtype if frames.dtype == np.uint8 else np.float32)
    for i in range(F):
        a = frames[i].astype(np.float32)
        if tx is not None:                                 # columns -> rows so that every tap reads whole contiguous rows
            a = _apply_taps_rows(np.ascontiguousarray(a.transpose(1, 0, 2)), *tx).transpose(1, 0, 2)
        if ty is not None:
            a = _apply_taps_rows(np.ascontiguousarray(a), *ty)
        out[i] = np.clip(np.rint(a), 0, 255) if frames.dtype == np.uint8 else a
    return out


VIDEO_SUFFIXES = (".mp4", ".mov", ".mkv", ".webm", ".avi", ".m4v", ".gif")


def decode_video_ffmpeg(path: Union[str, Path], frame_cap: Optional[int] = None) -> np.ndarray:
    """Decode a video FILE to (F,H,W,3) uint8 RGB frames through an ffmpeg child process (the reference decodes with
    cv2.VideoCapture, utils.py:587-606; neither cv2 nor PyAV exists in this image, and output already leaves through an
    ffmpeg pipe, generate.py:1833-1893): ffprobe gives the frame size, ffmpeg streams rawvideo rgb24 on stdout.
    FileNotFoundError when the binaries are not installed, ValueError when nothing decodes (utils.py:588-589,608-609)."""
    import json
    import shutil
    import subprocess
    ffmpeg, ffprobe = shutil.which("ffmpeg"), shutil.which("ffprobe")
    if ffmpeg is None or ffprobe is None:
        raise FileNotFoundError("ffmpeg / ffprobe not found: video files cannot be decoded here; pass a directory of frames, "
                                "an .npy array of (F,H,W,3) frames or a pixel tensor instead")
    pr = subprocess.run([ffprobe, "-v", "error", "-select_streams", "v:0", "-show_entries", "stream=width,height", "-of", "json",
                         str(path)], capture_output=True, text=True)
    try:
        st = json.loads(pr.stdout)["streams"][0]
        w, h = int(st["width"]), int(st["height"])
    except (ValueError, KeyError, IndexError):
        raise ValueError(f"Unable to open video: {path}")
    # -noautorotate: ffprobe reports the CODED frame size; with rotation side data ffmpeg would otherwise hand back
    # transposed frames that reshape into garbage (cv2.VideoCapture's orientation handling is not reproduced: parity unpinned)
    cmd = [ffmpeg, "-v", "error", "-noautorotate", "-i", str(path)] + (["-frames:v", str(int(frame_cap))] if frame_cap else []) + \
          ["-f", "rawvideo", "-pix_fmt", "rgb24", "-"]
    out = subprocess.run(cmd, capture_output=True)
    n = len(out.stdout) // (w * h * 3)
    if out.returncode != 0 or n == 0:
        raise ValueError(f"No frames decoded from video: {path}")
    return np.frombuffer(out.stdout[: n * w * h * 3], dtype=np.uint8).reshape(n, h, w, 3)


def load_frames(source: Union[str, Path, np.ndarray], height: int, width: int, frame_cap: Optional[int] = None) -> np.ndarray:
    """Frame sequence for video conditioning (utils.py:578-613, which decodes with cv2 and resizes every decoded uint8
    frame with INTER_AREA): a video file (through ffmpeg, where installed), a directory of images (sorted), or an
    (F,H,W,3) uint8/float array or .npy file.  (F,H,W,3) in [0,1]."""
    if isinstance(source, (str, Path)) and Path(source).is_file() and Path(source).suffix.lower() in VIDEO_SUFFIXES:
        source = decode_video_ffmpeg(source, frame_cap)
    if isinstance(source, (str, Path)) and Path(source).is_dir():
        files = sorted(p for p in Path(source).iterdir() if p.suffix.lower() in (".png", ".jpg", ".jpeg", ".bmp"))
        if not files:
            raise ValueError(f"No frames decoded from video: {source}")
        return np.stack([load_image(f, height, width) for f in files[:frame_cap]], 0)
    arr = np.load(source) if isinstance(source, (str, Path)) else np.asarray(source)
    if arr.ndim != 4 or arr.shape[-1] != 3:
        raise ValueError(f"expected (F,H,W,3) frames, got {arr.shape}")
    arr = arr[:frame_cap]
    if arr.shape[1:3] != (height, width):
        arr = resize_area(arr, height, width)              # on the decoded uint8 frames, as utils.py:597-598
    if arr.dtype == np.uint8:
        arr = arr.astype(np.float32) / 255.0
    return arr.astype(np.float32)


def resize_conditioning(frames01: np.ndarray, height: int, width: int, is_video: bool) -> torch.Tensor:
    """(F,H,W,3) in [0,1] -> (1,3,F,height,width) in [-1,1], resized the way the reference prepares an already
    decoded source of another size: images through uint8 + LANCZOS (prepare_image_for_encoding, utils.py:643-661:
    `(image*255).astype(uint8)`, i.e. truncation), video frames with cv2.INTER_AREA on the float frames
    (prepare_video_for_encoding, utils.py:699-705; resize_area above)."""
    if is_video:
        arr = resize_area(np.asarray(frames01, np.float32), height, width)
    else:
        from PIL import Image
        arr = np.stack([np.asarray(Image.fromarray((f * 255).astype(np.uint8)).resize((width, height), Image.Resampling.LANCZOS)).astype(np.float32) / 255.0
                        for f in frames01], 0)
    t = torch.from_numpy(np.ascontiguousarray(arr)).permute(3, 0, 1, 2)[None]
    return t * 2.0 - 1.0


def frames_to_conditioning(frames: np.ndarray) -> torch.Tensor:
    """(F,H,W,3) in [0,1] -> (1,3,F,H,W) in [-1,1]; F is trimmed to 1+8k as the encoder requires
    (video_vae.py:332-337)."""
    f = frames.shape[0]
    keep = 1 + ((f - 1) // 8) * 8
    t = torch.from_numpy(np.ascontiguousarray(frames[:keep])).permute(3, 0, 1, 2)[None]
    return t * 2.0 - 1.0


def ffmpeg_command(ffmpeg: str, width: int, height: int, fps: float, path: Union[str, Path], codec: str = "libx264",
                   preset: str = "veryfast", crf: int = 18) -> list:
    """Raw RGB24 frames on stdin -> yuv420p video file (the encoder settings of generate.py:1833-1877)."""
    return [ffmpeg, "-y", "-f", "rawvideo", "-pix_fmt", "rgb24", "-s", f"{width}x{height}", "-r", str(fps), "-i", "-", "-an",
            "-c:v", codec, "-preset", preset, "-crf", str(crf), "-pix_fmt", "yuv420p", str(path)]


def write_video_ffmpeg(video_np: np.ndarray, path: Union[str, Path], fps: float, codec: str = "libx264",
                       preset: str = "veryfast", crf: int = 18) -> None:
    """(F,H,W,3) uint8 -> video file through an ffmpeg child process fed frame by frame (generate.py:1833-1893:
    FileNotFoundError when ffmpeg is not installed, RuntimeError with ffmpeg's stderr when it fails).  There is no
    OpenCV fallback here: callers that cannot encode write .npy frames instead."""
    import shutil
    import subprocess
    if video_np.ndim != 4 or video_np.shape[-1] != 3 or video_np.dtype != np.uint8:
        raise ValueError(f"expected (F,H,W,3) uint8 frames, got {video_np.shape} {video_np.dtype}")
    ffmpeg = shutil.which("ffmpeg")
    if ffmpeg is None:
        raise FileNotFoundError("ffmpeg not found")
    h, w = video_np.shape[1], video_np.shape[2]
    proc = subprocess.Popen(ffmpeg_command(ffmpeg, w, h, fps, path, codec, preset, crf), stdin=subprocess.PIPE,
                            stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    try:
        for frame in video_np:
            proc.stdin.write(np.ascontiguousarray(frame).tobytes())
        proc.stdin.close()
    except BrokenPipeError:
        pass                                            # ffmpeg died early: its stderr says why
    err = proc.stderr.read() if proc.stderr is not None else b""
    if proc.wait() != 0:
        raise RuntimeError(err.decode(errors="ignore"))

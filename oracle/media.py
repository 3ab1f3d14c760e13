"""CPU restatement of the conditioning-frame area resize.  TEST INFRASTRUCTURE - see oracle/__init__.py.

The reference resizes decoded video frames with ``cv2.resize(f, (W,H), interpolation=cv2.INTER_AREA)`` on float32 frames
(mlx_video/utils.py:699-705).  OpenCV is a third-party dependency that is not importable in this image (``opencv-python``,
pinned >=4.12 in pyproject.toml), so this restates its published algorithm for DOWNscaling (modules/imgproc/src/resize.cpp:
computeResizeAreaTab + resizeArea_): along each axis destination pixel d covers the source interval [d*scale,(d+1)*scale);
whole source pixels inside it weigh 1/cell, the two partially covered end pixels their overlap/cell, cell = min(scale,
ssize - d*scale); ends shorter than 1e-3 are dropped.  Parity status: for integer factors the result is the plain block mean
(a closed-form known answer, checked in the tests); for fractional factors and for enlarging "parity unpinned" - no cv2 here to confirm it."""
import math

import numpy as np


def area_weights(ssize: int, dsize: int) -> np.ndarray:
    """(dsize, ssize) float32 weight matrix of one axis (computeResizeAreaTab).  Enlarging (dsize > ssize) is what cv2.resize
    does with INTER_AREA there: the linear path with area-mode coefficients (resize.cpp: sx = floor(dx*scale), fx = (dx+1) -
    (sx+1)*inv_scale, clamped to 0 when negative else its fractional part; the last source pixel repeats) - an integer
    enlargement therefore replicates pixels."""
    scale = ssize / dsize
    A = np.zeros((dsize, ssize), dtype=np.float32)
    if dsize > ssize:
        for d in range(dsize):
            sx = math.floor(d * scale)
            fx = (d + 1) - (sx + 1) / scale
            fx = 0.0 if fx <= 0 else fx - math.floor(fx)
            if sx >= ssize - 1:
                sx, fx = ssize - 1, 0.0
            A[d, sx] += np.float32(1.0 - fx)
            if fx > 0:
                A[d, sx + 1] += np.float32(fx)
        return A
    for d in range(dsize):
        f1 = d * scale
        f2 = f1 + scale
        cell = min(scale, ssize - f1)
        s1, s2 = math.ceil(f1), math.floor(f2)
        s2 = min(s2, ssize - 1)
        s1 = min(s1, s2)
        if s1 - f1 > 1e-3:
            A[d, s1 - 1] = np.float32((s1 - f1) / cell)
        for s in range(s1, s2):
            A[d, s] = np.float32(1.0 / cell)
        if f2 - s2 > 1e-3:
            A[d, s2] = np.float32(min(min(f2 - s2, 1.0), cell) / cell)
    return A


def resize_area(frames: np.ndarray, oh: int, ow: int) -> np.ndarray:
    """(..., H, W) float32 -> (..., oh, ow) float32, horizontal pass then vertical pass (resizeArea_)."""
    H, W = frames.shape[-2:]
    Ax, Ay = area_weights(W, ow), area_weights(H, oh)
    h = np.einsum("...yx,dx->...yd", frames.astype(np.float32), Ax, optimize=True).astype(np.float32)
    return np.einsum("...yd,ey->...ed", h, Ay, optimize=True).astype(np.float32)

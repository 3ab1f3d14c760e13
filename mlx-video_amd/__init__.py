"""ltx355 — MI355X-native LTX-2 denoise step + video VAE behind the mlx_video.generate surface.

Compute lives in ``csrc/`` (hand-written gfx950 HIP kernels) and is reached only through the
C ABI of ``libltxk.so`` (``include/ltxk.h``).  PyTorch is used for device memory, streams and
``torch.distributed`` — never for the math of the hot path.
"""
__version__ = "0.1.0"

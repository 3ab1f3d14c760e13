"""CPU oracle for the LTX-2 denoise step + video VAE hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package (``mlx-video_amd/``); only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use it, and only as the checker.

PARITY UNPINNED for the floating-point DiT forward / attention / AdaLN / Euler /
full VAE decode: the reference executes on Apple MLX (mlx 0.30.1, uv.lock:760-761),
which is not installed here and cannot be fetched, and the reference's own tests
hold no numeric fixtures for those stages.  What IS pinned, by the
closed-form known answers in the reference's tests (tests/test_oracle_kat.py):
cfg formula, scheduler invariants, position-grid shape/causal-fix/bounds, token
counts, SPLIT-rope shapes, depth-to-space shapes, chunked==regular, trapezoid
mask properties, tiling presets, LoRA ``I+1``.

Each function cites the reference file:line it restates (paths relative to the
reference checkout root).
"""

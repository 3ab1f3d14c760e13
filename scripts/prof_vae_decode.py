"""A few 33x512x512 decodes for rocprofv3 --kernel-trace --stats (per-kernel time of the VAE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import video_vae
print(video_vae.bench_decode(torch.device("cuda:0"), 5, 16, 16, iters=int(sys.argv[1]) if len(sys.argv) > 1 else 5))

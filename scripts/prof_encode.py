"""VAE encode of a 65x384x384 clip (config 5 conditioning) and the host-side resize in front of it, timed apart."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from mlx_video_amd.video_vae import VideoEncoder
from mlx_video_amd import media
import numpy as np
dev = torch.device('cuda:0')
from mlx_video_amd.video_vae import random_encoder_weights
enc = VideoEncoder(random_encoder_weights(dev))
print('enc', enc is not None)
x = (torch.rand(1, 3, 65, 384, 384) * 2 - 1)
t0 = time.perf_counter(); xd = x.to(dev).to(torch.bfloat16); torch.cuda.synchronize(); print('upload', time.perf_counter() - t0)
for i in range(3):
    t0 = time.perf_counter(); z = enc(xd); torch.cuda.synchronize(); print('encode', time.perf_counter() - t0, tuple(z.shape))
fr = np.random.rand(65, 768, 768, 3).astype(np.float32)
t0 = time.perf_counter(); y = media.resize_conditioning(fr, 384, 384, True); print('host resize 65x768x768 -> 384', time.perf_counter() - t0, tuple(y.shape))

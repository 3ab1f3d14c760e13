"""DVFS probe for the VAE decode: the same launches on all-zero weights and latents (every instruction still executes) -
how much of the conv time is the power limit and how much is structure."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import video_vae as V, ops
dev = torch.device("cuda:0")
for zero in (False, True, False, True):
    W = V.random_decoder_weights(dev)
    if zero:
        W = {k: (torch.zeros_like(v) if v.dtype == torch.bfloat16 and v.dim() >= 2 else v) for k, v in W.items()}
    dec = V.LTX2VideoDecoder(W)
    lat = torch.zeros((1, 128, 5, 16, 16), device=dev, dtype=torch.bfloat16) if zero else torch.randn((1, 128, 5, 16, 16), device=dev).to(torch.bfloat16)
    dec(lat); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        dec(lat)
    torch.cuda.synchronize()
    print(f"{'zeros ' if zero else 'random'}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per decode", flush=True)

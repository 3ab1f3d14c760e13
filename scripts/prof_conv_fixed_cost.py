"""Fixed cost per tile of the conv kernels: time the 33x128x128-voxel, Cout=128 convolution at Cin = 64 / 128 / 256 (18 / 36 /
72 stages of the kw kernel per tile) and fit t = rounds * (a + b * stages)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import video_vae as V
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
res = {}
for cin in (64, 128, 256):
    x = torch.randn((1, 33, 128, 128, cin), generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn((128, 3, 3, 3, cin), generator=g, device=dev) * 0.02).to(torch.bfloat16)
    b = torch.zeros(128, device=dev, dtype=torch.bfloat16)
    if os.environ.get("CONV_ZERO_DATA") == "1":          # DVFS probe: all-zero operands (same instruction stream)
        x.zero_(); w.zero_()
    for _ in range(3): V.conv3d(x, w, b, False, V.PAD_REFLECT)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st):
            for _ in range(10): V.conv3d(x, w, b, False, V.PAD_REFLECT)
    gr.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); gr.replay(); torch.cuda.synchronize()
    res[cin] = (time.perf_counter() - t0) / 10 * 1e6
    print(f"Cin={cin}: {res[cin]:8.1f} us", flush=True)
b = (res[256] - res[64]) / (72 - 18)
a = res[128] - 36 * b
print(f"per launch: fixed {a:.1f} us + {b:.2f} us per stage; per tile-round (8.25 rounds): fixed {a/8.25:.2f} us + {b/8.25:.3f} us per stage; fixed share at Cin=128: {a/res[128]:.1%}")

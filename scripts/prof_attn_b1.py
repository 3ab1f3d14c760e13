import math, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mlx_video_amd import ops
dev = torch.device("cuda:0")
for (B, Tq, Tk) in ((1, 1280, 1280), (1, 1280, 1024), (1, 3328, 3328), (1, 3328, 1024), (2, 1280, 1280)):
    H, D = 32, 4096
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn((B * Tq, D), generator=g, device=dev).to(torch.bfloat16)
    k = torch.randn((B * Tk, D), generator=g, device=dev).to(torch.bfloat16)
    vt = torch.randn((B, D, (Tk + 63) // 64 * 64), generator=g, device=dev).to(torch.bfloat16)
    out = torch.empty((B * Tq, D), dtype=torch.bfloat16, device=dev)
    for _ in range(3): ops.flash_attn(q, k, vt, out, B, H, Tq, Tk, 1 / math.sqrt(128))
    torch.cuda.synchronize()
    st = torch.cuda.Stream(); gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st):
            for _ in range(20): ops.flash_attn(q, k, vt, out, B, H, Tq, Tk, 1 / math.sqrt(128))
    gr.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); gr.replay(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"B={B} Tq={Tq} Tk={Tk}: {dt*1e6:8.1f} us", flush=True)

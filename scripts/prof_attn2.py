"""Attention timing at chosen (Tq, Tk) pairs via graph replay (host launch rate excluded): argv = iters Tq:Tk ..."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import ops
dev = torch.device("cuda:0")
iters = int(sys.argv[1])
B, H, D = 2, 32, 4096
for spec in sys.argv[2:]:
    Tq, Tk = map(int, spec.split(":"))
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn((B * Tq, D), generator=g, device=dev).to(torch.bfloat16)
    k = torch.randn((B * Tk, D), generator=g, device=dev).to(torch.bfloat16)
    vt = torch.randn((B, D, (Tk + 63) // 64 * 64), generator=g, device=dev).to(torch.bfloat16)
    out = torch.empty((B * Tq, D), dtype=torch.bfloat16, device=dev)
    if os.environ.get("ATTN_ZERO_DATA") == "1":          # DVFS probe (all-zero operands; the softmax still runs every instruction)
        q.zero_(); k.zero_(); vt.zero_()
    for _ in range(3):
        ops.flash_attn(q, k, vt, out, B, H, Tq, Tk, 1 / math.sqrt(128))
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st):
            for _ in range(iters):
                ops.flash_attn(q, k, vt, out, B, H, Tq, Tk, 1 / math.sqrt(128))
    gr.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    gr.replay(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"Tq={Tq} Tk={Tk}: {dt*1e6:8.1f} us  {4.0*B*H*Tq*Tk*128/dt/1e12:7.1f} TFLOP/s", flush=True)

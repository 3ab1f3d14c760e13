"""Per-launch-shape time of one DiT forward (eager launches, HIP events around every libltxk call), B=1 against B=2:
which launches make the CFG-pair split (one B=1 forward per rank) less efficient than the B=2 forward.
  python scripts/prof_forward_shapes.py [N | FxHxW] [layers] [batches, e.g. 1,2]
N = 1280 / 3328: latent (N/256) x 16 x 16 (configs 2 / 4); FxHxW e.g. 9x12x12 (N=1296, configs 3 / 5 stage 1), 9x24x24 (N=5184, stage 2)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import ops
from mlx_video_amd.ltx_model import LTXModel, LTXModelConfig, TimestepPlan, precompute_freqs_cis
from mlx_video_amd.schedulers import create_position_grid
shape = sys.argv[1] if len(sys.argv) > 1 else "1280"
Fl, Hl, Wl = (int(v) for v in shape.split("x")) if "x" in shape else (int(shape) // 256, 16, 16)
N = Fl * Hl * Wl
L = int(sys.argv[2]) if len(sys.argv) > 2 else 8
BATCHES = tuple(int(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 else (1, 2)
dev = torch.device("cuda:0")
model = LTXModel.random_init(LTXModelConfig(num_layers=L), dev, seed=1234)
rec = []
orig_gemm, orig_fa = ops.gemm, ops.flash_attn
def gemm(a, w, bias, **kw):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); r = orig_gemm(a, w, bias, **kw); e.record()
    rec.append((f"gemm M={a.shape[0]} N={w.shape[0]} K={a.shape[1]} epi={kw.get('epilogue', 0)}{' T' if kw.get('out_tokens_per_batch') else ''}{' ss' if kw.get('sumsq') is not None else ''}", 2.0 * a.shape[0] * w.shape[0] * a.shape[1], s, e))
    return r
def fa(q, k, vt, out, B, H, Tq, Tk, scale, **kw):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); r = orig_fa(q, k, vt, out, B, H, Tq, Tk, scale, **kw); e.record()
    rec.append((f"attn B={B} Tq={Tq} Tk={Tk}", 4.0 * B * H * Tq * Tk * 128, s, e))
    return r
ops.gemm, ops.flash_attn = gemm, fa
import mlx_video_amd.ltx_model as lm
res = {}
for B in BATCHES:
    g = torch.Generator(device=dev).manual_seed(1)
    lat = torch.randn((B, N, 128), generator=g, device=dev).to(torch.bfloat16)
    ctx = torch.randn((B, 1024, 3840), generator=g, device=dev).to(torch.bfloat16)
    pos = create_position_grid(1, Fl, Hl, Wl).to(dev)
    pe = precompute_freqs_cis(pos, 4096, 10000.0, (20, 2048, 2048), 32)
    plan = TimestepPlan(torch.tensor([0.7], device=dev).to(torch.bfloat16), torch.zeros(B * N, dtype=torch.int32, device=dev))
    for it in range(3):
        rec.clear()
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); model.forward_tokens(lat, plan, ctx, pe); t1.record()
        torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for name, fl, s, e in rec:
        d = agg.setdefault(name, [0, 0.0, 0.0]); d[0] += 1; d[1] += s.elapsed_time(e) * 1e3; d[2] += fl
    res[B] = (agg, t0.elapsed_time(t1))
for B in BATCHES:
    agg, tot = res[B]
    print(f"==== N={N} B={B}: forward {tot:.2f} ms over {L} blocks (eager, instrumented), per block:")
    tot_us = tot_fl = 0.0
    for name, (n, us, fl) in agg.items():
        if n >= L:
            print(f"  {name:48s} x{n // L:2d}/block  {us / n:8.1f} us  {fl / us / 1e6:7.1f} TF/s")
            tot_us += us / L; tot_fl += fl / L
    print(f"  GEMM + attention launches of one block: {tot_us:8.1f} us  {tot_fl / tot_us / 1e6:7.1f} TF/s")

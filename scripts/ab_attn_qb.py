"""Query blocks per wave: the shipped attention kernel (2 blocks of 16 rows per wave, 128-row workgroups) against the 3-block form
(48 rows per wave, 192-row workgroups; LTXK_FA_QB=3 in the A/B build) in ONE process, interleaved graph replays, random data.
Same bits expected (the key order inside a tile and the tile order are the same): checked against the no-tail-split launch.
  python3 scripts/ab_attn_qb.py [rounds]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import _lib, ops
dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
H, D = 32, 4096
SHAPES = [(2, 1024, 1280), (2, 1536, 1280), (2, 1280, 1280), (2, 1280, 1024), (2, 1296, 1296), (2, 1296, 1024), (2, 5184, 5184), (2, 5184, 1024), (2, 3328, 3328), (2, 3328, 1024),
          (1, 1280, 1280), (1, 1280, 1024), (1, 1296, 1296), (1, 3328, 3328), (1, 5184, 5184), (1, 5184, 1024), (2, 320, 320), (2, 640, 640), (4, 1280, 1280)]
ALT = os.environ.get("AB_ALT_LIB")           # optional second build of the library: its mixed grid runs as variant "0alt"
alt = _lib._open(ALT) if ALT else None
if os.environ.get("AB_SHAPES"):
    SHAPES = [tuple(int(v) for v in s_.split(":")) for s_ in os.environ["AB_SHAPES"].split()]
with _lib.use_library(_lib.AB_LIB_PATH) as main_lib:
    for B, Tq, Tk in SHAPES:
        g = torch.Generator(device=dev).manual_seed(Tq + Tk)
        q = torch.randn((B * Tq, D), generator=g, device=dev).to(torch.bfloat16)
        k = torch.randn((B * Tk, D), generator=g, device=dev).to(torch.bfloat16)
        vt = torch.randn((B, D, (Tk + 63) // 64 * 64), generator=g, device=dev).to(torch.bfloat16)
        outs, graphs = {}, {}
        for qb, ts in (("2", True), ("2nosplit", False), ("3", True), ("0mix", True)) + ((("0alt", True),) if alt else ()):
            os.environ["LTXK_FA_QB"] = qb[0]
            _lib._lib = alt if qb == "0alt" else main_lib
            out = torch.zeros((B * Tq, D), dtype=torch.bfloat16, device=dev)
            fn = lambda o=out, t=ts: ops.flash_attn(q, k, vt, o, B, H, Tq, Tk, 1 / math.sqrt(128), tail_split=t)
            fn(); torch.cuda.synchronize()
            outs[qb] = out.clone()
            st = torch.cuda.Stream(); gr = torch.cuda.CUDAGraph()
            with torch.cuda.stream(st):
                with torch.cuda.graph(gr, stream=st):
                    for _ in range(20): fn()
            graphs[qb] = gr
        _lib._lib = main_lib
        same = torch.equal(outs["3"], outs["2nosplit"]) and torch.equal(outs["0mix"], outs["2nosplit"]) and (alt is None or torch.equal(outs["0alt"], outs["2nosplit"]))
        ts = {n: [] for n in graphs}
        for r in range(rounds):
            for n, gr in graphs.items():
                gr.replay(); torch.cuda.synchronize()
                t0 = time.perf_counter(); gr.replay(); torch.cuda.synchronize()
                ts[n].append((time.perf_counter() - t0) / 20 * 1e6)
        fl = 4.0 * B * H * Tq * Tk * 128
        line = f"B={B} Tq={Tq:5d} Tk={Tk:5d}  QB3 == mix == QB2(no tail split) bits: {same}  "
        for n in graphs:
            v = sorted(ts[n])[len(ts[n]) // 2]
            line += f"| {n:9s} {v:8.1f} us {fl / v / 1e6:7.1f} TF/s {v / (B * H * Tq) * 1e3:6.3f} ns/row "
        print(line, flush=True)

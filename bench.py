#!/usr/bin/env python3
"""bench.py — LTX-2 19B dev denoise step (512x512x33, CFG 4.0, bf16) + video-VAE decode on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...; started without a
  launcher, `--gpus N` starts its own N ranks from a parent that never touches the GPU, or exits non-zero)

One "step" = the whole body of denoise_dev's loop (mlx_video/generate.py:1227-1304) at
BASELINE.json configs[1]: latent (1,128,5,16,16) -> N=1280 tokens, 48 blocks, D=4096, text
context 2x(1024x3840), cfg_batch (pos/neg branches as one B=2 forward, the CLI default,
generate.py:4643), CFG 4.0, x0 + Euler.  Nothing is cached across steps in the timed region:
caption projection and the text-context K/V projections are recomputed every forward exactly
as the reference does (SURVEY.md §8d counts them in the 69.7 TFLOP/step).  Synthetic inputs,
random-init weights of the exact architecture (no checkpoints exist offline).

Multi-GPU (weak scaling): every rank denoises its own seed with a full weight replica — the
path shards over independent forwards only; no data-path collective.  The
CFG-pair split with an RCCL all-gather of velocities (config 4) is `--shard cfgpair`.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_DENSE_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level table)
PEAK_HBM_GBS = 8000.0


def dit_forward_flops(N: int, B: int = 1, D: int = 4096, FF: int = 16384, S: int = 1024, L: int = 48, U: int = 0) -> float:
    """SURVEY.md §8d algorithmic FLOPs (2*MAC) of one DiT forward, per batch row, times B.  The survey's formula prices
    the AdaLN / timestep MLPs per TOKEN (2N(256D+D^2+6D^2), what the reference executes); with ``U`` > 0 they are priced
    for the U distinct timestep rows this implementation actually computes - used for achieved TFLOP/s."""
    per_block = 12 * N * D * D + 4 * N * D * FF + 4 * S * D * D + 4 * N * N * D + 4 * N * S * D
    ada_rows = N if U <= 0 else U / float(B)
    extra = 2 * ada_rows * (256 * D + D * D + 6 * D * D) + 2 * S * (3840 * D + D * D) + 4 * N * 128 * D
    return float(B) * (L * per_block + extra)


def _strip_comments(text: str) -> str:
    """C/C++ source without // and /* */ comments and blank lines (string literals respected): a comment edit must not
    invalidate the PMC artefacts stamped with source_sha()."""
    out, i, n, state = [], 0, len(text), 0          # state: 0 code, 1 string, 2 char, 3 line comment, 4 block comment
    while i < n:
        c, d = text[i], text[i + 1] if i + 1 < n else ""
        if state == 0:
            if c == "/" and d == "/":
                state, i = 3, i + 2
                continue
            if c == "/" and d == "*":
                state, i = 4, i + 2
                continue
            if c == '"':
                state = 1
            elif c == "'":
                state = 2
            out.append(c)
        elif state in (1, 2):
            out.append(c)
            if c == "\\" and i + 1 < n:
                out.append(d)
                i += 1
            elif (state == 1 and c == '"') or (state == 2 and c == "'"):
                state = 0
        elif state == 3:
            if c == "\n":
                out.append(c)
                state = 0
        elif state == 4 and c == "*" and d == "/":
            state, i = 0, i + 1
        i += 1
    return "\n".join(ln.rstrip() for ln in "".join(out).splitlines() if ln.strip())


def source_sha() -> str:
    """Hash of the kernel sources (comments and blank lines stripped) the loaded libltxk.so was built from (the .so itself
    is not tracked): PMC artefacts under profiles/ carry the hash they were measured on, and are only quoted while it
    still matches."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "mlx-video_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode())
            h.update(_strip_comments(open(os.path.join(csrc, name), encoding="utf-8").read()).encode())
    return h.hexdigest()[:16]


_PMC_CACHE = {}


def pmc_entries():
    """Newest profiles/r*_pmc_traffic.json measured on the kernel sources that are loaded now (source_sha match), or {}: the
    separate rocprofv3 --pmc passes of scripts/pmc_families.py (FETCH_SIZE / WRITE_SIZE gfx950-corrected, MFMA-busy, clock)."""
    if "v" not in _PMC_CACHE:
        import glob
        _PMC_CACHE["v"] = ({}, None)
        sha = source_sha()
        for pf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
            try:
                pm = json.load(open(pf))
            except (OSError, ValueError):
                continue
            if pm.get("source_sha") == sha:
                _PMC_CACHE["v"] = (pm.get("kernels", {}), os.path.basename(pf))
                break
    return _PMC_CACHE["v"]


def family_traffic(family: str, prefer: str):
    """`roofline.traffic` of one kernel family: the PMC record of its launch shape `prefer` (per launch, like `achieved`), plus
    MFMA-busy / clock / traffic ratio of every profiled shape of the family.  None when no PMC file matches the sources."""
    kern, fname = pmc_entries()
    fam = {n: v for n, v in kern.items() if v.get("family") == family and "bytes_per_launch" in v}
    if not fam:
        return None
    name = prefer if prefer in fam else sorted(fam)[0]
    k = fam[name]
    t = {"bytes_per_launch": k["bytes_per_launch"], "algorithmic_bytes": k["algorithmic_bytes"],
         "traffic_over_algorithmic": k["bytes_per_launch"] / k["algorithmic_bytes"], "kernel": f"{k.get('kernel')} [{k.get('shape')}]",
         "source_sha": source_sha(), "source": f"profiles/{fname} (rocprofv3 --pmc FETCH_SIZE x2 on gfx950 / WRITE_SIZE, separate passes)"}
    by = {n: v for n, v in fam.items() if v.get("mfma_busy_frac_of_simd_cycles")}
    if by:
        t["mfma_busy_frac_by_shape"] = {n: round(v["mfma_busy_frac_of_simd_cycles"], 3) for n, v in by.items()}
        t["clock_GHz_by_shape"] = {n: round(v["clock_GHz_profiled"], 2) for n, v in by.items() if v.get("clock_GHz_profiled")}
        t["traffic_over_algorithmic_by_shape"] = {n: round(v["bytes_per_launch"] / v["algorithmic_bytes"], 2) for n, v in fam.items()}
    return t


def cpu_baseline_block(threads: int):
    """CPU baseline (kind "port"): the oracle's transformer block, fp32, at the bench shape
    (B=2 CFG pair, N=1280, S=1024, D=4096) — REPS of the 48 blocks (about 10 s of host work), scaled to steps/s."""
    from oracle import dit as O
    torch.set_num_threads(threads)
    cfg = O.DiTConfig(num_layers=1)
    W = O.make_weights(cfg, seed=1234, dtype=torch.float32)
    g = torch.Generator().manual_seed(42)
    B, N, S, D = 2, 1280, 1024, cfg.dim
    x = torch.randn(B, N, D, generator=g)
    ts = torch.randn(B, N, 6 * D, generator=g) * 0.02
    ctx = torch.randn(B, S, D, generator=g)
    pos = torch.from_numpy(O.create_position_grid(1, 5, 16, 16))
    cos, sin = O.precompute_freqs_cis(pos, D)
    cos, sin = cos.expand(B, -1, -1, -1), sin.expand(B, -1, -1, -1)
    REPS = 5
    t0 = time.perf_counter()
    for _ in range(REPS):
        x = O.transformer_block(x, ts, ctx, (cos, sin), W, 0, cfg, O.F32)
    dt = (time.perf_counter() - t0) / REPS
    # "port": this repository's own torch-CPU fp32 restatement of the reference step (oracle/), NOT the reference's MLX-CPU
    # path - MLX is not installable here
    res = {"value": 1.0 / (dt * 48), "unit": "steps/s", "cores": threads, "kind": "port",
           "sample": f"{REPS} of 48 DiT blocks (oracle fp32, B=2 CFG pair, N=1280, S=1024, D=4096) = {dt * REPS:.1f} s, scaled x48/{REPS}"}
    del W, x, ts, ctx
    # ---- VAE leg (BASELINE.md "CPU-baseline plan"): the oracle's fp32 decode of a (1,128,2,4,4) latent -> 9 x 128 x 128 frames,
    # full depth (5 res blocks per stage), 0.212 TFLOP; decoder.py:361-450 restated in oracle/vae.py ----
    from oracle import vae as OV
    Wv = OV.make_decoder_weights(seed=1234, dtype=torch.float32)
    zl = torch.randn(1, 128, 2, 4, 4, generator=g)
    OV.vae_decode(zl, Wv, O.F32)                                     # warm-up (thread pool, allocator)
    VREPS = 2
    t0 = time.perf_counter()
    for _ in range(VREPS):
        vid = OV.vae_decode(zl, Wv, O.F32)
    dtv = (time.perf_counter() - t0) / VREPS
    res["vae"] = {"value": vid.shape[2] / dtv, "unit": "frames/s", "cores": threads, "kind": "port", "seconds_per_decode": dtv,
                  "tflops": 0.212 / dtv,
                  "sample": f"{VREPS} fp32 decodes of a (1,128,2,4,4) latent -> {vid.shape[2]} x {vid.shape[3]} x {vid.shape[4]} frames "
                            f"(oracle/vae.py::vae_decode, 5 res blocks per stage, 0.212 TFLOP each); the GPU leg decodes 33 x 512 x 512 "
                            f"(11.26 TFLOP) - compare TFLOP/s, not frames/s"}
    del Wv
    # ---- config 1 of BASELINE.json as BASELINE.md describes it: ONE dev step at 128x128x9 (N=32 tokens, CFG pair = 2 forwards,
    # context 1024x3840), full-width blocks, L=2 end to end (prepare + blocks + head + CFG + x0 + Euler); the 48-layer figure
    # adds 23 more pairs of blocks at the measured per-block time (13 B fp32 parameters = 52 GB are not materialised) ----
    cfg2 = O.DiTConfig(num_layers=2)
    W2 = O.make_weights(cfg2, seed=1234, dtype=torch.float32)
    lat = torch.randn(1, 128, 2, 4, 4, generator=g)
    cpos, cneg = torch.randn(1, 1024, cfg2.caption_channels, generator=g), torch.randn(1, 1024, cfg2.caption_channels, generator=g)
    pos1 = O.create_position_grid(1, 2, 4, 4)
    sig = [1.0, 0.0]
    t0 = time.perf_counter()
    O.denoise_dev(lat, pos1, cpos, cneg, W2, cfg2, sig, O.F32, 4.0)
    dt2 = time.perf_counter() - t0
    cfg0 = O.DiTConfig(num_layers=0)
    W0 = {k: v for k, v in W2.items() if not k.startswith("transformer_blocks.")}
    t0 = time.perf_counter()
    O.denoise_dev(lat, pos1, cpos, cneg, W0, cfg0, sig, O.F32, 4.0)
    dt0 = time.perf_counter() - t0
    res["dit_config1"] = {"seconds_per_step_L2": dt2, "seconds_per_step_L0_prepare_and_head": dt0,
                          "steps_per_s_L48_extrapolated": 1.0 / (dt0 + (dt2 - dt0) * 24.0), "cores": threads, "kind": "port",
                          "sample": "one dev step 128x128x9 (N=32, S=1024, CFG 4.0 = 2 forwards), oracle fp32, L=2 measured; "
                                    "L=48 = prepare/head time + 24 x the two blocks' time"}
    return res


def time_forward(model, B: int, N: int, reps: int = 3, dev=None):
    """Median ms of ONE DiT forward (all L blocks + prepare + head, text K/V recomputed) at batch B, N tokens per row, as a
    captured-graph replay.  Used for the one-GPU prediction of the CFG-pair split: a pair rank runs the B=1 forward."""
    from mlx_video_amd.ltx_model import TimestepPlan, precompute_freqs_cis
    from mlx_video_amd.schedulers import create_position_grid
    g = torch.Generator(device=dev).manual_seed(7)
    lat = torch.randn((B, N, 128), generator=g, device=dev).to(torch.bfloat16)
    ctx = torch.randn((B, 1024, 3840), generator=g, device=dev).to(torch.bfloat16)
    pos = create_position_grid(1, N // 256, 16, 16).to(dev)
    pe = precompute_freqs_cis(pos, 4096, 10000.0, (20, 2048, 2048), 32)
    plan = TimestepPlan(torch.tensor([0.7], device=dev).to(torch.bfloat16), torch.zeros(B * N, dtype=torch.int32, device=dev))
    for _ in range(2):
        model.forward_tokens(lat, plan, ctx, pe)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        model.forward_tokens(lat, plan, ctx, pe)
    gr.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        gr.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    del gr
    return sorted(ts)[len(ts) // 2]


def visible_gpu_count() -> int:
    """GPUs this process may use, counted WITHOUT touching the HIP runtime (torch.cuda.device_count() falls back to
    hipGetDeviceCount - which initialises it - whenever amdsmi is unavailable): the KFD topology lists every agent, GPUs are
    the nodes with simd_count > 0; HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES narrow that.  -1 = unknown."""
    import glob
    try:
        n = 0
        for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            props = dict(ln.split(None, 1) for ln in open(f).read().splitlines() if " " in ln)
            n += int(props.get("simd_count", "0")) > 0
    except (OSError, ValueError):
        return -1
    if n == 0:
        return -1
    # a container that was handed some of the host's GPUs still sees the whole topology; the render nodes it can open are its own
    try:
        rn = len(glob.glob("/dev/dri/renderD*"))
        if rn > 0:
            n = min(n, rn)
    except OSError:
        pass
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` started WITHOUT torch.distributed.run (no RANK in the environment): this parent - which
    never touches the GPU - starts the N ranks itself (one child process per GPU, the same environment torchrun would give
    them), relays rank 0's JSON line and exits non-zero as soon as any rank fails (the others are then terminated: a rank
    that dies early must not leave the rest waiting for the process-group timeout).  It never prints a line of its own, so
    an N-GPU request cannot come back as an n_gpus=1 measurement.  (Under `rocprofv3 -- python bench.py --gpus N` the
    profiler's preloaded library has initialised the GPU in this parent already: profile multi-GPU runs with an external
    launcher, one rocprofv3 per rank - profiles/README.md.)"""
    import socket
    import subprocess
    rehearsal = os.environ.get("LTXK_BENCH_REHEARSAL") == "1"
    have = visible_gpu_count()
    if have < 0:
        have = torch.cuda.device_count()
    if have < n and not rehearsal:
        print(f"[bench] --gpus {n} but only {have} GPU(s) are visible; refusing to measure fewer", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # rank 0's stdout is drained by a thread so that polling every child never blocks on a full pipe
    import threading
    out0 = []
    rd = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.2)
        for pr in list(live):
            code = pr.poll()
            if code is not None:
                live.remove(pr)
                rc = rc or code
    for pr in live:                      # a rank failed: stop the others (exact PIDs started above)
        pr.terminate()
    for pr in live:
        try:
            pr.wait(timeout=30)
        except subprocess.TimeoutExpired:
            pr.kill()
    rd.join(timeout=30)
    if rc == 0:
        # rank 0's result line only (a rehearsal over gloo also prints connection chatter on stdout)
        lines = [ln for ln in "".join(out0).splitlines() if ln.startswith("{") and '"metric"' in ln]
        if not lines:
            print("[bench] rank 0 printed no result line", file=sys.stderr)
            return 3
        print(lines[-1], flush=True)
    else:
        print(f"[bench] a rank failed (exit code {rc}); no result line", file=sys.stderr)
    return rc


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--layers", type=int, default=48, help="debug only; the reported config is always 48")
    ap.add_argument("--frames", type=int, default=33)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--shard", choices=["seeds", "cfgpair"], default="seeds")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of the captured step graph")
    ap.add_argument("--cache-context", action="store_true", help="reuse ctx K/V across steps (reported separately)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # LTXK_BENCH_REHEARSAL=1 (builder's one-GPU box only): all ranks share cuda:0 and talk over gloo, to execute the N>1
    # code paths of this file without an 8-GPU node; numbers from such a run are not throughput measurements.
    rehearsal = os.environ.get("LTXK_BENCH_REHEARSAL") == "1"
    dev = torch.device("cuda:0" if rehearsal else f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    dist = None
    if world > 1 or "RANK" in os.environ:          # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", str(world))
        if rehearsal:
            dist_mod.init_process_group("gloo")
        else:
            dist_mod.init_process_group("nccl", device_id=dev)       # "nccl" is RCCL on ROCm
        dist = dist_mod

    from mlx_video_amd import ops
    from mlx_video_amd.denoise import denoise_dev
    from mlx_video_amd.ltx_model import LTXModel, LTXModelConfig
    from mlx_video_amd.schedulers import create_position_grid, ltx2_scheduler

    cfg = LTXModelConfig(num_layers=args.layers)
    model = LTXModel.random_init(cfg, dev, seed=1234)
    Fl, Hl, Wl = 1 + (args.frames - 1) // 8, args.height // 32, args.width // 32
    N = Fl * Hl * Wl
    g = torch.Generator(device=dev).manual_seed(42 + (rank if args.shard == "seeds" else rank // 2))
    latents = torch.randn((1, 128, Fl, Hl, Wl), generator=g, device=dev).to(torch.bfloat16)
    g2 = torch.Generator(device=dev).manual_seed(43)
    ctx_pos = torch.randn((1, 1024, 3840), generator=g2, device=dev).to(torch.bfloat16)
    ctx_neg = torch.randn((1, 1024, 3840), generator=g2, device=dev).to(torch.bfloat16)
    positions = create_position_grid(1, Fl, Hl, Wl).to(dev)
    sig_all = ltx2_scheduler(40, N)

    pg_shard = None
    if args.shard == "cfgpair" and world > 1:
        from mlx_video_amd.sharding import CfgPairSharding
        pg_shard = CfgPairSharding(dist, rank, world)

    graph_cache = {}

    def run_steps(k: int, start: int = 0, graph: bool = True):
        # k consecutive steps of the 40-step schedule (sigma values only select scalars; cost is step-invariant).
        # The step is replayed from one captured hipGraph (built during warm-up, like the reference's
        # mx.compile'd step_fn, generate.py:1109-1177); every kernel still runs every step.
        s = sig_all[start:start + k + 1].clone()
        if pg_shard is not None:
            return pg_shard.denoise_dev(latents, positions, ctx_pos, ctx_neg, model, s, cfg_scale=4.0)
        return denoise_dev(latents, positions, ctx_pos, ctx_neg, model, s, cfg_scale=4.0, compile_step=True,
                           cfg_batch=True, use_graph=graph and not args.no_graph, graph_cache=graph_cache,
                           cache_context=args.cache_context)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    try:
        run_steps(max(args.warmup, 1) if not args.no_graph else args.warmup)      # W untimed steps (>= 1: builds the step graph)
    except RuntimeError as e:
        # Only a CAPTURE failure may fall back to eager launches (a failed capture must not cost the measurement); a kernel
        # fault also surfaces as RuntimeError and must end the run - a headline from a poisoned context is worthless.
        msg = str(e).lower()
        if args.no_graph or not any(w in msg for w in ("capture", "captur", "graph")):
            raise
        print(f"[bench] step-graph capture failed ({e}); eager launches", file=sys.stderr, flush=True)
        args.no_graph = True
        graph_cache.clear()
        probe = torch.ones(1024, device=dev)
        if float((probe * 2).sum().item()) != 2048.0:            # the context must still compute (and synchronise) correctly
            raise SystemExit("[bench] GPU context unhealthy after the failed capture")
        run_steps(args.warmup)
    # ---- timed region: exactly K steps, barrier + synchronize on both sides, MAX over ranks ----
    barrier()
    t0 = time.perf_counter()
    out = run_steps(args.steps, start=min(args.warmup, 40 - args.steps))
    barrier()
    dt = time.perf_counter() - t0
    if not torch.isfinite(out.float()).all():
        raise SystemExit("non-finite latents after the timed steps")
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # ---- the same K steps again with a HIP event pair around every kernel launch (on the launch stream):
    # per-kernel durations for the roofline object.  ~2000 extra event records per step perturb the
    # step time by a few %, so this pass is reported next to, not as, `value`.
    ops.TIMER = ops.KernelTimer()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    run_steps(args.steps, start=min(args.warmup, 40 - args.steps), graph=False)
    torch.cuda.synchronize()
    dt_instr = time.perf_counter() - t1
    timer, ops.TIMER = ops.TIMER, None
    # ---- optional extra: the same steps with the step-invariant caption projection + text-context K/V
    # computed once and reused (an algorithmic change relative to the reference, reported separately) ----
    dt_cached = None
    if not args.cache_context and pg_shard is None:
        gc2 = {}
        s_c = sig_all[:args.steps + 1].clone()
        kwc = dict(cfg_scale=4.0, compile_step=True, cfg_batch=True, use_graph=True, graph_cache=gc2, cache_context=True)
        denoise_dev(latents, positions, ctx_pos, ctx_neg, model, s_c[:3], **kwc)       # builds the graph
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        denoise_dev(latents, positions, ctx_pos, ctx_neg, model, s_c, **kwc)          # incl. the once-per-call text K/V
        torch.cuda.synchronize()
        dt_cached = time.perf_counter() - t2
        del gc2

    # ---- second line of the same run (N even, >= 2): BASELINE config 4's CFG-pair sharding - ranks (2i, 2i+1) denoise
    # seed i, one forward each, RCCL all-gather of the two velocities per step, fused tail on both (sharding.py) ----
    cfgpair = None
    if dist is not None and world >= 2 and world % 2 == 0 and pg_shard is None and os.environ.get("LTXK_BENCH_CFGPAIR", "1") != "0":
        from mlx_video_amd.sharding import CfgPairSharding
        try:
            sh = CfgPairSharding(dist, rank, world)
            g3 = torch.Generator(device=dev).manual_seed(42 + rank // 2)
            lat_p = torch.randn((1, 128, Fl, Hl, Wl), generator=g3, device=dev).to(torch.bfloat16)
            s_p = sig_all[:args.steps + 1].clone()
            sh.denoise_dev(lat_p, positions, ctx_pos, ctx_neg, model, s_p[:3], cfg_scale=4.0)     # builds the graphs
            barrier()
            t3 = time.perf_counter()
            out_p = sh.denoise_dev(lat_p, positions, ctx_pos, ctx_neg, model, s_p, cfg_scale=4.0)
            barrier()
            dtp = time.perf_counter() - t3
            tp = torch.tensor([dtp], device=dev, dtype=torch.float64)
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)
            dtp = float(tp.item())
            cfgpair = {"value": (world // 2) * args.steps / dtp, "unit": "denoise steps/s (whole job; one seed per rank PAIR)",
                       "ms_per_step": 1000.0 * dtp / args.steps, "pairs": world // 2, "finite": bool(torch.isfinite(out_p.float()).all()),
                       "collective": "all_gather_into_tensor of 2 x (1,N,128) bf16 per step per pair (RCCL)"}
        except dist.DistError as e:
            # Only a failure of the process-group machinery (group creation, the collective's backend, the network) may cost
            # the secondary line alone.  Anything else - a HIP fault, a capture error, a Python bug - ends the run: the
            # VAE leg below must not execute on a context a kernel has just poisoned.
            cfgpair = {"error": repr(e)[:300]}

    # ---- one-GPU prediction of the CFG-pair split (N=1 only): a pair rank runs ONE B=1 forward per step where this
    # rank runs the B=2 forward; expected 2-GPU efficiency of the split = t(B=2) / (2 * t(B=1)), before any RCCL cost.
    # Plus the 2-seed batch (B=4, M=5120: every N=4096 launch then fills whole rounds of 320x256 tiles) as an EXTRA line.
    cfgpair_pred = None
    batch2 = None
    if world == 1 and args.layers == 48 and os.environ.get("LTXK_BENCH_EXTRAS", "1") != "0":
        try:
            cfgpair_pred = {}
            for n_tok in (1280, 3328):
                t1, t2 = time_forward(model, 1, n_tok, dev=dev), time_forward(model, 2, n_tok, dev=dev)
                cfgpair_pred[f"N{n_tok}"] = {"forward_b1_ms": t1, "forward_b2_ms": t2, "cfgpair_predicted_efficiency": t2 / (2.0 * t1)}
            t4 = time_forward(model, 4, 1280, dev=dev)
            batch2 = {"forward_b4_ms": t4, "steps_per_s_two_seeds_batched": 2.0 / (t4 * 1e-3),
                      "vs_two_sequential_b2_forwards": 2.0 * cfgpair_pred["N1280"]["forward_b2_ms"] / t4,
                      "note": "two seeds' CFG pairs as ONE B=4 forward (M=5120), forward only (no step tail); reported beside, never instead of, the B=2 headline"}
        except Exception as e:
            cfgpair_pred = {"error": repr(e)[:300]}

    seeds = world if pg_shard is None else max(world // 2, 1)
    steps_per_s = seeds * args.steps / dt
    fams = timer.summary()

    result = {
        "metric": "denoise steps/sec/GPU + VAE frames/sec, LTX-2 19B 512x512x33 bf16",
        "value": steps_per_s,
        "unit": "denoise steps/s (whole job; one step = 2 CFG forwards + CFG + x0 + Euler)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt / args.steps,
        "ms_per_step_instrumented": 1000.0 * dt_instr / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic (random-init weights of the LTX-2 video DiT, N(0,1) latents/context)",
        "config": {"workload": f"LTX-2 19B dev {args.width}x{args.height}x{args.frames}, CFG 4.0, cfg_batch, "
                               f"N={N} tokens, L={args.layers}, ctx 1024x3840, {args.shard} sharding",
                   "global_batch": seeds, "tokens": N, "parallelism": f"{args.shard}{world}",
                   "ctx_kv_cached": bool(args.cache_context), "step_graph": not args.no_graph},
    }
    step_flops = dit_forward_flops(N, B=2, L=args.layers)                 # SURVEY.md §8d figure (69.7 TFLOP at N=1280)
    exec_flops = dit_forward_flops(N, B=2, L=args.layers, U=1)            # with the AdaLN MLPs priced for the U=1 row computed
    if dt_cached is not None:
        result["value_ctx_kv_cached"] = seeds * args.steps / dt_cached
        result["ms_per_step_ctx_kv_cached"] = 1000.0 * dt_cached / args.steps
    result["step_tflop"] = step_flops / 1e12
    result["step_tflop_executed"] = exec_flops / 1e12
    result["achieved_tflops_per_gpu"] = exec_flops * args.steps / dt / 1e12 * (1 if pg_shard is None else 0.5)
    # ---- roofline: one entry per kernel family, measured live (HIP events around every launch of the instrumented
    # pass, on the launch stream).  achieved = algorithmic FLOPs (or bytes) of the family / its summed launch time.
    rooflines = {}
    if "gemm_bf16" in fams:
        gm = fams["gemm_bf16"]
        ach = gm["flops"] / (gm["ms"] * 1e-3) / 1e12
        # L2->fabric bytes of the dominant GEMM launch (FF1) from separate --pmc passes (gfx950-corrected); quoted only while the
        # kernel sources still hash to what the passes were measured on
        traffic = family_traffic("gemm_bf16", "gemm_ff1_gelu")
        rooflines["gemm_bf16"] = {"kernel": "ltxk::gemm_bf16_kernel + gemm_bf16_big_kernel (all Linear layers; algorithmic FLOPs = sum 2*M*N*K per launch)",
                                  "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
                                  "frac": ach / PEAK_BF16_DENSE_TFLOPS, "traffic": traffic,
                                  "launches": gm["launches"], "avg_ms": gm["ms"] / gm["launches"],
                                  # context, not a second roofline: bf16 GEMMs on random data are power-limited on this chip
                                  "power_limited_reference": {"tuned_bf16_gemm_random_data_tflops": 1247.0,
                                                              "mfma_only_register_loop_random_data_tflops": 2030.0,
                                                              "same_instruction_mix_synthetic_loop_tflops": {"160x256 tile": 1210.0, "320x256 tile": 1410.0},
                                                              "source": "MI355X_MICROARCH.md (DVFS give-back); scripts/mfma_power_probe.hip -> "
                                                                        "profiles/r02_mfma_power_probe.log; these binaries on all-zero operands: "
                                                                        "1433-1542 TF/s (profiles/r02_gemm_zero_vs_random.log)"}}
    if "flash_attn" in fams:
        fa = fams["flash_attn"]
        ach = fa["flops"] / (fa["ms"] * 1e-3) / 1e12
        rooflines["flash_attn"] = {"kernel": "ltxk::flash_attn16_kernel (4*B*H*Tq*Tk*128 FLOP per launch)", "bound": "mfma", "achieved": ach,
                                   "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_DENSE_TFLOPS,
                                   "traffic": family_traffic("flash_attn", "attn_1280x1280"),
                                   "launches": fa["launches"], "avg_ms": fa["ms"] / fa["launches"]}
        result["attention_tflops"] = ach
    for k in ("rmsnorm_modulate", "qknorm_rope"):
        if k in fams and fams[k]["ms"] > 0:
            gbs = fams[k]["bytes"] / (fams[k]["ms"] * 1e-3) / 1e9
            rooflines[k] = {"kernel": f"ltxk {k} (algorithmic bytes: read x [+ tables], write y)", "bound": "hbm", "achieved": gbs,
                            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                            "traffic": family_traffic(k, "norm_mod" if k == "rmsnorm_modulate" else "qknorm_k"),
                            "launches": fams[k]["launches"], "avg_ms": fams[k]["ms"] / fams[k]["launches"]}
            result[f"{k}_GBs"] = gbs
    if "gemm_bf16" in rooflines:
        result["roofline"] = rooflines["gemm_bf16"]          # the dominant kernel (85 % of the step)
    result["roofline_by_family"] = rooflines
    result["kernel_breakdown_ms_per_step"] = {k: v["ms"] / args.steps for k, v in fams.items()}
    result["kernel_source_sha"] = source_sha()
    if cfgpair is not None:
        result["cfgpair"] = cfgpair
    if cfgpair_pred is not None:
        result["cfgpair_one_gpu_prediction"] = cfgpair_pred
        if "N1280" in cfgpair_pred:
            result["forward_b1_ms"] = cfgpair_pred["N1280"]["forward_b1_ms"]
            # the split is what BASELINE config 4 uses (512x512x97: N=3328 tokens per forward); N=1280 is this bench's own shape
            result["cfgpair_predicted_efficiency"] = {"N1280": cfgpair_pred["N1280"]["cfgpair_predicted_efficiency"],
                                                      "N3328_config4": cfgpair_pred.get("N3328", {}).get("cfgpair_predicted_efficiency")}
    if batch2 is not None:
        result["two_seed_batch"] = batch2

    if not args.no_vae:
        try:
            from mlx_video_amd import video_vae
            vres = video_vae.bench_decode(dev, Fl, Hl, Wl)
            result.update(vres)
            if "vae_conv3d_tflops" in vres:
                result["roofline_by_family"]["conv3d_k3"] = {
                    "kernel": "ltxk::conv3d_k3_kw_kernel + conv3d_k3_kernel (2*27*Cin*Cout*voxels FLOP per launch)", "bound": "mfma",
                    "achieved": vres["vae_conv3d_tflops"], "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
                    "frac": vres["vae_conv3d_tflops"] / PEAK_BF16_DENSE_TFLOPS, "traffic": family_traffic("conv3d_k3", "conv128"),
                    "launches": vres.get("vae_conv3d_launches"),
                    "avg_ms": (vres["vae_kernel_breakdown_ms"]["conv3d_k3"] / vres["vae_conv3d_launches"]
                               if vres.get("vae_conv3d_launches") and "vae_kernel_breakdown_ms" in vres else None)}
            if "vae_pixelnorm_GBs" in vres:
                result["roofline_by_family"]["pixelnorm_act"] = {
                    "kernel": "ltxk::pixelnorm_act_kernel (algorithmic bytes: read x, write y)", "bound": "hbm", "achieved": vres["vae_pixelnorm_GBs"],
                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": vres["vae_pixelnorm_GBs"] / PEAK_HBM_GBS,
                    "traffic": family_traffic("pixelnorm_act", "pixelnorm128"), "launches": vres.get("vae_pixelnorm_launches"),
                    "avg_ms": vres["vae_kernel_breakdown_ms"]["pixelnorm_act"] / vres["vae_pixelnorm_launches"]}
        except ImportError:
            result["vae_decode_fps"] = None

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        del model
        torch.cuda.empty_cache()
        result["cpu_baseline"] = cpu_baseline_block(min(len(os.sched_getaffinity(0)), 16))
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

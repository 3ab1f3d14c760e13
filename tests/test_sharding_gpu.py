"""CfgPairSharding's HIP path (forward graph | collective | tail graph) on real kernels.

* ``test_cfg_pair_hip_one_gpu_gloo``: two processes share cuda:0 and exchange over gloo - everything of the sharded
  loop except RCCL itself runs (device-side step scalars, captured forward / tail graphs, preallocated receive
  buffer); each rank must reproduce the single-process ``denoise_dev(cfg_batch=False)`` latents BIT FOR BIT.
* ``test_cfg_pair_hip_two_gpus_rccl``: the same over RCCL with one rank per GPU; skipped below two devices (the
  builder's box has one; the driver's 8-GPU node runs it)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(dev):
    from oracle import dit as O
    from mlx_video_amd.ltx_model import LTXModel, LTXModelConfig
    from mlx_video_amd.schedulers import create_position_grid, ltx2_scheduler
    cfg = O.DiTConfig(num_layers=2, heads=4, caption_channels=256)
    W = O.make_weights(cfg, seed=7)
    mc = LTXModelConfig(num_attention_heads=4, num_layers=2, caption_channels=256, cross_attention_dim=cfg.dim)
    model = LTXModel(mc, {k: v.to(dev) for k, v in W.items()})
    g = torch.Generator().manual_seed(42)
    lat = torch.randn(1, 128, 2, 4, 4, generator=g).to(torch.bfloat16).to(dev)
    cp = torch.randn(1, 64, 256, generator=g).to(torch.bfloat16).to(dev)
    cn = torch.randn(1, 64, 256, generator=g).to(torch.bfloat16).to(dev)
    sig = ltx2_scheduler(STEPS, 32)
    pos = create_position_grid(1, 2, 4, 4).to(dev)
    return model, lat, pos, cp, cn, sig


def _worker(rank, world, port, backend, one_gpu, q):
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dev = torch.device("cuda:0" if one_gpu else f"cuda:{rank}")
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        from mlx_video_amd.denoise import denoise_dev
        from mlx_video_amd.sharding import CfgPairSharding
        model, lat, pos, cp, cn, sig = _setup(dev)
        ref = denoise_dev(lat, pos, cp, cn, model, sig, cfg_scale=4.0, compile_step=True, cfg_batch=False)
        sh = CfgPairSharding(dist, rank, world)
        out_eager = sh.denoise_dev(lat, pos, cp, cn, model, sig, cfg_scale=4.0, use_graph=False)
        out_graph = sh.denoise_dev(lat, pos, cp, cn, model, sig, cfg_scale=4.0, use_graph=True)
        out_graph2 = sh.denoise_dev(lat, pos, cp, cn, model, sig, cfg_scale=4.0, use_graph=True)     # replay of the cached graphs
        torch.cuda.synchronize()
        q.put((rank, bool(torch.equal(out_eager, ref)), bool(torch.equal(out_graph, ref)), bool(torch.equal(out_graph2, ref)),
               float((out_graph.float() - ref.float()).abs().max())))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                                    # surface the failure instead of a queue timeout
        q.put((rank, False, False, False, repr(e)))
        raise


def _run(backend, one_gpu):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, backend, one_gpu, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
    for rank, eq_eager, eq_graph, eq_replay, info in res:
        assert eq_eager, f"rank {rank}: eager sharded loop differs from the single-process loop ({info})"
        assert eq_graph and eq_replay, f"rank {rank}: graph-replayed sharded loop differs from the single-process loop ({info})"
    for p in procs:
        assert p.exitcode == 0


def test_cfg_pair_hip_one_gpu_gloo():
    _run("gloo", True)


def test_cfg_pair_hip_two_gpus_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one rank per GPU over RCCL)")
    _run("nccl", False)


def test_bench_n2_code_path_rehearsal():
    """bench.py's N>1 code - the self-launcher, init_process_group, per-pair new_group, the weak-scaling timing with its
    barriers / MAX over ranks and the cfgpair leg - executed as a fresh child process with two ranks sharing this box's
    one GPU over gloo (LTXK_BENCH_REHEARSAL=1; the numbers of such a run are not measurements).  On the driver's 8-GPU node
    the same code runs over RCCL; this test keeps it from being executed there for the first time."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["LTXK_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--layers", "2", "--steps", "2", "--warmup", "1",
                        "--no-vae", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["value"] > 0 and res["scaling"] == "weak"
    assert res["cfgpair"].get("finite") is True and res["cfgpair"]["pairs"] == 1, res["cfgpair"]


def test_bench_under_torchrun_one_rank_over_rccl():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 ...`: the launcher form the driver uses, with the REAL
    backend - init_process_group("nccl", device_id=...) is RCCL on ROCm - on this box's one GPU: process-group start-up, the barriers
    and the MAX all_reduce of the timed region, teardown.  (What a one-GPU box can execute of the RCCL path; pair groups and the
    all_gather need two devices: test_cfg_pair_hip_two_gpus_rccl.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "LTXK_BENCH_REHEARSAL")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--layers", "2", "--steps", "2",
                        "--warmup", "1", "--no-vae", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["config"]["parallelism"] == "seeds1"

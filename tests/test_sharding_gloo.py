"""N>1 path on CPU: world_size-2, -4 and -8 gloo process groups exercise CfgPairSharding's partition and
velocity exchange with stand-in forward/tail functions (the HIP kernels need a GPU; what is checked here
is that the sharded loop reproduces the single-process loop bit for bit and stays replicated)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _fwd(tok, s, ctx):
    return torch.tanh(tok * (0.5 + s) + ctx.mean(dim=1, keepdim=True)[..., :128])


def _tail(vp, vn, lat, cfg, s, sn):
    v = vp + (cfg - 1.0) * (vp - vn)
    b, c = lat.shape[:2]
    vel = v.permute(0, 2, 1).reshape(lat.shape)
    x0 = lat - s * vel
    return x0 + sn * (lat - x0) / s if sn > 0 else x0


def _tokens(lat):
    b, c = lat.shape[:2]
    return lat.reshape(b, c, -1).permute(0, 2, 1).contiguous()


def _inputs(seed):
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn(1, 128, 2, 2, 2, generator=g)
    cp = torch.randn(1, 4, 128, generator=g)
    cn = torch.randn(1, 4, 128, generator=g)
    return lat, cp, cn


def _single(seed, sig):
    lat, cp, cn = _inputs(seed)
    for i in range(len(sig) - 1):
        from mlx_video_amd.sharding import _bf16_round
        s, sn = _bf16_round(float(sig[i])), _bf16_round(float(sig[i + 1]))
        tok = _tokens(lat)
        lat = _tail(_fwd(tok, s, cp), _fwd(tok, s, cn), lat, 4.0, s, sn)
    return lat


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mlx_video_amd.sharding import CfgPairSharding
    sh = CfgPairSharding(dist, rank, world)
    lat, cp, cn = _inputs(100 + sh.pair)
    sig = torch.tensor([1.0, 0.7, 0.3, 0.0])
    out = sh.denoise_dev(lat, None, cp, cn, None, sig, cfg_scale=4.0, forward_fn=_fwd, tail_fn=_tail, tokens_fn=_tokens)
    q.put((rank, sh.pair, sh.branch, out))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 4, 8])          # 8 = config 4's actual layout: 4 seeds x CFG pair
def test_cfg_pair_sharding_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sig = torch.tensor([1.0, 0.7, 0.3, 0.0])
    for rank, pair, branch, out in res:
        assert pair == rank // 2 and branch == rank % 2
        assert torch.equal(out, _single(100 + pair, sig)), f"rank {rank} diverged from the unsharded loop"
    for i in range(0, world, 2):
        assert torch.equal(res[i][3], res[i + 1][3])          # latents stay replicated inside a pair


def test_cfg_pair_sharding_rejects_odd_world():
    from mlx_video_amd.sharding import CfgPairSharding
    with pytest.raises(ValueError, match="even world size"):
        CfgPairSharding(None, 0, 3)


def test_bench_gpus_n_without_launcher_never_reports_one_gpu():
    """`python bench.py --gpus 8` with no RANK in the environment starts its own ranks or fails; on a machine with fewer
    GPUs (this container has none) it must exit non-zero WITHOUT printing a result line."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LTXK_BENCH_REHEARSAL")}
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("8 GPUs visible: the launcher would really start the bench")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "n_gpus" not in r.stdout and "refusing" in r.stderr


def test_bench_self_launch_fails_fast_when_a_rank_dies():
    """The self-launcher polls every child: ranks that die at start-up (no GPU in this container: set_device raises) end
    the run at once with a non-zero code and no result line, instead of leaving the others to the process-group timeout."""
    import subprocess, sys, time
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the rehearsal ranks would really run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["LTXK_BENCH_REHEARSAL"] = "1"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--layers", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and '"metric"' not in r.stdout and "a rank failed" in r.stderr
    assert time.time() - t0 < 120

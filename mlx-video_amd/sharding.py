"""Multi-GPU sharding of the denoise loop.  The path shards only over independent
forwards: seeds, and the positive/negative CFG branches of one seed (generate.py:1239-1255).

``CfgPairSharding``: ranks (2i, 2i+1) form a pair that denoises seed i; rank parity picks the branch
(0 = positive context, 1 = negative).  Per step each rank runs ONE forward, the pair all-gathers the
two (B,N,128) velocities (RCCL over xGMI on GPUs; <= 0.85 MB/rank at N=3328, latency-bound — one
all_gather_into_tensor into a preallocated (2,B,N,128) buffer, not a ring of chunks), then both ranks
redundantly run the fused CFG + x0 + Euler kernel, so the latents stay replicated without a broadcast.
One process per GPU; the weights are a full replica per rank (26 GB of 288 GB).

The step is latency-bound, so the HIP path keeps the host out of it: the forward (with the device-side
step-scalar node, see denoise._StepGraph) is one captured hipGraph, the fused tail a second one, and the
only eager call between them is the collective itself — no per-step allocation, host->device copy or
Python tensor op.  No 1/2/4/8-GPU curve has been measured on hardware yet."""
from __future__ import annotations

from typing import Callable, Optional

import torch


def _bf16_round(x: float) -> float:
    return float(torch.tensor(x, dtype=torch.float32).to(torch.bfloat16).to(torch.float32))


class CfgPairSharding:
    def __init__(self, dist, rank: int, world: int):
        if world < 2 or world % 2:
            raise ValueError(f"CFG-pair sharding needs an even world size >= 2, got {world}")
        self.dist, self.rank, self.world = dist, rank, world
        self.pair, self.branch = rank // 2, rank % 2
        # every rank must create every group, in the same order
        groups = [dist.new_group([2 * i, 2 * i + 1]) for i in range(world // 2)]
        self.group = groups[self.pair]
        self._vbuf: Optional[torch.Tensor] = None
        self._graphs: dict = {}

    def exchange(self, v_local: torch.Tensor):
        """all-gather the pair's velocities -> (v_pos, v_neg); the (2,...) receive buffer is allocated once."""
        v_local = v_local.contiguous()
        b = v_local.shape[0]
        want = (2 * b,) + tuple(v_local.shape[1:])              # concatenation along dim 0 (the form every backend takes)
        if (self._vbuf is None or tuple(self._vbuf.shape) != want or self._vbuf.dtype != v_local.dtype
                or self._vbuf.device != v_local.device):
            self._vbuf = torch.empty(want, dtype=v_local.dtype, device=v_local.device)
        self.dist.all_gather_into_tensor(self._vbuf, v_local, group=self.group)
        return self._vbuf[:b], self._vbuf[b:]

    def denoise_dev(self, latents, positions, text_embeddings_pos, text_embeddings_neg, transformer, sigmas,
                    cfg_scale: float = 4.0, state=None, forward_fn: Optional[Callable] = None,
                    tail_fn: Optional[Callable] = None, tokens_fn: Optional[Callable] = None, use_graph: bool = True):
        """Sharded twin of denoise.denoise_dev (compiled-step sigma semantics).  ``forward_fn(tok, sigma_bf16,
        ctx) -> velocity`` / ``tail_fn(v_pos, v_neg, latents, cfg, s, s_next) -> latents`` / ``tokens_fn(latents)``
        default to the HIP path; the CPU (gloo) tests inject stand-ins to exercise the exchange logic."""
        sig = [float(s) for s in (sigmas.tolist() if torch.is_tensor(sigmas) else sigmas)]
        if forward_fn is None:
            return self._denoise_hip(latents, positions, text_embeddings_pos, text_embeddings_neg, transformer, sig,
                                     cfg_scale, state, use_graph)
        ctx_local = text_embeddings_pos if self.branch == 0 else text_embeddings_neg
        for i in range(len(sig) - 1):
            s_bf, sn_bf = _bf16_round(sig[i]), _bf16_round(sig[i + 1])
            v_local = forward_fn(tokens_fn(latents), s_bf, ctx_local)
            v_pos, v_neg = self.exchange(v_local)
            latents = tail_fn(v_pos, v_neg, latents, cfg_scale, s_bf, sn_bf)
        return latents

    # ------------------------------------------------------------------ HIP path
    def _denoise_hip(self, latents, positions, ctx_pos, ctx_neg, transformer, sig, cfg_scale, state, use_graph):
        from . import ops
        from .denoise import GRAPH_MAX_STEPS, _StepPlan
        from .ltx_model import TimestepPlan, precompute_freqs_cis
        BF16 = torch.bfloat16
        latents = (state.latent if state is not None else latents).to(BF16).contiguous()
        pe = precompute_freqs_cis(positions[:1].contiguous(), transformer.inner_dim, transformer.positional_embedding_theta,
                                  transformer.positional_embedding_max_pos, transformer.num_attention_heads)
        plan = _StepPlan(latents, state, 1, sig)
        ctx = (ctx_pos if self.branch == 0 else ctx_neg).to(BF16).contiguous()
        nst = len(sig) - 1
        if not use_graph or nst > GRAPH_MAX_STEPS:
            for i in range(nst):
                v_local = transformer.forward_tokens(ops.latent_to_tokens(latents), plan.timestep_plan(i), ctx, pe)
                v_pos, v_neg = self.exchange(v_local)
                latents = ops.cfg_euler_step(v_pos, v_neg, latents, cfg_scale, plan.sig_bf[i], plan.sig_bf[i + 1], plan.clean,
                                             plan.mask_tok_f32)
            return latents
        key = (tuple(latents.shape), tuple(ctx.shape), id(transformer), float(cfg_scale), state is not None, plan.U)
        ent = self._graphs.get(key)
        if ent is None:
            ent = _ShardGraphs(self, latents, plan, transformer, ctx, cfg_scale)
            self._graphs[key] = ent
        return ent.run(latents, plan, ctx, pe)


class _ShardGraphs:
    """Forward graph | eager all_gather_into_tensor | tail graph, with persistent buffers refreshed per call."""

    def __init__(self, sh: CfgPairSharding, latents, plan, transformer, ctx, cfg_scale):
        from .denoise import GRAPH_MAX_STEPS
        BF16 = torch.bfloat16
        dev = latents.device
        self.sh, self.tr, self.cfg_scale = sh, transformer, cfg_scale
        self.lat_buf = torch.empty_like(latents)
        self.ctx = torch.empty_like(ctx)
        self.clean = torch.empty_like(plan.clean) if plan.clean is not None else None
        self.mask_tok = torch.empty_like(plan.mask_tok_f32) if plan.mask_tok_f32 is not None else None
        self.tok2row = torch.empty_like(plan.tok2row)
        self.pe = None
        self.ts_all = torch.zeros((GRAPH_MAX_STEPS, plan.U), dtype=BF16, device=dev)
        self.sig_all = torch.zeros((GRAPH_MAX_STEPS, 2), dtype=torch.float32, device=dev)
        self.step = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.ts_buf = torch.zeros((plan.U,), dtype=BF16, device=dev)
        self.sig_buf = torch.zeros((2,), dtype=torch.float32, device=dev)
        b, _, f, h, w = latents.shape
        self.b = b
        self.vbuf = torch.empty((2 * b, f * h * w, transformer.config.out_channels), dtype=BF16, device=dev)
        self.v_local = None
        self.g_fwd = self.g_tail = None

    def _fwd(self):
        from . import ops
        from .ltx_model import TimestepPlan
        ops.step_scalars(self.ts_all, self.sig_all, self.step, self.ts_buf, self.sig_buf)
        return self.tr.forward_tokens(ops.latent_to_tokens(self.lat_buf), TimestepPlan(self.ts_buf, self.tok2row), self.ctx, self.pe)

    def _tail(self):
        from . import ops
        ops.cfg_euler_step(self.vbuf[:self.b], self.vbuf[self.b:], self.lat_buf, self.cfg_scale, 1.0, 0.0, self.clean, self.mask_tok,
                           out=self.lat_buf, sigmas_dev=self.sig_buf)

    def _gather(self):
        self.sh.dist.all_gather_into_tensor(self.vbuf, self.v_local, group=self.sh.group)

    def run(self, latents, plan, ctx, pe):
        self.lat_buf.copy_(latents)
        self.ctx.copy_(ctx)
        if self.clean is not None:
            self.clean.copy_(plan.clean)
            self.mask_tok.copy_(plan.mask_tok_f32)
        self.tok2row.copy_(plan.tok2row)
        if self.pe is None:
            self.pe = (pe[0].clone(), pe[1].clone())
        else:
            self.pe[0].copy_(pe[0]); self.pe[1].copy_(pe[1])
        nst = plan.ts_host.shape[0]
        self.ts_all[:nst].copy_(plan.ts_host)
        self.sig_all[:nst].copy_(plan.sig_host)
        self.step.zero_()
        start = 0
        if self.g_fwd is None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.v_local = self._fwd()                # real step 0, eager
            torch.cuda.current_stream().wait_stream(side)
            self._gather()
            self._tail()
            torch.cuda.synchronize()
            self.g_fwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_fwd, capture_error_mode="thread_local"):
                self.v_local = self._fwd()                # static output tensor of the captured forward
            self.g_tail = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_tail, capture_error_mode="thread_local"):
                self._tail()
            start = 1
        for _ in range(start, nst):
            self.g_fwd.replay()
            self._gather()
            self.g_tail.replay()
        return self.lat_buf.clone()

"""Multi-GPU sharding of the denoise loop.  The path shards only over independent
forwards: seeds, and the positive/negative CFG branches of one seed (generate.py:1239-1255).

``CfgPairSharding``: ranks (2i, 2i+1) form a pair that denoises seed i; rank parity picks the branch
(0 = positive context, 1 = negative).  Per step each rank runs ONE forward, the pair all-gathers the
two (B,N,128) velocities (RCCL over xGMI on GPUs; <= 0.85 MB/rank at N=3328, latency-bound — one
all_gather, not a ring of chunks), then both ranks redundantly run the fused CFG + x0 + Euler kernel,
so the latents stay replicated without a broadcast.  One process per GPU; the weights are a full
replica per rank (26 GB of 288 GB)."""
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

    def exchange(self, v_local: torch.Tensor):
        """all-gather the pair's velocities -> (v_pos, v_neg)."""
        bufs = [torch.empty_like(v_local), torch.empty_like(v_local)]
        self.dist.all_gather(bufs, v_local.contiguous(), group=self.group)
        return bufs[0], bufs[1]

    def denoise_dev(self, latents, positions, text_embeddings_pos, text_embeddings_neg, transformer, sigmas,
                    cfg_scale: float = 4.0, state=None, forward_fn: Optional[Callable] = None,
                    tail_fn: Optional[Callable] = None, tokens_fn: Optional[Callable] = None):
        """Sharded twin of denoise.denoise_dev (compiled-step sigma semantics).  ``forward_fn(tok, sigma_bf16,
        ctx) -> velocity`` / ``tail_fn(v_pos, v_neg, latents, cfg, s, s_next) -> latents`` / ``tokens_fn(latents)``
        default to the HIP path; the CPU (gloo) tests inject stand-ins to exercise the exchange logic."""
        if forward_fn is None:
            from . import ops
            from .denoise import _StepPlan
            from .ltx_model import precompute_freqs_cis
            latents = (state.latent if state is not None else latents).to(torch.bfloat16).contiguous()
            pe = precompute_freqs_cis(positions[:1].contiguous(), transformer.inner_dim, transformer.positional_embedding_theta,
                                      transformer.positional_embedding_max_pos, transformer.num_attention_heads)
            plan = _StepPlan(latents, state, 1)
            ctx = (text_embeddings_pos if self.branch == 0 else text_embeddings_neg).to(torch.bfloat16).contiguous()

            def forward_fn(tok, s_bf, _ctx):
                return transformer.forward_tokens(tok, plan.timestep_plan(s_bf), ctx, pe)

            def tail_fn(vp, vn, lat, cfg, s, sn):
                return ops.cfg_euler_step(vp, vn, lat, cfg, s, sn, plan.clean, plan.mask_tok_f32)

            tokens_fn = ops.latent_to_tokens
        ctx_local = text_embeddings_pos if self.branch == 0 else text_embeddings_neg
        sig = [float(s) for s in sigmas.tolist()]
        for i in range(len(sig) - 1):
            s_bf, sn_bf = _bf16_round(sig[i]), _bf16_round(sig[i + 1])
            v_local = forward_fn(tokens_fn(latents), s_bf, ctx_local)
            v_pos, v_neg = self.exchange(v_local)
            latents = tail_fn(v_pos, v_neg, latents, cfg_scale, s_bf, sn_bf)
        return latents

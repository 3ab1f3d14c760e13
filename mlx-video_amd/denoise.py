"""Denoise loops: ``denoise_dev`` (CFG) and ``denoise_distilled`` (mlx_video/generate.py:1060-1327,
564-881), video branch.  The step body is: tokens <- latent transpose, DiT forward(s), then one
fused kernel for CFG + x0 + mask blend + Euler (ltxk_cfg_euler_step).

Sigma handling follows the reference exactly (SURVEY.md §7 "bf16-quantised timesteps"):
timesteps = bf16(sigma)*mask always (generate.py:1084,1237); with ``compile_step`` x0 and Euler
use the bf16-rounded sigma (1160-1174), without it x0 uses bf16(sigma) and Euler the Python
float (1288,1293-1301)."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import ops
from .conditioning import LatentState
from .ltx_model import LTXModel, TimestepPlan, precompute_freqs_cis

BF16 = torch.bfloat16


def _bf16_round(x: float) -> float:
    return float(torch.tensor(x, dtype=torch.float32).to(BF16).to(torch.float32))


class _StepPlan:
    """Per-loop constants: token->row map of the (per-frame) denoise mask, RoPE table."""

    def __init__(self, latents: torch.Tensor, state: Optional[LatentState], batch_rep: int):
        b, c, f, h, w = latents.shape
        n = f * h * w
        dev = latents.device
        if state is not None:
            m = state.denoise_mask.reshape(b, 1, f, 1, 1).to(BF16)
            mask_tok = m.expand(b, 1, f, h, w).reshape(b, n)
        else:
            mask_tok = torch.ones((b, n), dtype=BF16, device=dev)
        vals, inv = torch.unique(mask_tok.reshape(-1), sorted=True, return_inverse=True)
        self.mask_vals = vals.to(torch.float32).cpu().tolist()        # U distinct mask values (host)
        self.tok2row = inv.to(torch.int32).repeat(batch_rep).contiguous()
        self.mask_tok_f32 = mask_tok.to(torch.float32).contiguous() if state is not None else None
        self.clean = state.clean_latent.to(BF16).contiguous() if state is not None else None
        self.dev = dev

        self._ts_buf = torch.zeros((len(self.mask_vals),), dtype=BF16, device=dev)      # stable address (graph replay)
        self._sig_buf = torch.zeros((2,), dtype=torch.float32, device=dev)

    def timestep_values(self, sigma_bf16: float) -> torch.Tensor:
        # timesteps = sigma_bf16 * mask (bf16 multiply), one value per distinct mask entry
        return torch.tensor([sigma_bf16 * m for m in self.mask_vals], dtype=torch.float32).to(BF16)

    def timestep_plan(self, sigma_bf16: float) -> TimestepPlan:
        return TimestepPlan(self.timestep_values(sigma_bf16).to(self.dev), self.tok2row)

    def set_step_scalars(self, sigma_bf16: float, sigma_next_bf16: float) -> TimestepPlan:
        """Write this step's scalars into the persistent device buffers the captured graph reads."""
        self._ts_buf.copy_(self.timestep_values(sigma_bf16))
        self._sig_buf.copy_(torch.tensor([sigma_bf16, sigma_next_bf16], dtype=torch.float32))
        return TimestepPlan(self._ts_buf, self.tok2row)


def denoise_dev(latents: torch.Tensor, positions: torch.Tensor, text_embeddings_pos: torch.Tensor,
                text_embeddings_neg: torch.Tensor, transformer: LTXModel, sigmas: torch.Tensor,
                cfg_scale: float = 4.0, verbose: bool = False, state: Optional[LatentState] = None,
                eval_interval: int = 1, compile_step: bool = False, compile_shapeless: bool = False,
                cfg_batch: bool = False, ui_phase: str = "denoise", use_graph: bool = False,
                graph_cache: Optional[dict] = None) -> torch.Tensor:
    """generate.py:1060-1327.  latents (B,128,F,H,W) bf16 on the GPU; returns the same shape.
    ``use_graph``: capture the whole step (forward(s) + fused tail, ~1000 launches) once as a hipGraph and
    replay it per step — the analogue of the reference's mx.compile'd step_fn (generate.py:1109-1177);
    requires compile_step semantics (bf16 sigmas) and gives bit-identical results to the eager path.
    ``graph_cache`` (a dict owned by the caller) keeps the captured graph across calls with the same
    shapes/contexts, like a compiled function that is traced once."""
    if state is not None:
        latents = state.latent
    latents = latents.to(BF16).contiguous()
    sig = [float(s) for s in sigmas.tolist()]
    use_cfg = cfg_scale != 1.0
    cfg_batch = cfg_batch and use_cfg
    pe = precompute_freqs_cis(positions[:1].contiguous(), transformer.inner_dim, transformer.positional_embedding_theta,
                              transformer.positional_embedding_max_pos, transformer.num_attention_heads)
    plan = _StepPlan(latents, state, 2 if cfg_batch else 1)
    ctx_pos = text_embeddings_pos.to(BF16).contiguous()
    ctx_neg = text_embeddings_neg.to(BF16).contiguous() if use_cfg else None
    ctx_cat = torch.cat([ctx_pos, ctx_neg], 0).contiguous() if cfg_batch else None
    b = latents.shape[0]
    if use_graph and compile_step:
        key = (tuple(latents.shape), bool(cfg_batch), bool(use_cfg), float(cfg_scale), text_embeddings_pos.data_ptr(),
               text_embeddings_neg.data_ptr(), id(transformer), None if state is None else state.clean_latent.data_ptr())
        cache = graph_cache if graph_cache is not None else {}
        ent = cache.get(key)
        if ent is None:
            ent = _StepGraph(latents, plan, transformer, ctx_pos, ctx_neg, ctx_cat, pe, cfg_scale, cfg_batch, use_cfg)
            cache[key] = ent
        return ent.run(latents, sig)
    for i in range(len(sig) - 1):
        s_bf, sn_bf = _bf16_round(sig[i]), _bf16_round(sig[i + 1])
        tp = plan.timestep_plan(s_bf)
        if cfg_batch:
            tok = ops.latent_to_tokens(latents, rep=2)
            v = transformer.forward_tokens(tok, tp, ctx_cat, pe)
            v_pos, v_neg = v[:b], v[b:]
        else:
            tok = ops.latent_to_tokens(latents, rep=1)
            v_pos = transformer.forward_tokens(tok, tp, ctx_pos, pe)
            v_neg = transformer.forward_tokens(tok, tp, ctx_neg, pe) if use_cfg else None
        # x0 uses the bf16 sigma in both paths; Euler: bf16 sigmas if compiled else Python floats.
        # The fused kernel takes one sigma for x0 and the ratio terms; when they differ (eager path)
        # x0 and Euler run as two launches of the same kernel.
        if compile_step or (s_bf == sig[i] and sn_bf == sig[i + 1]):
            latents = ops.cfg_euler_step(v_pos, v_neg, latents, cfg_scale, s_bf, sn_bf, plan.clean, plan.mask_tok_f32)
        else:
            latents = _eager_tail(v_pos, v_neg, latents, cfg_scale, s_bf, sig[i], sig[i + 1], plan)
    return latents


class _StepGraph:
    """One denoise step as a hipGraph.  Per step only {timestep values, sigma, sigma_next} change; they live
    in persistent device buffers refreshed by two tiny H2D copies before each replay.  The first step of the
    first run executes eagerly (it doubles as the allocator warm-up torch requires before capture), then the
    step is captured (host-only) and every later step is a replay."""

    def __init__(self, latents, plan, transformer, ctx_pos, ctx_neg, ctx_cat, pe, cfg_scale, cfg_batch, use_cfg):
        self.plan, self.tr, self.pe = plan, transformer, pe
        self.ctx_pos, self.ctx_neg, self.ctx_cat = ctx_pos, ctx_neg, ctx_cat      # keep alive: the graph holds raw pointers
        self.cfg_scale, self.cfg_batch, self.use_cfg = cfg_scale, cfg_batch, use_cfg
        self.b = latents.shape[0]
        self.lat_buf = latents.clone()
        self.graph = None
        self.out = None

    def _step(self, tp):
        if self.cfg_batch:
            v = self.tr.forward_tokens(ops.latent_to_tokens(self.lat_buf, rep=2), tp, self.ctx_cat, self.pe)
            v_pos, v_neg = v[:self.b], v[self.b:]
        else:
            tok = ops.latent_to_tokens(self.lat_buf, rep=1)
            v_pos = self.tr.forward_tokens(tok, tp, self.ctx_pos, self.pe)
            v_neg = self.tr.forward_tokens(tok, tp, self.ctx_neg, self.pe) if self.use_cfg else None
        return ops.cfg_euler_step(v_pos, v_neg, self.lat_buf, self.cfg_scale, 1.0, 0.0, self.plan.clean,
                                  self.plan.mask_tok_f32, sigmas_dev=self.plan._sig_buf)

    def run(self, latents, sig):
        self.lat_buf.copy_(latents)
        start = 0
        if self.graph is None:
            tp = self.plan.set_step_scalars(_bf16_round(sig[0]), _bf16_round(sig[1]))
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                first = self._step(tp)                    # real step 0, eager
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = self._step(tp)
            self.lat_buf.copy_(first)
            start = 1
        for i in range(start, len(sig) - 1):
            self.plan.set_step_scalars(_bf16_round(sig[i]), _bf16_round(sig[i + 1]))
            self.graph.replay()
            self.lat_buf.copy_(self.out)
        return self.lat_buf.clone()


def _eager_tail(v_pos, v_neg, latents, cfg_scale, s_bf, s, s_next, plan):
    """generate.py:1283-1301: x0 = x - bf16(sigma)*v (rounded to bf16), then fp32 Euler with the
    un-rounded Python-float sigmas."""
    x0 = ops.cfg_euler_step(v_pos, v_neg, latents, cfg_scale, s_bf, 0.0, plan.clean, plan.mask_tok_f32)
    if s_next <= 0:
        return x0
    # Euler only: out = x0 + s_next*(x - x0)/s, expressed through the same kernel with v := (x-x0)
    # is not possible without extra rounding, so use the dedicated form: sigma=s, velocity-free.
    return ops.euler_only(latents, x0, s, s_next)


def denoise_distilled(latents: torch.Tensor, positions: torch.Tensor, text_embeddings: torch.Tensor,
                      transformer: LTXModel, sigmas: Sequence[float], verbose: bool = False,
                      state: Optional[LatentState] = None, audio_latents=None, audio_positions=None,
                      audio_embeddings=None, eval_interval: int = 1, compile_step: bool = False,
                      compile_shapeless: bool = False, fp32_euler: bool = True,
                      ui_phase: str = "denoise") -> Tuple[torch.Tensor, None]:
    """generate.py:564-881, video branch (no CFG)."""
    if audio_latents is not None:
        raise ValueError("audio latents are not supported: the audio branch is out of scope (SURVEY.md §2a #3)")
    if not fp32_euler:
        raise ValueError("fp32_euler=False (bf16 Euler, LTX_FP32_EULER=0) is not implemented")
    sig_t = torch.tensor([float(s) for s in sigmas], dtype=torch.float32)
    out = denoise_dev(latents, positions, text_embeddings, text_embeddings, transformer, sig_t, cfg_scale=1.0,
                      state=state, compile_step=compile_step)
    return out, None

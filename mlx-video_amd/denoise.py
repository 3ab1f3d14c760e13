"""Denoise loops: ``denoise_dev`` (CFG) and ``denoise_distilled`` (mlx_video/generate.py:1060-1327,
564-881), video branch.  The step body is: tokens <- latent transpose, DiT forward(s), then one
fused kernel for CFG + x0 + mask blend + Euler (ltxk_cfg_euler_step).

Sigma handling follows the reference exactly (SURVEY.md §7 "bf16-quantised timesteps"):
timesteps = bf16(sigma)*mask always (generate.py:1084,1237); with ``compile_step`` x0 and Euler
use the bf16-rounded sigma (1160-1174), without it x0 uses bf16(sigma) and Euler the Python
float (1288,1293-1301).  ``fp32_euler=False`` (LTX_FP32_EULER=0) only changes the compiled
distilled step (generate.py:741-748): the Euler update then runs op by op in bf16."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .conditioning import LatentState
from .ltx_model import ContextKV, LTXModel, TimestepPlan, precompute_freqs_cis

BF16 = torch.bfloat16
GRAPH_MAX_STEPS = 256        # rows of the per-step scalar tables a captured step graph reads


def _bf16_round(x: float) -> float:
    return float(torch.tensor(x, dtype=torch.float32).to(BF16).to(torch.float32))


class _StepPlan:
    """Per-call constants: the U distinct values of the (per-frame) denoise mask, the token -> row map,
    and the per-step scalar tables {timestep values, sigma, sigma_next}."""

    def __init__(self, latents: torch.Tensor, state: Optional[LatentState], batch_rep: int, sig: Sequence[float]):
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
        self.U = len(self.mask_vals)
        self.tok2row = inv.to(torch.int32).repeat(batch_rep).contiguous()
        self.mask_tok_f32 = mask_tok.to(torch.float32).contiguous() if state is not None else None
        self.clean = state.clean_latent.to(BF16).contiguous() if state is not None else None
        self.dev = dev
        # timesteps = bf16(sigma) * mask as a bf16 multiply (generate.py:1084,1237): one value per distinct mask entry
        nst = len(sig) - 1
        self.sig_bf = [_bf16_round(s) for s in sig]
        ts = torch.tensor([[self.sig_bf[i] * mv for mv in self.mask_vals] for i in range(nst)], dtype=torch.float32).to(BF16)
        self.ts_host = ts.reshape(nst, self.U)
        self.sig_host = torch.tensor([[self.sig_bf[i], self.sig_bf[i + 1]] for i in range(nst)], dtype=torch.float32).reshape(nst, 2)
        self._ts_dev: Optional[torch.Tensor] = None

    def timestep_plan(self, i: int) -> TimestepPlan:
        if self._ts_dev is None:
            self._ts_dev = self.ts_host.to(self.dev)              # one upload per call; rows are views
        return TimestepPlan(self._ts_dev[i], self.tok2row)


def _denoise(latents: torch.Tensor, positions: torch.Tensor, ctx_pos_in: torch.Tensor, ctx_neg_in: Optional[torch.Tensor],
             transformer: LTXModel, sig: List[float], cfg_scale: float, state: Optional[LatentState], compile_step: bool,
             cfg_batch: bool, use_graph: bool, graph_cache: Optional[dict], cache_context: bool, bf16_euler: bool) -> torch.Tensor:
    if state is not None:
        latents = state.latent
    latents = latents.to(BF16).contiguous()
    use_cfg = cfg_scale != 1.0
    cfg_batch = cfg_batch and use_cfg
    bf16_euler = bf16_euler and compile_step          # the eager body always updates in fp32 (generate.py:835-849)
    if len(sig) < 2:
        return latents
    pe = precompute_freqs_cis(positions[:1].contiguous(), transformer.inner_dim, transformer.positional_embedding_theta,
                              transformer.positional_embedding_max_pos, transformer.num_attention_heads)
    plan = _StepPlan(latents, state, 2 if cfg_batch else 1, sig)
    ctx_pos = ctx_pos_in.to(BF16).contiguous()
    ctx_neg = ctx_neg_in.to(BF16).contiguous() if use_cfg else None
    b = latents.shape[0]
    if use_graph and compile_step and len(sig) - 1 <= GRAPH_MAX_STEPS:
        key = (tuple(latents.shape), bool(cfg_batch), bool(use_cfg), float(cfg_scale), tuple(ctx_pos.shape), id(transformer),
               state is not None, plan.U, bool(bf16_euler), bool(cache_context))
        cache = graph_cache if graph_cache is not None else {}
        ent = cache.get(key)
        if ent is None:
            ent = _StepGraph(latents, plan, transformer, ctx_pos, cfg_scale, cfg_batch, use_cfg, bf16_euler, cache_context)
            cache[key] = ent
        return ent.run(latents, plan, ctx_pos, ctx_neg, pe)
    ctx_cat = torch.cat([ctx_pos, ctx_neg], 0).contiguous() if cfg_batch else None
    kv_pos = kv_neg = kv_cat = None
    if cache_context:
        if cfg_batch:
            kv_cat = transformer.prepare_context(ctx_cat)
        else:
            kv_pos = transformer.prepare_context(ctx_pos)
            kv_neg = transformer.prepare_context(ctx_neg) if use_cfg else None
    for i in range(len(sig) - 1):
        s_bf, sn_bf = plan.sig_bf[i], plan.sig_bf[i + 1]
        tp = plan.timestep_plan(i)
        if cfg_batch:
            tok = ops.latent_to_tokens(latents, rep=2)
            v = transformer.forward_tokens(tok, tp, ctx_cat, pe, kv_cat)
            v_pos, v_neg = v[:b], v[b:]
        else:
            tok = ops.latent_to_tokens(latents, rep=1)
            v_pos = transformer.forward_tokens(tok, tp, ctx_pos, pe, kv_pos)
            v_neg = transformer.forward_tokens(tok, tp, ctx_neg, pe, kv_neg) if use_cfg else None
        # x0 uses the bf16 sigma in both paths; Euler: bf16 sigmas if compiled else Python floats.
        # The fused kernel takes one sigma for x0 and the ratio terms; when they differ (eager path)
        # x0 and Euler run as two launches.
        if compile_step or (s_bf == sig[i] and sn_bf == sig[i + 1]):
            latents = ops.cfg_euler_step(v_pos, v_neg, latents, cfg_scale, s_bf, sn_bf, plan.clean, plan.mask_tok_f32,
                                         bf16_euler=bf16_euler)
        else:
            latents = _eager_tail(v_pos, v_neg, latents, cfg_scale, s_bf, sig[i], sig[i + 1], plan)
    return latents


def denoise_dev(latents: torch.Tensor, positions: torch.Tensor, text_embeddings_pos: torch.Tensor,
                text_embeddings_neg: torch.Tensor, transformer: LTXModel, sigmas: torch.Tensor,
                cfg_scale: float = 4.0, verbose: bool = False, state: Optional[LatentState] = None,
                eval_interval: int = 1, compile_step: bool = False, compile_shapeless: bool = False,
                cfg_batch: bool = False, ui_phase: str = "denoise", use_graph: bool = False,
                graph_cache: Optional[dict] = None, cache_context: bool = False) -> torch.Tensor:
    """generate.py:1060-1327.  latents (B,128,F,H,W) bf16 on the GPU; returns the same shape.
    ``use_graph``: capture the whole step (forward(s) + fused tail, ~1000 launches) once as a hipGraph and
    replay it per step — the analogue of the reference's mx.compile'd step_fn (generate.py:1109-1177);
    requires compile_step semantics (bf16 sigmas) and gives bit-identical results to the eager path.
    ``graph_cache`` (a dict owned by the caller) keeps the captured graph across calls of the same geometry,
    like a compiled function that is traced once: the graph owns persistent copies of every per-call input
    (latents, contexts, clean latent, mask, token->row map, RoPE table) and each call REFRESHES them, so a new
    prompt / mask / conditioning never sees stale data whatever addresses the new tensors happen to get.
    ``cache_context``: compute the text-only part of the forward (caption projection, cross-attention K/V)
    once per call instead of every step — an algorithmic change relative to the reference."""
    sig = [float(s) for s in (sigmas.tolist() if torch.is_tensor(sigmas) else sigmas)]
    return _denoise(latents, positions, text_embeddings_pos, text_embeddings_neg, transformer, sig, cfg_scale, state,
                    compile_step, cfg_batch, use_graph, graph_cache, cache_context, False)


class _StepGraph:
    """One denoise step as a hipGraph.  First node: ltxk_step_scalars, which reads this step's {timestep values,
    sigma, sigma_next} from device tables and advances a device-side step counter — so the schedule runs as bare
    graph replays with no host->device traffic in between; the fused tail updates the latents in place.  The
    first step of the first run executes eagerly (it doubles as the warm-up torch requires before capture:
    allocator, lazy hipFuncSetAttribute calls), then the step is captured and every later step is a replay."""

    def __init__(self, latents, plan: _StepPlan, transformer: LTXModel, ctx_pos, cfg_scale, cfg_batch, use_cfg, bf16_euler,
                 cache_context):
        dev = latents.device
        self.tr = transformer                       # strong reference: id(transformer) in the cache key stays unique
        self.cfg_scale, self.cfg_batch, self.use_cfg, self.bf16_euler = cfg_scale, cfg_batch, use_cfg, bf16_euler
        self.b = latents.shape[0]
        self.lat_buf = torch.empty_like(latents)
        reps = 2 if cfg_batch else 1
        self.ctx_a = torch.empty((reps * ctx_pos.shape[0],) + tuple(ctx_pos.shape[1:]), dtype=BF16, device=dev)   # pos (| neg)
        self.ctx_b = torch.empty_like(ctx_pos) if (use_cfg and not cfg_batch) else None                            # neg
        self.clean = torch.empty_like(plan.clean) if plan.clean is not None else None
        self.mask_tok = torch.empty_like(plan.mask_tok_f32) if plan.mask_tok_f32 is not None else None
        self.tok2row = torch.empty_like(plan.tok2row)
        self.pe: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
        self.ts_all = torch.zeros((GRAPH_MAX_STEPS, plan.U), dtype=BF16, device=dev)
        self.sig_all = torch.zeros((GRAPH_MAX_STEPS, 2), dtype=torch.float32, device=dev)
        self.step = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.ts_buf = torch.zeros((plan.U,), dtype=BF16, device=dev)
        self.sig_buf = torch.zeros((2,), dtype=torch.float32, device=dev)
        self.kv_a: Optional[ContextKV] = None
        self.kv_b: Optional[ContextKV] = None
        self.cache_context = cache_context
        self.graph = None

    def _load(self, latents, plan: _StepPlan, ctx_pos, ctx_neg, pe) -> None:
        """Refresh every per-call input in the buffers the captured kernels read."""
        self.lat_buf.copy_(latents)
        if self.cfg_batch:
            n = ctx_pos.shape[0]
            self.ctx_a[:n].copy_(ctx_pos)
            self.ctx_a[n:].copy_(ctx_neg)
        else:
            self.ctx_a.copy_(ctx_pos)
            if self.ctx_b is not None:
                self.ctx_b.copy_(ctx_neg)
        if self.clean is not None:
            self.clean.copy_(plan.clean)
            self.mask_tok.copy_(plan.mask_tok_f32)
        self.tok2row.copy_(plan.tok2row)
        cos, sin = pe
        if self.pe is None:
            self.pe = (cos.clone(), sin.clone())
        else:
            self.pe[0].copy_(cos)
            self.pe[1].copy_(sin)
        nst = plan.ts_host.shape[0]
        self.ts_all[:nst].copy_(plan.ts_host)
        self.sig_all[:nst].copy_(plan.sig_host)
        self.step.zero_()
        if self.cache_context:
            self.kv_a = self.tr.prepare_context(self.ctx_a, out=self.kv_a)
            if self.ctx_b is not None:
                self.kv_b = self.tr.prepare_context(self.ctx_b, out=self.kv_b)

    def _step(self):
        ops.step_scalars(self.ts_all, self.sig_all, self.step, self.ts_buf, self.sig_buf)
        tp = TimestepPlan(self.ts_buf, self.tok2row)
        if self.cfg_batch:
            v = self.tr.forward_tokens(ops.latent_to_tokens(self.lat_buf, rep=2), tp, self.ctx_a, self.pe, self.kv_a)
            v_pos, v_neg = v[:self.b], v[self.b:]
        else:
            tok = ops.latent_to_tokens(self.lat_buf, rep=1)
            v_pos = self.tr.forward_tokens(tok, tp, self.ctx_a, self.pe, self.kv_a)
            v_neg = self.tr.forward_tokens(tok, tp, self.ctx_b, self.pe, self.kv_b) if self.use_cfg else None
        ops.cfg_euler_step(v_pos, v_neg, self.lat_buf, self.cfg_scale, 1.0, 0.0, self.clean, self.mask_tok, out=self.lat_buf,
                           sigmas_dev=self.sig_buf, bf16_euler=self.bf16_euler)

    def run(self, latents, plan: _StepPlan, ctx_pos, ctx_neg, pe) -> torch.Tensor:
        self._load(latents, plan, ctx_pos, ctx_neg, pe)
        nst = plan.ts_host.shape[0]
        start = 0
        if self.graph is None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._step()                              # real step 0, eager (device step counter 0 -> 1)
            torch.cuda.current_stream().wait_stream(side)
            # thread_local: a process-group watchdog thread (RCCL, N > 1) polling its events must not invalidate the capture
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self._step()
            start = 1
        for _ in range(start, nst):
            self.graph.replay()
        return self.lat_buf.clone()


def _eager_tail(v_pos, v_neg, latents, cfg_scale, s_bf, s, s_next, plan):
    """generate.py:1283-1301: x0 = x - bf16(sigma)*v (rounded to bf16), then fp32 Euler with the
    un-rounded Python-float sigmas."""
    x0 = ops.cfg_euler_step(v_pos, v_neg, latents, cfg_scale, s_bf, 0.0, plan.clean, plan.mask_tok_f32)
    if s_next <= 0:
        return x0
    return ops.euler_only(latents, x0, s, s_next)


def denoise_distilled(latents: torch.Tensor, positions: torch.Tensor, text_embeddings: torch.Tensor,
                      transformer: LTXModel, sigmas: Sequence[float], verbose: bool = False,
                      state: Optional[LatentState] = None, audio_latents=None, audio_positions=None,
                      audio_embeddings=None, eval_interval: int = 1, compile_step: bool = False,
                      compile_shapeless: bool = False, fp32_euler: bool = True,
                      ui_phase: str = "denoise", use_graph: bool = False, graph_cache: Optional[dict] = None,
                      cache_context: bool = False) -> Tuple[torch.Tensor, None]:
    """generate.py:564-881, video branch (no CFG).  ``fp32_euler=False`` selects the bf16 Euler update of the
    compiled step (generate.py:741-748); the un-compiled loop body always updates in fp32 (835-849)."""
    if audio_latents is not None:
        raise ValueError("audio latents are not supported: the audio branch is out of scope (SURVEY.md §2a #3)")
    sig = [float(s) for s in (sigmas.tolist() if torch.is_tensor(sigmas) else sigmas)]
    out = _denoise(latents, positions, text_embeddings, None, transformer, sig, 1.0, state, compile_step, False,
                   use_graph, graph_cache, cache_context, not fp32_euler)
    return out, None

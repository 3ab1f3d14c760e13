"""Host-side mirror of the reference's DiT seam: ``LTXModel.__call__(video: Modality, audio=None)``
(mlx_video/models/ltx/ltx.py:459-506, transformer.py:13-22,247-261,342-347).

Same names, argument meaning and error behaviour as the reference; every numeric op is a
libltxk HIP kernel (``ops``).  Video-only (the audio / cross-modal branches are out of scope,
SURVEY.md §2a #3).

Data layout in HBM (per GPU, bf16 unless noted):
  * weights: one contiguous (out,in) matrix per Linear, with to_q|to_k of the self-attention
    packed into one (8192,4096) panel so q and k come out of one GEMM; 26 GB for L=48;
  * tokens: (B*N, 4096) row-major residual stream, updated in place by the GEMM epilogues;
  * AdaLN: instead of the reference's per-token (B,N,6*4096) tensor (63 MB at N=1280) the
    ``U`` distinct timestep rows are embedded once — (L,U,6,4096) — and every kernel indexes it
    with a (B*N) int32 token->row map.  Values are identical: each row of the reference's
    per-token GEMM depends on that token's timestep only.
  * V is produced transposed, (B, 4096, N_pad64), so attention reads it key-contiguous.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops

BF16 = torch.bfloat16


@dataclass(frozen=True)
class Modality:
    """transformer.py:13-22."""
    latent: torch.Tensor                 # (B,N,128)
    timesteps: torch.Tensor              # (B,N)  sigma*mask, model dtype
    positions: Optional[torch.Tensor]    # (B,3,N,2) float32
    context: torch.Tensor                # (B,S,3840)
    enabled: bool = True
    context_mask: Optional[torch.Tensor] = None
    positional_embeddings: Optional[Tuple[torch.Tensor, torch.Tensor]] = None


@dataclass
class LTXModelConfig:
    """Video half of config.py:93-129 with the constants of generate.py:2866-2881."""
    num_attention_heads: int = 32
    attention_head_dim: int = 128
    in_channels: int = 128
    out_channels: int = 128
    num_layers: int = 48
    cross_attention_dim: int = 4096
    caption_channels: int = 3840
    positional_embedding_theta: float = 10000.0
    positional_embedding_max_pos: Sequence[int] = (20, 2048, 2048)
    use_middle_indices_grid: bool = True
    rope_type: str = "split"
    double_precision_rope: bool = True
    timestep_scale_multiplier: int = 1000
    norm_eps: float = 1e-6

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim


@dataclass
class TimestepPlan:
    """The U distinct timestep values of a forward and the token -> row map."""
    values: torch.Tensor     # (U,) bf16
    tok2row: torch.Tensor    # (B*N,) int32

    @staticmethod
    def from_timesteps(timesteps: torch.Tensor) -> "TimestepPlan":
        vals, inv = torch.unique(timesteps.reshape(-1), sorted=True, return_inverse=True)
        return TimestepPlan(vals.to(BF16).contiguous(), inv.to(torch.int32).contiguous())


def precompute_freqs_cis(positions: torch.Tensor, dim: int, theta: float = 10000.0,
                         max_pos: Sequence[int] = (20, 2048, 2048), num_attention_heads: int = 32,
                         ) -> Tuple[torch.Tensor, torch.Tensor]:
    """rope.py:364-416 -> 419-529 (SPLIT, double_precision path, middle-indices grid).
    positions (B,3,N,2) float32 on device.  Returns cos, sin (B,H,N,dim/H/2) float32.  The
    per-index frequency vector theta^linspace(0,1,n)*pi/2 (682 floats) is a host table; the
    (N x dim/2) trig table is a HIP kernel."""
    if positions.dtype == BF16:      # rope.py:433-445: warn, then compute from the (already rounded) values in float32
        import warnings
        warnings.warn("Position grid has dtype bfloat16, which causes precision loss in RoPE. "
                      "Use float32 for position grids to avoid quality degradation.", UserWarning, stacklevel=2)
    if positions.dtype != torch.float32:
        positions = positions.to(torch.float32)
    b, nd, n, two = positions.shape
    if nd != 3 or two != 2:
        raise ValueError(f"positions must be (B,3,N,2), got {tuple(positions.shape)}")
    n_freq = max(dim // (2 * nd), 1)
    lin = torch.linspace(0.0, 1.0, n_freq, dtype=torch.float32)
    freq = (torch.pow(torch.tensor(theta, dtype=torch.float32), lin) * (math.pi / 2)).to(positions.device)
    cos_l, sin_l = [], []
    for i in range(b):
        c, s = ops.rope_table(positions[i].contiguous(), freq, num_attention_heads, dim, max_pos)
        cos_l.append(c)
        sin_l.append(s)
    return torch.stack(cos_l), torch.stack(sin_l)


class _Block:
    __slots__ = ("wqkv", "bqkv", "wqn", "wkn", "wqkn", "wo", "bo", "wq2", "bq2", "wkv2", "bkv2",
                 "wqn2", "wkn2", "wo2", "bo2", "w1", "b1", "w2", "b2")


class ContextKV:
    """Step-invariant text-side tensors of ONE context: the caption projection (ltx.py:77-89) and every
    block's cross-attention K (q/k-normed) and V^T (attention.py:123-131 with context=text).  Built by
    ``LTXModel.prepare_context`` and owned by the caller: ``prepare_context(context, out=kv)`` recomputes
    it IN PLACE for a new prompt, so a captured step graph that reads these buffers stays valid."""

    def __init__(self, shape: Tuple[int, int, int]):
        self.shape = tuple(shape)           # (B,S,caption_channels) it was built for
        self.ctx: Optional[torch.Tensor] = None
        self.kv: List[tuple] = []


class LTXModel:
    """Velocity model.  ``model(video=Modality(...)) -> (velocity (B,N,128), None)``."""

    def __init__(self, config: LTXModelConfig, weights: Dict[str, torch.Tensor], fuse: int = 15):
        self.config = config
        self.inner_dim = config.inner_dim
        self.num_attention_heads = config.num_attention_heads
        self.positional_embedding_theta = config.positional_embedding_theta
        self.positional_embedding_max_pos = list(config.positional_embedding_max_pos)
        self.use_middle_indices_grid = config.use_middle_indices_grid
        self.rope_type = config.rope_type
        self.timestep_scale_multiplier = config.timestep_scale_multiplier
        # A/B switches of the launch structure (scripts/ab_step.py; all three forms give the same roundings):
        #   1: q|k|v and text k|v as ONE GEMM launch with a split output;  2: row sums of squares carried by the GEMM
        #   epilogues into the norm kernels;  4: q_norm + RoPE of q applied inside the attention kernel (needs 2);
        #   8: self-attention q|k and v as TWO launches after all - q|k (N=8192) then fills exactly one round of the
        #   320x256-tile kernel, 152 + 74 us against 239 us for the one q|k|v launch on 160x256 tiles at M=2560
        #   (profiles/r02_gemm_big_tile_ab.log); the text k|v pair stays one launch (it takes the big tile as it is)
        self.fuse = int(fuse)
        # every token through the token->row map even when all tokens share one timestep row (A/B runs only)
        self.tok2row_always = False
        # ops.flash_attn(tail_split=): False makes a forward's bits independent of the batch it runs in (B=1 per CFG-pair
        # rank == row b of the B=2 cfg_batch forward), at ~5 % of attention time at N=1280 (attention.hip)
        self.attn_tail_split = True
        self._pack(weights)
        # the split-K scratch of ops.gemm (small-M launches) must exist before anyone captures a forward into a hipGraph: allocated
        # inside a capture it would come from that graph's private pool
        ops._gemm_workspace(self.tables.device)

    # ------------------------------------------------------------------ weights
    def _pack(self, W: Dict[str, torch.Tensor]) -> None:
        """Builds the packed panels from ``W`` WITHOUT modifying it (a second model - e.g. the LoRA-merged
        stage-2 transformer, generate.py:3229-3237 - can be constructed from the same dict)."""
        cfg = self.config
        missing = [k for k in self.expected_keys(cfg) if k not in W]
        if missing:   # strict load (ltx.py:874-881)
            raise ValueError(f"Missing {len(missing)} parameters in checkpoint, e.g. {missing[:4]}")

        def g(k):
            t = W[k]
            if t.dtype != BF16 or not t.is_cuda:
                raise TypeError(f"weight {k}: expected a bf16 device tensor")
            return t.contiguous()

        self.patchify_w, self.patchify_b = g("patchify_proj.weight"), g("patchify_proj.bias")
        p = "adaln_single.emb.timestep_embedder"
        self.t1_w, self.t1_b = g(f"{p}.linear1.weight"), g(f"{p}.linear1.bias")
        self.t2_w, self.t2_b = g(f"{p}.linear2.weight"), g(f"{p}.linear2.bias")
        self.ada_w, self.ada_b = g("adaln_single.linear.weight"), g("adaln_single.linear.bias")
        self.c1_w, self.c1_b = g("caption_projection.linear1.weight"), g("caption_projection.linear1.bias")
        self.c2_w, self.c2_b = g("caption_projection.linear2.weight"), g("caption_projection.linear2.bias")
        self.head_table = g("scale_shift_table").reshape(1, 2, -1)
        self.out_w, self.out_b = g("proj_out.weight"), g("proj_out.bias")
        self.blocks: List[_Block] = []
        tables = []
        for i in range(cfg.num_layers):
            pre = f"transformer_blocks.{i}"
            b = _Block()
            # one (3D,D) panel for to_q|to_k|to_v: a single GEMM launch writes q|k row-major and V^T (split output)
            b.wqkv = torch.cat([g(f"{pre}.attn1.to_{n}.weight") for n in "qkv"], 0)
            b.bqkv = torch.cat([g(f"{pre}.attn1.to_{n}.bias") for n in "qkv"], 0)
            b.wqn, b.wkn = g(f"{pre}.attn1.q_norm.weight"), g(f"{pre}.attn1.k_norm.weight")
            b.wqkn = torch.cat([b.wqn, b.wkn], 0)
            b.wo, b.bo = g(f"{pre}.attn1.to_out.weight"), g(f"{pre}.attn1.to_out.bias")
            b.wq2, b.bq2 = g(f"{pre}.attn2.to_q.weight"), g(f"{pre}.attn2.to_q.bias")
            b.wkv2 = torch.cat([g(f"{pre}.attn2.to_k.weight"), g(f"{pre}.attn2.to_v.weight")], 0)     # text k | V^T, one launch
            b.bkv2 = torch.cat([g(f"{pre}.attn2.to_k.bias"), g(f"{pre}.attn2.to_v.bias")], 0)
            b.wqn2, b.wkn2 = g(f"{pre}.attn2.q_norm.weight"), g(f"{pre}.attn2.k_norm.weight")
            b.wo2, b.bo2 = g(f"{pre}.attn2.to_out.weight"), g(f"{pre}.attn2.to_out.bias")
            b.w1, b.b1 = g(f"{pre}.ff.proj_in.weight"), g(f"{pre}.ff.proj_in.bias")
            b.w2, b.b2 = g(f"{pre}.ff.proj_out.weight"), g(f"{pre}.ff.proj_out.bias")
            tables.append(g(f"{pre}.scale_shift_table"))
            self.blocks.append(b)
        self.tables = torch.stack(tables, 0).contiguous()      # (L,6,D)

    def weight_views(self) -> Dict[str, torch.Tensor]:
        """Checkpoint key -> the (out,in) matrix as it lives inside THIS model: the packed q|k|v and text k|v panels are
        returned as their row ranges.  Writing through these views changes the model (lora.apply_lora_to_weights(...,
        in_place=True): the stage-2 transformer of the distilled pipeline without a second 21-GB replica, generate.py:3229-3283)."""
        D = self.inner_dim
        v = {"patchify_proj.weight": self.patchify_w, "adaln_single.emb.timestep_embedder.linear1.weight": self.t1_w,
             "adaln_single.emb.timestep_embedder.linear2.weight": self.t2_w, "adaln_single.linear.weight": self.ada_w,
             "caption_projection.linear1.weight": self.c1_w, "caption_projection.linear2.weight": self.c2_w, "proj_out.weight": self.out_w}
        for i, b in enumerate(self.blocks):
            pre = f"transformer_blocks.{i}"
            for j, n in enumerate("qkv"):
                v[f"{pre}.attn1.to_{n}.weight"] = b.wqkv[j * D:(j + 1) * D]
            v[f"{pre}.attn1.to_out.weight"], v[f"{pre}.attn2.to_q.weight"] = b.wo, b.wq2
            v[f"{pre}.attn2.to_k.weight"], v[f"{pre}.attn2.to_v.weight"] = b.wkv2[:D], b.wkv2[D:]
            v[f"{pre}.attn2.to_out.weight"], v[f"{pre}.ff.proj_in.weight"], v[f"{pre}.ff.proj_out.weight"] = b.wo2, b.w1, b.w2
        return v

    @staticmethod
    def expected_keys(cfg: LTXModelConfig) -> List[str]:
        keys = []
        for n in ("patchify_proj", "adaln_single.emb.timestep_embedder.linear1",
                  "adaln_single.emb.timestep_embedder.linear2", "adaln_single.linear",
                  "caption_projection.linear1", "caption_projection.linear2", "proj_out"):
            keys += [f"{n}.weight", f"{n}.bias"]
        keys.append("scale_shift_table")
        for i in range(cfg.num_layers):
            pre = f"transformer_blocks.{i}"
            for a in ("attn1", "attn2"):
                for nm in ("to_q", "to_k", "to_v", "to_out"):
                    keys += [f"{pre}.{a}.{nm}.weight", f"{pre}.{a}.{nm}.bias"]
                keys += [f"{pre}.{a}.q_norm.weight", f"{pre}.{a}.k_norm.weight"]
            keys += [f"{pre}.ff.proj_in.weight", f"{pre}.ff.proj_in.bias",
                     f"{pre}.ff.proj_out.weight", f"{pre}.ff.proj_out.bias", f"{pre}.scale_shift_table"]
        return keys

    @staticmethod
    def sanitize(weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Checkpoint key map of ltx.py:508-533 (PyTorch LTX-2 names -> module names)."""
        out = {}
        for key, value in weights.items():
            if (not key.startswith("model.diffusion_model.") or "audio_embeddings_connector" in key
                    or "video_embeddings_connector" in key):
                continue
            k = key.replace("model.diffusion_model.", "")
            k = k.replace(".to_out.0.", ".to_out.")
            k = k.replace(".ff.net.0.proj.", ".ff.proj_in.").replace(".ff.net.2.", ".ff.proj_out.")
            k = k.replace(".audio_ff.net.0.proj.", ".audio_ff.proj_in.").replace(".audio_ff.net.2.", ".audio_ff.proj_out.")
            k = k.replace(".linear_1.", ".linear1.").replace(".linear_2.", ".linear2.")
            out[k] = value
        return out

    @staticmethod
    def random_weights(config: LTXModelConfig, device, seed: int = 1234) -> Dict[str, torch.Tensor]:
        """Random weights of the exact architecture, generated on the device (synthetic bench;
        SURVEY.md §8d): Linear N(0,0.02^2), biases 0.01*N(0,1), tables N(0,0.02^2),
        q/k-norm weights 1+0.1*N(0,1).  Module-name keys (what ``sanitize`` produces)."""
        g = torch.Generator(device=device).manual_seed(seed)
        D, FF = config.inner_dim, config.inner_dim * 4
        W: Dict[str, torch.Tensor] = {}

        def rn(*shape, std=1.0, mean=0.0):
            return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * std + mean).to(BF16)

        def lin(name, o, i):
            W[f"{name}.weight"] = rn(o, i, std=0.02)
            W[f"{name}.bias"] = rn(o, std=0.01)

        lin("patchify_proj", D, config.in_channels)
        lin("adaln_single.emb.timestep_embedder.linear1", D, 256)
        lin("adaln_single.emb.timestep_embedder.linear2", D, D)
        lin("adaln_single.linear", 6 * D, D)
        lin("caption_projection.linear1", D, config.caption_channels)
        lin("caption_projection.linear2", D, D)
        W["scale_shift_table"] = rn(2, D, std=0.02)
        lin("proj_out", config.out_channels, D)
        for i in range(config.num_layers):
            pre = f"transformer_blocks.{i}"
            for a in ("attn1", "attn2"):
                for nm in ("to_q", "to_k", "to_v", "to_out"):
                    lin(f"{pre}.{a}.{nm}", D, D)
                W[f"{pre}.{a}.q_norm.weight"] = rn(D, std=0.1, mean=1.0)
                W[f"{pre}.{a}.k_norm.weight"] = rn(D, std=0.1, mean=1.0)
            lin(f"{pre}.ff.proj_in", FF, D)
            lin(f"{pre}.ff.proj_out", D, FF)
            W[f"{pre}.scale_shift_table"] = rn(6, D, std=0.02)
        return W

    @classmethod
    def random_init(cls, config: LTXModelConfig, device, seed: int = 1234, fuse: int = 15) -> "LTXModel":
        return cls(config, cls.random_weights(config, device, seed), fuse=fuse)

    # ------------------------------------------------------------------ forward
    def _prepare_context(self, context: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ltx.py:77-89: caption_projection, (B,S,3840) -> (B*S,D)."""
        b, s, c = context.shape
        h = ops.gemm(context.reshape(b * s, c), self.c1_w, self.c1_b, epilogue=ops.EPI_BIAS_GELU)
        return ops.gemm(h, self.c2_w, self.c2_b, out=out)

    def _context_kv(self, blk: _Block, ctx: torch.Tensor, b: int, s: int, sp: int, out: Optional[tuple] = None):
        D, H, eps = self.inner_dim, self.num_attention_heads, self.config.norm_eps
        if out is None:
            k2 = torch.empty((b * s, D), dtype=BF16, device=ctx.device)
            vt2 = torch.zeros((b, D, sp), dtype=BF16, device=ctx.device) if sp != s else \
                torch.empty((b, D, sp), dtype=BF16, device=ctx.device)
            ss = torch.empty((b * s, D // 64), dtype=torch.float32, device=ctx.device)
        else:
            k2, vt2, ss = out
        # k (row-major, with its per-row sums of squares) and V^T from one launch over the packed k|v panel
        st = ss if self.fuse & 2 else None
        if self.fuse & 1:
            ops.gemm(ctx, blk.wkv2, blk.bkv2, out=k2, out2=vt2, n_split=D, out_tokens_per_batch=s, sumsq=st)
        else:
            ops.gemm(ctx, blk.wkv2[:D], blk.bkv2[:D], out=k2, sumsq=st)
            ops.gemm(ctx, blk.wkv2[D:], blk.bkv2[D:], out=vt2, out_tokens_per_batch=s)
        ops.qknorm_rope(k2, 1, D, blk.wkn2, None, None, s, H, eps, sumsq=st)
        return k2, vt2, ss

    def prepare_context(self, context: torch.Tensor, out: Optional[ContextKV] = None) -> ContextKV:
        """Everything of the forward that depends on the text context only (3.37 TFLOP at S=1024, SURVEY.md
        §8d).  The reference recomputes it in every forward; a denoise loop may hoist it (an algorithmic
        change, always reported separately).  ``out``: refresh an existing object in place."""
        context = context.to(BF16).contiguous()
        b, s, _ = context.shape
        sp = (s + 63) // 64 * 64
        kv = out if out is not None else ContextKV(context.shape)
        if kv.shape != tuple(context.shape):
            raise ValueError(f"ContextKV was built for context {kv.shape}, got {tuple(context.shape)}")
        kv.ctx = self._prepare_context(context, kv.ctx)
        new = [self._context_kv(blk, kv.ctx, b, s, sp, kv.kv[i] if kv.kv else None) for i, blk in enumerate(self.blocks)]
        kv.kv = new
        return kv

    def forward_tokens(self, latent: torch.Tensor, plan: TimestepPlan, context: torch.Tensor,
                       pe: Tuple[torch.Tensor, torch.Tensor], ctx_kv: Optional[ContextKV] = None,
                       hidden: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        """latent (B,N,128) bf16; context (B,S,3840) bf16; pe = (cos,sin) each (1|B,H,N,64) fp32
        (one table shared by every batch row, as in cfg_batch where it is a broadcast,
        generate.py:1196-1202).  ``ctx_kv``: a ContextKV of THIS context (prepare_context); without it the
        caption projection and the text K/V are recomputed here, as the reference does every forward.
        ``hidden``: a list that receives a copy of the residual stream (B,N,D) after every block (the reference's
        debug taps, transformer.py; used by the per-layer parity tests)."""
        cfg = self.config
        D, H, eps = self.inner_dim, self.num_attention_heads, cfg.norm_eps
        B, N, C = latent.shape
        M = B * N
        S = context.shape[1]
        dev = latent.device
        cos, sin = pe
        if cos.dim() == 4:
            cos, sin = cos[0], sin[0]
        cos, sin = cos.contiguous(), sin.contiguous()
        U = plan.values.numel()
        # one timestep row for every token (an unconditioned CFG pair): no per-token row gather - the kernels then skip a
        # dependent load in front of every modulation / gate read
        tok2row = plan.tok2row if (U > 1 or self.tok2row_always) else None
        scale = 1.0 / math.sqrt(cfg.attention_head_dim)

        # --- prepare (ltx.py:129-158) ---
        # Row statistics travel with the residual stream: every GEMM that writes x also emits the per-row sums of
        # squares of what it stored (64-column partials), so the rms_norm that follows does not re-reduce the row.
        P = D // 64
        xss = torch.empty((M, P), dtype=torch.float32, device=dev)
        x = ops.gemm(latent.reshape(M, C), self.patchify_w, self.patchify_b, sumsq=xss)
        tproj = ops.timestep_embed(plan.values, 256, float(cfg.timestep_scale_multiplier))
        h = ops.gemm(tproj, self.t1_w, self.t1_b, epilogue=ops.EPI_BIAS_SILU)
        emb = ops.gemm(h, self.t2_w, self.t2_b)                           # embedded_timestep (U,D)
        ada = ops.gemm(ops.silu(emb), self.ada_w, self.ada_b)             # (U,6D)
        # (L,U,6,D): shift, 1+scale, gate, shift, 1+scale, gate - the (1 + scale) factor is the same for every token of a row
        # (without the carried row statistics the self-reducing norm kernel takes the raw scale and adds 1 itself)
        mods = ops.ada_combine(self.tables, ada, cfg.num_layers, U, 6, D, one_plus_mask=0b010010 if self.fuse & 2 else 0)
        head = ops.ada_combine(self.head_table, emb.repeat(1, 2), 1, U, 2, D)[0]   # (U,2,D): shift, scale

        if ctx_kv is not None:
            if ctx_kv.shape != tuple(context.shape):
                raise ValueError(f"ctx_kv was built for context {ctx_kv.shape}, got {tuple(context.shape)}")
            ctx = ctx_kv.ctx
        else:
            ctx = self._prepare_context(context)

        np64 = (N + 63) // 64 * 64
        sp64 = (S + 63) // 64 * 64
        vt = torch.zeros((B, D, np64), dtype=BF16, device=dev) if np64 != N else \
            torch.empty((B, D, np64), dtype=BF16, device=dev)
        qk = torch.empty((M, 2 * D), dtype=BF16, device=dev)
        qkss = torch.empty((M, 2 * P), dtype=torch.float32, device=dev)
        nx = torch.empty((M, D), dtype=BF16, device=dev)
        att = torch.empty((M, D), dtype=BF16, device=dev)
        q2 = torch.empty((M, D), dtype=BF16, device=dev)
        q2ss = torch.empty((M, P), dtype=torch.float32, device=dev)
        hff = torch.empty((M, 4 * D), dtype=BF16, device=dev)
        ms = 6 * D
        kv_buf = None
        if ctx_kv is None:                  # text K / V^T of the current block, recomputed every forward (one buffer set)
            # (computing block li+1's text K / V^T on a side stream beside block li - they do not depend on the token
            # stream - measured 1.7 % SLOWER eagerly and unchanged in a captured graph: the main kernels leave no CU idle
            # long enough for a 156-KiB-LDS GEMM workgroup; profiles/r02_launch_structure_ab.log.  k's q_norm + RoPE on a side
            # stream beside the V^T GEMM - a memory-bound kernel that needs no LDS - measured the same way: 1.269 against
            # 1.264 ms per block.  Forked graph branches cost more than they overlap.  Nor does one GRID for two independent GEMMs
            # (q|k on 320x256 tiles with v's 160x256 tiles back-filling behind them; text k|v with q2): 1269.6 us per block
            # either way, big tiles first or last - the tail of one launch is not where the time goes.)
            kv_buf = (torch.empty((B * S, D), dtype=BF16, device=dev),
                      torch.zeros((B, D, sp64), dtype=BF16, device=dev) if sp64 != S else torch.empty((B, D, sp64), dtype=BF16, device=dev),
                      torch.empty((B * S, D // 64), dtype=torch.float32, device=dev))

        fq, fs, fp = self.fuse & 1, self.fuse & 2, (self.fuse & 6) == 6
        ts_ = self.attn_tail_split
        s_x, s_qk, s_q2 = (xss, qkss, q2ss) if fs else (None, None, None)
        for li, blk in enumerate(self.blocks):
            mod = mods[li]                                           # (U,6,D): shift, 1+scale, gate x2
            # self-attention (transformer.py:248-254).  q|k|v from one launch: q,k row-major with their row statistics,
            # V^T transposed.  k is normalised + rotated in place; q stays RAW in HBM - the attention kernel normalises
            # and rotates its Q fragments in registers (attention.py:129-136).
            ops.rmsnorm_modulate(x, eps, mod[:, 1], mod[:, 0], ms, tok2row, out=nx, sumsq=s_x, scale_is_one_plus=bool(fs))
            # (fuse bit 8: q|k on the 320x256 tile with v as its own launch - a gain only where M is a whole number of 320-row
            # tiles (M=1280: 35.4 against 35.6 ms per forward, M=2560: 60.3 against 61.0); every other row count is 1-4 % faster
            # with q|k|v as ONE launch (M=1296: 39.4 against 41.2 ms, 3328: 87.6 / 89.2, 5184: 141.5 / 143.8, 6656: 161.5 / 164.7;
            # scripts/exp_qkv_one_launch.py), and at small M every launch is a weight stream with ~5 us of fixed cost)
            if fq and (not (self.fuse & 8) or M <= ops.SPLITK_MAX_M or M % 320 != 0):
                ops.gemm(nx, blk.wqkv, blk.bqkv, out=qk, out2=vt, n_split=2 * D, out_tokens_per_batch=N, sumsq=s_qk)
            else:
                ops.gemm(nx, blk.wqkv[:2 * D], blk.bqkv[:2 * D], out=qk, sumsq=s_qk)
                ops.gemm(nx, blk.wqkv[2 * D:], blk.bqkv[2 * D:], out=vt, out_tokens_per_batch=N)
            if fp:
                ops.qknorm_rope(qk[:, D:], 1, D, blk.wkn, cos, sin, N, H, eps, sumsq=qkss[:, P:])
                ops.flash_attn(qk[:, :D], qk[:, D:], vt, att, B, H, N, N, scale, q_sumsq=qkss, q_norm_weight=blk.wqn,
                               cos=cos, sin=sin, eps=eps, tail_split=ts_)
            else:
                ops.qknorm_rope(qk, 2, D, blk.wqkn, cos, sin, N, H, eps, sumsq=s_qk)
                ops.flash_attn(qk[:, :D], qk[:, D:], vt, att, B, H, N, N, scale, tail_split=ts_)
            ops.gemm(att, blk.wo, blk.bo, epilogue=ops.EPI_BIAS_GATE_RES, out=x, resid=x,
                     gate=mod[:, 2], gate_row=tok2row, gate_stride=ms, sumsq=s_x)
            # text cross-attention (transformer.py:257-261)
            ops.rmsnorm_modulate(x, eps, out=nx, sumsq=s_x)
            ops.gemm(nx, blk.wq2, blk.bq2, out=q2, sumsq=s_q2)
            kv = ctx_kv.kv[li] if ctx_kv is not None else self._context_kv(blk, ctx, B, S, sp64, kv_buf)
            if fp:
                ops.flash_attn(q2, kv[0], kv[1], att, B, H, N, S, scale, q_sumsq=q2ss, q_norm_weight=blk.wqn2, eps=eps, tail_split=ts_)
            else:
                ops.qknorm_rope(q2, 1, D, blk.wqn2, None, None, N, H, eps, sumsq=s_q2)
                ops.flash_attn(q2, kv[0], kv[1], att, B, H, N, S, scale, tail_split=ts_)
            ops.gemm(att, blk.wo2, blk.bo2, epilogue=ops.EPI_BIAS_RES, out=x, resid=x, sumsq=s_x)
            # feed-forward (transformer.py:343-347)
            ops.rmsnorm_modulate(x, eps, mod[:, 4], mod[:, 3], ms, tok2row, out=nx, sumsq=s_x, scale_is_one_plus=bool(fs))
            ops.gemm(nx, blk.w1, blk.b1, epilogue=ops.EPI_BIAS_GELU, out=hff)
            ops.gemm(hff, blk.w2, blk.b2, epilogue=ops.EPI_BIAS_GATE_RES, out=x, resid=x,
                     gate=mod[:, 5], gate_row=tok2row, gate_stride=ms, sumsq=s_x)
            if hidden is not None:
                hidden.append(x.reshape(B, N, D).clone())

        # --- output head (ltx.py:432-457) ---
        ops.layernorm_modulate(x, eps, head[:, 1], head[:, 0], 2 * D, tok2row, out=nx)
        v = ops.gemm(nx, self.out_w, self.out_b)
        return v.reshape(B, N, cfg.out_channels)

    def __call__(self, video: Optional[Modality] = None, audio: Optional[Modality] = None):
        if audio is not None:
            raise ValueError("Audio is not enabled for this model")      # ltx.py:468-469
        if video is None:
            return None, None
        lat = video.latent
        if lat.dim() != 3 or lat.shape[-1] != self.config.in_channels:
            raise ValueError(f"latent must be (B,N,{self.config.in_channels}), got {tuple(lat.shape)}")
        if video.context_mask is not None:
            raise ValueError("context_mask is not supported on this path (the reference passes None, generate.py:800)")
        pe = video.positional_embeddings
        if pe is None:
            pe = precompute_freqs_cis(video.positions, self.inner_dim, self.positional_embedding_theta,
                                      self.positional_embedding_max_pos, self.num_attention_heads)
        plan = TimestepPlan.from_timesteps(video.timesteps.to(BF16))
        v = self.forward_tokens(lat.to(BF16).contiguous(), plan, video.context.to(BF16).contiguous(), pe)
        return v, None


class X0Model:
    """ltx.py:888-906: velocity -> denoised wrapper, x0 = latent - timesteps*velocity with the per-token
    timesteps as sigma (utils.py:404-440, fp32 then cast).  Only used by the reference's legacy
    ltx_pipelines/utils helpers; implemented with the step kernel per distinct sigma."""

    def __init__(self, velocity_model: LTXModel):
        self.velocity_model = velocity_model

    def __call__(self, video: Optional[Modality] = None, audio: Optional[Modality] = None):
        v, _ = self.velocity_model(video, audio)
        if v is None:
            return None, None
        lat = video.latent.to(BF16)
        B, N, C = lat.shape
        out = torch.empty_like(lat)
        ts = video.timesteps.to(BF16)
        for val in torch.unique(ts).tolist():
            # ltxk_cfg_euler_step with sigma_next = 0 returns x0 = bf16(x - sigma*v); it works on channels-first
            # latents, so view the (B,N,C) tokens as a (B*N, C, 1) "latent" with one position per token row
            sel = (ts == val)
            idx = sel.reshape(-1).nonzero().squeeze(1)
            xs = lat.reshape(B * N, C)[idx].contiguous().reshape(-1, C, 1)
            vs = v.reshape(B * N, C)[idx].contiguous().reshape(-1, 1, C)
            if float(val) == 0.0:
                out.reshape(B * N, C)[idx] = xs.reshape(-1, C)
                continue
            x0 = ops.cfg_euler_step(vs, None, xs, 1.0, float(val), 0.0)
            out.reshape(B * N, C)[idx] = x0.reshape(-1, C)
        return out, None

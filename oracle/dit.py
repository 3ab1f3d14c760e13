"""CPU restatement (torch, fp32 or fp64) of the LTX-2 DiT velocity forward and the
denoise-step algebra.  TEST INFRASTRUCTURE — see oracle/__init__.py.

Two precision policies:
  * ``Prec(emulate_bf16=False)``: pure fp32/fp64 math ("what the formulas say").
  * ``Prec(emulate_bf16=True)``: every op output is rounded to bf16 where the
    reference, running with bf16 arrays on MLX, materialises a bf16 array
    (Linear = fp32 accumulate + one rounding, fast rms_norm / sdpa = fp32 inside +
    one rounding, elementwise ops round per op).  This is the policy the HIP
    kernels are written to reproduce.

Weight keys follow the reference's sanitised names (mlx_video/models/ltx/ltx.py:508-533).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor


@dataclass(frozen=True)
class Prec:
    emulate_bf16: bool = False
    dtype: torch.dtype = torch.float32
    # sdpa rounding policy.  False: the softmax weights P stay fp32 until after P.V ("fp32 inside, one rounding", what
    # MLX documents for mx.fast.scaled_dot_product_attention).  True ("flash"): P is rounded to bf16 before P.V, as every
    # MFMA flash-attention kernel must do to feed the matrix cores - see sdpa() for the exact rounding points.
    flash_sdpa: bool = False

    def r(self, x: Tensor) -> Tensor:
        """Round to bf16 storage (if emulating) and return in the compute dtype."""
        if self.emulate_bf16:
            return x.to(torch.bfloat16).to(self.dtype)
        return x.to(self.dtype)


F32 = Prec(False, torch.float32)
F64 = Prec(False, torch.float64)
BF16 = Prec(True, torch.float32)
BF16_FLASH = Prec(True, torch.float32, flash_sdpa=True)


@dataclass(frozen=True)
class DiTConfig:
    """Constants of the video-only LTX-2 transformer (generate.py:2866-2881, config.py:93-129)."""
    num_layers: int = 48
    heads: int = 32
    d_head: int = 128
    in_channels: int = 128
    out_channels: int = 128
    caption_channels: int = 3840
    ff_mult: int = 4
    theta: float = 10000.0
    max_pos: Tuple[int, int, int] = (20, 2048, 2048)
    timestep_scale_multiplier: float = 1000.0
    norm_eps: float = 1e-6

    @property
    def dim(self) -> int:
        return self.heads * self.d_head


# --------------------------------------------------------------------------------------
# index / position math (bit-exact contracts)
# --------------------------------------------------------------------------------------

def create_position_grid(batch_size: int, num_frames: int, height: int, width: int,
                         temporal_scale: int = 8, spatial_scale: int = 32,
                         fps: float = 24.0, causal_fix: bool = True) -> np.ndarray:
    """generate.py:470-525.  Returns (B,3,N,2) float32 [start,end) pixel coords, token
    order n=(f*H+h)*W+w."""
    t, h, w = np.meshgrid(np.arange(num_frames), np.arange(height), np.arange(width), indexing="ij")
    starts = np.stack([t, h, w], axis=0)                      # (3,F,H,W)
    ends = starts + 1
    coords = np.stack([starts, ends], axis=-1).reshape(3, num_frames * height * width, 2)
    coords = np.tile(coords[None], (batch_size, 1, 1, 1))
    scale = np.array([temporal_scale, spatial_scale, spatial_scale]).reshape(1, 3, 1, 1)
    pix = (coords * scale).astype(np.float32)
    if causal_fix:
        pix[:, 0] = np.clip(pix[:, 0] + 1 - temporal_scale, a_min=0, a_max=None)
    pix[:, 0] = pix[:, 0] / fps
    return pix.astype(np.float32)


def latent_to_tokens(latents: Tensor) -> Tensor:
    """generate.py:792,1236: (B,C,F,H,W) -> (B,N,C)."""
    b, c = latents.shape[:2]
    return latents.reshape(b, c, -1).permute(0, 2, 1).contiguous()


def tokens_to_latent(tokens: Tensor, shape: Sequence[int]) -> Tensor:
    """generate.py:822,1287: (B,N,C) -> (B,C,F,H,W)."""
    b, c, f, h, w = shape
    return tokens.permute(0, 2, 1).reshape(b, c, f, h, w).contiguous()


# --------------------------------------------------------------------------------------
# RoPE (rope.py:419-529 table, rope.py:109-172 apply)
# --------------------------------------------------------------------------------------

def precompute_freqs_cis(positions: Tensor, dim: int, theta: float = 10000.0,
                         max_pos: Sequence[int] = (20, 2048, 2048), heads: int = 32,
                         ) -> Tuple[Tensor, Tensor]:
    """SPLIT rope, double_precision_rope=True path, use_middle_indices_grid=True
    (rope.py:419-529; the "double precision" variant is float32, rope.py:428-431).
    positions (B,3,N,2) float32 -> cos,sin (B,H,N,dim/H/2) float32."""
    pos = positions.to(torch.float32)
    n_dims = pos.shape[1]
    n_elem = 2 * n_dims
    num_indices = max(dim // n_elem, 1)
    # mx.linspace(0,1,num) in fp32, theta**lin * pi/2 in fp32 (rope.py:449-450)
    lin = torch.linspace(0.0, 1.0, num_indices, dtype=torch.float32)
    freq_idx = torch.pow(torch.tensor(theta, dtype=torch.float32), lin) * (math.pi / 2)
    mid = (pos[..., 0] + pos[..., 1]) / 2.0                               # (B,3,N)
    frac = torch.stack([mid[:, i, :] / max_pos[i] for i in range(n_dims)], dim=-1)  # (B,N,3)
    scaled = frac * 2 - 1
    freqs = scaled.unsqueeze(-1) * freq_idx.reshape(1, 1, 1, -1)           # (B,N,3,idx)
    freqs = freqs.transpose(-1, -2).reshape(freqs.shape[0], freqs.shape[1], -1)  # idx-major, dim-minor
    cos, sin = torch.cos(freqs), torch.sin(freqs)
    pad = dim // 2 - cos.shape[-1]
    if pad > 0:                                                            # FRONT pad (rope.py:504-509)
        cos = torch.cat([torch.ones(*cos.shape[:-1], pad), cos], dim=-1)
        sin = torch.cat([torch.zeros(*sin.shape[:-1], pad), sin], dim=-1)
    b, t = cos.shape[:2]
    cos = cos.reshape(b, t, heads, -1).transpose(1, 2).contiguous()
    sin = sin.reshape(b, t, heads, -1).transpose(1, 2).contiguous()
    return cos, sin


def apply_split_rotary_emb(x: Tensor, cos: Tensor, sin: Tensor, p: Prec) -> Tensor:
    """rope.py:109-172.  x (B,T,H*D); cos/sin (B,H,T,D/2).  fp32 math, one rounding."""
    b, h, t, half = cos.shape
    xh = x.reshape(b, t, h, 2, half).permute(0, 2, 1, 3, 4).to(p.dtype)   # (B,H,T,2,D/2)
    c, s = cos.to(p.dtype), sin.to(p.dtype)
    x1, x2 = xh[..., 0, :], xh[..., 1, :]
    o1 = x1 * c - s * x2
    o2 = x2 * c + s * x1
    out = torch.stack([o1, o2], dim=-2).reshape(b, h, t, 2 * half)
    return p.r(out.permute(0, 2, 1, 3).reshape(b, t, h * 2 * half))


# --------------------------------------------------------------------------------------
# small ops
# --------------------------------------------------------------------------------------

def linear(x: Tensor, w: Tensor, b: Optional[Tensor], p: Prec) -> Tensor:
    """nn.Linear: x@W^T+b, weight (out,in); fp32 accumulate, one rounding."""
    y = x.to(p.dtype) @ w.to(p.dtype).t()
    if b is not None:
        y = y + b.to(p.dtype)
    return p.r(y)


def silu(x: Tensor, p: Prec) -> Tensor:
    return p.r(x * torch.sigmoid(x))


def gelu_tanh(x: Tensor, p: Prec) -> Tensor:
    """nn.gelu_approx (feed_forward.py:12, text_projection.py:19)."""
    return p.r(0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x * x * x))))


def rms_norm(x: Tensor, p: Prec, eps: float = 1e-6, weight: Optional[Tensor] = None) -> Tensor:
    """utils.py:398-400 (weight=1) and nn.RMSNorm (attention.py:96-97).  fp32 inside."""
    xf = x.to(p.dtype)
    y = xf * torch.rsqrt(torch.mean(xf * xf, dim=-1, keepdim=True) + eps)
    if weight is not None:
        y = y * weight.to(p.dtype)
    return p.r(y)


def layer_norm_noaffine(x: Tensor, p: Prec, eps: float = 1e-6) -> Tensor:
    """nn.LayerNorm(affine=False) (ltx.py:300)."""
    xf = x.to(p.dtype)
    mu = xf.mean(dim=-1, keepdim=True)
    var = ((xf - mu) ** 2).mean(dim=-1, keepdim=True)
    return p.r((xf - mu) * torch.rsqrt(var + eps))


def get_timestep_embedding(t: Tensor, dim: int = 256) -> Tensor:
    """utils.py:486-526 with flip_sin_to_cos=True, downscale_freq_shift=0 (adaln.py:70).
    Always fp32: [cos | sin](t * exp(-ln(1e4) * i / half))."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half
    emb = t.reshape(-1, 1).to(torch.float32) * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def adaln_single(t: Tensor, W: Dict[str, Tensor], prefix: str, p: Prec) -> Tuple[Tensor, Tensor]:
    """adaln.py:9-48,81-85,134-138.  t: 1-D timesteps (already *1000).  Returns
    (scale_shift (T,6D), embedded (T,D))."""
    proj = p.r(get_timestep_embedding(t))                    # astype(hidden_dtype)
    h = linear(proj, W[f"{prefix}.emb.timestep_embedder.linear1.weight"],
               W[f"{prefix}.emb.timestep_embedder.linear1.bias"], p)
    h = silu(h, p)
    emb = linear(h, W[f"{prefix}.emb.timestep_embedder.linear2.weight"],
                 W[f"{prefix}.emb.timestep_embedder.linear2.bias"], p)
    ss = linear(silu(emb, p), W[f"{prefix}.linear.weight"], W[f"{prefix}.linear.bias"], p)
    return ss, emb


def sdpa(q: Tensor, k: Tensor, v: Tensor, heads: int, p: Prec) -> Tensor:
    """attention.py:13-53: softmax(q k^T / sqrt(dh)) v, no mask on this path."""
    b, tq, d = q.shape
    tk = k.shape[1]
    dh = d // heads
    qh = q.reshape(b, tq, heads, dh).transpose(1, 2).to(p.dtype)
    kh = k.reshape(b, tk, heads, dh).transpose(1, 2).to(p.dtype)
    vh = v.reshape(b, tk, heads, dh).transpose(1, 2).to(p.dtype)
    if p.flash_sdpa:
        # "flash" policy: the documented rounding points of an MFMA attention kernel, independent of its tiling:
        #   S = q.k^T (fp32);  c = fp32(scale*log2 e);  M = ceil(rowmax(S)*c)  - an INTEGER offset in the exp2 domain;
        #   P = exp2(fma(S, c, -M)) (fp32);  l = sum P (fp32, un-rounded P);  O = (bf16(P) @ V) / l, one rounding to bf16.
        # Because M is an integer, bf16(exp2(x - M)) = 2^-M * bf16(exp2(x)) exactly: WHICH integer a kernel uses (the
        # row's final max, or the running max of the key tile it is processing) does not change a single rounded P, so
        # a tiled online-softmax kernel differs from this formula by fp32 summation order only.
        c = np.float32(np.float32(1.0 / math.sqrt(dh)) * np.float32(1.4426950408889634))
        s = (qh @ kh.transpose(-1, -2)).to(torch.float32)
        M = torch.ceil(s.amax(dim=-1, keepdim=True) * float(c))
        pw = torch.exp2((s.double() * float(c) - M.double()).to(torch.float32))      # fma: one rounding of the exponent
        l = pw.sum(dim=-1, keepdim=True)
        o = (pw.to(torch.bfloat16).to(torch.float32) @ vh.to(torch.float32)) / l
        return p.r(o.to(p.dtype).transpose(1, 2).reshape(b, tq, d))
    s = (qh @ kh.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
    o = torch.softmax(s, dim=-1) @ vh
    return p.r(o.transpose(1, 2).reshape(b, tq, d))


def attention(x: Tensor, W: Dict[str, Tensor], prefix: str, heads: int, p: Prec,
              context: Optional[Tensor] = None, pe: Optional[Tuple[Tensor, Tensor]] = None,
              eps: float = 1e-6, taps: Optional[dict] = None, tag: str = "") -> Tensor:
    """attention.py:102-142.  ``taps``: receives every intermediate (test instrumentation)."""
    ctx = x if context is None else context
    q = linear(x, W[f"{prefix}.to_q.weight"], W[f"{prefix}.to_q.bias"], p)
    k = linear(ctx, W[f"{prefix}.to_k.weight"], W[f"{prefix}.to_k.bias"], p)
    v = linear(ctx, W[f"{prefix}.to_v.weight"], W[f"{prefix}.to_v.bias"], p)
    if taps is not None:
        taps[f"{tag}q_raw"], taps[f"{tag}k_raw"], taps[f"{tag}v"] = q, k, v
    q = rms_norm(q, p, eps, W[f"{prefix}.q_norm.weight"])    # over the full inner dim
    k = rms_norm(k, p, eps, W[f"{prefix}.k_norm.weight"])
    if pe is not None:
        q = apply_split_rotary_emb(q, pe[0], pe[1], p)
        k = apply_split_rotary_emb(k, pe[0], pe[1], p)
    o = sdpa(q, k, v, heads, p)
    if taps is not None:
        taps[f"{tag}q"], taps[f"{tag}k"], taps[f"{tag}att"] = q, k, o
    return linear(o, W[f"{prefix}.to_out.weight"], W[f"{prefix}.to_out.bias"], p)


def feed_forward(x: Tensor, W: Dict[str, Tensor], prefix: str, p: Prec, taps: Optional[dict] = None) -> Tensor:
    """feed_forward.py:35-40."""
    h = linear(x, W[f"{prefix}.proj_in.weight"], W[f"{prefix}.proj_in.bias"], p)
    h = gelu_tanh(h, p)
    if taps is not None:
        taps["hff"] = h
    return linear(h, W[f"{prefix}.proj_out.weight"], W[f"{prefix}.proj_out.bias"], p)


def ada_values(table: Tensor, ts: Tensor, lo: int, hi: int, p: Prec) -> List[Tensor]:
    """transformer.py:135-177: table[k] + t[:,:,k,:] (bf16 add), k in [lo,hi)."""
    b, n, _ = ts.shape
    k = table.shape[0]
    t4 = ts.reshape(b, n, k, -1)
    return [p.r(table[i].to(p.dtype)[None, None, :] + t4[:, :, i, :].to(p.dtype)) for i in range(lo, hi)]


def modulate(x: Tensor, scale: Tensor, shift: Tensor, p: Prec) -> Tensor:
    """x * (1 + scale) + shift, rounding per op (transformer.py:253,346)."""
    return p.r(p.r(x * p.r(1.0 + scale)) + shift)


def transformer_block(x: Tensor, ts: Tensor, context: Tensor, pe, W: Dict[str, Tensor], i: int,
                      cfg: DiTConfig, p: Prec, taps: Optional[dict] = None) -> Tensor:
    """transformer.py:247-261,342-347 (video branch only).  ``taps``: receives every intermediate."""
    pre = f"transformer_blocks.{i}"
    table = W[f"{pre}.scale_shift_table"]
    shift_msa, scale_msa, gate_msa = ada_values(table, ts, 0, 3, p)
    nx = modulate(rms_norm(x, p, cfg.norm_eps), scale_msa, shift_msa, p)
    if taps is not None:
        taps["x0"], taps["nx1"] = x, nx
    x = p.r(x + p.r(attention(nx, W, f"{pre}.attn1", cfg.heads, p, pe=pe, eps=cfg.norm_eps, taps=taps, tag="a1.") * gate_msa))
    nx = rms_norm(x, p, cfg.norm_eps)
    if taps is not None:
        taps["x1"], taps["nx2"] = x, nx
    x = p.r(x + attention(nx, W, f"{pre}.attn2", cfg.heads, p, context=context, eps=cfg.norm_eps, taps=taps, tag="a2."))
    shift_mlp, scale_mlp, gate_mlp = ada_values(table, ts, 3, 6, p)
    nx = modulate(rms_norm(x, p, cfg.norm_eps), scale_mlp, shift_mlp, p)
    if taps is not None:
        taps["x2"], taps["nx3"] = x, nx
    x = p.r(x + p.r(feed_forward(nx, W, f"{pre}.ff", p, taps=taps) * gate_mlp))
    if taps is not None:
        taps["x3"] = x
    return x


def ltx_forward(latent: Tensor, timesteps: Tensor, context: Tensor, pe: Tuple[Tensor, Tensor],
                W: Dict[str, Tensor], cfg: DiTConfig, p: Prec,
                return_hidden: bool = False):
    """LTXModel.__call__ video-only (ltx.py:459-506 -> 129-158, 422-457).
    latent (B,N,128), timesteps (B,N) (sigma*mask, in the model dtype), context (B,S,3840),
    pe = (cos,sin) (B|1,H,N,64).  Returns velocity (B,N,128)."""
    latent = p.r(latent)
    context = p.r(context)
    b, n, _ = latent.shape
    x = linear(latent, W["patchify_proj.weight"], W["patchify_proj.bias"], p)
    t = p.r(p.r(timesteps) * cfg.timestep_scale_multiplier)            # ltx.py:68 (stays bf16)
    ss, emb = adaln_single(t.reshape(-1), W, "adaln_single", p)        # per token (ltx.py:69)
    ts = ss.reshape(b, n, -1)
    emb = emb.reshape(b, n, -1)
    ctx = linear(context, W["caption_projection.linear1.weight"], W["caption_projection.linear1.bias"], p)
    ctx = gelu_tanh(ctx, p)
    ctx = linear(ctx, W["caption_projection.linear2.weight"], W["caption_projection.linear2.bias"], p)
    ctx = ctx.reshape(b, -1, x.shape[-1])
    cos, sin = pe
    if cos.shape[0] != b:
        cos, sin = cos.expand(b, *cos.shape[1:]), sin.expand(b, *sin.shape[1:])
    hidden = []
    for i in range(cfg.num_layers):
        x = transformer_block(x, ts, ctx, (cos, sin), W, i, cfg, p)
        if return_hidden:
            hidden.append(x)
    # _process_output (ltx.py:432-457): order shift, scale
    tab = W["scale_shift_table"].to(p.dtype)
    shift = p.r(tab[0][None, None, :] + emb)
    scale = p.r(tab[1][None, None, :] + emb)
    x = layer_norm_noaffine(x, p, cfg.norm_eps)
    x = modulate(x, scale, shift, p)
    v = linear(x, W["proj_out.weight"], W["proj_out.bias"], p)
    if return_hidden:
        return v, hidden
    return v


# --------------------------------------------------------------------------------------
# step algebra (generate.py:1283-1301, utils.py:404-440, conditioning/latent.py:180-196)
# --------------------------------------------------------------------------------------

def cfg_combine(v_pos: Tensor, v_neg: Tensor, scale: float, p: Prec) -> Tensor:
    """generate.py:1255: v+ + (s-1)(v+ - v-), rounding per op."""
    return p.r(v_pos + p.r((scale - 1.0) * p.r(v_pos - v_neg)))


def cfg_delta(cond: Tensor, uncond: Tensor, scale: float) -> Tensor:
    """generate.py:382-393."""
    return (scale - 1.0) * (cond - uncond)


def to_denoised(noisy: Tensor, velocity: Tensor, sigma: float, p: Prec) -> Tensor:
    """utils.py:404-440: fp32 x - sigma*v, one rounding."""
    return p.r(noisy.to(p.dtype) - float(sigma) * velocity.to(p.dtype))


def apply_denoise_mask(denoised: Tensor, clean: Tensor, mask: Tensor, p: Prec) -> Tensor:
    """conditioning/latent.py:180-196."""
    return p.r(p.r(denoised * mask) + p.r(clean * p.r(1.0 - mask)))


def euler_step(latents: Tensor, denoised: Tensor, sigma: float, sigma_next: float, p: Prec) -> Tensor:
    """generate.py:1293-1301: fp32 x0 + s'*(x-x0)/s; last step returns x0."""
    if sigma_next > 0:
        lf, df = latents.to(p.dtype), denoised.to(p.dtype)
        return p.r(df + float(sigma_next) * (lf - df) / float(sigma))
    return denoised


def bf16_round_scalar(x: float) -> float:
    return float(torch.tensor(x, dtype=torch.float32).to(torch.bfloat16).to(torch.float32))


def denoise_dev(latents: Tensor, positions: np.ndarray, ctx_pos: Tensor, ctx_neg: Tensor,
                W: Dict[str, Tensor], cfg: DiTConfig, sigmas: Sequence[float], p: Prec,
                cfg_scale: float = 4.0, clean_latent: Optional[Tensor] = None,
                denoise_mask: Optional[Tensor] = None, compiled: bool = True, bf16_euler: bool = False) -> Tensor:
    """denoise_dev loop (generate.py:1060-1327).  ``compiled=True`` follows the mx.compile'd
    step (bf16-rounded sigma in x0 and Euler, generate.py:1109-1174); ``False`` follows the
    eager body (Python-float sigma in Euler, 1293-1301).  Timesteps are sigma_bf16*mask in
    both (1084,1237).  ``cfg_scale == 1`` is the distilled (no-CFG) loop, generate.py:564-881, whose compiled
    step with ``bf16_euler`` (fp32_euler=False) evaluates the Euler update op by op in bf16 (generate.py:748)."""
    b, c, f, h, w = latents.shape
    n = f * h * w
    pe = precompute_freqs_cis(torch.from_numpy(positions), cfg.dim, cfg.theta, cfg.max_pos, cfg.heads)
    if denoise_mask is not None:
        mask_tok = denoise_mask.reshape(b, 1, f, 1, 1).expand(b, 1, f, h, w).reshape(b, n)
    else:
        mask_tok = torch.ones(b, n)
    x = p.r(latents)
    use_cfg = cfg_scale != 1.0
    for i in range(len(sigmas) - 1):
        s, s_next = float(sigmas[i]), float(sigmas[i + 1])
        s_m = bf16_round_scalar(s) if p.emulate_bf16 else s
        sn_m = bf16_round_scalar(s_next) if p.emulate_bf16 else s_next
        tok = latent_to_tokens(x)
        ts = p.r(s_m * mask_tok)
        v = ltx_forward(tok, ts, ctx_pos, pe, W, cfg, p)
        if use_cfg:
            vn = ltx_forward(tok, ts, ctx_neg, pe, W, cfg, p)
            v = cfg_combine(v, vn, cfg_scale, p)
        vel = tokens_to_latent(v, x.shape)
        x0 = to_denoised(x, vel, s_m, p)
        if denoise_mask is not None:
            x0 = apply_denoise_mask(x0, p.r(clean_latent), p.r(denoise_mask), p)
        if compiled and bf16_euler:
            x = p.r(x0 + p.r(p.r(sn_m * p.r(x - x0)) / s_m))
        elif compiled:
            x = p.r(x0 + sn_m * (x - x0) / s_m)
        else:
            x = euler_step(x, x0, s, s_next, p)
    return x


# --------------------------------------------------------------------------------------
# synthetic weights
# --------------------------------------------------------------------------------------

def make_weights(cfg: DiTConfig, seed: int = 1234, dtype=torch.bfloat16) -> Dict[str, Tensor]:
    """Seeded synthetic checkpoint with the reference's sanitised key names.  Linear weights
    N(0,0.02^2), biases 0.01*N(0,1), scale_shift tables N(0,0.02^2) (zero-init in the
    reference would hide the AdaLN path), q/k norm weights 1+0.1*N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    D, FF = cfg.dim, cfg.dim * cfg.ff_mult
    W: Dict[str, Tensor] = {}

    def lin(name, out_f, in_f):
        W[f"{name}.weight"] = (torch.randn(out_f, in_f, generator=g) * 0.02).to(dtype)
        W[f"{name}.bias"] = (torch.randn(out_f, generator=g) * 0.01).to(dtype)

    lin("patchify_proj", D, cfg.in_channels)
    lin("adaln_single.emb.timestep_embedder.linear1", D, 256)
    lin("adaln_single.emb.timestep_embedder.linear2", D, D)
    lin("adaln_single.linear", 6 * D, D)
    lin("caption_projection.linear1", D, cfg.caption_channels)
    lin("caption_projection.linear2", D, D)
    W["scale_shift_table"] = (torch.randn(2, D, generator=g) * 0.02).to(dtype)
    lin("proj_out", cfg.out_channels, D)
    for i in range(cfg.num_layers):
        pre = f"transformer_blocks.{i}"
        for a in ("attn1", "attn2"):
            for nm in ("to_q", "to_k", "to_v", "to_out"):
                lin(f"{pre}.{a}.{nm}", D, D)
            W[f"{pre}.{a}.q_norm.weight"] = (1.0 + 0.1 * torch.randn(D, generator=g)).to(dtype)
            W[f"{pre}.{a}.k_norm.weight"] = (1.0 + 0.1 * torch.randn(D, generator=g)).to(dtype)
        lin(f"{pre}.ff.proj_in", FF, D)
        lin(f"{pre}.ff.proj_out", D, FF)
        W[f"{pre}.scale_shift_table"] = (torch.randn(6, D, generator=g) * 0.02).to(dtype)
    return W

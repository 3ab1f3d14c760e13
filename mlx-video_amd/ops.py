"""torch.Tensor -> libltxk C-ABI call shims.  One function per entry point of include/ltxk.h.
Tensors carry device memory only; all math happens in the HIP kernels.  Every call is queued
on torch's current HIP stream (so a torch.cuda.graph capture records the whole step)."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from ._lib import AttnArgs, GemmArgs, check

EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_SILU, EPI_BIAS_GATE_RES, EPI_BIAS_RES, EPI_SCALE_RES = 0, 1, 2, 3, 4, 5
BF16 = torch.bfloat16


class KernelTimer:
    """Optional per-launch HIP-event timing (bench.py roofline leg).  Events are recorded on the
    stream the kernels are launched on; nothing is synchronised until ``summary()``."""

    def __init__(self):
        self.records = []       # (family, algorithmic flops, algorithmic bytes, start, end)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for fam, flops, nbytes, s, e in self.records:
            d = out.setdefault(fam, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out


TIMER: Optional[KernelTimer] = None


class _timed:
    __slots__ = ("fam", "flops", "nbytes", "s")

    def __init__(self, fam, flops=0.0, nbytes=0.0):
        self.fam, self.flops, self.nbytes = fam, flops, nbytes

    def __enter__(self):
        if TIMER is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.s.record()

    def __exit__(self, *a):
        if TIMER is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            TIMER.records.append((self.fam, self.flops, self.nbytes, self.s, e))


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype, name: str) -> None:
    if not t.is_cuda:
        raise _lib.LtxkError(f"{name}: expected a device tensor (the product path has no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")


_GEMM_WS = {}
GEMM_WORKSPACE_BYTES = 64 << 20
SPLITK_MAX_M = 640          # rows up to which a launch is offered the split-K scratch (the library decides; gemm.hip)


def _gemm_workspace(dev: torch.device) -> torch.Tensor:
    """Per-device fp32 scratch handed to ltxk_gemm_bf16 for its split-K form at small M (caller-owned and reused by every
    call on the stream - launches on one stream are ordered; the library never allocates)."""
    key = dev.index if dev.index is not None else 0
    if key not in _GEMM_WS:
        _GEMM_WS[key] = torch.empty(GEMM_WORKSPACE_BYTES // 4, dtype=torch.float32, device=dev)
    return _GEMM_WS[key]


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], *, epilogue: int = EPI_BIAS,
         out: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None,
         gate: Optional[torch.Tensor] = None, gate_row: Optional[torch.Tensor] = None,
         gate_stride: int = 0, out_tokens_per_batch: int = 0, alpha: float = 1.0,
         out2: Optional[torch.Tensor] = None, n_split: int = 0, sumsq: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = epi(a @ w.T + bias).  a (M,K) (row stride may exceed K), w (N,K) contiguous.
    ``n_split``/``out2``: columns >= n_split go transposed per batch (out_tokens_per_batch tokens) to out2
    (B, N-n_split, ld).  ``sumsq``: (M, >= cols/64) fp32, receives the per-64-column sums of squares of the stored
    row-major outputs."""
    _req(a, BF16, "gemm.a"); _req(w, BF16, "gemm.w")
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K or not w.is_contiguous() or a.stride(1) != 1:
        raise ValueError(f"gemm: bad operand layout a{tuple(a.shape)} w{tuple(w.shape)}")
    if out is None:
        if out_tokens_per_batch:
            raise ValueError("gemm: transposed output needs a preallocated `out`")
        out = torch.empty((M, N), dtype=BF16, device=a.device)
    if n_split:
        if out2 is None or out2.dtype != BF16:
            raise ValueError("gemm: split output needs a bf16 `out2`")
    if sumsq is not None and (sumsq.dtype != torch.float32 or sumsq.stride(-1) != 1):
        raise TypeError("gemm: sumsq must be float32 with unit inner stride")
    args = GemmArgs()
    args.A, args.W, args.bias, args.out = _p(a), _p(w), _p(bias), _p(out)
    args.resid, args.gate, args.gate_row = _p(resid), _p(gate), _p(gate_row)
    args.M, args.N, args.K = M, N, K
    args.lda = a.stride(0)
    args.ldo = out.stride(-2)
    args.ldr = resid.stride(0) if resid is not None else 0
    args.gate_stride = gate_stride
    args.epilogue = epilogue
    args.out_tokens_per_batch = out_tokens_per_batch
    args.alpha = alpha
    args.out2, args.n_split, args.ldo2 = _p(out2), n_split, (out2.stride(-2) if out2 is not None else 0)
    args.sumsq, args.sumsq_ld = _p(sumsq), (sumsq.stride(0) if sumsq is not None else 0)
    if M <= SPLITK_MAX_M:
        ws = _gemm_workspace(a.device)
        args.workspace, args.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    with _timed("gemm_bf16", 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N)):
        check(_lib.load().ltxk_gemm_bf16(ctypes.byref(args), _stream()), "ltxk_gemm_bf16")
    return out


def flash_attn(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, B: int, H: int,
               Tq: int, Tk: int, scale: float, q_sumsq: Optional[torch.Tensor] = None,
               q_norm_weight: Optional[torch.Tensor] = None, cos: Optional[torch.Tensor] = None,
               sin: Optional[torch.Tensor] = None, eps: float = 1e-6, tail_split: bool = True) -> torch.Tensor:
    """q (B*Tq, >=H*128) view, k (B*Tk, ...) view, vt (B, H*128, ldvt), out (B*Tq, H*128).  With ``q_sumsq``
    (B*Tq, >= H*2) the raw q projection is normalised (q_norm_weight) and rotated (cos/sin (H,Tq,64)) inside the kernel.
    ``tail_split=False``: LTXK_ATTN_NO_TAIL_SPLIT (results independent of how many (batch, head) pairs share the launch)."""
    a = AttnArgs()
    a.flags = 0 if tail_split else _lib.ATTN_NO_TAIL_SPLIT
    a.q, a.k, a.vt, a.out = _p(q), _p(k), _p(vt), _p(out)
    a.ldq, a.ldk, a.ldvt, a.ldo = q.stride(0), k.stride(0), vt.stride(-2), out.stride(0)
    a.B, a.H, a.Tq, a.Tk, a.scale = B, H, Tq, Tk, scale
    if q_sumsq is not None:
        _req(q_norm_weight, BF16, "flash_attn.q_norm_weight")
        if q_sumsq.dtype != torch.float32 or q_sumsq.stride(-1) != 1:
            raise TypeError("flash_attn: q_sumsq must be float32 with unit inner stride")
        if cos is not None and (cos.dtype != torch.float32 or not cos.is_contiguous() or not sin.is_contiguous()):
            raise TypeError("flash_attn: cos/sin must be contiguous float32 (H,Tq,64)")
        a.q_sumsq, a.q_sumsq_ld, a.q_sumsq_n = _p(q_sumsq), q_sumsq.stride(0), H * 2
        a.q_norm_weight, a.cos, a.sin, a.eps = _p(q_norm_weight), _p(cos), _p(sin), eps
    with _timed("flash_attn", 4.0 * B * H * Tq * Tk * 128, 2.0 * B * H * 128 * (2 * Tq + 2 * Tk)):
        check(_lib.load().ltxk_flash_attn(ctypes.byref(a), _stream()), "ltxk_flash_attn")
    return out


def rmsnorm_modulate(x: torch.Tensor, eps: float, scale: Optional[torch.Tensor] = None,
                     shift: Optional[torch.Tensor] = None, mod_stride: int = 0,
                     mod_row: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                     sumsq: Optional[torch.Tensor] = None, scale_is_one_plus: bool = False) -> torch.Tensor:
    """``sumsq`` (M, D/64) fp32: the rows' sums of squares in 64-column partials (gemm(..., sumsq=)); without it the
    kernel reduces the row itself.  ``scale_is_one_plus``: scale holds bf16(1+scale) (needs sumsq)."""
    _req(x, BF16, "rmsnorm_modulate.x")
    M, D = x.shape
    if out is None:
        out = torch.empty_like(x)
    if sumsq is not None:
        if sumsq.dtype != torch.float32 or sumsq.stride(-1) != 1 or sumsq.shape[0] != M:
            raise TypeError("rmsnorm_modulate: sumsq must be (M, >= D/64) float32")
        with _timed("rmsnorm_modulate", 0.0, 4.0 * M * D):
            check(_lib.load().ltxk_rmsnorm_modulate_ss(_p(x), _p(out), M, D, eps, _p(sumsq), sumsq.stride(0), D // 64, _p(scale),
                                                       _p(shift), mod_stride, _p(mod_row), int(scale_is_one_plus), _stream()),
                  "ltxk_rmsnorm_modulate_ss")
        return out
    if scale_is_one_plus:
        raise ValueError("rmsnorm_modulate: scale_is_one_plus needs sumsq")
    with _timed("rmsnorm_modulate", 0.0, 4.0 * M * D):
        check(_lib.load().ltxk_rmsnorm_modulate(_p(x), _p(out), M, D, eps, _p(scale), _p(shift), mod_stride,
                                                _p(mod_row), _stream()), "ltxk_rmsnorm_modulate")
    return out


def layernorm_modulate(x: torch.Tensor, eps: float, scale: Optional[torch.Tensor] = None,
                       shift: Optional[torch.Tensor] = None, mod_stride: int = 0,
                       mod_row: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(x, BF16, "layernorm_modulate.x")
    M, D = x.shape
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().ltxk_layernorm_modulate(_p(x), _p(out), M, D, eps, _p(scale), _p(shift), mod_stride,
                                              _p(mod_row), _stream()), "ltxk_layernorm_modulate")
    return out


def qknorm_rope(buf: torch.Tensor, nseg: int, D: int, weight: torch.Tensor, cos: Optional[torch.Tensor],
                sin: Optional[torch.Tensor], T: int, H: int, eps: float, sumsq: Optional[torch.Tensor] = None) -> torch.Tensor:
    """In place on the first nseg*D columns of buf (M, ld).  ``sumsq`` (M, >= nseg*D/64) fp32: precomputed partial
    sums of squares of those columns (gemm(..., sumsq=))."""
    _req(buf, BF16, "qknorm_rope.buf")
    if cos is not None and (cos.dtype != torch.float32 or not cos.is_contiguous()):
        raise TypeError("qknorm_rope: cos/sin must be contiguous float32 (H,T,64)")
    if sumsq is not None:
        if sumsq.dtype != torch.float32 or sumsq.stride(-1) != 1:
            raise TypeError("qknorm_rope: sumsq must be float32 with unit inner stride")
        with _timed("qknorm_rope", 0.0, 4.0 * buf.shape[0] * nseg * D + (8.0 * buf.shape[0] * D // 2 if cos is not None else 0.0)):
            check(_lib.load().ltxk_qknorm_rope_ss(_p(buf), buf.stride(0), buf.shape[0], nseg, D, _p(weight), _p(cos), _p(sin),
                                                  T, H, eps, _p(sumsq), sumsq.stride(0), _stream()), "ltxk_qknorm_rope_ss")
        return buf
    with _timed("qknorm_rope", 0.0, 4.0 * buf.shape[0] * nseg * D + (8.0 * buf.shape[0] * D // 2 if cos is not None else 0.0)):
        check(_lib.load().ltxk_qknorm_rope(_p(buf), buf.stride(0), buf.shape[0], nseg, D, _p(weight), _p(cos), _p(sin),
                                           T, H, eps, _stream()), "ltxk_qknorm_rope")
    return buf


def timestep_embed(t: torch.Tensor, dim: int = 256, mult: float = 1.0) -> torch.Tensor:
    _req(t, BF16, "timestep_embed.t")
    out = torch.empty((t.numel(), dim), dtype=BF16, device=t.device)
    check(_lib.load().ltxk_timestep_embed(_p(t), _p(out), t.numel(), dim, mult, _stream()), "ltxk_timestep_embed")
    return out


def rope_table(positions: torch.Tensor, freq: torch.Tensor, H: int, dim: int, max_pos) -> tuple:
    """positions (3,T,2) fp32, freq (n_freq) fp32 -> cos, sin (H,T,dim/2/H) fp32."""
    T = positions.shape[1]
    per_head = dim // 2 // H
    cos = torch.empty((H, T, per_head), dtype=torch.float32, device=positions.device)
    sin = torch.empty_like(cos)
    mp = (ctypes.c_float * 3)(*[float(v) for v in max_pos])
    check(_lib.load().ltxk_rope_table(_p(positions), _p(freq), _p(cos), _p(sin), T, H, dim, freq.numel(), mp,
                                      _stream()), "ltxk_rope_table")
    return cos, sin


def ada_combine(table: torch.Tensor, ada: torch.Tensor, L: int, U: int, K: int, D: int, one_plus_mask: int = 0) -> torch.Tensor:
    """table (L,K,D), ada (U,K*D) -> (L,U,K,D); rows k with bit k of one_plus_mask set hold bf16(1 + value)."""
    _req(table, BF16, "ada_combine.table"); _req(ada, BF16, "ada_combine.ada")
    out = torch.empty((L, U, K, D), dtype=BF16, device=ada.device)
    check(_lib.load().ltxk_ada_combine(_p(table), _p(ada), _p(out), L, U, K, D, one_plus_mask, _stream()), "ltxk_ada_combine")
    return out


def silu(x: torch.Tensor) -> torch.Tensor:
    _req(x, BF16, "silu.x")
    out = torch.empty_like(x)
    check(_lib.load().ltxk_silu(_p(x), _p(out), x.numel(), _stream()), "ltxk_silu")
    return out


def latent_to_tokens(latent: torch.Tensor, rep: int = 1) -> torch.Tensor:
    """(B,C,F,H,W) or (B,C,S) -> (rep*B,S,C)."""
    _req(latent, BF16, "latent_to_tokens.latent")
    B, C = latent.shape[:2]
    S = latent.numel() // (B * C)
    lat = latent.contiguous()
    out = torch.empty((rep * B, S, C), dtype=BF16, device=latent.device)
    check(_lib.load().ltxk_latent_to_tokens(_p(lat), _p(out), B, C, S, rep, _stream()), "ltxk_latent_to_tokens")
    return out


def cfg_euler_step(v_pos: torch.Tensor, v_neg: Optional[torch.Tensor], latent: torch.Tensor, cfg_scale: float,
                   sigma: float, sigma_next: float, clean: Optional[torch.Tensor] = None,
                   mask_tok: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                   sigmas_dev: Optional[torch.Tensor] = None, bf16_euler: bool = False) -> torch.Tensor:
    """v_* (B,S,C) tokens; latent (B,C,...) channels-first; mask_tok (B,S) float32.  With ``sigmas_dev`` (2
    float32 on the device) the scalars are read from device memory (hipGraph replay).  ``out`` may be
    ``latent`` itself (each element is read and written by one thread).  ``bf16_euler``: the reference's
    fp32_euler=False compiled step (generate.py:741-748)."""
    _req(latent, BF16, "cfg_euler_step.latent")
    B, C = latent.shape[:2]
    S = latent.numel() // (B * C)
    if out is None:
        out = torch.empty_like(latent)
    if sigmas_dev is not None:
        check(_lib.load().ltxk_cfg_euler_step_dev(_p(v_pos), _p(v_neg), _p(latent), _p(out), _p(clean), _p(mask_tok),
                                                  B, C, S, cfg_scale, _p(sigmas_dev), int(bf16_euler), _stream()),
              "ltxk_cfg_euler_step_dev")
        return out
    check(_lib.load().ltxk_cfg_euler_step(_p(v_pos), _p(v_neg), _p(latent), _p(out), _p(clean), _p(mask_tok),
                                          B, C, S, cfg_scale, sigma, sigma_next, int(bf16_euler), _stream()),
          "ltxk_cfg_euler_step")
    return out


def step_scalars(ts_all: torch.Tensor, sig_all: torch.Tensor, step: torch.Tensor, ts: torch.Tensor, sig: torch.Tensor) -> None:
    """ts_all (steps,U) bf16, sig_all (steps,2) fp32, step (1) int32 -> ts (U), sig (2); step += 1 (device side)."""
    _req(ts_all, BF16, "step_scalars.ts_all")
    n, U = ts_all.shape
    if sig_all.dtype != torch.float32 or tuple(sig_all.shape) != (n, 2) or step.dtype != torch.int32 or ts.numel() != U:
        raise TypeError("step_scalars: bad table shapes / dtypes")
    check(_lib.load().ltxk_step_scalars(_p(ts_all), _p(sig_all), _p(step), _p(ts), _p(sig), U, n, _stream()), "ltxk_step_scalars")


def euler_only(latent: torch.Tensor, denoised: torch.Tensor, sigma: float, sigma_next: float) -> torch.Tensor:
    _req(latent, BF16, "euler_only.latent"); _req(denoised, BF16, "euler_only.denoised")
    out = torch.empty_like(latent)
    check(_lib.load().ltxk_euler_step(_p(latent.contiguous()), _p(denoised.contiguous()), _p(out), latent.numel(),
                                      sigma, sigma_next, _stream()), "ltxk_euler_step")
    return out


def resize_area(x: torch.Tensor, oh: int, ow: int) -> torch.Tensor:
    """(..., H, W) fp32 / bf16 device tensor -> (..., oh, ow) bf16: cv2.INTER_AREA downscale of every (H,W) plane
    (prepare_video_for_encoding, utils.py:699-705)."""
    if not x.is_cuda or x.dtype not in (torch.float32, BF16):
        raise _lib.LtxkError("resize_area: expected a float32 / bfloat16 device tensor")
    x = x.contiguous()
    H, W = x.shape[-2:]
    planes = x.numel() // (H * W)
    out = torch.empty(tuple(x.shape[:-2]) + (oh, ow), dtype=BF16, device=x.device)
    check(_lib.load().ltxk_resize_area(_p(x), int(x.dtype == torch.float32), _p(out), planes, H, W, oh, ow, _stream()), "ltxk_resize_area")
    return out

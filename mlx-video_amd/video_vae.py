"""Host-side mirror of the reference's video-VAE seams over the libltxk C ABI:

  * ``LTX2VideoDecoder.__call__(sample, causal=False, timestep=None, ...)`` and ``decode_tiled``
    (mlx_video/models/ltx/video_vae/decoder.py:361-531, tiling.py:279-509)
  * ``VideoEncoder.__call__(sample)`` (video_vae/video_vae.py:321-372)

Module-level interface is the reference's: channels-first (B,C,F,H,W) in and out.  Inside, the
volume lives channels-last (B,D,H,W,C) so that a voxel is one contiguous row: the convolutions
are implicit GEMMs whose A-rows are gathered with the halo resolved in the load address (no
padded copies), and the norm/activation kernels are one-row-per-voxel streams.
"""
from __future__ import annotations


import ctypes
import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import _lib, ops
from ._lib import Conv3dArgs, check

BF16 = torch.bfloat16
PAD_ZEROS, PAD_REFLECT = 0, 1
DEC_CH = (1024, 512, 256, 128)


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


_ZERO_PAGES: Dict[int, torch.Tensor] = {}
_WORKSPACES: Dict[int, torch.Tensor] = {}
SPLITK_WORKSPACE_BYTES = 96 << 20


def _workspace(dev) -> torch.Tensor:
    """Per-device fp32 scratch handed to the conv kernel for split-K (caller-owned, reused by every call
    on the stream; the library never allocates)."""
    key = dev.index if dev.index is not None else 0
    if key not in _WORKSPACES:
        _WORKSPACES[key] = torch.empty(SPLITK_WORKSPACE_BYTES // 4, dtype=torch.float32, device=dev)
    return _WORKSPACES[key]


def _zero_page(dev) -> torch.Tensor:
    key = dev.index if dev.index is not None else 0
    if key not in _ZERO_PAGES:
        _ZERO_PAGES[key] = torch.zeros(256, dtype=torch.uint8, device=dev)
    return _ZERO_PAGES[key]


# ------------------------------------------------------------------------------------ op shims
def conv3d(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, causal: bool, pad_mode: int,
           resid: Optional[torch.Tensor] = None, act: Optional[dict] = None, keep_out: bool = True):
    """x (B,D,H,W,Cin) bf16 channels-last; w (Cout,3,3,3,Cin), or (Cout,3,3,Cin) for a per-frame 3x3 kernel;
    returns (B,D,H,W,Cout).  ``act`` = dict(eps, silu, scale, shift): the PixelNorm (+ modulation) + SiLU that follows
    this conv is applied to its output rows in the epilogue (Cout 128 / 256) and returned as a second tensor:
    ``(out, act_out)``; ``keep_out=False`` skips the raw output (``out`` is None) when nothing else reads it."""
    B, D, H, W, Cin = x.shape
    Cout = w.shape[0]
    taps_d = 3 if w.dim() == 5 else 1
    if tuple(w.shape[1:]) != ((3, 3, 3, Cin) if taps_d == 3 else (3, 3, Cin)):
        raise ValueError(f"conv3d: weight {tuple(w.shape)} does not match Cin={Cin}")
    out = torch.empty((B, D, H, W, Cout), dtype=BF16, device=x.device) if (keep_out or act is None) else None
    a = Conv3dArgs()
    a.x, a.w, a.bias, a.out, a.resid = _p(x), _p(w), _p(b), _p(out), _p(resid)
    a.zero_page = _p(_zero_page(x.device))
    a.B, a.D, a.H, a.W, a.Cin, a.Cout = B, D, H, W, Cin, Cout
    a.causal, a.pad_mode, a.taps_d = int(causal), pad_mode, taps_d
    ws = _workspace(x.device)
    a.workspace, a.workspace_bytes = _p(ws), ws.numel() * 4
    act_out = None
    if act is not None:
        act_out = torch.empty((B, D, H, W, Cout), dtype=BF16, device=x.device)
        a.act_out, a.act_scale, a.act_shift = _p(act_out), _p(act.get("scale")), _p(act.get("shift"))
        a.act_eps, a.act_silu = float(act["eps"]), int(bool(act.get("silu", True)))
    V = B * D * H * W
    ntap = 27 if taps_d == 3 else 9
    with ops._timed("conv3d_k3", 2.0 * ntap * Cin * Cout * V, 2.0 * V * (Cin + Cout) + 2.0 * ntap * Cin * Cout):
        check(_lib.load().ltxk_conv3d_k3_bf16(ctypes.byref(a), _stream()), "ltxk_conv3d_k3_bf16")
    return out if act is None else (out, act_out)


def conv_act_fusable(Cout: int, voxels: int, force: bool = False) -> bool:
    """The conv epilogue can carry the following PixelNorm + SiLU when its tile holds whole rows (Cout 128 / 256) and the
    launch would not be split along K anyway (more than 128 tiles: the small 1024 / 512-channel volumes keep split-K).
    OFF by default (``LTX2VideoDecoder.fuse_act = True`` turns it on): measured on the 33x512x512 decode, same box, interleaved -
    12.34 ms separate vs 12.47 ms fused: the normalisation's ~25 VALU per element cost the MFMA kernel's serial epilogue
    0.77 ms, more than the 0.66 ms the HBM-bound PixelNorm launches took beside nothing (profiles/r02_vae_fusion_ab.log)."""
    if not force or Cout not in (128, 256):
        return False
    bm = 256 if Cout <= 128 else 160
    return (voxels + bm - 1) // bm > 128


def pixelnorm_act(x: torch.Tensor, eps: float, silu: bool, scale: Optional[torch.Tensor] = None,
                  shift: Optional[torch.Tensor] = None) -> torch.Tensor:
    B = x.shape[0]
    C = x.shape[-1]
    V = x.numel() // C
    out = torch.empty_like(x)
    with ops._timed("pixelnorm_act", 0.0, 4.0 * V * C):
        check(_lib.load().ltxk_pixelnorm_act(_p(x), _p(out), V, C, eps, _p(scale), _p(shift), V // B, int(silu),
                                             _stream()), "ltxk_pixelnorm_act")
    return out


def d2s_add(conv: torch.Tensor, xin: Optional[torch.Tensor]) -> torch.Tensor:
    B, D, H, W, C8 = conv.shape
    Co = C8 // 8
    out = torch.empty((B, 2 * D - 1, 2 * H, 2 * W, Co), dtype=BF16, device=conv.device)
    with ops._timed("d2s_add", 0.0, 2.0 * conv.numel() * 2 + (2.0 * xin.numel() if xin is not None else 0)):
        check(_lib.load().ltxk_d2s_add(_p(conv), _p(xin), _p(out), B, D, H, W, Co, xin.shape[-1] if xin is not None else 0,
                                       _stream()), "ltxk_d2s_add")
    return out


# ------------------------------------------------------------------------------------ tiling config (host math)
def compute_trapezoidal_mask_1d(length: int, ramp_left: int, ramp_right: int, left_starts_from_0: bool = False) -> torch.Tensor:
    """tiling.py:17-62."""
    if length <= 0:
        raise ValueError("Mask length must be positive.")
    ramp_left = max(0, min(ramp_left, length))
    ramp_right = max(0, min(ramp_right, length))
    mask = [1.0] * length
    if ramp_left > 0:
        n = ramp_left + 1 if left_starts_from_0 else ramp_left + 2
        ramp = [i / (n - 1) for i in range(n)][:-1]
        if not left_starts_from_0:
            ramp = ramp[1:]
        for i in range(min(ramp_left, len(ramp))):
            mask[i] *= ramp[i]
    for i in range(ramp_right):
        mask[length - ramp_right + i] *= (ramp_right - i) / (ramp_right + 1)
    return torch.tensor(mask, dtype=torch.float32).clamp_(0, 1)


@dataclass(frozen=True)
class SpatialTilingConfig:
    tile_size_in_pixels: int
    tile_overlap_in_pixels: int = 0

    def __post_init__(self):
        if self.tile_size_in_pixels < 64:
            raise ValueError(f"tile_size_in_pixels must be at least 64, got {self.tile_size_in_pixels}")
        if self.tile_size_in_pixels % 32 != 0:
            raise ValueError(f"tile_size_in_pixels must be divisible by 32, got {self.tile_size_in_pixels}")
        if self.tile_overlap_in_pixels % 32 != 0:
            raise ValueError(f"tile_overlap_in_pixels must be divisible by 32, got {self.tile_overlap_in_pixels}")
        if self.tile_overlap_in_pixels >= self.tile_size_in_pixels:
            raise ValueError(f"Overlap must be less than tile size, got {self.tile_overlap_in_pixels} and {self.tile_size_in_pixels}")


@dataclass(frozen=True)
class TemporalTilingConfig:
    tile_size_in_frames: int
    tile_overlap_in_frames: int = 0

    def __post_init__(self):
        if self.tile_size_in_frames < 16:
            raise ValueError(f"tile_size_in_frames must be at least 16, got {self.tile_size_in_frames}")
        if self.tile_size_in_frames % 8 != 0:
            raise ValueError(f"tile_size_in_frames must be divisible by 8, got {self.tile_size_in_frames}")
        if self.tile_overlap_in_frames % 8 != 0:
            raise ValueError(f"tile_overlap_in_frames must be divisible by 8, got {self.tile_overlap_in_frames}")
        if self.tile_overlap_in_frames >= self.tile_size_in_frames:
            raise ValueError(f"Overlap must be less than tile size, got {self.tile_overlap_in_frames} and {self.tile_size_in_frames}")


@dataclass(frozen=True)
class TilingConfig:
    """tiling.py:104-211."""
    spatial_config: Optional[SpatialTilingConfig] = None
    temporal_config: Optional[TemporalTilingConfig] = None

    @classmethod
    def default(cls):
        return cls(SpatialTilingConfig(512, 64), TemporalTilingConfig(64, 24))

    @classmethod
    def spatial_only(cls, tile_size: int = 512, overlap: int = 64):
        return cls(SpatialTilingConfig(tile_size, overlap), None)

    @classmethod
    def temporal_only(cls, tile_size: int = 64, overlap: int = 24):
        return cls(None, TemporalTilingConfig(tile_size, overlap))

    @classmethod
    def aggressive(cls):
        return cls(SpatialTilingConfig(256, 64), TemporalTilingConfig(32, 8))

    @classmethod
    def conservative(cls):
        return cls(SpatialTilingConfig(768, 64), TemporalTilingConfig(96, 24))

    @classmethod
    def auto(cls, height: int, width: int, num_frames: int, spatial_threshold: int = 512, temporal_threshold: int = 65):
        needs_s = height > spatial_threshold or width > spatial_threshold
        needs_t = num_frames > temporal_threshold
        if not needs_s and not needs_t:
            return None
        est_gb = (3 * num_frames * height * width * 4) / (1024 ** 3)
        if est_gb > 2.0 or (height * width > 768 * 1024 and num_frames > 100):
            return cls.aggressive()
        sc = tc = None
        if needs_s:
            m = max(height, width)
            sc = SpatialTilingConfig(512 if 768 < m <= 1024 else 384, 64)
        if needs_t:
            ts, ov = (32, 8) if num_frames > 200 else ((48, 16) if num_frames > 100 else (64, 24))
            tc = TemporalTilingConfig(ts, ov)
        return cls(sc, tc)


@dataclass
class DimensionIntervals:
    starts: List[int]
    ends: List[int]
    left_ramps: List[int]
    right_ramps: List[int]


def split_in_spatial(size: int, overlap: int, dimension_size: int) -> DimensionIntervals:
    """tiling.py:223-235."""
    if dimension_size <= size:
        return DimensionIntervals([0], [dimension_size], [0], [0])
    n = (dimension_size + size - 2 * overlap - 1) // (size - overlap)
    starts = [i * (size - overlap) for i in range(n)]
    ends = [s + size for s in starts]
    ends[-1] = dimension_size
    return DimensionIntervals(starts, ends, [0] + [overlap] * (n - 1), [overlap] * (n - 1) + [0])


def split_in_temporal(size: int, overlap: int, dimension_size: int) -> DimensionIntervals:
    """tiling.py:238-254: later tiles start one latent frame early (causal first frame)."""
    if dimension_size <= size:
        return DimensionIntervals([0], [dimension_size], [0], [0])
    iv = split_in_spatial(size, overlap, dimension_size)
    starts = [s - (1 if i else 0) for i, s in enumerate(iv.starts)]
    lefts = [r + (1 if i else 0) for i, r in enumerate(iv.left_ramps)]
    return DimensionIntervals(starts, iv.ends, lefts, iv.right_ramps)


def map_temporal_slice(begin, end, left_ramp, right_ramp, scale):
    start, stop = begin * scale, 1 + (end - 1) * scale
    lr = 1 + (left_ramp - 1) * scale if left_ramp > 0 else 0
    return slice(start, stop), compute_trapezoidal_mask_1d(stop - start, lr, right_ramp * scale, True)


def map_spatial_slice(begin, end, left_ramp, right_ramp, scale):
    start, stop = begin * scale, end * scale
    return slice(start, stop), compute_trapezoidal_mask_1d(stop - start, left_ramp * scale, right_ramp * scale, False)


def decode_with_tiling(decoder_fn, latents: torch.Tensor, tiling_config: TilingConfig, spatial_scale: int = 32,
                       temporal_scale: int = 8, causal: bool = False, timestep=None, chunked_conv: bool = False,
                       on_frames_ready: Optional[Callable] = None, noise_fn: Optional[Callable] = None) -> torch.Tensor:
    """tiling.py:279-509: decode t/h/w tiles, fp32 accumulate tile*mask and mask, divide.  ``noise_fn(shape)``: the
    noise source of a timestep-conditioned decoder, called once per tile with the tile's latent shape (the reference
    draws mx.random.normal inside every decoder call, decoder.py:381-385)."""
    b, c, fl, hl, wl = latents.shape
    out_f, out_h, out_w = 1 + (fl - 1) * temporal_scale, hl * spatial_scale, wl * spatial_scale
    sc, tc = tiling_config.spatial_config, tiling_config.temporal_config
    s_tile, s_ov = (sc.tile_size_in_pixels // spatial_scale, sc.tile_overlap_in_pixels // spatial_scale) if sc else (max(hl, wl), 0)
    t_tile, t_ov = (tc.tile_size_in_frames // temporal_scale, tc.tile_overlap_in_frames // temporal_scale) if tc else (fl, 0)
    tiv = split_in_temporal(t_tile, t_ov, fl)
    hiv = split_in_spatial(s_tile, s_ov, hl)
    wiv = split_in_spatial(s_tile, s_ov, wl)
    dev = latents.device
    acc = torch.zeros((b, 3, out_f, out_h, out_w), dtype=torch.float32, device=dev)
    wsum = torch.zeros((b, 1, out_f, out_h, out_w), dtype=torch.float32, device=dev)
    lib = _lib.load()
    emitted = 0

    def finalize(lo: int, hi: int) -> torch.Tensor:
        a = acc[:, :, lo:hi].contiguous()
        w = wsum[:, :, lo:hi].contiguous()
        o = torch.empty(a.shape, dtype=BF16, device=dev)
        check(lib.ltxk_tile_blend_finalize(_p(a), _p(w), _p(o), b, 3, (hi - lo) * out_h * out_w, _stream()),
              "ltxk_tile_blend_finalize")
        return o

    for ti in range(len(tiv.starts)):
        tsl, tmask = map_temporal_slice(tiv.starts[ti], tiv.ends[ti], tiv.left_ramps[ti], tiv.right_ramps[ti], temporal_scale)
        for hi_ in range(len(hiv.starts)):
            hsl, hmask = map_spatial_slice(hiv.starts[hi_], hiv.ends[hi_], hiv.left_ramps[hi_], hiv.right_ramps[hi_], spatial_scale)
            for wi in range(len(wiv.starts)):
                wsl, wmask = map_spatial_slice(wiv.starts[wi], wiv.ends[wi], wiv.left_ramps[wi], wiv.right_ramps[wi], spatial_scale)
                tile_lat = latents[:, :, tiv.starts[ti]:tiv.ends[ti], hiv.starts[hi_]:hiv.ends[hi_], wiv.starts[wi]:wiv.ends[wi]].contiguous()
                kw = {"noise_fn": noise_fn} if noise_fn is not None else {}
                tile = decoder_fn(tile_lat, causal=causal, timestep=timestep, debug=False, chunked_conv=chunked_conv, **kw)
                _, _, dt_, dh_, dw_ = tile.shape
                at, ah, aw = min(dt_, tsl.stop - tsl.start), min(dh_, hsl.stop - hsl.start), min(dw_, wsl.stop - wsl.start)
                mt, mh, mw = tmask[:at].to(dev), hmask[:ah].to(dev), wmask[:aw].to(dev)
                check(lib.ltxk_tile_blend_accum(_p(tile), dt_, dh_, dw_, at, ah, aw, _p(mt), _p(mh), _p(mw), _p(acc), _p(wsum),
                                                b, 3, out_f, out_h, out_w, tsl.start, hsl.start, wsl.start, _stream()),
                      "ltxk_tile_blend_accum")
        if on_frames_ready is not None and len(tiv.starts) > 1 and ti < len(tiv.starts) - 1:
            nxt = tiv.starts[ti + 1]
            nxt_out = 0 if nxt == 0 else 1 + (nxt - 1) * temporal_scale
            if nxt_out > emitted:
                on_frames_ready(finalize(emitted, nxt_out), emitted)
                emitted = nxt_out
    out = finalize(0, out_f)
    if on_frames_ready is not None and emitted < out_f:
        on_frames_ready(out[:, :, emitted:], emitted)
    return out


# ------------------------------------------------------------------------------------ decoder
class LTX2VideoDecoder:
    """decoder.py:237-531.  ``weights``: bf16 device tensors, keys as the reference's module tree
    after its key remap (decoder.py:544-591): conv_in.conv.{weight,bias}, up_blocks.{0,2,4,6}.
    res_blocks.{i}.conv{1,2}.conv.*, up_blocks.{1,3,5}.conv.*, conv_out.conv.*, latents_mean/std,
    and with timestep conditioning the time embedders + scale_shift tables.  Conv weights in the
    MLX layout (O,kD,kH,kW,I)."""

    def __init__(self, weights: Dict[str, torch.Tensor], timestep_conditioning: bool = False,
                 num_layers_per_block: int = 5, patch_size: int = 4):
        self.W = {k: v.contiguous() for k, v in weights.items()}
        self.timestep_conditioning = timestep_conditioning
        self.num_layers_per_block = num_layers_per_block
        self.patch_size = patch_size
        self.decode_noise_scale = 0.025
        self.decode_timestep = 0.05
        self.fuse_act = False          # PixelNorm + SiLU in the conv epilogues (conv_act_fusable): built, verified, slower
        need = ["conv_in.conv.weight", "conv_out.conv.weight", "up_blocks.1.conv.weight", "latents_mean", "latents_std"]
        missing = [k for k in need if k not in self.W]
        if missing:
            raise ValueError(f"Missing VAE decoder parameters: {missing}")
        self.latents_mean = self.W["latents_mean"]
        self.latents_std = self.W["latents_std"]

    @staticmethod
    def remap_decoder_key(key: str) -> str:
        """decoder.py:544-591: diffusers-style names -> this module tree."""
        parts = key.split(".")
        if len(parts) >= 4 and parts[0] == "mid_block" and parts[1] == "resnets":
            return ".".join(["up_blocks", "0", "res_blocks", parts[2]] + parts[3:])
        if len(parts) >= 3 and parts[0] == "up_blocks":
            try:
                b = int(parts[1])
            except ValueError:
                return key
            if len(parts) >= 4 and parts[2] == "resnets":
                return ".".join(["up_blocks", str(2 * b + 2), "res_blocks", parts[3]] + parts[4:])
            if len(parts) >= 5 and parts[2] == "upsamplers" and parts[3] == "0":
                return ".".join(["up_blocks", str(2 * b + 1)] + parts[4:])
        return key

    def _time_embed(self, t: torch.Tensor, prefix: str) -> torch.Tensor:
        W = self.W
        proj = ops.timestep_embed(t, 256, 1.0)
        h = ops.gemm(proj, W[f"{prefix}.timestep_embedder.linear_1.weight"], W[f"{prefix}.timestep_embedder.linear_1.bias"],
                     epilogue=ops.EPI_BIAS_SILU)
        return ops.gemm(h, W[f"{prefix}.timestep_embedder.linear_2.weight"], W[f"{prefix}.timestep_embedder.linear_2.bias"])

    def __call__(self, sample: torch.Tensor, causal: bool = False, timestep: Optional[torch.Tensor] = None,
                 debug: bool = False, chunked_conv: bool = False, noise: Optional[torch.Tensor] = None,
                 noise_fn: Optional[Callable] = None) -> torch.Tensor:
        """(B,128,F',H',W') -> (B,3,8(F'-1)+1,32H',32W') bf16.  ``chunked_conv`` is accepted for
        interface parity; the temporal chunking of convolution.py:168-222 / sampling.py:199-275 is
        an MLX memory workaround whose result equals the unchunked op.  A timestep-conditioned decoder needs
        ``noise`` (this call's tensor) or ``noise_fn(shape)`` (drawn here, as the reference does per call)."""
        W = self.W
        if sample.dim() != 5 or sample.shape[1] != 128:
            raise ValueError(f"latents must be (B,128,F,H,W), got {tuple(sample.shape)}")
        B, C, Fl, Hl, Wl = sample.shape
        x = sample.to(BF16).contiguous()
        tc = self.timestep_conditioning
        st = None
        if tc:
            if noise is None and noise_fn is not None:
                noise = noise_fn(tuple(sample.shape))
            if noise is None:
                raise ValueError("timestep-conditioned decode needs an explicit `noise` tensor (the reference draws "
                                 "mx.random.normal, decoder.py:381-385; MLX's stream is not reproducible here): "
                                 "pass noise= or noise_fn=")
            nz = noise.to(BF16).contiguous()
            tval = self.decode_timestep if timestep is None else float(timestep.reshape(-1)[0])
            st = torch.full((B,), tval * 1000.0, dtype=torch.float32, device=x.device).to(BF16)
        S = Fl * Hl * Wl
        xcl = torch.empty((B, Fl, Hl, Wl, C), dtype=BF16, device=x.device)
        check(_lib.load().ltxk_latent_denorm_cl(_p(x), _p(nz) if tc else None, float(self.decode_noise_scale) if tc else 0.0,
                                                _p(self.latents_mean), _p(self.latents_std), _p(xcl), B, C, S, _stream()),
              "ltxk_latent_denorm_cl")
        x = conv3d(xcl, W["conv_in.conv.weight"], W["conv_in.conv.bias"], causal, PAD_REFLECT)
        sc = sh = None                          # modulation of the last PixelNorm (before conv_out)
        if tc:
            emb = self._time_embed(st, "last_time_embedder")
            ada = ops.ada_combine(W["last_scale_shift_table"].reshape(1, 2, 128), emb, 1, B, 2, 128)[0]
            sh, sc = ada[:, 0].contiguous(), ada[:, 1].contiguous()
        h_final = None
        for bi in range(7):
            pre = f"up_blocks.{bi}"
            if bi % 2 == 0:
                c = x.shape[-1]
                mods = None
                if tc:
                    emb = self._time_embed(st, f"{pre}.time_embedder")                      # (B,4C)
                nl = self.num_layers_per_block
                all_mods = []
                for li in range(nl):
                    if tc:
                        ada = ops.ada_combine(W[f"{pre}.res_blocks.{li}.scale_shift_table"].reshape(1, 4, c), emb, 1, B, 4, c)[0]   # (B,4,C)
                        all_mods.append([ada[:, i].contiguous() for i in range(4)])          # shift1, scale1, shift2, scale2
                    else:
                        all_mods.append([None] * 4)
                # PixelNorm + SiLU ride in the epilogue of the conv that produces their input where the tile holds whole
                # rows (128 / 256 channels): conv1 then writes ONLY its normalised output, conv2 writes the residual
                # stream and, for the next block, its normalised copy.
                fuse = conv_act_fusable(c, x.numel() // c, force=self.fuse_act)    # off by default: measured slower, see conv_act_fusable
                h_next = None
                for li in range(nl):
                    rp = f"{pre}.res_blocks.{li}"
                    mods = all_mods[li]
                    h = h_next if h_next is not None else pixelnorm_act(x, 1e-8, True, mods[1], mods[0])
                    h_next = None
                    if fuse:
                        _, h = conv3d(h, W[f"{rp}.conv1.conv.weight"], W[f"{rp}.conv1.conv.bias"], causal, PAD_REFLECT,
                                      act=dict(eps=1e-8, silu=True, scale=mods[3], shift=mods[2]), keep_out=False)
                        if li + 1 < nl:
                            nm = all_mods[li + 1]
                            x, h_next = conv3d(h, W[f"{rp}.conv2.conv.weight"], W[f"{rp}.conv2.conv.bias"], causal, PAD_REFLECT, resid=x,
                                               act=dict(eps=1e-8, silu=True, scale=nm[1], shift=nm[0]))
                        elif bi == 6:            # the decoder's last res block: its successor is the PixelNorm in front of conv_out
                            x, h_final = conv3d(h, W[f"{rp}.conv2.conv.weight"], W[f"{rp}.conv2.conv.bias"], causal, PAD_REFLECT, resid=x,
                                                act=dict(eps=1e-8, silu=True, scale=sc, shift=sh))
                        else:
                            x = conv3d(h, W[f"{rp}.conv2.conv.weight"], W[f"{rp}.conv2.conv.bias"], causal, PAD_REFLECT, resid=x)
                    else:
                        h = conv3d(h, W[f"{rp}.conv1.conv.weight"], W[f"{rp}.conv1.conv.bias"], causal, PAD_REFLECT)
                        h = pixelnorm_act(h, 1e-8, True, mods[3], mods[2])
                        x = conv3d(h, W[f"{rp}.conv2.conv.weight"], W[f"{rp}.conv2.conv.bias"], causal, PAD_REFLECT, resid=x)
            else:
                cv = conv3d(x, W[f"{pre}.conv.weight"], W[f"{pre}.conv.bias"], causal, PAD_REFLECT)
                x = d2s_add(cv, x)
        x = h_final if h_final is not None else pixelnorm_act(x, 1e-8, True, sc, sh)
        y = conv3d(x, W["conv_out.conv.weight"], W["conv_out.conv.bias"], causal, PAD_REFLECT)
        _, Fo, Ho, Wo, _ = y.shape
        P = self.patch_size
        video = torch.empty((B, 3, Fo, Ho * P, Wo * P), dtype=BF16, device=y.device)
        check(_lib.load().ltxk_unpatchify_cf(_p(y), _p(video), B, Fo, Ho, Wo, 3, P, _stream()), "ltxk_unpatchify_cf")
        return video

    def decode_tiled(self, sample: torch.Tensor, tiling_config: Optional[TilingConfig] = None, tiling_mode: str = "auto",
                     causal: bool = False, timestep=None, debug: bool = False,
                     on_frames_ready: Optional[Callable] = None, noise_fn: Optional[Callable] = None) -> torch.Tensor:
        """decoder.py:452-531.  ``noise_fn``: see decode_with_tiling."""
        if tiling_config is None:
            tiling_config = TilingConfig.default()
        _, _, f, h, w = sample.shape
        need_s = need_t = False
        if tiling_config.spatial_config is not None:
            tl = tiling_config.spatial_config.tile_size_in_pixels // 32
            need_s = h > tl or w > tl
        if tiling_config.temporal_config is not None:
            need_t = f > tiling_config.temporal_config.tile_size_in_frames // 8
        if not need_s and not need_t:
            out = self(sample, causal=causal, timestep=timestep, noise_fn=noise_fn)
            if on_frames_ready is not None:
                try:
                    on_frames_ready(out, 0)
                except Exception:        # the reference swallows callback errors here (decoder.py:514-518)
                    pass
            return out
        return decode_with_tiling(self, sample, tiling_config, 32, 8, causal, timestep, False, on_frames_ready, noise_fn)


def to_uint8_frames(video: torch.Tensor) -> torch.Tensor:
    """generate.py:3894-3898: (B,3,F,H,W) bf16 -> (B,F,H,W,3) uint8."""
    B, C, F, H, W = video.shape
    out = torch.empty((B, F, H, W, C), dtype=torch.uint8, device=video.device)
    check(_lib.load().ltxk_to_uint8(_p(video.contiguous()), _p(out), B, C, F, H, W, _stream()), "ltxk_to_uint8")
    return out


# ------------------------------------------------------------------------------------ encoder
ENC_BLOCKS = [("res_x", 4), ("compress_space_res", (1, 2, 2)), ("res_x", 6), ("compress_time_res", (2, 1, 1)),
              ("res_x", 6), ("compress_all_res", (2, 2, 2)), ("res_x", 2), ("compress_all_res", (2, 2, 2)),
              ("res_x", 2)]       # encoder.py:95-105


class VideoEncoder:
    """video_vae.py:220-372 with the default block list of encoder.py:95-105: causal convs, zero
    spatial padding, PixelNorm(eps=1e-6), uniform log-variance head (only the 128 mean channels are
    computed: the 129th output channel never reaches the caller, video_vae.py:350-372)."""

    def __init__(self, weights: Dict[str, torch.Tensor], encoder_blocks=None, patch_size: int = 4):
        self.blocks = list(ENC_BLOCKS if encoder_blocks is None else encoder_blocks)
        self.patch_size = patch_size
        W = {k: v.contiguous() for k, v in weights.items()}
        for k in ("conv_in.weight", "conv_out.weight", "per_channel_statistics.mean", "per_channel_statistics.std"):
            if k not in W:
                raise ValueError(f"Missing VAE encoder parameter: {k}")
        # first conv: pad the 48 patchified channels to 64 with zero weights (same sums)
        w_in = W["conv_in.weight"]
        if w_in.shape[-1] % 64 != 0:
            cpad = (w_in.shape[-1] + 63) // 64 * 64
            wp = torch.zeros(w_in.shape[:-1] + (cpad,), dtype=BF16, device=w_in.device)
            wp[..., : w_in.shape[-1]] = w_in
            W["conv_in.weight"] = wp
        W["conv_out.weight_means"] = W["conv_out.weight"][:128].contiguous()
        W["conv_out.bias_means"] = W["conv_out.bias"][:128].contiguous()
        self.W = W

    def __call__(self, sample: torch.Tensor) -> torch.Tensor:
        W = self.W
        B, C, F, H, Wd = sample.shape
        if (F - 1) % 8 != 0:
            raise ValueError("Invalid number of frames: Encode input must have 1 + 8 * x frames "
                             f"(e.g., 1, 9, 17, ...). Got {F} frames.")
        P = self.patch_size
        lib = _lib.load()
        cpad = W["conv_in.weight"].shape[-1]
        x = torch.empty((B, F, H // P, Wd // P, cpad), dtype=BF16, device=sample.device)
        check(lib.ltxk_patchify_cl(_p(sample.to(BF16).contiguous()), _p(x), B, C, F, H, Wd, P, cpad, _stream()), "ltxk_patchify_cl")
        x = conv3d(x, W["conv_in.weight"], W["conv_in.bias"], True, PAD_ZEROS)
        for bi, (kind, arg) in enumerate(self.blocks):
            pre = f"down_blocks.{bi}"
            if kind == "res_x":
                for li in range(arg):
                    rp = f"{pre}.res_blocks.{li}"
                    h = pixelnorm_act(x, 1e-6, True)
                    h = conv3d(h, W[f"{rp}.conv1.weight"], W[f"{rp}.conv1.bias"], True, PAD_ZEROS)
                    h = pixelnorm_act(h, 1e-6, True)
                    x = conv3d(h, W[f"{rp}.conv2.weight"], W[f"{rp}.conv2.bias"], True, PAD_ZEROS, resid=x)
            else:
                st, sh, sw = arg
                if st == 2:
                    x = torch.cat([x[:, :1], x], dim=1)                      # duplicate first frame (sampling.py:78-81)
                Bx, Dp, Hp, Wp, Cx = x.shape
                if Dp % st or Hp % sh or Wp % sw:
                    raise ValueError(f"encoder volume {Dp}x{Hp}x{Wp} not divisible by stride {arg}")
                cv = conv3d(x, W[f"{pre}.conv.weight"], W[f"{pre}.conv.bias"], True, PAD_ZEROS)
                Cc = cv.shape[-1]
                out = torch.empty((Bx, Dp // st, Hp // sh, Wp // sw, Cc * st * sh * sw), dtype=BF16, device=x.device)
                check(lib.ltxk_s2d_skip(_p(cv), _p(x), _p(out), Bx, Dp, Hp, Wp, Cc, Cx, st, sh, sw, Cx // Cc, _stream()),
                      "ltxk_s2d_skip")
                x = out
        x = pixelnorm_act(x, 1e-6, True)
        y = conv3d(x, W["conv_out.weight_means"], W["conv_out.bias_means"], True, PAD_ZEROS)
        Bx, Dl, Hl, Wl, _ = y.shape
        lat = torch.empty((Bx, 128, Dl, Hl, Wl), dtype=BF16, device=y.device)
        check(lib.ltxk_latent_norm_cf(_p(y), 128, _p(W["per_channel_statistics.mean"]), _p(W["per_channel_statistics.std"]),
                                      _p(lat), Bx, 128, Dl * Hl * Wl, _stream()), "ltxk_latent_norm_cf")
        return lat


# ------------------------------------------------------------------------------------ synthetic weights / bench / smoke
def random_decoder_weights(dev, seed: int = 1234, timestep_conditioning: bool = False, layers: int = 5) -> Dict[str, torch.Tensor]:
    g = torch.Generator(device=dev).manual_seed(seed)
    W: Dict[str, torch.Tensor] = {}

    def conv(name, o, i):
        W[f"{name}.weight"] = (torch.randn((o, 3, 3, 3, i), generator=g, device=dev) / math.sqrt(27 * i)).to(BF16)
        W[f"{name}.bias"] = (torch.randn((o,), generator=g, device=dev) * 0.01).to(BF16)

    W["latents_mean"] = torch.zeros(128, dtype=BF16, device=dev)
    W["latents_std"] = torch.ones(128, dtype=BF16, device=dev)
    conv("conv_in.conv", 1024, 128)
    for bi in range(7):
        c = DEC_CH[bi // 2]
        if bi % 2 == 0:
            for li in range(layers):
                conv(f"up_blocks.{bi}.res_blocks.{li}.conv1.conv", c, c)
                conv(f"up_blocks.{bi}.res_blocks.{li}.conv2.conv", c, c)
        else:
            conv(f"up_blocks.{bi}.conv", c * 4, c)
    conv("conv_out.conv", 48, 128)
    return W


def random_encoder_weights(dev, seed: int = 4321, blocks=None) -> Dict[str, torch.Tensor]:
    """Random weights of the encoder architecture (encoder.py:95-105), generated on the device (synthetic runs)."""
    blocks = ENC_BLOCKS if blocks is None else blocks
    g = torch.Generator(device=dev).manual_seed(seed)
    W: Dict[str, torch.Tensor] = {}

    def conv(name, o, i):
        W[f"{name}.weight"] = (torch.randn((o, 3, 3, 3, i), generator=g, device=dev) / math.sqrt(27 * i)).to(BF16)
        W[f"{name}.bias"] = (torch.randn((o,), generator=g, device=dev) * 0.01).to(BF16)

    conv("conv_in", 128, 48)
    ch = 128
    for bi, (kind, arg) in enumerate(blocks):
        pre = f"down_blocks.{bi}"
        if kind == "res_x":
            for li in range(arg):
                conv(f"{pre}.res_blocks.{li}.conv1", ch, ch)
                conv(f"{pre}.res_blocks.{li}.conv2", ch, ch)
        else:
            conv(f"{pre}.conv", ch * 2 // (arg[0] * arg[1] * arg[2]), ch)
            ch *= 2
    conv("conv_out", 129, ch)
    W["per_channel_statistics.mean"] = torch.zeros(128, dtype=BF16, device=dev)
    W["per_channel_statistics.std"] = torch.ones(128, dtype=BF16, device=dev)
    return W


def decode_flops(Fl: int, Hl: int, Wl: int, layers: int = 5) -> float:
    """2*27*Cin*Cout*voxels summed over the decoder's convolutions."""
    d, h, w = Fl, Hl, Wl
    total = 2.0 * 27 * 128 * 1024 * d * h * w
    for bi in range(7):
        c = DEC_CH[bi // 2]
        if bi % 2 == 0:
            total += 2 * layers * 2.0 * 27 * c * c * d * h * w
        else:
            total += 2.0 * 27 * c * (4 * c) * d * h * w
            d, h, w = 2 * d - 1, 2 * h, 2 * w
    total += 2.0 * 27 * 128 * 48 * d * h * w
    return total


def bench_decode(dev, Fl: int, Hl: int, Wl: int, iters: int = 3) -> dict:
    """VAE-decode leg of bench.py: frames/s of the non-tiled decode (TilingConfig.auto is None at
    512x512x33), latents resident in HBM."""
    import time
    dec = LTX2VideoDecoder(random_decoder_weights(dev))
    g = torch.Generator(device=dev).manual_seed(7)
    lat = torch.randn((1, 128, Fl, Hl, Wl), generator=g, device=dev).to(BF16)
    dec(lat)
    torch.cuda.synchronize()
    # clean pass: no per-launch events inside the timed region (this is the reported frames/s)
    prev, ops.TIMER = ops.TIMER, None
    t0 = time.perf_counter()
    for _ in range(iters):
        v = dec(lat)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    # instrumented pass: a HIP event pair around every launch, for the per-family roofline only
    ops.TIMER = ops.KernelTimer()
    for _ in range(iters):
        dec(lat)
    fams = ops.TIMER.summary()
    ops.TIMER = prev
    frames = v.shape[2]
    out = {"vae_decode_fps": frames / dt, "vae_decode_ms": dt * 1e3, "vae_decode_tflop": decode_flops(Fl, Hl, Wl) / 1e12,
           "vae_decode_tflops_achieved": decode_flops(Fl, Hl, Wl) / dt / 1e12}
    if "conv3d_k3" in fams:
        cf = fams["conv3d_k3"]
        out["vae_conv3d_tflops"] = cf["flops"] / (cf["ms"] * 1e-3) / 1e12
        out["vae_conv3d_launches"] = cf["launches"] // iters
        out["vae_kernel_breakdown_ms"] = {k: v_["ms"] / iters for k, v_ in fams.items()}
    if "pixelnorm_act" in fams and fams["pixelnorm_act"]["ms"] > 0:          # the decoder's HBM-bound kernel (r + w of the volume)
        pn = fams["pixelnorm_act"]
        out["vae_pixelnorm_GBs"] = pn["bytes"] / (pn["ms"] * 1e-3) / 1e9
        out["vae_pixelnorm_launches"] = pn["launches"] // iters
    return out

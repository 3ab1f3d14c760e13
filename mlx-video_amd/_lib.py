"""ctypes binding of libltxk.so (include/ltxk.h).  Fails loudly when the HIP extension is
missing: there is no CPU fallback on the product path."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LTXK_LIB", os.path.join(_HERE, "libltxk.so"))


class LtxkError(RuntimeError):
    pass


class GemmArgs(Structure):
    _fields_ = [
        ("A", c_void_p), ("W", c_void_p), ("bias", c_void_p), ("out", c_void_p),
        ("resid", c_void_p), ("gate", c_void_p), ("gate_row", c_void_p),
        ("M", c_int32), ("N", c_int32), ("K", c_int32),
        ("lda", c_int32), ("ldo", c_int32), ("ldr", c_int32), ("gate_stride", c_int32),
        ("epilogue", c_int32), ("out_tokens_per_batch", c_int32), ("alpha", c_float),
        ("out2", c_void_p), ("n_split", c_int32), ("ldo2", c_int32), ("sumsq", c_void_p), ("sumsq_ld", c_int32),
        ("workspace", c_void_p), ("workspace_bytes", c_int64),
    ]


class AttnArgs(Structure):
    _fields_ = [
        ("q", c_void_p), ("k", c_void_p), ("vt", c_void_p), ("out", c_void_p),
        ("ldq", c_int32), ("ldk", c_int32), ("ldvt", c_int32), ("ldo", c_int32),
        ("B", c_int32), ("H", c_int32), ("Tq", c_int32), ("Tk", c_int32), ("scale", c_float),
        ("q_sumsq", c_void_p), ("q_sumsq_ld", c_int32), ("q_sumsq_n", c_int32),
        ("q_norm_weight", c_void_p), ("cos", c_void_p), ("sin", c_void_p), ("eps", c_float), ("flags", c_int32),
    ]


class Conv3dArgs(Structure):
    _fields_ = [
        ("x", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("out", c_void_p),
        ("resid", c_void_p), ("zero_page", c_void_p),
        ("B", c_int32), ("D", c_int32), ("H", c_int32), ("W", c_int32),
        ("Cin", c_int32), ("Cout", c_int32),
        ("causal", c_int32), ("pad_mode", c_int32),
        ("workspace", c_void_p), ("workspace_bytes", c_int64), ("taps_d", c_int32),
        ("act_out", c_void_p), ("act_scale", c_void_p), ("act_shift", c_void_p), ("act_eps", c_float), ("act_silu", c_int32),
    ]


# name -> (restype, argtypes); mirrors include/ltxk.h one to one.
SIGNATURES = {
    "ltxk_version": (c_int32, []),
    "ltxk_last_error": (c_char_p, []),
    "ltxk_abi_sizeof": (c_int32, [c_int32]),
    "ltxk_gemm_bf16": (c_int32, [POINTER(GemmArgs), c_void_p]),
    "ltxk_flash_attn": (c_int32, [POINTER(AttnArgs), c_void_p]),
    "ltxk_flash_attn_bf16": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32,
                                       c_int32, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "ltxk_rmsnorm_modulate": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p, c_void_p, c_int32,
                                        c_void_p, c_void_p]),
    "ltxk_layernorm_modulate": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p, c_void_p, c_int32,
                                          c_void_p, c_void_p]),
    "ltxk_qknorm_rope": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                   c_int32, c_int32, c_float, c_void_p]),
    "ltxk_timestep_embed": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p]),
    "ltxk_rope_table": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                  POINTER(c_float), c_void_p]),
    "ltxk_ada_combine": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_uint32, c_void_p]),
    "ltxk_rmsnorm_modulate_ss": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p, c_int32, c_int32, c_void_p,
                                           c_void_p, c_int32, c_void_p, c_int32, c_void_p]),
    "ltxk_qknorm_rope_ss": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                      c_int32, c_int32, c_float, c_void_p, c_int32, c_void_p]),
    "ltxk_silu": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    "ltxk_latent_to_tokens": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "ltxk_conv3d_k3_bf16": (c_int32, [POINTER(Conv3dArgs), c_void_p]),
    "ltxk_pixelnorm_act": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p, c_void_p, c_int64,
                                     c_int32, c_void_p]),
    "ltxk_d2s_add": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                               c_void_p]),
    "ltxk_s2d_skip": (c_int32, [c_void_p, c_void_p, c_void_p] + [c_int32] * 10 + [c_void_p]),
    "ltxk_latent_denorm_cl": (c_int32, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int64,
                                        c_void_p]),
    "ltxk_latent_norm_cf": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int64,
                                      c_void_p]),
    "ltxk_unpatchify_cf": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                     c_void_p]),
    "ltxk_patchify_cl": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                   c_void_p]),
    "ltxk_to_uint8": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "ltxk_resize_area": (c_int32, [c_void_p, c_int32, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "ltxk_groupnorm_act": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32,
                                     c_float, c_int32, c_void_p]),
    "ltxk_tile_blend_accum": (c_int32, [c_void_p] + [c_int32] * 6 + [c_void_p] * 5 + [c_int32] * 8 + [c_void_p]),
    "ltxk_tile_blend_finalize": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int64, c_void_p]),
    "ltxk_step_scalars": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "ltxk_euler_step": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_void_p]),
    "ltxk_cfg_euler_step_dev": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                          c_int32, c_float, c_void_p, c_int32, c_void_p]),
    "ltxk_cfg_euler_step": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                      c_int32, c_float, c_float, c_float, c_int32, c_void_p]),
}

_lib = None
AB_LIB_PATH = os.path.join(_HERE, "libltxk_ab.so")
ATTN_NO_TAIL_SPLIT = 1        # ltxk.h: LTXK_ATTN_NO_TAIL_SPLIT


def _open(path: str) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise LtxkError(
            f"{path} is missing: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C mlx-video_amd/csrc`). "
            "There is no CPU fallback for the product path.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    for which, st in enumerate((GemmArgs, Conv3dArgs, AttnArgs)):
        if lib.ltxk_abi_sizeof(which) != ctypes.sizeof(st):
            raise LtxkError(f"{path} is stale: sizeof({st.__name__}) is {lib.ltxk_abi_sizeof(which)} in the library, "
                            f"{ctypes.sizeof(st)} in this binding; rebuild it (make -C mlx-video_amd/csrc)")
    return lib


def load() -> ctypes.CDLL:
    """Load libltxk.so; raise LtxkError with the build recipe if it is not there."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


class use_library:
    """``with use_library(AB_LIB_PATH):`` routes every ops call inside the block to another build of the same ABI - the
    -DLTXK_AB measurement build, whose launch-form switches are environment variables (csrc/common.h).  For scripts/ and
    for the tests that compare two launch forms of one kernel bit for bit; the product path never enters it."""

    def __init__(self, path: str = AB_LIB_PATH):
        self.lib = _open(path)

    def __enter__(self):
        global _lib
        self.prev, _lib = _lib, self.lib
        return self.lib

    def __exit__(self, *a):
        global _lib
        _lib = self.prev


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().ltxk_last_error()
        raise LtxkError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")

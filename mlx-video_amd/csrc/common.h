// Shared device/host helpers for libltxk (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ltxk.h"

typedef __bf16 bf16;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LTXK_WAVE 64

// A/B switches.  The product library (libltxk.so) reads NO environment variable and keeps no mutable global state
// (include/ltxk.h, Conventions): LTXK_AB_INT(name, default) is the constant `default` there.  The measurement build
// (`make ab` -> libltxk_ab.so, -DLTXK_AB; loaded only by scripts/ and by the tests that compare two launch forms of one
// kernel bit for bit, through mlx_video_amd._lib.use_library) reads the named variable on every call.
#ifdef LTXK_AB
#include <stdlib.h>
#define LTXK_AB_INT(name, dflt) ([] { const char* e__ = getenv(name); return e__ ? atoi(e__) : (dflt); }())
#else
#define LTXK_AB_INT(name, dflt) (dflt)
#endif

// Thread-local error string (never throws across the ABI).
void ltxk_set_error(const char* fmt, ...);

#define LTXK_CHECK_ARG(cond, ...)                  \
  do {                                             \
    if (!(cond)) {                                 \
      ltxk_set_error(__VA_ARGS__);                 \
      return LTXK_EINVAL;                          \
    }                                              \
  } while (0)

#define LTXK_CHECK_LAUNCH(name)                                                  \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      ltxk_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
      return LTXK_ELAUNCH;                                                       \
    }                                                                            \
  } while (0)

// Round fp32 to bf16 storage and back: the "materialise a bf16 array" point of the reference.
// Written as an opaque v_cvt_pk_bf16_f32 (RNE): with the plain cast pair hipcc (ROCm 7.2,
// -ffp-contract=fast) narrows fptrunc(fmul(fpext a, fpext b)) to a bf16 multiply, re-promotes
// it and then contracts it with the following add into one v_fmac_f32 — silently dropping the
// rounding of the product.
__device__ __forceinline__ float rbf(float x) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(r) : "v"(x));
  return __uint_as_float(r << 16);
}

__device__ __forceinline__ float gelu_tanh_f(float x) {
  // nn.gelu_approx: 0.5x(1+tanh(u)), u = sqrt(2/pi)(x+0.044715x^3).  0.5(1+tanh u) = 1/(1+exp(-2u)), so
  // gelu = x * rcp(1 + exp2(x*(c1 + c2*x^2))) with c1 = -2*sqrt(2/pi)*log2(e), c2 = 0.044715*c1:
  // 7 VALU ops (v_exp_f32 + v_rcp_f32, 1 ulp each — far below the bf16 the result is rounded to) instead of
  // the 14 of the textbook form; the FF1 GEMM epilogue is VALU-bound.
  const float t = __builtin_fmaf(-0.10294324f, x * x, -2.3022082f) * x;
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}

__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

// LDS-DMA of one 1-KiB piece: lane i's 16 bytes at gsrc land at lds_dst + 16*i (lds_dst wave-uniform).
// Issued as inline asm on purpose: hipcc (ROCm 7.2) models the builtin as a FLAT access that may
// touch LDS, which degrades every later ds_read wait in the loop to `s_waitcnt lgkmcnt(0)` — the
// fragment prefetch two MFMA groups ahead would then be drained at each use.  Hidden in asm, the
// compiler emits counted lgkmcnt waits for its ds_reads; the DMA itself is retired by the counted
// vmcnt in wait_stage_and_barrier().  M0 (the DMA's LDS base) is written and restored inside the
// one statement that uses it.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_dst;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}

// The same with the source split into a wave-uniform 64-bit base (SGPR pair) and a per-lane 32-bit byte offset:
// loops that walk a tensor tile by tile advance the base with scalar adds and keep the lane offsets constant.
__device__ __forceinline__ void glds16_s(uint64_t base, unsigned lane_off, void* lds_dst) {
  const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_dst;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(lane_off), "s"(base), "s"(dst)
               : "memory");
}

// Cross-lane exchanges as single VALU instructions (gfx950: v_permlane32_swap / v_permlane16_swap) instead of the
// ds_bpermute_b32 that __shfl_xor(x, 32 | 16) compiles to - an LDS instruction with an lgkmcnt wait on the critical path
// of whatever consumes it.  With both operands a copy of x, after the swap the pair holds (x of this lane, x of lane ^ 32)
// - resp. lane ^ 16 - in some order, in every lane.
__device__ __forceinline__ void lane_xor32_pair(float x, float& a, float& b) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void lane_xor16_pair(float x, float& a, float& b) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
// v_max_f32 without the canonicalising self-max hipcc puts in front of fmaxf's operands (inputs here are never NaN)
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float lane_xor32_max(float x) { float a, b; lane_xor32_pair(x, a, b); return vmax(a, b); }
__device__ __forceinline__ float lane_xor16_max(float x) { float a, b; lane_xor16_pair(x, a, b); return vmax(a, b); }
__device__ __forceinline__ float lane_xor32_sum(float x) { float a, b; lane_xor32_pair(x, a, b); return a + b; }
__device__ __forceinline__ float lane_xor16_sum(float x) { float a, b; lane_xor16_pair(x, a, b); return a + b; }

// v + (v of lane ^ OFF) without an LDS round trip: OFF = 32 / 16 by v_permlane32/16_swap, OFF = 8, 4, 2, 1 by a DPP row rotate
// (row_ror:OFF adds the lane OFF places up in its row of 16; once the value is symmetric under the larger offsets - as it is
// in a butterfly that runs 32, 16, 8, 4, 2, 1 - that lane holds the same number as lane ^ OFF).  Same operands, same single add
// as `v += __shfl_xor(v, OFF, 64)`, which hipcc lowers to ds_bpermute_b32 + s_waitcnt lgkmcnt(0): bit-identical results.
template <int OFF>
__device__ __forceinline__ float lane_butterfly_add(float v) {
  if constexpr (OFF == 32) return lane_xor32_sum(v);
  else if constexpr (OFF == 16) return lane_xor16_sum(v);
  else {
    static_assert(OFF == 8 || OFF == 4 || OFF == 2 || OFF == 1, "butterfly offsets are powers of two up to 32");
    const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + OFF, 0xf, 0xf, false);      // row_ror:OFF
    return v + __int_as_float(r);
  }
}

// Sum over the wave, the same value (bit for bit) in every lane: butterfly 32, 16, 8, 4, 2, 1.
__device__ __forceinline__ float wave_sum(float v) {
  v = lane_butterfly_add<32>(v);
  v = lane_butterfly_add<16>(v);
  v = lane_butterfly_add<8>(v);
  v = lane_butterfly_add<4>(v);
  v = lane_butterfly_add<2>(v);
  v = lane_butterfly_add<1>(v);
  return v;
}

// Sum over aligned groups of LPR lanes, butterfly from LPR/2 down.  LPR = 16, 32 or 64 take the single-instruction exchanges
// above.  The DPP row rotate equals lane ^ OFF only once the value is symmetric under every larger offset inside the row of
// 16, which a butterfly that STARTS below 8 never establishes (LPR = 8: row_ror:4 would mix the row's two groups), so
// narrower groups keep the plain __shfl_xor exchange.
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
  static_assert(LPR == 1 || LPR == 2 || LPR == 4 || LPR == 8 || LPR == 16 || LPR == 32 || LPR == 64, "group width");
  if constexpr (LPR >= 16) {
    if constexpr (LPR >= 64) v = lane_butterfly_add<32>(v);
    if constexpr (LPR >= 32) v = lane_butterfly_add<16>(v);
    v = lane_butterfly_add<8>(v);
    v = lane_butterfly_add<4>(v);
    v = lane_butterfly_add<2>(v);
    v = lane_butterfly_add<1>(v);
  } else {
#pragma unroll
    for (int off = LPR / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  }
  return v;
}

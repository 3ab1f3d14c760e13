// MFMA GEMM main loop shared by the dense (Linear) and implicit (conv3d) kernels.
//
//   C[m,n] = sum_k A[m,k] * W[n,k]          A rows and W rows are both k-contiguous bf16
//
// Geometry (gfx950, wave64): one workgroup = 8 waves = WM (rows) x WN (cols), WM*WN = 8;
// tile BM x BN x 64 with BM = 16*TT*WM, BN = 16*NT*WN; per wave TT x NT MFMA tiles of
// v_mfma_f32_16x16x32_bf16.  (TT=5, WN=4, NT=4): 160x256, the Linear layers at M=2560;
// (TT=4, WN=2, NT=4): 256x128, the 128-channel convolutions; (TT=5, WN=4, NT=2): 160x128 with 80x32 per wave, the
// Linear layers whose 256-column tiling leaves CUs idle (M=1280, N=4096: exactly 256 tiles; round 4).
// Staging: global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip) into a 3-deep LDS ring; one
// raw s_barrier per K-step, counted vmcnt so that two K-steps stay in flight across it.
// LDS image: 128-byte rows ([row][64 bf16]) with the 16-byte chunk index XOR-swizzled by
// (row & 7): the DMA destination is lane-linear, so the permutation is applied to the per-lane
// SOURCE address and again on the ds_read_b128 fragment reads (conflict-free for the 16x16x32
// operand map).
#pragma once
#include "common.h"

namespace ltxk {

template <int V> struct IntC { static constexpr int value = V; };

// A global pointer the optimiser cannot see through (an empty asm: no instruction, no wait): `load(c ? a : dummy)` used
// only when c holds otherwise gets unfolded into a load under `if (c)`, which breaks a counted-vmcnt prologue that
// assumes a fixed number of loads per lane (gemm.hip).
// (The pointer comes back as an explicit global-address-space pointer: a generic one would turn the load into flat_load,
// which hipcc waits for with vmcnt(0).)
template <class T>
__device__ __forceinline__ const __attribute__((address_space(1))) T* opaque_gptr(const T* p) {
  uintptr_t u = (uintptr_t)p;
  asm volatile("" : "+v"(u));
  return (const __attribute__((address_space(1))) T*)u;
}

constexpr int GEMM_BK = 64;
constexpr int GEMM_THREADS = 512;

template <int TT, int WN, int NT = 4>
struct GemmGeom {
  static constexpr int WM = 8 / WN;
  static constexpr int BM = 16 * TT * WM;
  static constexpr int BN = 16 * NT * WN;
  static_assert(BN % 64 == 0, "whole 1-KiB W pieces per wave");
  static constexpr int W_PIECES = BN / 8;          // 1 KiB pieces (8 rows x 128 B)
  static constexpr int W_PER_WAVE = W_PIECES / 8;
  static constexpr int W_STAGE_BYTES = BN * GEMM_BK * 2;
  static constexpr int A_PIECES = BM / 8;
  static constexpr int A_BASE = A_PIECES / 8;      // pieces per wave (floor)
  static constexpr int A_REM = A_PIECES % 8;       // first A_REM waves take one more
  static constexpr int MAXA = A_BASE + (A_REM ? 1 : 0);
  static constexpr int STAGE_BYTES = W_STAGE_BYTES + BM * GEMM_BK * 2;
  static constexpr int STAGES = 3;
  static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS ring exceeds 160 KiB");
};

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Each XCD owns a band
// of CT/8 column tiles and walks it in patches of 8 row tiles x 4 column tiles per 32-workgroup round: the round then
// pulls 8 A panels + 4 W panels through that XCD's L2 (against 16 + 2 for "all row tiles x 2 columns": -25 % of the
// L2->fabric reads, and the chip is power-limited in these launches - the first form of this patch order was worth
// < 1 % on FF1 in round 1, the 16x16 case below 0.8 % of the whole block in round 2).  Column bands that are not a
// multiple of 4 wide (q|k|v: 6) finish with a 2-column strip.  Placement affects speed only.
__device__ __forceinline__ void map_tile(int bid, int RT, int CT, int& rt, int& ct) {
  if ((CT & 15) == 0) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int cpx = CT >> 3;                 // column tiles owned by this XCD
    if (cpx == 2 && RT == 16) {
      // 16 x 16 tiles = ONE round (FF2, out, q2, o2 at M=2560, N=4096): two XCDs share a 4-column group, one row half each
      rt = (xcd & 1) * 8 + (idx >> 2);
      ct = (xcd >> 1) * 4 + (idx & 3);
    } else if ((RT & 7) == 0) {
      const int full = (cpx >> 2) * 4 * RT;  // tiles of the band's whole 4-column groups
      if (idx < full) {
        const int per_cg = 4 * RT;           // workgroups per column group (all row tiles)
        const int cg = idx / per_cg, j = idx - cg * per_cg;
        const int rg = j >> 5, k = j & 31;   // row group of 8, position inside the 8x4 patch
        rt = rg * 8 + (k >> 2);
        ct = xcd * cpx + cg * 4 + (k & 3);
      } else {
        const int remc = cpx & 3, j = idx - full;      // the remaining 1-3 columns, all row tiles
        rt = j / remc;
        ct = xcd * cpx + (cpx >> 2) * 4 + (j - rt * remc);
      }
    } else {
      const int pair = idx / (2 * RT), j = idx - pair * 2 * RT;
      rt = j >> 1;
      ct = xcd * cpx + pair * 2 + (j & 1);
    }
  } else {
    rt = bid % RT;
    ct = bid / RT;
  }
}

// Wait until this wave's loads of the current K-step have landed, leaving `keep` younger
// LDS-DMA instructions (the next K-step's) in flight, then rendezvous.
template <int KEEP>
__device__ __forceinline__ void wait_keep_and_barrier() {
  static_assert(KEEP >= 0 && KEEP <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(KEEP) : "memory");
}

__device__ __forceinline__ void wait_stage_and_barrier(int keep) {
  switch (keep) {
    case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
  }
}

// Operand roles.  SWAP=false: acc tile = D[n (4 regs)][row (lane&15)]  (W is the MFMA A operand), so a lane
// holds 4 consecutive output columns of one row: 8-byte row-major stores.  SWAP=true: acc tile =
// D[row (4 regs)][n (lane&15)]: 4 consecutive rows of one column, used for the transposed (V^T) output.
// Software-pipelined K-step: the 2*TT MFMA groups (4 MFMAs each: one A-row fragment x 4 W
// fragments) run in a fixed order pinned by sched_barrier; fragment ds_read_b128s are issued two
// groups ahead of their use, and this wave's LDS-DMA pieces of the stage two K-steps ahead are
// sprinkled one (or PPG) per group instead of being issued as a burst in front of the MFMAs — the
// DMA issue cost (~60-180 cycles per 1-KiB piece) then hides under the matrix pipe.
// issue(i): launch this wave's i-th LDS-DMA piece of the prefetched stage (no-op for i >= count).
template <int TT, int WN, bool SWAP, int NT = 4, class IssueFn>
__device__ __forceinline__ void mma_stage_pipelined(const char* st, int wm, int wn, int lane,
                                                    f32x4 (&acc)[TT][NT], IssueFn&& issue) {
  using G = GemmGeom<TT, WN, NT>;
  constexpr int NG = 2 * TT;                                   // MFMA groups per K-step
  constexpr int MAXP = G::W_PER_WAVE + G::MAXA;                // LDS-DMA pieces per wave per stage
  constexpr int PPG = (MAXP + NG - 1) / NG;
  const char* wb = st + (wn * 16 * NT) * 128 + (lane & 15) * 128;
  const char* ab = st + G::W_STAGE_BYTES + (wm * TT * 16) * 128 + (lane & 15) * 128;
  const int koff0 = (((lane >> 4)) ^ (lane & 7)) << 4;
  const int koff1 = (((4 + (lane >> 4))) ^ (lane & 7)) << 4;
  bf16x8 wf[2][NT], af[2][TT];
  // flat read order: W0[0..NT-1], A0[0..TT-1], W1[0..NT-1], A1[0..TT-1]
  auto rd = [&](int idx) {
    // idx is a compile-time constant after unrolling
    const int ks = idx / (NT + TT), r = idx % (NT + TT);
    const int ko = ks ? koff1 : koff0;
    if (r < NT) wf[ks][r] = *(const bf16x8*)(wb + r * 2048 + ko);
    else af[ks][r - NT] = *(const bf16x8*)(ab + (r - NT) * 2048 + ko);
  };
  constexpr int TOTAL = 2 * (NT + TT);
  auto need = [](int g) { return (g / TT) * (NT + TT) + NT + (g % TT) + 1; };
  int issued = 0;
#pragma unroll
  for (int i = 0; i < TOTAL; ++i)
    if (i < need(1 < NG ? 1 : 0)) { rd(i); issued = i + 1; }
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int target = need(g + 2 < NG ? g + 2 : NG - 1);
#pragma unroll
    for (int i = 0; i < TOTAL; ++i)
      if (i >= issued && i < target) rd(i);
    issued = target > issued ? target : issued;
#pragma unroll
    for (int q = 0; q < PPG; ++q) issue(g * PPG + q);
    const int ks = g / TT, tt = g % TT;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      acc[tt][nt] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][tt], wf[ks][nt], acc[tt][nt], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][nt], af[ks][tt], acc[tt][nt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Rotated software pipeline: like mma_stage_pipelined, but the last two MFMA groups of every
// K-step are deferred until AFTER the next K-step's barrier (their operands are already in
// registers), so they execute while the first fragment reads of the new stage are in flight: the
// LDS cold start after each barrier (all 8 waves reading at once) no longer idles the matrix pipe.
// Requires TT >= 2 (both deferred groups then share the ks=1 W fragments).
// Fragment reads run this many MFMA groups ahead of their use.  1, not 2 (round 2, interleaved A/B on one box,
// profiles/r02_gemm_prefetch_depth_ab.log): the N=4096 launches 85.5 -> 79.7 us (bias) / 89.7 -> 85.2 us (gate+residual),
// FF2 299 -> 290 us, the whole block -1.1 %, the VAE decode -0.7 %; 3 is slower than 2.  With two waves per SIMD the
// other wave covers an LDS read that is one group (64 matrix-pipe cycles) ahead, and the shorter lead leaves fewer
// reads queued in front of each barrier.
#ifndef LTXK_PREFETCH_GROUPS
#define LTXK_PREFETCH_GROUPS 1
#endif
// DG = number of trailing MFMA groups of a K-step that are deferred past the next barrier (2 <= DG <= TT:
// all deferred groups use the ks=1 W fragments).  DG=2 hides the LDS cold start; DG=TT additionally puts a
// wave half a K-step out of phase with a DG=2 wave — used for waves 4-7, the SIMD partners of waves 0-3, so
// that the two waves of a SIMD do not hit their LDS-read bursts, DMA issues and MFMA-dense stretches together.
template <int TT, int WN, bool SWAP, int DG = 2, int NT = 4>
struct MmaPipe {
  using G = GemmGeom<TT, WN, NT>;
  static constexpr int PD = LTXK_PREFETCH_GROUPS;   // fragment reads run PD MFMA groups ahead of their use
  static constexpr int NG = 2 * TT;
  static constexpr int MAXP = G::W_PER_WAVE + G::MAXA;
#ifdef LTXK_DMA_PER_GROUP
  static constexpr int PPG = LTXK_DMA_PER_GROUP;                // A/B: bunch the stage's LDS-DMA pieces into the first groups
#else
  static constexpr int PPG = (MAXP + NG - 1) / NG;
#endif
  static constexpr int TOTAL = 2 * (NT + TT);
  static_assert(TT >= 2 && DG >= 2 && DG <= TT, "deferred groups must all lie in the ks=1 half");
  // fragment registers persist across K-steps: after step() wf[1][*] and af[1][TT-DG..TT-1] hold the operands
  // of the deferred groups; the next step() consumes them before overwriting them.
  bf16x8 wf[2][NT], af[2][TT];

  __device__ __forceinline__ void init() {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) wf[1][i][j] = (bf16)0.f;
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) af[1][i][j] = (bf16)0.f;
  }

  static __device__ __forceinline__ void group(const bf16x8& a, const bf16x8 (&w)[NT], f32x4 (&acc)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      acc[nt] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, w[nt], acc[nt], 0, 0, 0)
                     : __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nt], a, acc[nt], 0, 0, 0);
  }

  template <class IssueFn>
  __device__ __forceinline__ void step(const char* st, int wm, int wn, int lane, f32x4 (&acc)[TT][NT], IssueFn&& issue) {
    const char* wb = st + (wn * 16 * NT) * 128 + (lane & 15) * 128;
    const char* ab = st + G::W_STAGE_BYTES + (wm * TT * 16) * 128 + (lane & 15) * 128;
    const int koff0 = (((lane >> 4)) ^ (lane & 7)) << 4;
    const int koff1 = (((4 + (lane >> 4))) ^ (lane & 7)) << 4;
    auto rd = [&](int idx) {
      const int ks = idx / (NT + TT), r = idx % (NT + TT);
      const int ko = ks ? koff1 : koff0;
#ifdef LTXK_PROBE_FEWER_LDS_READS      // energy probe only (WRONG results): the ks=1 fragments are copies of the ks=0 ones
      if (ks == 1) { if (r < NT) wf[1][r] = wf[0][r]; else af[1][r - NT] = af[0][r - NT]; return; }
#endif
      if (r < NT) wf[ks][r] = *(const bf16x8*)(wb + r * 2048 + ko);
      else af[ks][r - NT] = *(const bf16x8*)(ab + (r - NT) * 2048 + ko);
    };
    // reads needed by real group g of THIS stage (flat order W0, A0[*], W1, A1[*]); g >= NG: everything
    auto need = [](int g) { return g < 0 ? 0 : (g >= NG ? TOTAL : (g / TT) * (NT + TT) + NT + (g % TT) + 1); };
    int issued = 0;
    // virtual groups: the DG groups deferred from the previous K-step (operands already in registers)
#pragma unroll
    for (int v = 0; v < DG; ++v) {
      int target = need(v - DG + PD);
      if (target > NT + TT) target = NT + TT;        // the ks=1 registers still feed the deferred groups
#pragma unroll
      for (int i = 0; i < TOTAL; ++i)
        if (i >= issued && i < target) rd(i);
      issued = target > issued ? target : issued;
#pragma unroll
      for (int q = 0; q < PPG; ++q) issue(v * PPG + q);
      group(af[1][TT - DG + v], wf[1], acc[TT - DG + v]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int g = 0; g < NG - DG; ++g) {
      const int target = need(g + PD);
#pragma unroll
      for (int i = 0; i < TOTAL; ++i)
        if (i >= issued && i < target) rd(i);
      issued = target > issued ? target : issued;
#pragma unroll
      for (int q = 0; q < PPG; ++q) issue((g + DG) * PPG + q);
      group(af[g / TT][g % TT], wf[g / TT], acc[g % TT]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // operands of the groups deferred to the next step must leave this stage's LDS slot before the barrier
#pragma unroll
    for (int i = 0; i < TOTAL; ++i)
      if (i >= issued) rd(i);
  }

  __device__ __forceinline__ void finish(f32x4 (&acc)[TT][NT]) {
#pragma unroll
    for (int v = 0; v < DG; ++v) group(af[1][TT - DG + v], wf[1], acc[TT - DG + v]);
  }
};

}  // namespace ltxk

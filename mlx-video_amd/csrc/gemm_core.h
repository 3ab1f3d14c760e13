// MFMA GEMM main loop shared by the dense (Linear) and implicit (conv3d) kernels.
//
//   C[m,n] = sum_k A[m,k] * W[n,k]          A rows and W rows are both k-contiguous bf16
//
// Geometry (gfx950, wave64): one workgroup = 8 waves = WM (rows) x WN (cols), WM*WN = 8;
// tile BM x BN x 64 with BM = 16*TT*WM, BN = 64*WN; per wave TT x 4 MFMA tiles of
// v_mfma_f32_16x16x32_bf16.  (TT=5, WN=4): 160x256, the Linear layers at M=2560;
// (TT=4, WN=2): 256x128, the 128-channel convolutions.
// Staging: global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip) into a 3-deep LDS ring; one
// raw s_barrier per K-step, counted vmcnt so that two K-steps stay in flight across it.
// LDS image: 128-byte rows ([row][64 bf16]) with the 16-byte chunk index XOR-swizzled by
// (row & 7): the DMA destination is lane-linear, so the permutation is applied to the per-lane
// SOURCE address and again on the ds_read_b128 fragment reads (conflict-free for the 16x16x32
// operand map).
#pragma once
#include "common.h"

namespace ltxk {

constexpr int GEMM_BK = 64;
constexpr int GEMM_THREADS = 512;

template <int TT, int WN>
struct GemmGeom {
  static constexpr int WM = 8 / WN;
  static constexpr int BM = 16 * TT * WM;
  static constexpr int BN = 64 * WN;
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

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an
// L2), so give each XCD a contiguous band of column tiles and walk it two column tiles at a
// time over all row tiles: the 32 workgroups resident on an XCD then share 2 W panels and RT
// A panels through that XCD's L2.  Placement affects speed only.
__device__ __forceinline__ void map_tile(int bid, int RT, int CT, int& rt, int& ct) {
  if ((CT & 15) == 0) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int cpx = CT >> 3;
    const int pair = idx / (2 * RT), j = idx - pair * 2 * RT;
    rt = j >> 1;
    ct = xcd * cpx + pair * 2 + (j & 1);
  } else {
    rt = bid % RT;
    ct = bid / RT;
  }
}

// Wait until this wave's loads of the current K-step have landed, leaving `keep` younger
// LDS-DMA instructions (the next K-step's) in flight, then rendezvous.
__device__ __forceinline__ void wait_stage_and_barrier(int keep) {
  switch (keep) {
    case 0: asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(7)\n\ts_barrier" ::: "memory"); break;
  }
}

// One K-step of MFMAs for this wave out of LDS stage `st`.
// SWAP=false: acc tile = D[n (4 regs)][tok (lane&15)]  (W is the MFMA A operand)
// SWAP=true : acc tile = D[tok (4 regs)][n (lane&15)]
template <int TT, int WN, bool SWAP>
__device__ __forceinline__ void mma_stage(const char* st, int wm, int wn, int lane,
                                          f32x4 (&acc)[TT][4]) {
  using G = GemmGeom<TT, WN>;
  const char* wb = st + (wn * 64) * 128;
  const char* ab = st + G::W_STAGE_BYTES + (wm * TT * 16) * 128;
  const int rowoff = (lane & 15) * 128;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int koff = (((ks * 4 + (lane >> 4)) ^ (lane & 7)) << 4);
    bf16x8 wf[4], af[TT];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wf[nt] = *(const bf16x8*)(wb + nt * 2048 + rowoff + koff);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) af[tt] = *(const bf16x8*)(ab + tt * 2048 + rowoff + koff);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[tt][nt] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tt], wf[nt], acc[tt][nt], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[tt], acc[tt][nt], 0, 0, 0);
  }
}

}  // namespace ltxk

// Fused softmax(QK^T/sqrt(dh))V for dh=128, bf16, no mask: ltxk_flash_attn_bf16.
// Replaces mx.fast.scaled_dot_product_attention (attention.py:47) and the head reshapes
// around it (attention.py:24-33,50-51).
//
// Structure (gfx950): workgroup = 4 waves, each wave owns 32 query rows; K/V tiles of 64 keys
// are staged by LDS-DMA (global_load_lds_dwordx4) into a 2-deep LDS ring, one barrier per tile.
// Per wave and tile:
//   S^T[key][q] = K . Q^T    (v_mfma_f32_32x32x16_bf16, K fragment = A operand from LDS, Q
//                             fragment = B operand held in registers for the whole kernel)
//   online softmax over keys: the key axis is the accumulator's register axis, so the row
//                             max / sum are lane-local plus one exchange with lane^32
//   O^T[d][q] += V^T . P^T    (the S^T accumulator, exponentiated and packed to bf16, IS the
//                             B operand of the second product: no LDS round trip for P; K rows
//                             are read in the permuted order fa_pi so that each P operand covers
//                             8 contiguous keys = one 16-byte V^T chunk)
// V is consumed transposed (V^T: [d][key], key-contiguous); the V projection GEMM writes it
// in that layout directly (ltxk_gemm_bf16 out_tokens_per_batch).
#include "common.h"
#include <stdlib.h>

namespace ltxk {

constexpr int FA_QW = 32;         // query rows per wave
constexpr int FA_BK = 64;         // keys per tile
constexpr int FA_DH = 128;
constexpr float FA_DEFER = 6.0f;   // defer the online-softmax rescale while the row max grows < 2^6
constexpr int FA_K_BYTES = FA_BK * FA_DH * 2;    // 16 KiB: [64 keys][256 B]
constexpr int FA_V_BYTES = FA_DH * FA_BK * 2;    // 16 KiB: [128 d][128 B]
constexpr int FA_STAGE = FA_K_BYTES + FA_V_BYTES;
constexpr int FA_LDS = 2 * FA_STAGE;

struct FaParams {
  const bf16* q; const bf16* k; const bf16* vt; bf16* out;
  int ldq, ldk, ldvt, ldo;
  int B, H, Tq, Tk;
  float c;   // scale * log2(e)
  int QT;    // query tiles per (batch, head)
  int xcd;   // 1: all query tiles of a (batch, head) run on one XCD (its K / V^T stay in that XCD's L2)
  int n_full, rem;   // tiles run as full workgroups / tiles split over two tail workgroups each
};

// Workgroups are dealt round-robin over the 8 XCDs (private 4 MiB L2 each).  With the plain order the QT query
// tiles of one (batch, head) land on all 8 XCDs and every L2 fetches that head's K and V^T: 8x the traffic.
// Here workgroup i (XCD i%8, i/8-th on it) takes head  xcd + 8*((i/8)/QT), tile (i/8)%QT.
__device__ __forceinline__ void fa_map(const FaParams& p, int i, int& bh, int& qt) {
  if (p.xcd) {
    const int x = i & 7, j = i >> 3;
    bh = x + 8 * (j / p.QT);
    qt = j % p.QT;
  } else {
    bh = i / p.QT;
    qt = i % p.QT;
  }
}

// Key order inside a 32-key block.  The S^T accumulator of lane half hh holds rows i = (j&3) + 8*(j>>2) + 4*hh
// (j = register): registers 8s..8s+7 - one B operand of the P.V product - are rows 16s + {0..3, 8..11} + 4hh.
// Lane r therefore reads K row pi(r) (bits 2 and 3 of r swapped), which makes S^T row i the key pi(i) and those
// eight registers the eight CONTIGUOUS keys 16s + 8hh + 0..7: the matching V^T fragment is one conflict-free
// ds_read_b128 (with the natural order it was two ds_read_b64 that only ever touched half the banks: 2-way
// conflicts, and the kernel was LDS-cycle bound).
__device__ __forceinline__ int fa_pi(int r) { return (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1); }
// key (minus 8*hh) of accumulator register j
__device__ __forceinline__ constexpr int fa_acc_key(int j) { return (j & 3) + 4 * ((j >> 2) & 1) + 16 * (j >> 3); }

__device__ __forceinline__ void fa_glds16(const void* g, void* l) { glds16(g, l); }

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));

// One workgroup's work.  KS = 1: the wave owns 32 query rows against all 64 keys of every tile.  KS = 2 ("tail"
// workgroups, see flash_attn_kernel): waves pair up on the same 32 query rows and split each tile's keys 32 / 32
// (kh = which half); the pair's (O, m, l) are merged through LDS at the end.
template <int NW, int KS>
__device__ __forceinline__ void fa_body(const FaParams& p, char* smem, int bh, int q0, int kh, int wave, int lane) {
  constexpr int NKB = 2 / KS;
  const int r = lane & 31, hh = lane >> 5;
  const int b = bh / p.H, h = bh - b * p.H;

  // ---- Q fragments (B operand: lane (q=r, half hh) holds d = 16*ks + 8*hh + j) ----
  // NW == 4: the wave's 32 x 256-byte Q block comes in as 8 LDS-DMA pieces of 4 whole rows (coalesced 256-byte
  // segments) into a private 8 KiB corner of ring slot 1 - free until tile 1 is fetched, which happens only after
  // the loop's first barrier - and the fragments are read from there; loading fragments straight from global
  // touches 32 rows x 32 bytes per instruction.
  bf16x8 qf[8];
  if (NW == 4) {
    char* qreg = smem + FA_STAGE + wave * 8192;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = 4 * j + (lane >> 4);
      int qrow = q0 + row;
      qrow = qrow < p.Tq ? qrow : p.Tq - 1;
      glds16(p.q + ((size_t)b * p.Tq + qrow) * p.ldq + h * FA_DH + (((lane & 15) ^ (row & 15)) << 3), qreg + j * 1024);
    }
  } else {
    int qrow = q0 + r;
    qrow = qrow < p.Tq ? qrow : p.Tq - 1;
    const bf16* qp = p.q + ((size_t)b * p.Tq + qrow) * p.ldq + h * FA_DH + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
  }

  // ---- loader addressing ----
  // K piece = 4 keys x 256 B; lane i -> key row (i>>4), 16-B slot (i&15) holding chunk slot^(row&15)
  const int k_lrow = lane >> 4, k_slot = lane & 15;
  // V^T piece = 8 d-rows x 128 B; lane i -> d row (i>>3), slot (i&7) holding chunk slot^((d>>1)&7)
  const int v_lrow = lane >> 3, v_slot = lane & 7;
  const bf16* kbase = p.k + (size_t)b * p.Tk * p.ldk + h * FA_DH;
  const bf16* vbase = p.vt + (size_t)bh * FA_DH * p.ldvt;

  // i-th K / V^T LDS-DMA piece of this wave for tile t (pieces dealt round-robin over the NW waves; a wave
  // with no i-th piece issues nothing: trip counts are wave-uniform)
  auto issue_k = [&](int i, int t, int st) __attribute__((always_inline)) {
    const int piece = wave + i * NW;
    if (piece < 16) {
      const int row = piece * 4 + k_lrow;               // key within tile
      int key = t * FA_BK + row;
      key = key < p.Tk ? key : p.Tk - 1;
      glds16(kbase + (size_t)key * p.ldk + (k_slot ^ (row & 15)) * 8, smem + st * FA_STAGE + piece * 1024);
    }
  };
  auto issue_v = [&](int i, int t, int st) __attribute__((always_inline)) {
    const int piece = wave + i * NW;
    if (piece < 16) {
      const int d = piece * 8 + v_lrow;
      glds16(vbase + (size_t)d * p.ldvt + t * FA_BK + (v_slot ^ ((d >> 1) & 7)) * 8,
             smem + st * FA_STAGE + FA_K_BYTES + piece * 1024);
    }
  };
  constexpr int NPW = (16 + NW - 1) / NW;               // pieces per wave per operand (4)

  f32x16 o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) o[i][j] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int nt = (p.Tk + FA_BK - 1) / FA_BK;
#pragma unroll
  for (int i = 0; i < NPW; ++i) { issue_k(i, 0, 0); issue_v(i, 0, 0); }
  if (NW == 4) {
    // the Q pieces were issued first: they have landed once only the 2*NPW K/V pieces are outstanding
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPW) : "memory");
    const char* qreg = smem + FA_STAGE + wave * 8192 + r * 256;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qreg + (((ks * 2 + hh) ^ (r & 15)) << 4));
  }
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const bool pre = t + 1 < nt;
    const int tn = pre ? t + 1 : t, stn = (t + 1) & 1;   // last tile: harmless re-load into the free slot
    const char* sk = smem + (t & 1) * FA_STAGE;
    const char* sv = sk + FA_K_BYTES;

    // ---- S^T = K . Q^T in groups of 4 MFMAs.  The K fragments of group g+1 are requested from LDS before the
    // MFMAs of group g issue (sched_barrier pins that order; the compiler then emits counted lgkmcnt waits):
    // left to itself hipcc reads each fragment right before its MFMA and every MFMA eats an LDS round trip.
    // The next tile's K pieces are issued between the groups (not as a burst).
    f32x16 s[NKB];
#pragma unroll
    for (int kbi = 0; kbi < NKB; ++kbi)
#pragma unroll
      for (int j = 0; j < 16; ++j) s[kbi][j] = 0.f;
    {
      constexpr int NG = 2 * NKB;
      bf16x8 kfr[2][4];
      auto load_k = [&](bf16x8* dst, int g) __attribute__((always_inline)) {
        const int row = (KS == 1 ? (g >> 1) : kh) * 32 + fa_pi(r);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int chunk = ((g & 1) * 4 + i) * 2 + hh;
          dst[i] = *(const bf16x8*)(sk + row * 256 + ((chunk ^ (row & 15)) << 4));
        }
      };
      load_k(kfr[0], 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) load_k(kfr[(g + 1) & 1], g + 1);
        // both operands of the next tile are requested during QK^T, as early in the tile as the ring allows (its K and
        // V^T halves were last read one tile ago): pieces issued during P.V had only ~0.3 us to land before the wait
        if (KS == 1) { issue_k(g, tn, stn); issue_v(g, tn, stn); }
        else { issue_k(2 * g, tn, stn); issue_k(2 * g + 1, tn, stn); issue_v(2 * g, tn, stn); issue_v(2 * g + 1, tn, stn); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          s[g >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[g & 1][i], qf[(g & 1) * 4 + i], s[g >> 1], 0, 0, 0);
      }
    }
    // V^T fragments of the first P.V group: requested now, they land under the softmax
    constexpr int NVG = 2 * NKB;                 // V^T fragments (= MFMAs) per 32-wide d block
    bf16x8 vfr[2][NVG];
    auto load_v = [&](bf16x8* dst, int db) __attribute__((always_inline)) {
      const int d = db * 32 + r;
      const char* vrow = sv + d * 128;
      const int sw = (d >> 1) & 7;
#pragma unroll
      for (int kbi = 0; kbi < NKB; ++kbi)
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
          const int cc = (KS == 1 ? kbi : kh) * 4 + 2 * sidx + hh;
          dst[kbi * 2 + sidx] = *(const bf16x8*)(vrow + ((cc ^ sw) << 4));
        }
    };
    load_v(vfr[0], 0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- mask the ragged key tail (last tile only) ----
    if (t == nt - 1 && (p.Tk & (FA_BK - 1)) != 0) {
      const int kbase_i = t * FA_BK + 8 * hh;
#pragma unroll
      for (int kbi = 0; kbi < NKB; ++kbi)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int key = kbase_i + (KS == 1 ? kbi : kh) * 32 + fa_acc_key(j);
          if (key >= p.Tk) s[kbi][j] = -1e30f;
        }
    }
    // ---- online softmax (key axis = registers + lane^32) ----
    float mx = s[0][0];
#pragma unroll
    for (int kbi = 0; kbi < NKB; ++kbi)
#pragma unroll
      for (int j = 0; j < 16; ++j) mx = fmaxf(mx, s[kbi][j]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // Deferred rescale: the running max is only raised (and O, l rescaled: 65 VALU ops per lane) when some
    // row's max grew by more than 2^FA_DEFER in the exponent domain; otherwise P is taken against the old
    // max and stays <= 2^FA_DEFER (bf16 keeps its relative precision, l accumulates in fp32).  All of this
    // tile's P is exponentiated after the decision and the previous tile's P.V is complete, so everything
    // scaled against the old max is rescaled exactly once.
    if (__any((mx - m_run) * p.c > FA_DEFER)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.c);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[db][j] *= alpha;
    }
    const float mc = m_run * p.c;
    float psum = 0.f;
    bf16x8 pb[NKB][2];
#pragma unroll
    for (int kbi = 0; kbi < NKB; ++kbi)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kbi][j], p.c, -mc));
        psum += e;
        pb[kbi][j >> 3][j & 7] = (bf16)e;
      }
    l_run += psum;

    // ---- O^T += V^T . P^T ; V^T fragment of half hh for P registers 8s..8s+7 = keys 16s + 8hh + 0..7;
    // one group per 32-wide d block, the next block's fragments requested before this block's MFMAs
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      if (db + 1 < 4) load_v(vfr[(db + 1) & 1], db + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kbi = 0; kbi < NKB; ++kbi)
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx)
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[db & 1][kbi * 2 + sidx], pb[kbi][sidx], o[db], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (KS == 2) {
    // ---- merge the key halves of each wave pair through LDS (the K/V ring is dead now) ----
    __syncthreads();
    float* xo = (float*)(smem + (wave >> 1) * (17 * 1024));
    if (kh == 1) {
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int j = 0; j < 16; ++j) xo[(db * 16 + j) * 64 + lane] = o[db][j];
      xo[4096 + lane] = m_run;
      xo[4096 + 64 + lane] = l_tot;
    }
    __syncthreads();
    if (kh == 1) return;
    const float m1 = xo[4096 + lane], l1 = xo[4096 + 64 + lane];
    const float m = fmaxf(m_run, m1);
    const float a0 = __builtin_amdgcn_exp2f((m_run - m) * p.c), a1 = __builtin_amdgcn_exp2f((m1 - m) * p.c);
    l_tot = l_tot * a0 + l1 * a1;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int j = 0; j < 16; ++j) o[db][j] = o[db][j] * a0 + xo[(db * 16 + j) * 64 + lane] * a1;
  }

  // ---- epilogue: O[q][d] = O^T[d][q] / l, transposed through a wave-private 8 KiB LDS image so that every
  // store instruction writes four whole 256-byte rows (16 B per lane) instead of 64 scattered 8-byte pieces
  const float inv = 1.0f / l_tot;
  if (KS == 1) __syncthreads();                       // every wave has left the K / V^T ring
  char* stg = smem + (KS == 1 ? wave * 8192 : 2 * 17 * 1024 + (wave >> 1) * 8192);
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4v v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (bf16)(o[db][g * 4 + j] * inv);
      *(bf16x4v*)(stg + r * 256 + (((db * 4 + g) ^ (r & 15)) << 4) + hh * 8) = v;      // row q=r, d = 32db+8g+4hh..+3
    }
  {
    bf16* ob = p.out + ((size_t)b * p.Tq) * p.ldo + h * FA_DH + (lane & 15) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = i * 4 + (lane >> 4);
      const bf16x8 v = *(const bf16x8*)(stg + row * 256 + (((lane & 15) ^ (row & 15)) << 4));
      if (q0 + row < p.Tq) *(bf16x8*)(ob + (size_t)(q0 + row) * p.ldo) = v;
    }
  }
}

// NW waves per workgroup (4 or 5): 128 or 160 query rows.  Two workgroups per CU are resident (64 KiB LDS
// each): 512 slots.  At Tq=1280, B*H=64 the 128-row form has 640 tiles: a full round plus a quarter-filled
// one that takes as long.  So only the first n_full tiles (whole rounds) run as 128-row workgroups; each of
// the remaining `rem` tiles is split over two "tail" workgroups of 64 rows whose wave pairs halve the keys
// (fa_body<KS=2>): the short round then does half the work per wave with twice the workgroups.
// (The 160-row form gives exactly 512 workgroups but loads the SIMDs 3,3,2,2 and measures slower.)
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void flash_attn_kernel(FaParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bh, qt;
  if (NW != 4 || (int)blockIdx.x < p.n_full) {
    fa_map(p, blockIdx.x, bh, qt);
    fa_body<NW, 1>(p, smem, bh, qt * (FA_QW * NW) + wave * FA_QW, 0, wave, lane);
  } else {
    const int ti = blockIdx.x - p.n_full;
    const int tile = p.n_full + ((p.rem & 7) == 0 ? ti % p.rem : ti >> 1);     // keeps tile%8 == blockIdx%8 (same XCD)
    const int half = (p.rem & 7) == 0 ? ti / p.rem : ti & 1;
    fa_map(p, tile, bh, qt);
    fa_body<NW, 2>(p, smem, bh, qt * 128 + half * 64 + (wave >> 1) * FA_QW, wave & 1, wave, lane);
  }
}


// ---------------------------------------------------------------------------------------
// Ping-pong form: one workgroup of 8 waves (256 query rows) per CU, K / V^T tiles in a 4-deep LDS ring
// (128 KiB, tiles fetched three ahead).  Per tile a wave has a VALU phase X (online softmax of S(t): ~1000
// cycles, half of it quarter-rate v_exp) and an MFMA phase Y (O += P(t).V(t), then S(t+1) = K(t+1).Q^T:
// 32 MFMAs, ~1000 cycles).  Waves 0-3 (one per SIMD) and waves 4-7 (the second wave of each SIMD) run half a
// tile apart - group B passes one extra barrier up front - so on every SIMD one wave's softmax runs under
// the other wave's MFMAs instead of both waves doing the same phase at the same time.  Each K / V^T tile
// now feeds 256 query rows: half the L2->LDS traffic of the 128-row form.
//
// Barrier pairing (A = waves 0-3, B = waves 4-7):   A: b0 X0 | Y0 | X1 | Y1 | ... X(n-1) | Y(n-1) | -
//                                                   B: b0 -  | X0 | Y0 | X1 | ...          Y(n-2)| X(n-1) | Y(n-1)
// Tile u (u >= 3) is fetched during Y(u-3) into ring slot u&3 (tile u-1's slot was last read in Y(u-1)...
// i.e. slot (u&3) held tile u-4, last read in Y(u-4), which every wave has left).  Before the barrier that
// opens A's Y(t) every wave has waited (vmcnt(4): all but its newest tile) for its pieces of tile t+1.
// ---------------------------------------------------------------------------------------
constexpr int PP_NS = 4;
constexpr int PP_LDS = PP_NS * FA_STAGE;
constexpr int PP_BQ = 256;

#define PP_BARRIER()                                   \
  do {                                                 \
    __builtin_amdgcn_sched_barrier(0);                 \
    asm volatile("s_barrier" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);                 \
  } while (0)

__global__ __launch_bounds__(512, 2) void flash_attn_pp_kernel(FaParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int r = lane & 31, hh = lane >> 5;
  int bh, qt;
  fa_map(p, blockIdx.x, bh, qt);
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = qt * PP_BQ + wave * FA_QW;

  int qrow = q0 + r;
  qrow = qrow < p.Tq ? qrow : p.Tq - 1;
  const bf16* qp = p.q + ((size_t)b * p.Tq + qrow) * p.ldq + h * FA_DH + hh * 8;
  bf16x8 qf[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);

  const int k_lrow = lane >> 4, k_slot = lane & 15;
  const int v_lrow = lane >> 3, v_slot = lane & 7;
  const bf16* kbase = p.k + (size_t)b * p.Tk * p.ldk + h * FA_DH;
  const bf16* vbase = p.vt + (size_t)bh * FA_DH * p.ldvt;
  const int nt = (p.Tk + FA_BK - 1) / FA_BK;

  // piece j (0,1: K; 2,3: V^T) of this wave for tile t -> ring slot `slot` (16 + 16 pieces dealt over 8 waves)
  auto issue = [&](int j, int t, int slot) __attribute__((always_inline)) {
    t = t < nt ? t : nt - 1;          // past the end: harmless re-load into a free slot (keeps vmcnt constant)
    char* st = smem + slot * FA_STAGE;
    const int piece = wave + 8 * (j & 1);
    if (j < 2) {
      const int row = piece * 4 + k_lrow;
      int key = t * FA_BK + row;
      key = key < p.Tk ? key : p.Tk - 1;
      glds16(kbase + (size_t)key * p.ldk + (k_slot ^ (row & 15)) * 8, st + piece * 1024);
    } else {
      const int d = piece * 8 + v_lrow;
      glds16(vbase + (size_t)d * p.ldvt + t * FA_BK + (v_slot ^ ((d >> 1) & 7)) * 8, st + FA_K_BYTES + piece * 1024);
    }
  };

  f32x16 o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) o[i][j] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  f32x16 s[2];

  // fragment groups: the next group's LDS reads are requested before this group's 4 MFMAs (see fa_body)
  bf16x8 fr[2][4];
  auto load_k = [&](bf16x8* dst, int slot, int g) __attribute__((always_inline)) {
    const char* sk = smem + slot * FA_STAGE;
    const int row = (g >> 1) * 32 + fa_pi(r);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int chunk = ((g & 1) * 4 + i) * 2 + hh;
      dst[i] = *(const bf16x8*)(sk + row * 256 + ((chunk ^ (row & 15)) << 4));
    }
  };
  auto load_v = [&](bf16x8* dst, int slot, int db) __attribute__((always_inline)) {
    const int d = db * 32 + r;
    const char* vrow = smem + slot * FA_STAGE + FA_K_BYTES + d * 128;
    const int sw = (d >> 1) & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = *(const bf16x8*)(vrow + ((((i >> 1) * 4 + 2 * (i & 1) + hh) ^ sw) << 4));   // i = kb*2 + sidx
  };
  // S = K(slot) . Q^T; fr[first] already holds (or has in flight) group 0
  auto qk = [&](int slot, int first) __attribute__((always_inline)) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int j = 0; j < 16; ++j) s[kb][j] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g + 1 < 4) load_k(fr[(first + g + 1) & 1], slot, g + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s[g >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[(first + g) & 1][i], qf[(g & 1) * 4 + i], s[g >> 1], 0, 0, 0);
      }
    }
  };

#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) issue(j, t, t);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  PP_BARRIER();
  load_k(fr[0], 0, 0);
  qk(0, 0);
  if (grp == 1) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    PP_BARRIER();
  }

  for (int t = 0; t < nt; ++t) {
    // ================= X: online softmax of S(t) (VALU) =================
    if (t == nt - 1 && (p.Tk & (FA_BK - 1)) != 0) {
      const int kbase_i = t * FA_BK + 8 * hh;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int key = kbase_i + kb * 32 + fa_acc_key(j);
          if (key >= p.Tk) s[kb][j] = -1e30f;
        }
    }
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int j = 0; j < 16; ++j) mx = fmaxf(mx, s[kb][j]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (__any((mx - m_run) * p.c > FA_DEFER)) {      // deferred rescale, see flash_attn_kernel
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.c);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[db][j] *= alpha;
    }
    const float mc = m_run * p.c;
    float psum = 0.f;
    bf16x8 pb[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][j], p.c, -mc));
        psum += e;
        pb[kb][j >> 3][j & 7] = (bf16)e;
      }
    l_run += psum;
    if (grp == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    PP_BARRIER();

    // ================= Y: O^T += V^T(t) . P^T(t);  S(t+1) = K(t+1) . Q^T  (MFMA) =================
    load_v(fr[0], t & 3, 0);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      if (db + 1 < 4) load_v(fr[(db + 1) & 1], t & 3, db + 1);
      else if (t + 1 < nt) load_k(fr[0], (t + 1) & 3, 0);
      issue(db, t + 3, (t + 3) & 3);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[db & 1][i], pb[i >> 1][i & 1], o[db], 0, 0, 0);
      }
    }
    if (t + 1 < nt) qk((t + 1) & 3, 0);
    if (grp == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    PP_BARRIER();
  }
  if (grp == 0) PP_BARRIER();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qout = q0 + r;
  if (qout < p.Tq) {
    bf16* op = p.out + ((size_t)b * p.Tq + qout) * p.ldo + h * FA_DH + 4 * hh;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4v v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (bf16)(o[db][g * 4 + j] * inv);
        *(bf16x4v*)(op + db * 32 + g * 8) = v;
      }
  }
}

}  // namespace ltxk

extern "C" int ltxk_flash_attn_bf16(const void* q, int32_t ldq, const void* k, int32_t ldk,
                                    const void* vt, int32_t ldvt, void* out, int32_t ldo,
                                    int32_t B, int32_t H, int32_t Tq, int32_t Tk, float scale,
                                    void* stream) {
  using namespace ltxk;
  LTXK_CHECK_ARG(q && k && vt && out, "ltxk_flash_attn_bf16: null pointer");
  LTXK_CHECK_ARG(B > 0 && H > 0 && Tq > 0 && Tk > 0, "ltxk_flash_attn_bf16: bad dims");
  LTXK_CHECK_ARG(ldq >= H * FA_DH && ldk >= H * FA_DH && ldo >= H * FA_DH, "ltxk_flash_attn_bf16: row strides < H*128");
  LTXK_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldo % 8 == 0, "ltxk_flash_attn_bf16: row strides must be multiples of 8");
  const int tk_pad = (Tk + FA_BK - 1) / FA_BK * FA_BK;
  LTXK_CHECK_ARG(ldvt >= tk_pad && ldvt % 8 == 0, "ltxk_flash_attn_bf16: ldvt=%d must be >= %d (Tk rounded up to 64) and a multiple of 8", ldvt, tk_pad);
  LTXK_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt) & 15) == 0 && ((uintptr_t)out & 15) == 0, "ltxk_flash_attn_bf16: misaligned pointer");
  FaParams p;
  p.q = (const bf16*)q; p.k = (const bf16*)k; p.vt = (const bf16*)vt; p.out = (bf16*)out;
  p.ldq = ldq; p.ldk = ldk; p.ldvt = ldvt; p.ldo = ldo;
  p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk;
  p.c = scale * 1.4426950408889634f;
  static thread_local int attr_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev != attr_dev) {
    hipError_t e = hipFuncSetAttribute((const void*)flash_attn_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, FA_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)flash_attn_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, FA_LDS);
    if (e != hipSuccess) { ltxk_set_error("ltxk_flash_attn_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e)); return LTXK_ELAUNCH; }
    attr_dev = dev;
  }
  // Variants measured at B=2,H=32 (scripts/prof_attn.py; Tq=Tk=1280 / Tq=Tk=5184): 4-wave/64 KiB with the
  // tail split (default) 630 / 945 TF/s; without the split 600 / 920; 8-wave ping-pong (variant 8) 565 / 930;
  // 5-wave/160-row form (one exact round at 1280, but 10 waves per CU load the SIMDs 3,3,2,2) 554 / 580;
  // a 48-KiB/3-workgroup form (register-capped at 168, spilled) ~430, removed.  Ablating the ping-pong kernel at
  // Tq=Tk=5184 (982 us): no softmax 838, no DMA 858, neither 761, MFMAs + barriers only 636 us - i.e. the
  // MFMA stream alone already runs at a DVFS-lowered ~1.5-1.6 GHz, and LDS reads, DMA and softmax each add
  // 10-15 % on top; none of them alone is the bound.
  // LTXK_FA_VARIANT={4,5,8}, LTXK_FA_XCD={1,0}, LTXK_FA_SPLIT={1,0} select forms for A/B runs.
  static const int variant = [] { const char* e = getenv("LTXK_FA_VARIANT"); return e ? atoi(e) : 4; }();
  static const int xcd_map = [] { const char* e = getenv("LTXK_FA_XCD"); return e ? atoi(e) : 1; }();
  const int rows = variant == 5 ? 160 : variant == 8 ? PP_BQ : 128;
  p.QT = (Tq + rows - 1) / rows;
  p.xcd = (xcd_map && (B * H) % 8 == 0) ? 1 : 0;
  // read per call (not cached) so one process can A/B it: with the split on, a tile in the short round sums its
  // keys in a different order than the same rows would in a launch without a short round (e.g. B=1 vs B=2), so
  // batching changes low-order bits; LTXK_FA_SPLIT=0 restores batch-invariant results.
  const char* split_env = getenv("LTXK_FA_SPLIT");
  const int split = split_env ? atoi(split_env) : 1;
  static thread_local int slots = 0;
  if (slots == 0) {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    slots = 2 * cus;
  }
  const int tiles = p.QT * B * H;
  p.n_full = tiles; p.rem = 0;
  if (split && variant == 4 && tiles % slots != 0 && 2 * (tiles % slots) <= slots) {
    p.rem = tiles % slots;
    p.n_full = tiles - p.rem;
  }
  const dim3 grid((unsigned)(p.n_full + 2 * p.rem));
  if (variant == 8) {
    static thread_local int attr8_dev = -1;
    if (dev != attr8_dev) {
      hipError_t e = hipFuncSetAttribute((const void*)flash_attn_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);
      if (e != hipSuccess) { ltxk_set_error("ltxk_flash_attn_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e)); return LTXK_ELAUNCH; }
      attr8_dev = dev;
    }
    hipLaunchKernelGGL(flash_attn_pp_kernel, grid, dim3(512), PP_LDS, (hipStream_t)stream, p);
  } else if (variant == 5) {
    hipLaunchKernelGGL(flash_attn_kernel<5>, grid, dim3(320), FA_LDS, (hipStream_t)stream, p);
  } else {
    hipLaunchKernelGGL(flash_attn_kernel<4>, grid, dim3(256), FA_LDS, (hipStream_t)stream, p);
  }
  LTXK_CHECK_LAUNCH("ltxk_flash_attn_bf16");
  return LTXK_OK;
}

// Fused softmax(QK^T/sqrt(dh))V for dh=128, bf16, no mask: ltxk_flash_attn_bf16.
// Replaces mx.fast.scaled_dot_product_attention (attention.py:47) and the head reshapes
// around it (attention.py:24-33,50-51).
//
// Kernels of the same structure live here.  On v_mfma_f32_16x16x32_bf16 (fa_body16<KS, QB>, rounds 3-4; its header has the operand
// map): flash_attn16_mix_kernel - 192-row tiles (three 16-row query blocks per wave) mixed with 128-row tiles, what
// ltxk_flash_attn launches from 1.25 rounds of 128-row tiles up - and flash_attn16_kernel - 128-row tiles with a key-split tail,
// for smaller grids.  On v_mfma_f32_32x32x16_bf16: flash_attn_kernel (fa_body, rounds 1-2; kept for the A/B build,
// LTXK_FA_MFMA=32), described first.
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

namespace ltxk {

constexpr int FA_QW = 32;         // query rows per wave
constexpr int FA_BK = 64;         // keys per tile
constexpr int FA_DH = 128;
constexpr float FA_DEFER = 6.0f;   // defer the online-softmax rescale while the row max grows < 2^6
constexpr int FA_K_BYTES = FA_BK * FA_DH * 2;    // 16 KiB: [64 keys][256 B]
constexpr int FA_V_BYTES = FA_DH * FA_BK * 2;    // 16 KiB: [128 d][128 B]
constexpr int FA_STAGE = FA_K_BYTES + FA_V_BYTES;
constexpr int FA_LDS = 2 * FA_STAGE;
constexpr int FA_DEFAULT_MFMA = 16;   // MFMA shape of the shipped kernel: 32 = v_mfma_f32_32x32x16_bf16, 16 = v_mfma_f32_16x16x32_bf16 (fa_body16)

struct FaParams {
  const bf16* q; const bf16* k; const bf16* vt; bf16* out;
  int ldq, ldk, ldvt, ldo;
  int B, H, Tq, Tk;
  float c;   // scale * log2(e)
  int QT;    // query tiles per (batch, head)
  int xcd;   // 1: all query tiles of a (batch, head) run on one XCD (its K / V^T stay in that XCD's L2)
  int n_full, rem;   // tiles run as full workgroups / tiles split over two tail workgroups each
  int n_a, qta, qtb; // mixed grid (flash_attn16_mix_kernel): workgroups [0, n_a) are 192-row tiles (qta per (batch, head)), the rest 128-row tiles (qtb each)
  // fused query preparation (attention.py:129-136): q holds the RAW to_q output; its per-row sums of squares come as
  // q_ss_n fp32 partials per row (the GEMM's sumsq output), the kernel applies RMSNorm (all heads jointly) * weight and
  // the SPLIT rotation to its Q fragments in registers - the normalised / rotated q never makes a trip through HBM
  const float* q_ss; int q_ss_ld, q_ss_n;
  const bf16* q_w;
  const float* cosb; const float* sinb;
  float eps;
};

// RMSNorm * weight (+ SPLIT RoPE) of one wave's Q fragments, same op order and rounding points as qknorm_rope_kernel
// (elementwise.hip): x = bf16(q * rstd * w); o1 = bf16(x1*c - s*x2), o2 = bf16(x2*c + s*x1) with x2 = x1's partner 64
// channels up, which is fragment ks+4 of the same lane.
__device__ __forceinline__ void fa_prep_q(const FaParams& p, bf16x8 (&qf)[8], int b, int h, int qrow, int hh) {
  const int half = p.q_ss_n >> 1;
  const float* sp = p.q_ss + (size_t)(b * p.Tq + qrow) * p.q_ss_ld + hh * half;
  float ss = 0.f;
  for (int i = 0; i < half; i += 4) {
    const f32x4 v = *(const f32x4*)(sp + i);
    ss += v[0]; ss += v[1]; ss += v[2]; ss += v[3];
  }
  ss = lane_xor32_sum(ss);
  const float rstd = rsqrtf(ss / (float)(p.H * FA_DH) + p.eps);
  const bf16* wp = p.q_w + h * FA_DH + hh * 8;
  const size_t cso = ((size_t)h * p.Tq + qrow) * 64 + hh * 8;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const bf16x8 wa = *(const bf16x8*)(wp + ks * 16), wb = *(const bf16x8*)(wp + 64 + ks * 16);
    f32x4 c0, c1, s0, s1;
    if (p.cosb) {
      c0 = *(const f32x4*)(p.cosb + cso + ks * 16); c1 = *(const f32x4*)(p.cosb + cso + ks * 16 + 4);
      s0 = *(const f32x4*)(p.sinb + cso + ks * 16); s1 = *(const f32x4*)(p.sinb + cso + ks * 16 + 4);
    }
    bf16x8 oa, ob;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x1 = rbf((float)qf[ks][j] * rstd * (float)wa[j]);
      const float x2 = rbf((float)qf[ks + 4][j] * rstd * (float)wb[j]);
      if (p.cosb) {
        const float c = j < 4 ? c0[j & 3] : c1[j & 3], sn = j < 4 ? s0[j & 3] : s1[j & 3];
        oa[j] = (bf16)(x1 * c - sn * x2);
        ob[j] = (bf16)(x2 * c + sn * x1);
      } else {
        oa[j] = (bf16)x1;
        ob[j] = (bf16)x2;
      }
    }
    qf[ks] = oa;
    qf[ks + 4] = ob;
  }
}

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

  // i-th K / V^T LDS-DMA piece of this wave for tile t (pieces dealt round-robin over the NW waves).  The source is a
  // wave-uniform tile base (SGPR pair, advanced by scalar adds) plus a per-lane 32-bit byte offset that does not depend
  // on the tile: the loop spends no vector instruction on DMA addresses (they were 36 of ~250 VALU per tile).
  static_assert(16 % NW == 0, "pieces must divide evenly over the waves");
  unsigned koff[16 / NW], voff[16 / NW];
  const int last_ragged = (p.Tk & (FA_BK - 1)) != 0 ? (p.Tk - 1) / FA_BK : -1;    // tile whose key rows need the clamp
  auto set_koff = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 16 / NW; ++i) {
      const int row = (wave + i * NW) * 4 + k_lrow;                              // key row within the tile
      int key = row;
      if (t == last_ragged) key = (t * FA_BK + row < p.Tk ? t * FA_BK + row : p.Tk - 1) - t * FA_BK;
      koff[i] = (unsigned)key * (unsigned)p.ldk * 2u + ((unsigned)(k_slot ^ (row & 15)) << 4);
    }
  };
  set_koff(0);
#pragma unroll
  for (int i = 0; i < 16 / NW; ++i) {
    const int d = (wave + i * NW) * 8 + v_lrow;
    voff[i] = (unsigned)d * (unsigned)p.ldvt * 2u + ((unsigned)(v_slot ^ ((d >> 1) & 7)) << 4);
  }
  const uint64_t kstep = (uint64_t)FA_BK * p.ldk * 2, vstep = FA_BK * 2;
  auto issue_k = [&](int i, int t, int st) __attribute__((always_inline)) {
    glds16_s((uint64_t)(uintptr_t)kbase + (uint64_t)t * kstep, koff[i], smem + st * FA_STAGE + (wave + i * NW) * 1024);
  };
  auto issue_v = [&](int i, int t, int st) __attribute__((always_inline)) {
    glds16_s((uint64_t)(uintptr_t)vbase + (uint64_t)t * vstep, voff[i], smem + st * FA_STAGE + FA_K_BYTES + (wave + i * NW) * 1024);
  };
  constexpr int NPW = (16 + NW - 1) / NW;               // pieces per wave per operand (4)

  f32x16 o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) o[i][j] = 0.f;
  float mc_run = -1e30f, l_run = 0.f;   // running offset (exp2 domain, integer-valued once set), running sum

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
  if (p.q_ss) {
    const int qr = q0 + r < p.Tq ? q0 + r : p.Tq - 1;
    fa_prep_q(p, qf, b, h, qr, hh);
  }
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const bool pre = t + 1 < nt;
    const int tn = pre ? t + 1 : t, stn = (t + 1) & 1;   // last tile: harmless re-load into the free slot
    if (tn == last_ragged && pre) set_koff(tn);           // once per kernel: the ragged last tile clamps its key rows
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
    // Deferred rescale: the running max is only raised (and O, l rescaled: 65 VALU ops per lane) when some
    // row's max grew by more than 2^FA_DEFER in the exponent domain; otherwise P is taken against the old
    // max and stays <= 2^FA_DEFER (bf16 keeps its relative precision, l accumulates in fp32).  All of this
    // tile's P is exponentiated after the decision and the previous tile's P.V is complete, so everything
    // scaled against the old max is rescaled exactly once.
    // The offset is kept as an INTEGER in the exp2 domain (mc_run = ceil(max * c) at the time it was last raised):
    // bf16(exp2(x - M)) = 2^-M * bf16(exp2(x)) exactly for integer M, so the bf16 rounding of every P does not depend
    // on which tile's running max it was taken against, alpha below is an exact power of two, and the whole kernel
    // differs from  O = (bf16(P) @ V) / sum(P),  P = exp2(fma(S, c, -M))  by fp32 summation order only - whatever the
    // tile size, the deferral or the key split of the tail workgroups (oracle/dit.py::sdpa, "flash" policy).
    // The deferral test needs no cross-lane exchange: "some row's max grew past the threshold" is the same predicate whether
    // each lane tests its own half of the keys or the row's full max (the row max is the larger of its two halves, mc_run
    // is the same in both) - the exchange with lane ^ 32 happens only inside the (rare) rescale.
    if (__any(mx * p.c - mc_run > FA_DEFER)) {
      asm volatile("" ::: "memory");       // keeps this a real branch: hipcc otherwise if-converts it into 64 multiplies + selects per tile
      const float mxc = lane_xor32_max(mx) * p.c;
      const float m_new = fmaxf(mc_run, __builtin_ceilf(mxc));
      const float alpha = __builtin_amdgcn_exp2f(mc_run - m_new);
      mc_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[db][j] *= alpha;
    }
    const float mc = mc_run;
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

  float l_tot = lane_xor32_sum(l_run);
  if (KS == 2) {
    // ---- merge the key halves of each wave pair through LDS (the K/V ring is dead now) ----
    __syncthreads();
    float* xo = (float*)(smem + (wave >> 1) * (17 * 1024));
    if (kh == 1) {
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int j = 0; j < 16; ++j) xo[(db * 16 + j) * 64 + lane] = o[db][j];
      xo[4096 + lane] = mc_run;
      xo[4096 + 64 + lane] = l_tot;
    }
    __syncthreads();
    if (kh == 1) return;
    const float m1 = xo[4096 + lane], l1 = xo[4096 + 64 + lane];
    const float m = fmaxf(mc_run, m1);
    const float a0 = __builtin_amdgcn_exp2f(mc_run - m), a1 = __builtin_amdgcn_exp2f(m1 - m);      // exact powers of two
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

// =====================================================================================================================
// The same kernel on v_mfma_f32_16x16x32_bf16 (round 3).  MI355X_MICROARCH.md (DVFS give-back, item 7) and
// profiles/r02_mfma_power_probe.log price the 16x16x32 shape 12-15 % cheaper in held clock than 32x32x16 on random data at
// equal cycles per FLOP.  Geometry per wave is unchanged (32 query rows x 64-key tiles, 4 waves, 2 workgroups per CU);
// what changes is the operand map.  With g = lane >> 4, c = lane & 15:
//   S^T block (kb = 16-key block 0..3, qb = 16-query block 0..1) = K(kb) . Q^T(qb), 4 k-steps of 32 channels:
//     A = K fragment: lane (c,g) holds key row keymap(kb, c), channels 32ks + 8g .. +7          (one ds_read_b128)
//     B = Q fragment: lane (c,g) holds query 16qb + c, the same channels                            (registers, whole kernel)
//     D: lane (c,g) register j = S^T[row 4g + j of block kb][query 16qb + c]
//   The softmax axis (keys) is registers x the four lane groups g: row max / sum = lane-local + two exchanges (lane ^ 16,
//   lane ^ 32: v_permlane16_swap / v_permlane32_swap, one VALU instruction each - no LDS round trip).
//   O^T block (db = 16-channel block 0..7, qb) += V^T(db, kc) . P^T(kc, qb) over the two 32-key chunks kc of the tile:
//     B = P^T: lane (c,g) k-slots 8g .. 8g+7 = its own registers { S[2kc][qb][0..3], S[2kc+1][qb][0..3] } exponentiated
//     A = V^T fragment: lane (c,g) holds channel 16db + c, k-slots 8g .. 8g+7
//   keymap(kb, r) = 32 (kb >> 1) + 8 (r >> 2) + 4 (kb & 1) + (r & 3) makes those eight k-slots the eight CONTIGUOUS keys
//   32 kc + 8g + 0..7, so the V^T fragment is one conflict-free ds_read_b128 and P never touches LDS (as in the 32x32 form).
// LDS images: V^T as before; K rows are swizzled by kswz(row) = 4 ((row >> 3) & 3) + (row & 3) - the S^T row index c the
// lane reads it for - instead of row & 15: the 16 lanes of a ds_read_b128 group are {c in 0-3,12-15 of one g} + {c in 4-11
// of g ^ 1}, both sets closed under c ^ 1, so chunk ^ c is a bijection onto the 16 chunk positions.
// =====================================================================================================================
__device__ __forceinline__ constexpr int fa16_keymap(int kb, int r) { return 32 * (kb >> 1) + 8 * (r >> 2) + 4 * (kb & 1) + (r & 3); }
__device__ __forceinline__ int fa16_kswz(int row) { return 4 * ((row >> 3) & 3) + (row & 3); }

// RMSNorm * weight (+ SPLIT RoPE) of the wave's Q fragments in the 16x16x32 operand map: fragment (qb, ks) of lane (c,g)
// holds channels 32ks + 8g + j; the rotation partner 64 channels up is fragment ks + 2 of the same lane.  Same op order
// and rounding points as fa_prep_q / qknorm_rope_kernel.
template <int QB>
__device__ __forceinline__ void fa16_prep_q(const FaParams& p, bf16x8 (&qf)[QB][4], int b, int h, int q0, int c, int g) {
  const int half = p.q_ss_n >> 1;
  const bf16* wp = p.q_w + h * FA_DH + g * 8;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    int qrow = q0 + 16 * qb + c;
    qrow = qrow < p.Tq ? qrow : p.Tq - 1;
    const float* sp = p.q_ss + (size_t)(b * p.Tq + qrow) * p.q_ss_ld + (g & 1) * half;
    float ss = 0.f;
    for (int i = 0; i < half; i += 4) {
      const f32x4 v = *(const f32x4*)(sp + i);
      ss += v[0]; ss += v[1]; ss += v[2]; ss += v[3];
    }
    ss = lane_xor16_sum(ss);                      // half 0 + half 1 (lanes g and g ^ 1 hold one each): same sum as fa_prep_q
    const float rstd = rsqrtf(ss / (float)(p.H * FA_DH) + p.eps);
    const size_t cso = ((size_t)h * p.Tq + qrow) * 64 + g * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 wa = *(const bf16x8*)(wp + ks * 32), wb = *(const bf16x8*)(wp + 64 + ks * 32);
      f32x4 c0, c1, s0, s1;
      if (p.cosb) {
        c0 = *(const f32x4*)(p.cosb + cso + ks * 32); c1 = *(const f32x4*)(p.cosb + cso + ks * 32 + 4);
        s0 = *(const f32x4*)(p.sinb + cso + ks * 32); s1 = *(const f32x4*)(p.sinb + cso + ks * 32 + 4);
      }
      bf16x8 oa, ob;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x1 = rbf((float)qf[qb][ks][j] * rstd * (float)wa[j]);
        const float x2 = rbf((float)qf[qb][ks + 2][j] * rstd * (float)wb[j]);
        if (p.cosb) {
          const float cc = j < 4 ? c0[j & 3] : c1[j & 3], sn = j < 4 ? s0[j & 3] : s1[j & 3];
          oa[j] = (bf16)(x1 * cc - sn * x2);
          ob[j] = (bf16)(x2 * cc + sn * x1);
        } else {
          oa[j] = (bf16)x1;
          ob[j] = (bf16)x2;
        }
      }
      qf[qb][ks] = oa;
      qf[qb][ks + 2] = ob;
    }
  }
}

template <int KS, int QB = 2>
__device__ __forceinline__ void fa_body16(const FaParams& p, char* smem, int bh, int q0, int kh, int wave, int lane) {
  constexpr int NW = 4;
  constexpr int NKC = 2 / KS;                    // 32-key chunks of a tile this wave works on
  const int c = lane & 15, g = lane >> 4;
  const int b = bh / p.H, h = bh - b * p.H;

  // ---- Q.  QB = 2: the wave's 32 x 256-byte block by LDS-DMA into its private corner of ring slot 1 (rows swizzled by row & 15;
  // the slot is free until tile 1 is fetched, after the loop's first barrier).  QB = 3: 48 rows per wave would need 48 KiB of
  // staging - 80 KiB per workgroup, and two workgroups no longer share a CU - so the fragments come straight from global memory
  // (12 loads of 16 B per lane, once per kernel; hipcc's own counted waits stay correct with the LDS-DMA pieces issued behind them:
  // vmcnt retires in order and younger operations only make its waits conservative) ----
  bf16x8 qf[QB][4];
  if constexpr (QB == 2) {
    char* qreg = smem + FA_STAGE + wave * 8192;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = 4 * j + (lane >> 4);
      int qrow = q0 + row;
      qrow = qrow < p.Tq ? qrow : p.Tq - 1;
      glds16(p.q + ((size_t)b * p.Tq + qrow) * p.ldq + h * FA_DH + (((lane & 15) ^ (row & 15)) << 3), qreg + j * 1024);
    }
  } else {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      int qrow = q0 + 16 * qb + c;
      qrow = qrow < p.Tq ? qrow : p.Tq - 1;
      const bf16* qp = p.q + ((size_t)b * p.Tq + qrow) * p.ldq + h * FA_DH + g * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *(const bf16x8*)(qp + ks * 32);
    }
  }
  // ---- loader addressing: K piece = 4 key rows x 256 B (chunk position = chunk ^ kswz(row)); V^T piece = 8 d-rows x 128 B ----
  const int k_lrow = lane >> 4, k_slot = lane & 15;
  const int v_lrow = lane >> 3, v_slot = lane & 7;
  const bf16* kbase = p.k + (size_t)b * p.Tk * p.ldk + h * FA_DH;
  const bf16* vbase = p.vt + (size_t)bh * FA_DH * p.ldvt;
  unsigned koff[4], voff[4];
  const int last_ragged = (p.Tk & (FA_BK - 1)) != 0 ? (p.Tk - 1) / FA_BK : -1;
  auto set_koff = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave + i * NW) * 4 + k_lrow;
      int key = row;
      if (t == last_ragged) key = (t * FA_BK + row < p.Tk ? t * FA_BK + row : p.Tk - 1) - t * FA_BK;
      koff[i] = (unsigned)key * (unsigned)p.ldk * 2u + ((unsigned)(k_slot ^ fa16_kswz(row)) << 4);
    }
  };
  set_koff(0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int d = (wave + i * NW) * 8 + v_lrow;
    voff[i] = (unsigned)d * (unsigned)p.ldvt * 2u + ((unsigned)(v_slot ^ ((d >> 1) & 7)) << 4);
  }
  const uint64_t kstep = (uint64_t)FA_BK * p.ldk * 2, vstep = FA_BK * 2;
  auto issue_k = [&](int i, int t, int st) __attribute__((always_inline)) {
    glds16_s((uint64_t)(uintptr_t)kbase + (uint64_t)t * kstep, koff[i], smem + st * FA_STAGE + (wave + i * NW) * 1024);
  };
  auto issue_v = [&](int i, int t, int st) __attribute__((always_inline)) {
    glds16_s((uint64_t)(uintptr_t)vbase + (uint64_t)t * vstep, voff[i], smem + st * FA_STAGE + FA_K_BYTES + (wave + i * NW) * 1024);
  };

  f32x4 o[8][QB];
#pragma unroll
  for (int db = 0; db < 8; ++db)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int j = 0; j < 4; ++j) o[db][qb][j] = 0.f;
  float mc_run[QB], l_run[QB];                                     // per query block qb (query row 16qb + c)
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) { mc_run[qb] = -1e30f; l_run[qb] = 0.f; }

  const int nt = (p.Tk + FA_BK - 1) / FA_BK;
#pragma unroll
  for (int i = 0; i < 4; ++i) { issue_k(i, 0, 0); issue_v(i, 0, 0); }
  if constexpr (QB == 2) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");            // the 8 Q pieces have landed
    const char* qreg = smem + FA_STAGE + wave * 8192;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int row = 16 * qb + c;
        qf[qb][ks] = *(const bf16x8*)(qreg + row * 256 + (((4 * ks + g) ^ (row & 15)) << 4));
      }
  }
  if (p.q_ss) fa16_prep_q<QB>(p, qf, b, h, q0, c, g);

  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const bool pre = t + 1 < nt;
    const int tn = pre ? t + 1 : t, stn = (t + 1) & 1;
    if (tn == last_ragged && pre) set_koff(tn);
    const char* sk = smem + (t & 1) * FA_STAGE;
    const char* sv = sk + FA_K_BYTES;

    // ---- S^T = K . Q^T: one group of 8 MFMAs per 16-key block (4 K fragments x 2 query blocks); the next block's
    // fragments are requested before this block's MFMAs issue; one K and one V^T piece of the next tile per group ----
    f32x4 s[2 * NKC][QB];
#pragma unroll
    for (int i = 0; i < 2 * NKC; ++i)
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[i][qb][j] = 0.f;
    {
      constexpr int NG = 2 * NKC;
      constexpr int KBUF = QB >= 3 ? 1 : 2;      // QB = 3: 240+ live registers - ONE fragment set, each register refilled as soon as its MFMAs have issued
      bf16x8 kfr[KBUF][4];
      auto load_k1 = [&](bf16x8& dst, int gi, int ks) __attribute__((always_inline)) {
        const int kb = (KS == 1 ? gi : 2 * kh + gi);
        const int row = fa16_keymap(kb, c);
        dst = *(const bf16x8*)(sk + row * 256 + (((4 * ks + g) ^ c) << 4));      // swizzle = fa16_kswz(row) = c
      };
      auto load_k = [&](bf16x8* dst, int gi) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) load_k1(dst[ks], gi, ks);
      };
      load_k(kfr[0], 0);
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        if (KBUF == 2 && gi + 1 < NG) load_k(kfr[(gi + 1) & 1], gi + 1);
        if (KS == 1) { issue_k(gi, tn, stn); issue_v(gi, tn, stn); }
        else { issue_k(2 * gi, tn, stn); issue_k(2 * gi + 1, tn, stn); issue_v(2 * gi, tn, stn); issue_v(2 * gi + 1, tn, stn); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
          for (int qb = 0; qb < QB; ++qb)
            s[gi][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[gi % KBUF][ks], qf[qb][ks], s[gi][qb], 0, 0, 0);
          if (KBUF == 1 && gi + 1 < NG) {
            __builtin_amdgcn_sched_barrier(0);
            load_k1(kfr[0][ks], gi + 1, ks);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    // V^T fragments of the first P.V group (d block 0): requested now, they land under the softmax
    bf16x8 vfr[2][NKC];
    auto load_v = [&](bf16x8* dst, int db) __attribute__((always_inline)) {
      const int d = db * 16 + c;
      const char* vrow = sv + d * 128;
      const int sw = (d >> 1) & 7;
#pragma unroll
      for (int i = 0; i < NKC; ++i) {
        const int kc = KS == 1 ? i : kh;
        dst[i] = *(const bf16x8*)(vrow + (((4 * kc + g) ^ sw) << 4));
      }
    };
    if (QB < 3) load_v(vfr[0], 0);                 // (QB = 3: no register to spare under the softmax - requested after it)
    __builtin_amdgcn_sched_barrier(0);
    // ---- mask the ragged key tail (last tile only) ----
    if (t == nt - 1 && (p.Tk & (FA_BK - 1)) != 0) {
#pragma unroll
      for (int i = 0; i < 2 * NKC; ++i) {
        const int kb = KS == 1 ? i : 2 * kh + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int key = t * FA_BK + 32 * (kb >> 1) + 8 * g + 4 * (kb & 1) + j;      // fa16_keymap(kb, 4g + j)
          if (key >= p.Tk) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) s[i][qb][j] = -1e30f;
          }
        }
      }
    }
    // ---- online softmax per query block (key axis = registers x lane groups g) ----
    // The deferral decision is taken per 16-row query block (one wave-uniform branch each), not per wave: a row's sequence of
    // offsets M then depends only on its own MFMA block - the same 16 rows in every grid (tiles start at multiples of 64 rows) -
    // and not on which other blocks share its wave.  (exp2(fma(S, c, -M)) is 2^-M exp2(S c) only up to the fp32 rounding of the
    // fma, which depends on M: with a per-wave decision the 192-row and the 128-row forms differed by one bf16 ulp in a handful
    // of rows whenever a wave-mate's spike raised M early.)
    float mxc[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float mx = s[0][qb][0];
#pragma unroll
      for (int i = 0; i < 2 * NKC; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mx = fmaxf(mx, s[i][qb][j]);
      mxc[qb] = mx * p.c;                                   // this lane's keys only: enough for the deferral test (see fa_body)
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      if (__any(mxc[qb] - mc_run[qb] > FA_DEFER)) {
        asm volatile("" ::: "memory");
        const float m_new = fmaxf(mc_run[qb], __builtin_ceilf(lane_xor32_max(lane_xor16_max(mxc[qb]))));
        const float alpha = __builtin_amdgcn_exp2f(mc_run[qb] - m_new);
        mc_run[qb] = m_new;
        l_run[qb] *= alpha;
#pragma unroll
        for (int db = 0; db < 8; ++db)
#pragma unroll
          for (int j = 0; j < 4; ++j) o[db][qb][j] *= alpha;
      }
    }
    bf16x8 pb[NKC][QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const float mc = mc_run[qb];
      float psum = 0.f;
#pragma unroll
      for (int i = 0; i < 2 * NKC; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i][qb][j], p.c, -mc));
          psum += e;
          pb[i >> 1][qb][(i & 1) * 4 + j] = (bf16)e;
        }
      l_run[qb] += psum;
    }
    // ---- O^T += V^T . P^T: one group per 16-channel block (2 chunks x 2 query blocks), next block's fragments first ----
    if (QB >= 3) load_v(vfr[0], 0);
#pragma unroll
    for (int db = 0; db < 8; ++db) {
      if (db + 1 < 8) load_v(vfr[(db + 1) & 1], db + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NKC; ++i)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
          o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[db & 1][i], pb[i][qb], o[db][qb], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  float l_tot[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) l_tot[qb] = lane_xor32_sum(lane_xor16_sum(l_run[qb]));
  if constexpr (KS == 2) {
    static_assert(QB == 2, "the key-split tail workgroups are built for two query blocks per wave");
    // ---- merge the key halves of each wave pair through LDS (the K / V^T ring is dead now) ----
    __syncthreads();
    float* xo = (float*)(smem + (wave >> 1) * (17 * 1024));
    if (kh == 1) {
#pragma unroll
      for (int db = 0; db < 8; ++db)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
          for (int j = 0; j < 4; ++j) xo[((db * 2 + qb) * 4 + j) * 64 + lane] = o[db][qb][j];
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) { xo[4096 + qb * 128 + lane] = mc_run[qb]; xo[4096 + qb * 128 + 64 + lane] = l_tot[qb]; }
    }
    __syncthreads();
    if (kh == 1) return;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const float m1 = xo[4096 + qb * 128 + lane], l1 = xo[4096 + qb * 128 + 64 + lane];
      const float m = fmaxf(mc_run[qb], m1);
      const float a0 = __builtin_amdgcn_exp2f(mc_run[qb] - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
      l_tot[qb] = l_tot[qb] * a0 + l1 * a1;
#pragma unroll
      for (int db = 0; db < 8; ++db)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[db][qb][j] = o[db][qb][j] * a0 + xo[((db * 2 + qb) * 4 + j) * 64 + lane] * a1;
    }
  }
  // ---- epilogue: O[q][d] = O^T[d][q] / l through a wave-private 4*QB KiB LDS image, whole 256-byte rows out ----
  if (KS == 1) __syncthreads();
  char* stg = smem + (KS == 1 ? wave * (4096 * QB) : 2 * 17 * 1024 + (wave >> 1) * 8192);
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float inv = 1.0f / l_tot[qb];
    const int row = 16 * qb + c;
#pragma unroll
    for (int db = 0; db < 8; ++db) {
      bf16x4v v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (bf16)(o[db][qb][j] * inv);
      // channels 16db + 4g .. +3 of query row `row`: 16-byte chunk 2db + (g >> 1), half (g & 1)
      *(bf16x4v*)(stg + row * 256 + (((2 * db + (g >> 1)) ^ (row & 15)) << 4) + (g & 1) * 8) = v;
    }
  }
  {
    bf16* ob = p.out + ((size_t)b * p.Tq) * p.ldo + h * FA_DH + (lane & 15) * 8;
#pragma unroll
    for (int i = 0; i < 4 * QB; ++i) {
      const int row = i * 4 + (lane >> 4);
      const bf16x8 v = *(const bf16x8*)(stg + row * 256 + (((lane & 15) ^ (row & 15)) << 4));
      if (q0 + row < p.Tq) *(bf16x8*)(ob + (size_t)(q0 + row) * p.ldo) = v;
    }
  }
}

__global__ __launch_bounds__(256, 2) void flash_attn16_kernel(FaParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bh, qt;
  if ((int)blockIdx.x < p.n_full) {
    fa_map(p, blockIdx.x, bh, qt);
    fa_body16<1>(p, smem, bh, qt * 128 + wave * FA_QW, 0, wave, lane);
  } else {
    const int ti = blockIdx.x - p.n_full;
    const int tile = p.n_full + ((p.rem & 7) == 0 ? ti % p.rem : ti >> 1);
    const int half = (p.rem & 7) == 0 ? ti / p.rem : ti & 1;
    fa_map(p, tile, bh, qt);
    fa_body16<2>(p, smem, bh, qt * 128 + half * 64 + (wave >> 1) * FA_QW, wave & 1, wave, lane);
  }
}

// Query blocks per wave (round 4).  With 16-row MFMA blocks a wave can own two blocks (32 rows, 128-row workgroups: the kernel
// above) or three (48 rows, 192-row workgroups): every K / V^T fragment read then feeds three MFMAs instead of two, and a workgroup
// stages its K / V^T tiles for 1.5x the rows - measured 12 % cheaper per query row at equal fill (one full round of 512 workgroups:
// Tq=1536 in 60.7 us against Tq=1024 in 46.5 us at Tk=1280, B=2, H=32; 5184^2: 702 against 781 us = 1.25 PF/s;
// profiles/r04_attn_qb_ab.log), at 256 registers per lane with one K fragment set instead of two.  Four blocks need more than the
// 256 registers two waves per SIMD leave each (o alone is 128).  What a grid of 192-row tiles loses is balance: N=1280 is 6.67 of
// them.  So ONE launch mixes the two sizes: per (batch, head) `qta` tiles of 192 rows followed by `qtb` tiles of 128 rows, chosen by
// fa_pick_mix below; all 192-row workgroups come first in the grid, so that a CU that is handed two workgroups at a time gets its
// share of each kind (N=1280, B*H=64: 4 x (192 + 128) rows = 256 + 256 workgroups = one of each per CU = 80 rows per SIMD, the
// chip's exact share; placement is the dispatcher's and affects speed only).  No key split anywhere: a row's result does not
// depend on the batch it is launched in (the bits of LTXK_ATTN_NO_TAIL_SPLIT).
constexpr int FA_LDS_MIX = FA_LDS;                   // two workgroups per CU (the 192-row form's output staging, 4 x 12 KiB, reuses the dead ring)
__global__ __launch_bounds__(256, 2) void flash_attn16_mix_kernel(FaParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int bh, qt;
  if ((int)blockIdx.x < p.n_a) {
    p.QT = p.qta;
    fa_map(p, blockIdx.x, bh, qt);
    fa_body16<1, 3>(p, smem, bh, qt * 192 + wave * 48, 0, wave, lane);
  } else {
    p.QT = p.qtb;
    fa_map(p, blockIdx.x - p.n_a, bh, qt);
    fa_body16<1, 2>(p, smem, bh, p.qta * 192 + qt * 128 + wave * 32, 0, wave, lane);
  }
}

// (qta, qtb) for Tq rows per (batch, head): qtb in 0..6 tiles of 128 rows, the rest in 192-row tiles, minimising the busiest CU's
// load in a model of the launch - workgroups dealt in grid order to the least-loaded of `cus` CUs, a CU's time = the summed cost
// of its workgroups (two co-resident workgroups share its matrix pipe), cost = rows (x 0.875 for the 192-row form, measured) +
// a fixed part (prologue, Q staging, output tail ~ 2 key tiles' worth).  Returns the model's makespan.
static long fa_pick_mix(int Tq, int Tk, int BH, int cus, int& qta, int& qtb) {
  long best = -1;
  if (cus > 1024) cus = 1024;
  const long fixed = 8L * 128 * 128 / (Tk < 128 ? 128 : Tk);    // in rows x 8: the fixed part of a workgroup ~ 128 keys' worth of a 128-row tile
  const long ca = 192 * 7 + fixed, cb = 128 * 8 + fixed;        // 0.875 per row of a 192-row tile
  for (int nb = 0; nb <= 6; ++nb) {
    const int rest = Tq - 128 * nb;
    if (rest <= -128) break;                                     // more 128-row tiles than rows
    const int na = rest > 0 ? (rest + 191) / 192 : 0;
    long load[1024];
    const long wa = (long)na * BH, wb = (long)nb * BH;
    for (int c = 0; c < cus; ++c) load[c] = (wa / cus + (c < wa % cus ? 1 : 0)) * ca;      // 192-row workgroups, round-robin
    for (long i = 0; i < wb; ++i) {                              // 128-row workgroups onto the least-loaded CU
      int m = 0;
      for (int c = 1; c < cus; ++c)
        if (load[c] < load[m]) m = c;
      load[m] += cb;
    }
    long span = 0;
    for (int c = 0; c < cus; ++c) span = load[c] > span ? load[c] : span;
    if (best < 0 || span < best) { best = span; qta = na; qtb = nb; }
  }
  return best;
}

}  // namespace ltxk

extern "C" int ltxk_flash_attn(const ltxk_attn_args* a, void* stream) {
  using namespace ltxk;
  LTXK_CHECK_ARG(a != nullptr, "ltxk_flash_attn: null args");
  const void *q = a->q, *k = a->k, *vt = a->vt;
  void* out = a->out;
  const int32_t ldq = a->ldq, ldk = a->ldk, ldvt = a->ldvt, ldo = a->ldo, B = a->B, H = a->H, Tq = a->Tq, Tk = a->Tk;
  const float scale = a->scale;
  LTXK_CHECK_ARG(q && k && vt && out, "ltxk_flash_attn_bf16: null pointer");
  LTXK_CHECK_ARG(B > 0 && H > 0 && Tq > 0 && Tk > 0, "ltxk_flash_attn_bf16: bad dims");
  LTXK_CHECK_ARG(scale > 0.f, "ltxk_flash_attn_bf16: scale must be positive");
  LTXK_CHECK_ARG(ldq >= H * FA_DH && ldk >= H * FA_DH && ldo >= H * FA_DH, "ltxk_flash_attn_bf16: row strides < H*128");
  LTXK_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldo % 8 == 0, "ltxk_flash_attn_bf16: row strides must be multiples of 8");
  const int tk_pad = (Tk + FA_BK - 1) / FA_BK * FA_BK;
  LTXK_CHECK_ARG(ldvt >= tk_pad && ldvt % 8 == 0, "ltxk_flash_attn_bf16: ldvt=%d must be >= %d (Tk rounded up to 64) and a multiple of 8", ldvt, tk_pad);
  LTXK_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt) & 15) == 0 && ((uintptr_t)out & 15) == 0, "ltxk_flash_attn_bf16: misaligned pointer");
  if (a->q_sumsq) {
    LTXK_CHECK_ARG(a->q_norm_weight != nullptr, "ltxk_flash_attn: q_sumsq needs q_norm_weight");
    LTXK_CHECK_ARG(a->q_sumsq_n > 0 && a->q_sumsq_n % 8 == 0 && a->q_sumsq_n * 64 == H * FA_DH && a->q_sumsq_ld >= a->q_sumsq_n,
                   "ltxk_flash_attn: q_sumsq_n=%d must equal H*128/64 and be a multiple of 8", a->q_sumsq_n);
    LTXK_CHECK_ARG((((uintptr_t)a->q_sumsq | (uintptr_t)a->q_norm_weight) & 15) == 0 && a->q_sumsq_ld % 4 == 0, "ltxk_flash_attn: q_sumsq / q_norm_weight must be 16-byte aligned");
    LTXK_CHECK_ARG((a->cos == nullptr) == (a->sin == nullptr) && (((uintptr_t)a->cos | (uintptr_t)a->sin) & 15) == 0, "ltxk_flash_attn: cos and sin must both be set (16-byte aligned) or both NULL");
  }
  FaParams p;
  p.q = (const bf16*)q; p.k = (const bf16*)k; p.vt = (const bf16*)vt; p.out = (bf16*)out;
  p.ldq = ldq; p.ldk = ldk; p.ldvt = ldvt; p.ldo = ldo;
  p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk;
  p.c = scale * 1.4426950408889634f;
  p.q_ss = a->q_sumsq; p.q_ss_ld = a->q_sumsq_ld; p.q_ss_n = a->q_sumsq_n;
  p.q_w = (const bf16*)a->q_norm_weight; p.cosb = a->q_sumsq ? a->cos : nullptr; p.sinb = a->q_sumsq ? a->sin : nullptr; p.eps = a->eps;
  static thread_local int attr_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev != attr_dev) {
    hipError_t e = hipFuncSetAttribute((const void*)flash_attn_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, FA_LDS);
    if (e != hipSuccess) { ltxk_set_error("ltxk_flash_attn_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e)); return LTXK_ELAUNCH; }
    attr_dev = dev;
  }
  static thread_local int slots = 0;
  if (slots == 0) {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    slots = 2 * cus;
  }
  // Forms measured and removed again (numbers at B=2,H=32, Tq=Tk=1280 / 5184, round 1): an 8-wave ping-pong kernel with a
  // 4-deep 128 KiB ring 565 / 930 TF/s, a 5-wave 160-row form 554 / 580, a 48-KiB 3-workgroup form ~430, against 630 / 945
  // for this one.  Round 2 (same box, interleaved rounds; this kernel 52.6 us at Tq=1024,Tk=1280 and 864 us at 5184^2):
  // a 64-rows-per-wave, one-wave-per-SIMD kernel whose MFMA gaps carry the other 32-row block's softmax (accumulators
  // pinned by inline-asm MFMAs; git 25a0404, LTXK_FA_VARIANT=64) 57.5 / 897 us (798 vs 834 us at 5120^2, where its
  // 256-row tiles fill whole rounds); a 32-row kernel pipelined across 32-key halves 50.5 / 853 us; a fixed s_setprio
  // for the odd wave slot of each SIMD: no change.  PMC of the 64-row form: 35 issue cycles per MFMA (4.7 VALU + 1 LDS
  // read beside it), 18 more parked or stalled - the softmax placement is not what holds the loop at ~55 % MFMA-busy.
  // Packed fp32 softmax arithmetic (v_pk_fma_f32 / v_pk_add_f32, two scores per instruction): 79.1 against 73.8 us at 1280^2.
  // The short last round as a launch of its own with 8-wave workgroups (64 rows, 128-KiB ring, each 32-row group's four waves
  // splitting the keys four ways - halves of every tile x even / odd tiles): 80.1 against 73.4 us at 1280^2, 65.5 against 58.2 at
  // 1280x1024 - and without any tail split the launch takes 74.9 us: two workgroups on a CU share its matrix pipe, so the
  // short round's lone workgroups already run nearly twice as fast, and there is little left for a finer split to win.
  // Round 3: the 16x16x32 formulation (fa_body16) - first form equal at 5184^2 (834.7 vs 834.5 us: 13 % higher held clock spent
  // on 18 % more VALU issue), after dropping the cross-lane exchange from the deferral test and the packed fp32 adds 795 vs 830 us
  // at 5184^2 and 67.4 vs 70.1 us at 1280^2: it ships (profiles/r03_attn_*).  LTXK_FA_XCD={1,0}, LTXK_FA_MFMA={16,32} remain in the A/B build.
  const int xcd_map = LTXK_AB_INT("LTXK_FA_XCD", 1);
  p.QT = (Tq + 127) / 128;
  p.xcd = (xcd_map && (B * H) % 8 == 0) ? 1 : 0;
  // With the tail split on, a tile in the short round sums its keys in a different order than the same rows would in a
  // launch without a short round (e.g. B=1 vs B=2), so batching changes low-order bits; LTXK_ATTN_NO_TAIL_SPLIT in
  // args->flags restores batch-invariant results (an explicit ABI field because it changes output bits).
  const int split = (a->flags & LTXK_ATTN_NO_TAIL_SPLIT) ? 0 : 1;
  const int tiles = p.QT * B * H;
  p.n_full = tiles; p.rem = 0;
  if (split && tiles % slots != 0) {
    // r tiles in the short last round.  r <= slots/2: split them all (2r workgroups).  r > slots/2 (round 3; e.g. one
    // B=1 forward of the CFG-pair split: 320 tiles for 512 slots): split the slots - r tiles that fill the round exactly -
    // 2 (slots - r) half workgroups + (2r - slots) whole ones = slots workgroups, no CU left with two whole tiles beside
    // CUs that have one.
    const int r = tiles % slots;
    p.rem = 2 * r <= slots ? r : (LTXK_AB_INT("LTXK_FA_FILL", 1) ? slots - r : 0);
    p.n_full = tiles - p.rem;
  }
  p.n_a = 0; p.qta = 0; p.qtb = 0;
  // More than one 128-row tile per (batch, head): the mixed 192 / 128-row grid (flash_attn16_mix_kernel).  A/B build: LTXK_FA_QB=2
  // keeps the 128-row kernel with its tail split everywhere, 3 forces 192-row tiles only.
  // Which grid (same box, interleaved, B*H = 32 / 64 / 128; profiles/r04_attn_qb_ab.log): from 1.25 rounds of 128-row tiles up the
  // mixed grid wins by 8-15 % (1280^2 B=2: 58.8 -> 50.0 us, 1296^2: 68.8 -> 59.3, 5184^2: 781 -> 701, B=4 1280^2: 113.5 -> 102.4); below
  // that the chip has fewer workgroups than slots, a CU runs one workgroup - one wave per SIMD, ~0.8 of the shared rate - and the
  // 128-row kernel's key-split tail, which doubles the workgroups, is the faster form (B=1 1280^2: 33.7 against 37.4 us for 192-row tiles).
  const int qb_env = LTXK_AB_INT("LTXK_FA_QB", 0);
  const long tiles128 = (long)((Tq + 127) / 128) * B * H;
  if (Tq > 128 && qb_env != 2 && (qb_env == 3 || 4 * tiles128 >= 5L * slots) && LTXK_AB_INT("LTXK_FA_MFMA", FA_DEFAULT_MFMA) == 16) {
    static thread_local int attr_devm = -1;
    if (dev != attr_devm) {
      hipError_t e = hipFuncSetAttribute((const void*)flash_attn16_mix_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FA_LDS_MIX);
      if (e != hipSuccess) { ltxk_set_error("ltxk_flash_attn_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e)); return LTXK_ELAUNCH; }
      attr_devm = dev;
    }
    static thread_local int c_tq = 0, c_tk = 0, c_bh = 0, c_qta = 0, c_qtb = 0;      // last two decisions (a forward repeats two shapes)
    static thread_local int c2_tq = 0, c2_tk = 0, c2_bh = 0, c2_qta = 0, c2_qtb = 0;
    int qta, qtb;
    if (qb_env == 3) { qta = (Tq + 191) / 192; qtb = 0; }
    else if (Tq == c_tq && Tk == c_tk && B * H == c_bh) { qta = c_qta; qtb = c_qtb; }
    else if (Tq == c2_tq && Tk == c2_tk && B * H == c2_bh) { qta = c2_qta; qtb = c2_qtb; }
    else {
      fa_pick_mix(Tq, Tk, B * H, slots / 2, qta, qtb);
      c2_tq = c_tq; c2_tk = c_tk; c2_bh = c_bh; c2_qta = c_qta; c2_qtb = c_qtb;
      c_tq = Tq; c_tk = Tk; c_bh = B * H; c_qta = qta; c_qtb = qtb;
    }
    p.qta = qta; p.qtb = qtb; p.n_a = qta * B * H;
    p.n_full = (qta + qtb) * B * H; p.rem = 0;
    hipLaunchKernelGGL(flash_attn16_mix_kernel, dim3((unsigned)p.n_full), dim3(256), FA_LDS_MIX, (hipStream_t)stream, p);
    LTXK_CHECK_LAUNCH("ltxk_flash_attn_bf16");
    return LTXK_OK;
  }
  const dim3 grid((unsigned)(p.n_full + 2 * p.rem));
  if (LTXK_AB_INT("LTXK_FA_MFMA", FA_DEFAULT_MFMA) == 16) {
    static thread_local int attr_dev16 = -1;
    if (dev != attr_dev16) {
      hipError_t e = hipFuncSetAttribute((const void*)flash_attn16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FA_LDS);
      if (e != hipSuccess) { ltxk_set_error("ltxk_flash_attn_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e)); return LTXK_ELAUNCH; }
      attr_dev16 = dev;
    }
    hipLaunchKernelGGL(flash_attn16_kernel, grid, dim3(256), FA_LDS, (hipStream_t)stream, p);
    LTXK_CHECK_LAUNCH("ltxk_flash_attn_bf16");
    return LTXK_OK;
  }
  hipLaunchKernelGGL(flash_attn_kernel<4>, grid, dim3(256), FA_LDS, (hipStream_t)stream, p);
  LTXK_CHECK_LAUNCH("ltxk_flash_attn_bf16");
  return LTXK_OK;
}

extern "C" int ltxk_flash_attn_bf16(const void* q, int32_t ldq, const void* k, int32_t ldk,
                                    const void* vt, int32_t ldvt, void* out, int32_t ldo,
                                    int32_t B, int32_t H, int32_t Tq, int32_t Tk, float scale,
                                    void* stream) {
  ltxk_attn_args a = {};
  a.q = q; a.k = k; a.vt = vt; a.out = out;
  a.ldq = ldq; a.ldk = ldk; a.ldvt = ldvt; a.ldo = ldo;
  a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.scale = scale;
  return ltxk_flash_attn(&a, stream);
}

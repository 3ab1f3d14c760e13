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
//                             B operand of the second product: no LDS round trip for P; the
//                             V^T fragment is read with the matching key permutation)
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
};

__device__ __forceinline__ void fa_glds16(const void* g, void* l) { glds16(g, l); }

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));

// NW waves per workgroup (4 or 5): 128 or 160 query rows.  Two workgroups per CU are resident
// (64 KiB LDS each), so the grid should be a whole number of 512-slot rounds: at Tq=1280, B*H=64 the
// 160-row form gives exactly 512 workgroups where the 128-row form gives 640 (a 25 % second round).
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void flash_attn_kernel(FaParams p) {
  constexpr int FA_BQ = FA_QW * NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int bh = blockIdx.y;
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = blockIdx.x * FA_BQ + wave * FA_QW;

  // ---- Q fragments (B operand: lane (q=r, half hh) holds d = 16*ks + 8*hh + j) ----
  int qrow = q0 + r;
  qrow = qrow < p.Tq ? qrow : p.Tq - 1;
  const bf16* qp = p.q + ((size_t)b * p.Tq + qrow) * p.ldq + h * FA_DH + hh * 8;
  bf16x8 qf[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);

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
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const bool pre = t + 1 < nt;
    const int tn = pre ? t + 1 : t, stn = (t + 1) & 1;   // last tile: harmless re-load into the free slot
    const char* sk = smem + (t & 1) * FA_STAGE;
    const char* sv = sk + FA_K_BYTES;

    // ---- S^T = K . Q^T ; the next tile's K pieces are issued between the MFMAs (not as a burst) ----
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int j = 0; j < 16; ++j) s[kb][j] = 0.f;
      const int row = kb * 32 + r;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int chunk = ks * 2 + hh;
        const bf16x8 kf = *(const bf16x8*)(sk + row * 256 + ((chunk ^ (row & 15)) << 4));
        if ((ks & 3) == 1) issue_k(kb * 2 + (ks >> 2), tn, stn);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
      }
    }
    // ---- mask the ragged key tail (last tile only) ----
    if (t == nt - 1 && (p.Tk & (FA_BK - 1)) != 0) {
      const int kbase_i = t * FA_BK + 4 * hh;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int key = kbase_i + kb * 32 + (j & 3) + 8 * (j >> 2);
          if (key >= p.Tk) s[kb][j] = -1e30f;
        }
    }
    // ---- online softmax (key axis = registers + lane^32) ----
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int j = 0; j < 16; ++j) mx = fmaxf(mx, s[kb][j]);
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

    // ---- O^T += V^T . P^T ; V^T fragment element j of half hh = key 16s + 8(j>>2) + 4hh + (j&3)
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      const int d = db * 32 + r;
      const char* vrow = sv + d * 128 + hh * 8;
      const int sw = (d >> 1) & 7;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
          const int c0 = kb * 4 + 2 * sidx;
          const bf16x4v lo = *(const bf16x4v*)(vrow + ((c0 ^ sw) << 4));
          const bf16x4v hi = *(const bf16x4v*)(vrow + (((c0 + 1) ^ sw) << 4));
          bf16x8 vf;
          vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
          vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
          if (kb == 0 && sidx == 1) issue_v(db, tn, stn);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[kb][sidx], o[db], 0, 0, 0);
        }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: O[q][d] = O^T[d][q] / l ----
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


// ---------------------------------------------------------------------------------------
// 48-KiB variant: K tiles double-buffered (2 x 16 KiB), the V^T tile single-buffered (16 KiB) and
// re-filled right after the workgroup has finished P.V of the previous tile, so its DMA flies under
// this tile's QK^T + softmax.  Three workgroups (12 waves) fit a CU instead of two: at Tq=1280 and
// B*H=64 the 640 workgroups then run in one wave of 768 slots instead of 1.25 rounds of 512, and the
// third wave per SIMD gives the scheduler more to overlap softmax VALU with another wave's MFMAs.
// ---------------------------------------------------------------------------------------
constexpr int FA2_LDS = 2 * FA_K_BYTES + FA_V_BYTES;

__global__ __launch_bounds__(256, 3) void flash_attn_kernel_v2(FaParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int bh = blockIdx.y;
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = blockIdx.x * (FA_QW * NW) + wave * FA_QW;

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
  char* const sv = smem + 2 * FA_K_BYTES;

  auto issue_k = [&](int t) __attribute__((always_inline)) {   // 4 pieces per wave
    char* sk = smem + (t & 1) * FA_K_BYTES;
    const int key0 = t * FA_BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = wave * 4 + i;
      const int row = piece * 4 + k_lrow;
      int key = key0 + row;
      key = key < p.Tk ? key : p.Tk - 1;
      fa_glds16(kbase + (size_t)key * p.ldk + (k_slot ^ (row & 15)) * 8, sk + piece * 1024);
    }
  };
  auto issue_v = [&](int t) __attribute__((always_inline)) {   // 4 pieces per wave
    const int key0 = t * FA_BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = wave * 4 + i;
      const int d = piece * 8 + v_lrow;
      fa_glds16(vbase + (size_t)d * p.ldvt + key0 + (v_slot ^ ((d >> 1) & 7)) * 8, sv + piece * 1024);
    }
  };

  f32x16 o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) o[i][j] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int nt = (p.Tk + FA_BK - 1) / FA_BK;
  issue_k(0);
  for (int t = 0; t < nt; ++t) {
    // (A) everyone is done with P.V(t-1): V buffer and K buffer (t+1)&1 are free; K(t) has landed
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    issue_v(t);
    if (t + 1 < nt) issue_k(t + 1);
    const char* sk = smem + (t & 1) * FA_K_BYTES;

    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int j = 0; j < 16; ++j) s[kb][j] = 0.f;
      const int row = kb * 32 + r;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int chunk = ks * 2 + hh;
        const bf16x8 kf = *(const bf16x8*)(sk + row * 256 + ((chunk ^ (row & 15)) << 4));
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
      }
    }
    if (t == nt - 1 && (p.Tk & (FA_BK - 1)) != 0) {
      const int kbase_i = t * FA_BK + 4 * hh;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int key = kbase_i + kb * 32 + (j & 3) + 8 * (j >> 2);
          if (key >= p.Tk) s[kb][j] = -1e30f;
        }
    }
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int j = 0; j < 16; ++j) mx = fmaxf(mx, s[kb][j]);
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

    // (B) V(t) has landed in every wave's share (the K(t+1) pieces, issued after it, may still fly)
    if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      const int d = db * 32 + r;
      const char* vrow = sv + d * 128 + hh * 8;
      const int sw = (d >> 1) & 7;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
          const int c0 = kb * 4 + 2 * sidx;
          const bf16x4v lo = *(const bf16x4v*)(vrow + ((c0 ^ sw) << 4));
          const bf16x4v hi = *(const bf16x4v*)(vrow + (((c0 + 1) ^ sw) << 4));
          bf16x8 vf;
          vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
          vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[kb][sidx], o[db], 0, 0, 0);
        }
    }
  }

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
  LTXK_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldo % 4 == 0, "ltxk_flash_attn_bf16: row strides must be multiples of 8");
  const int tk_pad = (Tk + FA_BK - 1) / FA_BK * FA_BK;
  LTXK_CHECK_ARG(ldvt >= tk_pad && ldvt % 8 == 0, "ltxk_flash_attn_bf16: ldvt=%d must be >= %d (Tk rounded up to 64) and a multiple of 8", ldvt, tk_pad);
  LTXK_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt) & 15) == 0 && ((uintptr_t)out & 7) == 0, "ltxk_flash_attn_bf16: misaligned pointer");
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
  // Variants measured at B=2,H=32,Tq=Tk=1280 (scripts/prof_attn.py): 4-wave/64 KiB (default) 538 TF/s;
  // 5-wave/160-row form (removes the partial second round, but 10 waves per CU load the SIMDs 3,3,2,2)
  // ~12 % slower; 48-KiB/3-workgroup form (v2: register-capped at 168, spills) 430 TF/s.
  // LTXK_FA_VARIANT={4,5,2} selects one for A/B runs.
  static const int variant = [] { const char* e = getenv("LTXK_FA_VARIANT"); return e ? atoi(e) : 4; }();
  if (variant == 2) {
    static thread_local int attr2_dev = -1;
    if (dev != attr2_dev) {
      hipError_t e = hipFuncSetAttribute((const void*)flash_attn_kernel_v2, hipFuncAttributeMaxDynamicSharedMemorySize, FA2_LDS);
      if (e != hipSuccess) { ltxk_set_error("ltxk_flash_attn_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e)); return LTXK_ELAUNCH; }
      attr2_dev = dev;
    }
    dim3 grid((Tq + 127) / 128, B * H);
    hipLaunchKernelGGL(flash_attn_kernel_v2, grid, dim3(256), FA2_LDS, (hipStream_t)stream, p);
  } else if (variant == 5) {
    dim3 grid((Tq + 159) / 160, B * H);
    hipLaunchKernelGGL(flash_attn_kernel<5>, grid, dim3(320), FA_LDS, (hipStream_t)stream, p);
  } else {
    dim3 grid((Tq + 127) / 128, B * H);
    hipLaunchKernelGGL(flash_attn_kernel<4>, grid, dim3(256), FA_LDS, (hipStream_t)stream, p);
  }
  LTXK_CHECK_LAUNCH("ltxk_flash_attn_bf16");
  return LTXK_OK;
}

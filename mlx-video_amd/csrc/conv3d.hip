// 3x3x3 stride-1 convolution of the video VAE as an implicit GEMM on MFMA: ltxk_conv3d_k3_bf16.
// Replaces nn.Conv3d inside CausalConv3d (video_vae/convolution.py:111-118,161) including its
// halo handling (convolution.py:120-157): temporal pad by frame replication (causal: two copies
// of the first frame; non-causal: first + last), spatial pad reflect (decoder) or zeros
// (encoder).  No padded copy of the volume is ever materialised: the halo is resolved in the
// per-lane SOURCE address of the LDS-DMA tile loads (a row gather), zero padding reads a
// caller-provided zero page.
//
//   out[v, co] = bias[co] + sum_{tap, ci} x[src(v, tap), ci] * w[co, tap, ci]
//   M = B*D*H*W voxels (channels-last rows), N = Cout, K = 27*Cin walked tap-major in 64-wide
//   steps; same 3-stage LDS ring / MFMA loop as the Linear GEMM (gemm_core.h).
#include "gemm_core.h"
#include <stdlib.h>
#include <utility>

namespace ltxk {

struct ConvParams {
  const bf16* x; const bf16* w; const bf16* bias; bf16* out; const bf16* resid; const bf16* zero;
  int B, D, H, W, Cin, Cout;
  int causal, pad_mode;
  int M, RT, CT, cpb;   // cpb = Cin/64 K-steps per tap
  float* slab;          // split-K: fp32 partial slabs [S][M][Cout] (plain stores, summed in slice order)
  int S, kper;          // K slices per tile, K-steps per slice
  int ntaps, kd0;       // 27 taps from kd=0, or 9 taps at kd0=1 (per-frame 3x3 kernel)
  // fused PixelNorm (+ modulation) + SiLU of the output row (needs Cout == BN: the tile holds whole rows)
  bf16* act_out; const bf16* act_scale; const bf16* act_shift; float act_eps; int act_silu; int rows_per_batch;
  int xcd_order;        // A/B switch (LTXK_CONV_XCD, default 1)
  int m_base;           // first voxel row of this launch (the short last round runs as a second launch of half-height tiles)
};

template <int TT, int WN, bool RES, bool SPLIT>
__global__ __launch_bounds__(GEMM_THREADS) void conv3d_k3_kernel(ConvParams p) {
  using G = GemmGeom<TT, WN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware tile order (workgroups are dealt round-robin over the 8 XCDs): each XCD takes a CONTIGUOUS run of tiles, so
  // the tiles in flight on one XCD are spatial neighbours and share their halo rows (4 image rows read per 2 written,
  // x3 frames) and the weights through that XCD's L2 instead of each XCD fetching every halo itself.  Bijective for any
  // tile count: XCD x owns q + (x < r) tiles, q = tiles / 8, r = tiles % 8.
  int tile, slice = 0;
  if constexpr (SPLIT) {
    tile = blockIdx.x / p.S;
    slice = blockIdx.x - tile * p.S;
  } else {
    const int nt_all = p.RT * p.CT, q = nt_all >> 3, r = nt_all & 7;
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    tile = p.xcd_order ? x * q + (x < r ? x : r) + j : (int)blockIdx.x;
  }
  const int ct = tile % p.CT, rt = tile / p.CT;
  const int m0 = p.m_base + rt * G::BM, n0 = ct * G::BN;
  const int K = p.ntaps * p.Cin;

  const int lrow = lane >> 3;
  const int chunk = (lane & 7) ^ lrow;
  const bf16* wptr[G::W_PER_WAVE];
#pragma unroll
  for (int i = 0; i < G::W_PER_WAVE; ++i) {
    int r = n0 + (wave * G::W_PER_WAVE + i) * 8 + lrow;
    r = r < p.Cout ? r : p.Cout - 1;
    wptr[i] = p.w + (size_t)r * K + chunk * 8;
  }
  // uniform piece count per wave (see gemm.hip): waves past A_REM re-issue their last A piece
  const int nA = G::A_BASE + (wave < G::A_REM ? 1 : 0);
  const int a0 = wave * G::A_BASE + (wave < G::A_REM ? wave : G::A_REM);
  // The halo is separable: byte offset of (voxel, tap) = od[kd] + oh[kh] + ow[kw].  The nine per-piece
  // partial offsets are computed once; per tap the source address costs a few selects and adds instead
  // of re-deriving clamp/reflect/zero for every 64-channel K-step (the 128-channel stage would otherwise
  // be bound by address VALU, not by MFMA).  Offsets are 32-bit (volume < 4 GiB, checked on the host).
  unsigned od[G::MAXA][3], oh[G::MAXA][3], ow[G::MAXA][3];
  unsigned zbits[G::MAXA];
  int adst[G::MAXA];
  const unsigned rowB = (unsigned)p.Cin * 2u;
#pragma unroll
  for (int i = 0; i < G::MAXA; ++i) {
    const int pi = nA > 0 ? a0 + (i < nA ? i : nA - 1) : G::A_PIECES - 1;   // no own piece: re-issue the tile's last one
    int r = m0 + pi * 8 + lrow;
    r = r < p.M ? r : p.M - 1;
    const int vw = r % p.W; r /= p.W;
    const int vh = r % p.H; r /= p.H;
    const int vd = r % p.D;
    const int vb = r / p.D;
    adst[i] = G::W_STAGE_BYTES + (pi < G::A_PIECES ? pi : G::A_PIECES - 1) * 1024;
    unsigned zb = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int d = vd + k - (p.causal == 1 ? 2 : 1);
      if (p.causal == 2) {                                 // plain zero padding in time (latent upsampler)
        if (d < 0 || d >= p.D) { zb |= 64u << k; d = 0; }
      } else {
        d = d < 0 ? 0 : (d >= p.D ? p.D - 1 : d);          // frame replication
      }
      int h = vh + k - 1, w = vw + k - 1;
      if (p.pad_mode == LTXK_PAD_REFLECT) {
        h = h < 0 ? 1 : (h >= p.H ? p.H - 2 : h);
        w = w < 0 ? 1 : (w >= p.W ? p.W - 2 : w);
      } else {
        if (h < 0 || h >= p.H) { zb |= 1u << k; h = 0; }
        if (w < 0 || w >= p.W) { zb |= 8u << k; w = 0; }
      }
      od[i][k] = (unsigned)((vb * p.D + d) * p.H) * (unsigned)p.W * rowB;
      oh[i][k] = (unsigned)(h * p.W) * rowB;
      ow[i][k] = (unsigned)w * rowB + (unsigned)chunk * 16u;
    }
    zbits[i] = zb;
  }
  constexpr int PER_STAGE = G::W_PER_WAVE + G::MAXA;
  static_assert(PER_STAGE <= 7, "vmcnt immediates assume <= 7 pieces per stage");
  const char* zsrc = (const char*)p.zero + chunk * 16;
  const char* xb = (const char*)p.x;

  // per-tap source state of the prefetch stream
  unsigned off_tap[G::MAXA];
  bool ztap[G::MAXA];
  auto set_tap = [&](int tap) __attribute__((always_inline)) {
    const int kd = p.kd0 + tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    // 3-way selects as scalar bit masks (a ?: chain is turned into a scratch-memory table lookup by hipcc)
    const unsigned d0 = kd == 0 ? ~0u : 0u, d1 = kd == 1 ? ~0u : 0u, d2 = kd == 2 ? ~0u : 0u;
    const unsigned h0 = kh == 0 ? ~0u : 0u, h1 = kh == 1 ? ~0u : 0u, h2 = kh == 2 ? ~0u : 0u;
    const unsigned w0 = kw == 0 ? ~0u : 0u, w1 = kw == 1 ? ~0u : 0u, w2 = kw == 2 ? ~0u : 0u;
#pragma unroll
    for (int j = 0; j < G::MAXA; ++j) {
      const unsigned a = (od[j][0] & d0) | (od[j][1] & d1) | (od[j][2] & d2);
      const unsigned b = (oh[j][0] & h0) | (oh[j][1] & h1) | (oh[j][2] & h2);
      const unsigned c = (ow[j][0] & w0) | (ow[j][1] & w1) | (ow[j][2] & w2);
      off_tap[j] = a + b + c;
      ztap[j] = ((zbits[j] >> kh) & 1u) | ((zbits[j] >> (3 + kw)) & 1u) | ((zbits[j] >> (6 + kd)) & 1u);
    }
  };

  // i-th LDS-DMA piece of this wave for K-step kt (channel block cb of the current tap) into ring slot s
  auto issue_piece = [&](int i, int kt, int cb, int s) __attribute__((always_inline)) {
    char* base = smem + s * G::STAGE_BYTES;
    if (i < G::W_PER_WAVE) {
      glds16(wptr[i < G::W_PER_WAVE ? i : 0] + kt * GEMM_BK, base + (wave * G::W_PER_WAVE + i) * 1024);
    } else if (i < PER_STAGE) {
      const int j = i - G::W_PER_WAVE < G::MAXA ? i - G::W_PER_WAVE : 0;
      const char* src = ztap[j] ? zsrc : xb + (size_t)(off_tap[j] + (unsigned)cb * 128u);
      glds16(src, base + adst[j]);
    }
  };

  // Epilogue operands are prefetched under the main loop exactly as in gemm.hip (which see for the why and for the
  // compiler traps): the bias behind the first two stages' DMA, the residual tile in 16-row bands during the peeled
  // K-steps 2 .. 1+TT; counted vmcnt waits leave them in flight.
  constexpr bool HAS_RES = RES && !SPLIT;
  constexpr int NB = SPLIT ? 0 : 4;
  bf16x4 rres[HAS_RES ? TT : 1][4], bpre[4];
  auto load_res_band = [&](auto tt_c) __attribute__((always_inline)) {
    constexpr int tt = decltype(tt_c)::value;
    if constexpr (HAS_RES && tt < TT) {
      int m = m0 + wm * TT * 16 + tt * 16 + (lane & 15);
      m = m < p.M ? m : p.M - 1;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        int n = n0 + wn * 64 + nt * 16 + (lane >> 4) * 4;
        n = n < p.Cout ? n : p.Cout - 4;
        rres[tt][nt] = *opaque_gptr<bf16x4>((const bf16x4*)(p.resid + (size_t)m * p.Cout + n));
      }
    }
  };

  f32x4 acc[TT][4];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[tt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this workgroup's K-step range [kbeg, nk) (the whole K unless split)
  const int nk_all = p.ntaps * p.cpb;
  const int kbeg = SPLIT ? slice * p.kper : 0;
  int nk = SPLIT ? kbeg + p.kper : nk_all;
  nk = nk < nk_all ? nk : nk_all;
  // prologue: stages kbeg and kbeg+1
  int ptap = kbeg / p.cpb, pcb = kbeg - ptap * p.cpb;       // (tap, channel block) of the prefetch stream
  set_tap(ptap);
#pragma unroll
  for (int i = 0; i < PER_STAGE; ++i) issue_piece(i, kbeg, pcb, 0);
  if (kbeg + 1 < nk) {
    ++pcb;
    if (pcb == p.cpb) { pcb = 0; ++ptap; set_tap(ptap); }
  }
#pragma unroll
  for (int i = 0; i < PER_STAGE; ++i) issue_piece(i, kbeg + 1 < nk ? kbeg + 1 : kbeg, pcb, 1);
  if constexpr (!SPLIT) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      int n = n0 + wn * 64 + nt * 16 + (lane >> 4) * 4;
      n = n < p.Cout ? n : p.Cout - 4;
      bpre[nt] = *opaque_gptr<bf16x4>((const bf16x4*)(p.bias + n));
    }
  }
  int s = 0;
  MmaPipe<TT, WN, false> pipe;
  pipe.init();
  constexpr int PEEL = HAS_RES ? 3 + TT : 2;
  auto kstep = [&](int kt, auto kc) __attribute__((always_inline)) {
    constexpr int KC = decltype(kc)::value;                 // K-step index relative to kbeg in the peeled steps, -1 in the loop
    if constexpr (KC == 0 || KC == 1) wait_keep_and_barrier<PER_STAGE + NB>();
    else if constexpr (HAS_RES && KC >= 3 && KC < 3 + TT) wait_keep_and_barrier<PER_STAGE + 4>();
    else if constexpr (KC == -3) wait_keep_and_barrier<0>();   // last K-step: nothing younger than its own stage is in flight
    else wait_keep_and_barrier<PER_STAGE>();
    if constexpr (HAS_RES && KC >= 2 && KC < 2 + TT) load_res_band(IntC<(KC >= 2 ? KC - 2 : 0)>{});
    int s2 = s + 2;
    s2 = s2 >= 3 ? s2 - 3 : s2;
    // advance the prefetch stream to K-step kt+2 (held at the last K-step in the tail: harmless re-load)
    int kt2 = kt + 2;
    if (kt2 < nk) {
      ++pcb;
      if (pcb == p.cpb) { pcb = 0; ++ptap; set_tap(ptap); }
    } else {
      kt2 = nk - 1;
    }
    const int cb = pcb;
    // KC = -2 / -3: the last two K-steps of a range long enough to have them peeled request nothing (as in gemm.hip)
    pipe.step(smem + s * G::STAGE_BYTES, wm, wn, lane, acc, [&](int i) {
      if constexpr (KC != -2 && KC != -3) issue_piece(i, kt2, cb, s2);
    });
    s = s + 1 == 3 ? 0 : s + 1;
  };
  [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
    ((kbeg + I < nk ? kstep(kbeg + I, IntC<I>{}) : (void)0), ...);
  }(std::make_integer_sequence<int, PEEL>{});
  const bool tail = nk - kbeg >= PEEL + 2;
  const int nmain = tail ? nk - 2 : nk;
  for (int kt = kbeg + PEEL; kt < nmain; ++kt) kstep(kt, IntC<-1>{});
  if (tail) {
    kstep(nk - 2, IntC<-2>{});
    kstep(nk - 1, IntC<-3>{});
  }
  pipe.finish(acc);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (!SPLIT) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) asm volatile("" ::"v"(bpre[nt]));      // never dead, never dropped (see gemm.hip)
  }
  if constexpr (HAS_RES) {
    const int nsteps = nk - kbeg;
    if (nsteps < 2 + TT) {             // short K: the bands whose step never ran
      if (nsteps <= 2) load_res_band(IntC<0>{});
      if (nsteps <= 3) load_res_band(IntC<1>{});
      if (nsteps <= 4) load_res_band(IntC<2>{});
      if (nsteps <= 5) load_res_band(IntC<3>{});
      if (nsteps <= 6) load_res_band(IntC<4>{});
    }
  }

  // epilogue: acc[tt][nt][j]: voxel = lane&15, co = 4*(lane>>4) + j.  bf16 outputs go through a wave-private LDS
  // image (128-byte rows, chunks XOR-swizzled by row) and leave as whole 128-byte lines (see gemm.hip).
  const int nq = (lane >> 4) * 4;
  const bool wide = !SPLIT && (p.Cout & 7) == 0;
  const bool do_act = !SPLIT && p.act_out != nullptr, do_out = p.out != nullptr;
  char* stg = smem + wave * (TT * 16 * 128);
  if (wide) __syncthreads();
  // y of (tt, nt): the conv's bf16 output values (bias, residual), as floats
  auto out_vals = [&](int tt, int nt, float (&y)[4]) __attribute__((always_inline)) {
    const bf16x4 b = bpre[nt];
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = rbf(acc[tt][nt][j] + (float)b[j]);
    if constexpr (HAS_RES) {
      const bf16x4 r = rres[tt][nt];
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = rbf(y[j] + (float)r[j]);
    }
  };
  auto stage_or_store = [&](bf16* dst, int tt, int nt, int m, int n, bf16x4 o) __attribute__((always_inline)) {
    if (wide) {
      const int r = tt * 16 + (lane & 15), cg = lane >> 4;
      *(bf16x4*)(stg + r * 128 + (((nt * 2 + (cg >> 1)) ^ (r & 7)) << 4) + (cg & 1) * 8) = o;
    } else {
      *(bf16x4*)(dst + (size_t)m * p.Cout + n) = o;
    }
  };
  auto flush_lines = [&](bf16* dst) __attribute__((always_inline)) {
    if (!wide) return;
    const int c = lane & 7;
    const int n = n0 + wn * 64 + c * 8;
#pragma unroll
    for (int i = 0; i < TT * 2; ++i) {
      const int r = i * 8 + (lane >> 3);
      const int m = m0 + wm * TT * 16 + r;
      const bf16x8 v = *(const bf16x8*)(stg + r * 128 + ((c ^ (r & 7)) << 4));
      if (m < p.M && n < p.Cout) *(bf16x8*)(dst + (size_t)m * p.Cout + n) = v;
    }
  };
  float rowsq[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    rowsq[tt] = 0.f;
    const int m = m0 + wm * TT * 16 + tt * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wn * 64 + nt * 16 + nq;
      if (n >= p.Cout) continue;
      if constexpr (SPLIT) {
        *(f32x4*)(p.slab + ((size_t)slice * p.M + m) * p.Cout + n) = acc[tt][nt];
        continue;
      }
      float y[4];
      out_vals(tt, nt, y);
      if (do_act) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rowsq[tt] += rbf(y[j] * y[j]);          // PixelNorm's mean of squares, squares rounded as in ltxk_pixelnorm_act
      }
      if (do_out) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)y[j];
        stage_or_store(p.out, tt, nt, m, n, o);
      }
    }
  }
  if (do_out) flush_lines(p.out);
  if (do_act) {
    // Row statistic: the lane's 16 values -> the wave's 64 columns (4 lane groups) -> the tile's Cout columns (WN waves,
    // through LDS in wave order: deterministic).  The tile spans every channel (Cout == BN, checked on the host).
    float* part = (float*)(smem + 8 * (TT * 16 * 128));                       // behind the 8 staging images
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      float v = rowsq[tt];
      v = lane_xor16_sum(v);
      v = lane_xor32_sum(v);
      if (lane < 16) part[((wm * TT + tt) * 16 + lane) * WN + wn] = v;
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int m = m0 + wm * TT * 16 + tt * 16 + (lane & 15);
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < WN; ++w) tot += part[((wm * TT + tt) * 16 + (lane & 15)) * WN + w];
      const float mean = rbf(tot / (float)p.Cout);
      const float sd = rbf(sqrtf(rbf(mean + p.act_eps)));
      const float rsd = 1.0f / sd;
      if (m >= p.M) continue;
      const size_t mrow = p.act_scale ? (size_t)(m / p.rows_per_batch) * p.Cout : 0;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + wn * 64 + nt * 16 + nq;
        if (n >= p.Cout) continue;
        float y[4];
        out_vals(tt, nt, y);
        bf16x4 sc, sh;
        if (p.act_scale) {
          sc = *(const bf16x4*)(p.act_scale + mrow + n);
          sh = *(const bf16x4*)(p.act_shift + mrow + n);
        }
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float q0 = y[j] * rsd;                                          // x / sd: reciprocal + one Newton step (see ltxk_pixelnorm_act)
          float t = rbf(__builtin_fmaf(__builtin_fmaf(-q0, sd, y[j]), rsd, q0));
          if (p.act_scale) t = rbf(rbf(t * rbf(1.0f + (float)sc[j])) + (float)sh[j]);
          if (p.act_silu) t = silu_f(t);
          o[j] = (bf16)t;
        }
        stage_or_store(p.act_out, tt, nt, m, n, o);
      }
    }
    flush_lines(p.act_out);
  }
}

// split-K finalize: out = bf16(bf16(sum_s slab[s] + bias) [+ resid]), slabs summed in slice order (deterministic)
__global__ void conv_splitk_finalize_kernel(const float* __restrict__ slab, const bf16* __restrict__ bias,
                                            const bf16* __restrict__ resid, bf16* __restrict__ out, int M, int Cout, int S) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;        // one thread per 4 outputs
  const size_t total = (size_t)M * Cout / 4;
  if (idx >= total) return;
  const size_t e = idx * 4;
  const int n = (int)(e % Cout);
  f32x4 a = *(const f32x4*)(slab + e);
  for (int s = 1; s < S; ++s) {
    const f32x4 b = *(const f32x4*)(slab + (size_t)s * M * Cout + e);
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] += b[j];
  }
  const bf16x4 bs = *(const bf16x4*)(bias + n);
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float y = rbf(a[j] + (float)bs[j]);
    if (resid) y = y + (float)resid[e + j];
    o[j] = (bf16)y;
  }
  *(bf16x4*)(out + e) = o;
}

// =====================================================================================================================
// kw-reuse form (round 3).  The three taps kw = 0,1,2 of one (kd, kh) read the SAME image row shifted by one voxel, but the
// kernel above fetches the 256-row A tile afresh for each of the 27 taps.  Here a stage holds, for one (kd, kh) and one
// 32-channel block, a PANEL of the tile's voxel rows plus one halo voxel on either side of every image row the tile
// touches (tile row i of run r sits at panel row i + 2r + 1); tap kw reads panel rows shifted by kw.  A stage is then
//   A panel  (BM + 2*runs) x 64 B  =  17 KiB            (was 3 taps x 32 KiB per 64 channels: -65 % A bytes through LDS-DMA)
//   W        3 taps x 128 rows x 64 B  =  24 KiB
// and feeds 3 MFMA sub-steps (v_mfma_f32_16x16x32_bf16: 32 channels each) = 48 MFMAs per wave per barrier (was 32), with
// 41 KiB of LDS-DMA per stage against 48 KiB per 64-channel K-step: 0.85 KiB per MFMA instead of 1.5.  The MFMA power
// probe prices every 1-KiB LDS-DMA piece at ~30 matrix-pipe cycles plus clock.
// K is walked (kd, kh) -> 32-channel block -> kw: another summation order than the kernel above (same rounding points).
// LDS rows are 64 B; the 16-byte chunk index is XOR-ed with 2*((row >> 2) & 1): conflict-free ds_read_b128 for ANY 16
// consecutive rows (the panel rows of an MFMA block start at an arbitrary offset).
// Used for the 27-tap convolution when W >= 64 (at most BM/64 + 1 image rows per tile), no split-K, no fused activation.
// =====================================================================================================================
constexpr int KW_BN = 128;
template <int TT>
struct KwGeom {
  static constexpr int BM = 64 * TT;                 // 4 wave rows x TT MFMA row blocks
  static constexpr int RUNMAX = BM / 64 + 1;         // image rows a tile can touch when W >= 64
  static constexpr int PR = BM + 2 * RUNMAX;         // panel rows
  static constexpr int A_PIECES = (PR + 15) / 16;    // 1-KiB pieces of 16 rows x 64 B
  static constexpr int APW = (A_PIECES + 7) / 8;     // per wave (the last ones re-issue the final piece)
  static constexpr int A_BYTES = A_PIECES * 1024;
  static constexpr int W_BYTES = 3 * KW_BN * 64;
  static constexpr int STAGE = A_BYTES + W_BYTES;
  static constexpr int LDS = 3 * STAGE;
  static constexpr int PER_STAGE = 3 + APW;
};

// Rotated software pipeline of a stage with three MFMA sub-steps (kw = 0,1,2), after MmaPipe (gemm_core.h): the 3*TT MFMA
// groups (one A row block x 4 W column blocks) run in a fixed order pinned by sched_barrier; fragment ds_read_b128s are
// issued one group ahead of their use (flat order W0, A0[*], W1, A1[*], W2, A2[*]); the stage's LDS-DMA pieces are
// sprinkled one per group; and the last two groups of every stage are deferred until after the next barrier (their
// operands are in registers), so they execute while the first fragment reads of the new stage are in flight.
#ifndef LTXK_KW_PD
#define LTXK_KW_PD 1      // fragment reads run this many MFMA groups ahead of their use
#endif
#ifndef LTXK_KW_DG
#define LTXK_KW_DG 2      // trailing MFMA groups of a stage deferred past the next barrier
#endif
template <int TT, int NPIECES>
struct KwPipe {
  static constexpr int NS = 3, PD = LTXK_KW_PD, DG = LTXK_KW_DG;
  static constexpr int NG = NS * TT, RPS = 4 + TT, TOTAL = NS * RPS;
  static constexpr int PPG = (NPIECES + NG - 1) / NG;
  static_assert(TT >= 2, "deferred groups must lie in the last sub-step");
  bf16x8 wf[NS][4], af[NS][TT];

  __device__ __forceinline__ void init() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) wf[NS - 1][i][j] = (bf16)0.f;
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) af[NS - 1][i][j] = (bf16)0.f;
  }
  static __device__ __forceinline__ void group(const bf16x8& a, const bf16x8 (&w)[4], f32x4 (&acc)[4]) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nt], a, acc[nt], 0, 0, 0);
  }
  template <class RdFn, class IssueFn>
  __device__ __forceinline__ void step(RdFn&& rdf, IssueFn&& issue, f32x4 (&acc)[TT][4]) {
    auto rd = [&](int idx) __attribute__((always_inline)) {
      const int ks = idx / RPS, r = idx % RPS;
      if (r < 4) wf[ks][r] = rdf(ks, r);
      else af[ks][r - 4] = rdf(ks, r);
    };
    auto need = [](int g) { return g < 0 ? 0 : (g >= NG ? TOTAL : (g / TT) * RPS + 4 + (g % TT) + 1); };
    int issued = 0;
#pragma unroll
    for (int v = 0; v < DG; ++v) {                      // the groups deferred from the previous stage
      int target = need(v - DG + PD);
      if (target > (NS - 1) * RPS) target = (NS - 1) * RPS;     // the last sub-step's registers still feed the deferred groups
#pragma unroll
      for (int i = 0; i < TOTAL; ++i)
        if (i >= issued && i < target) rd(i);
      issued = target > issued ? target : issued;
#pragma unroll
      for (int q = 0; q < PPG; ++q) issue(v * PPG + q);
      group(af[NS - 1][TT - DG + v], wf[NS - 1], acc[TT - DG + v]);
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
#pragma unroll
    for (int i = 0; i < TOTAL; ++i)
      if (i >= issued) rd(i);
  }
  __device__ __forceinline__ void finish(f32x4 (&acc)[TT][4]) {
#pragma unroll
    for (int v = 0; v < DG; ++v) group(af[NS - 1][TT - DG + v], wf[NS - 1], acc[TT - DG + v]);
  }
};

__device__ __forceinline__ int kw_swz(int row) { return 2 * ((row >> 2) & 1); }

template <int TT, bool RES>
__global__ __launch_bounds__(GEMM_THREADS) void conv3d_k3_kw_kernel(ConvParams p) {
  using G = KwGeom<TT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 15, g = lane >> 4;
  int tile;
  {
    const int nt_all = p.RT * p.CT, q = nt_all >> 3, r = nt_all & 7;
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    tile = p.xcd_order ? x * q + (x < r ? x : r) + j : (int)blockIdx.x;
  }
  const int ct = tile % p.CT, rt = tile / p.CT;
  const int m0 = p.m_base + rt * G::BM, n0 = ct * KW_BN;
  const int Wd = p.W;
  const int w0 = m0 % Wd, vr0 = m0 / Wd;             // column / image-row id (over B*D*H) of the tile's first voxel
  const int nrows_img = p.B * p.D * p.H;
  const unsigned rowB = (unsigned)p.Cin * 2u;
  const int lrow4 = lane >> 2, slot4 = lane & 3;

  // ---- bias first: older than every LDS-DMA piece, so the counted vmcnt waits below cover it ----
  const int nq = g * 4;
  bf16x4 bpre[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    int n = n0 + wn * 64 + nt * 16 + nq;
    n = n < p.Cout ? n : p.Cout - 4;
    bpre[nt] = *opaque_gptr<bf16x4>((const bf16x4*)(p.bias + n));
  }

  // ---- W loader: piece i = tap kw of this (kd, kh), weight rows 16*wave .. +15 of the column tile ----
  int co = n0 + wave * 16 + lrow4;
  co = co < p.Cout ? co : p.Cout - 1;
  const char* wsrc = (const char*)(p.w + (size_t)co * 27 * p.Cin) + ((slot4 ^ kw_swz(wave * 16 + lrow4)) << 4);

  // ---- A loader: panel pieces wave, wave + 8, ... ; per panel row the separable source offsets (as in the kernel above) ----
  unsigned od[G::APW][3], oh[G::APW][3], owf[G::APW], zbits[G::APW];
  int adst[G::APW];
#pragma unroll
  for (int i = 0; i < G::APW; ++i) {
    int piece = wave + 8 * i;
    piece = piece < G::A_PIECES ? piece : G::A_PIECES - 1;
    const int prow = piece * 16 + lrow4;
    // run of this panel row: run r covers tile rows [S, E) and panel rows [S + 2r, E + 2r + 2)
    int r = 0, S = 0, E = Wd - w0 < G::BM ? Wd - w0 : G::BM;
#pragma unroll
    for (int rr = 0; rr < G::RUNMAX - 1; ++rr) {
      if (prow >= E + 2 * r + 2 && E < G::BM) {
        S = E;
        E = S + Wd < G::BM ? S + Wd : G::BM;
        ++r;
      }
    }
    int j = prow - (S + 2 * r);
    j = j < E - S + 1 ? j : E - S + 1;                  // rows past the panel's end repeat its last row
    int wq = (r == 0 ? w0 : 0) - 1 + j;                 // column of the source voxel: -1 and W are the halo
    int vr = vr0 + r;
    vr = vr < nrows_img ? vr : nrows_img - 1;
    const int vh = vr % p.H, vd = (vr / p.H) % p.D, vb = vr / (p.H * p.D);
    unsigned zb = 0;
    if (wq < 0 || wq >= Wd) {
      if (p.pad_mode == LTXK_PAD_REFLECT) wq = wq < 0 ? 1 : Wd - 2;
      else { zb |= 512u; wq = 0; }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int d = vd + k - (p.causal == 1 ? 2 : 1);
      if (p.causal == 2) {
        if (d < 0 || d >= p.D) { zb |= 64u << k; d = 0; }
      } else {
        d = d < 0 ? 0 : (d >= p.D ? p.D - 1 : d);
      }
      int h = vh + k - 1;
      if (p.pad_mode == LTXK_PAD_REFLECT) {
        h = h < 0 ? 1 : (h >= p.H ? p.H - 2 : h);
      } else if (h < 0 || h >= p.H) {
        zb |= 1u << k; h = 0;
      }
      od[i][k] = (unsigned)((vb * p.D + d) * p.H) * (unsigned)Wd * rowB;
      oh[i][k] = (unsigned)(h * Wd) * rowB;
    }
    owf[i] = (unsigned)wq * rowB + ((unsigned)(slot4 ^ kw_swz(prow)) << 4);
    zbits[i] = zb;
    adst[i] = piece * 1024;
  }
  const char* zsrc = (const char*)p.zero + slot4 * 16;
  const char* xb = (const char*)p.x;
  unsigned off_grp[G::APW];
  bool zgrp[G::APW];
  auto set_group = [&](int gi) __attribute__((always_inline)) {       // gi = kd * 3 + kh
    const int kd = gi / 3, kh = gi - 3 * kd;
    const unsigned d0 = kd == 0 ? ~0u : 0u, d1 = kd == 1 ? ~0u : 0u, d2 = kd == 2 ? ~0u : 0u;
    const unsigned h0 = kh == 0 ? ~0u : 0u, h1 = kh == 1 ? ~0u : 0u, h2 = kh == 2 ? ~0u : 0u;
#pragma unroll
    for (int j = 0; j < G::APW; ++j) {
      off_grp[j] = ((od[j][0] & d0) | (od[j][1] & d1) | (od[j][2] & d2)) + ((oh[j][0] & h0) | (oh[j][1] & h1) | (oh[j][2] & h2)) + owf[j];
      zgrp[j] = ((zbits[j] >> kh) & 1u) | ((zbits[j] >> (6 + kd)) & 1u) | ((zbits[j] >> 9) & 1u);
    }
  };
  // i-th LDS-DMA piece of this wave for stage (group gi, 32-channel block cb) into ring slot s
  auto issue_piece = [&](int i, int gi, int cb, int s) __attribute__((always_inline)) {
    char* base = smem + s * G::STAGE;
    if (i < 3) {
      glds16(wsrc + ((size_t)(gi * 3 + i) * p.Cin + cb * 32) * 2, base + G::A_BYTES + i * (KW_BN * 64) + wave * 1024);
    } else if (i < G::PER_STAGE) {
      const int j = i - 3 < G::APW ? i - 3 : 0;
      const char* src = zgrp[j] ? zsrc : xb + (size_t)(off_grp[j] + (unsigned)cb * 64u);
      glds16(src, base + adst[j]);
    }
  };

  // ---- fragment addresses (stage-invariant): A = panel rows of the wave's TT row blocks for kw = 0,1,2; W rows of its 4 column blocks ----
  unsigned aaddr[TT][3], waddr[4];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int i = wm * TT * 16 + tt * 16 + c;
    const int pb = i + 2 * ((w0 + i) / Wd);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) aaddr[tt][kw] = (unsigned)((pb + kw) * 64 + ((g ^ kw_swz(pb + kw)) << 4));
  }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int row = wn * 64 + nt * 16 + c;
    waddr[nt] = (unsigned)(G::A_BYTES + row * 64 + ((g ^ kw_swz(row)) << 4));
  }

  f32x4 acc[TT][4];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[tt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int cpb = p.Cin / 32;
  const int nk = 9 * cpb;
  int pgi = 0, pcb = 0;                                   // (group, channel block) of the prefetch stream
  set_group(0);
#pragma unroll
  for (int i = 0; i < G::PER_STAGE; ++i) issue_piece(i, 0, 0, 0);
  ++pcb;
  if (pcb == cpb) { pcb = 0; ++pgi; set_group(pgi); }
#pragma unroll
  for (int i = 0; i < G::PER_STAGE; ++i) issue_piece(i, pgi, pcb, 1);

  bf16x4 rres[RES ? TT : 1][4];
  int slot = 0;
  KwPipe<TT, G::PER_STAGE> pipe;
  pipe.init();
  auto stage = [&](auto mode_c) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_c)::value;         // 0: steady state; 1: next-to-last (nothing left to request); 2: last
    if constexpr (MODE == 2) wait_keep_and_barrier<0>();
    else wait_keep_and_barrier<G::PER_STAGE>();
    int s2 = slot + 2;
    s2 = s2 >= 3 ? s2 - 3 : s2;
    if constexpr (MODE == 0) {
      ++pcb;
      if (pcb == cpb) { pcb = 0; ++pgi; set_group(pgi); }
    }
    if constexpr (MODE == 2 && RES) {
      // the residual tile is requested behind the last barrier (no LDS-DMA is in flight any more) and lands under this
      // stage's 48 MFMAs
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        int m = m0 + wm * TT * 16 + tt * 16 + c;
        m = m < p.M ? m : p.M - 1;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          int n = n0 + wn * 64 + nt * 16 + nq;
          n = n < p.Cout ? n : p.Cout - 4;
          rres[tt][nt] = *opaque_gptr<bf16x4>((const bf16x4*)(p.resid + (size_t)m * p.Cout + n));
        }
      }
    }
    const char* st = smem + slot * G::STAGE;
    pipe.step([&](int ks, int r) __attribute__((always_inline)) -> bf16x8 {
                return r < 4 ? *(const bf16x8*)(st + waddr[r] + ks * (KW_BN * 64)) : *(const bf16x8*)(st + aaddr[r - 4 < TT ? r - 4 : 0][ks]);
              },
              [&](int i) __attribute__((always_inline)) {
                if constexpr (MODE == 0) issue_piece(i, pgi, pcb, s2);
              },
              acc);
    slot = slot + 1 == 3 ? 0 : slot + 1;
  };
  for (int s = 0; s < nk - 2; ++s) stage(IntC<0>{});
  stage(IntC<1>{});
  stage(IntC<2>{});
  pipe.finish(acc);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) asm volatile("" ::"v"(bpre[nt]));

  // ---- epilogue (as the kernel above): y = bf16(acc + bias) [+ resid], through a wave-private LDS image, whole lines out ----
  const bool wide = (p.Cout & 7) == 0;
  char* stg = smem + wave * (TT * 16 * 128);
  if (wide) __syncthreads();
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int m = m0 + wm * TT * 16 + tt * 16 + c;
    if (m >= p.M) continue;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wn * 64 + nt * 16 + nq;
      if (n >= p.Cout) continue;
      float y[4];
      const bf16x4 b = bpre[nt];
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = rbf(acc[tt][nt][j] + (float)b[j]);
      if constexpr (RES) {
        const bf16x4 r = rres[tt][nt];
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = rbf(y[j] + (float)r[j]);
      }
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (bf16)y[j];
      if (wide) {
        const int r = tt * 16 + c;
        *(bf16x4*)(stg + r * 128 + (((nt * 2 + (g >> 1)) ^ (r & 7)) << 4) + (g & 1) * 8) = o;
      } else {
        *(bf16x4*)(p.out + (size_t)m * p.Cout + n) = o;
      }
    }
  }
  if (wide) {
    const int cc = lane & 7;
    const int n = n0 + wn * 64 + cc * 8;
#pragma unroll
    for (int i = 0; i < TT * 2; ++i) {
      const int r = i * 8 + (lane >> 3);
      const int m = m0 + wm * TT * 16 + r;
      const bf16x8 v = *(const bf16x8*)(stg + r * 128 + ((cc ^ (r & 7)) << 4));
      if (m < p.M && n < p.Cout) *(bf16x8*)(p.out + (size_t)m * p.Cout + n) = v;
    }
  }
}

template <int TT, bool RES>
static int conv_launch_kw(const ConvParams& p0, hipStream_t stream) {
  using G = KwGeom<TT>;
  ConvParams p = p0;
  if (p.RT == 0) p.RT = (p.M - p.m_base + G::BM - 1) / G::BM;
  p.CT = (p.Cout + KW_BN - 1) / KW_BN;
  auto kern = conv3d_k3_kw_kernel<TT, RES>;
  static thread_local int attr_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev != attr_dev) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    if (e != hipSuccess) {
      ltxk_set_error("ltxk_conv3d_k3_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return LTXK_ELAUNCH;
    }
    attr_dev = dev;
  }
  hipLaunchKernelGGL(kern, dim3(p.RT * p.CT), dim3(GEMM_THREADS), G::LDS, stream, p);
  LTXK_CHECK_LAUNCH("ltxk_conv3d_k3_bf16(kw)");
  return LTXK_OK;
}

// whole rounds of 256-row tiles, then the rows of the short last round as 128-row tiles (same bits: a row's sum does not
// depend on the tile it is in)
template <bool RES>
static int conv_launch_kw_all(const ConvParams& p0, hipStream_t stream) {
  ConvParams p = p0;
  const int CT = (p.Cout + KW_BN - 1) / KW_BN;
  const int RT = (p.M + 255) / 256;
  const int tiles = RT * CT;
  const int rem_rt = (tiles % 256) / CT;
  const int rt_main = RT - rem_rt;
  const long tail_rows = (long)p.M - (long)rt_main * 256;
  const long tail_tiles = (tail_rows + 127) / 128 * CT;
  if (LTXK_AB_INT("LTXK_CONV_TAIL", 1) && tiles > 256 && rem_rt > 0 && rt_main > 0 && (long)rt_main * CT % 256 < CT && tail_tiles <= 256) {
    p.m_base = 0; p.RT = rt_main;
    int rc = conv_launch_kw<4, RES>(p, stream);
    if (rc != LTXK_OK) return rc;
    p.m_base = rt_main * 256; p.RT = (int)((tail_rows + 127) / 128);
    return conv_launch_kw<2, RES>(p, stream);
  }
  p.m_base = 0; p.RT = RT;
  return conv_launch_kw<4, RES>(p, stream);
}

template <int TT, int WN, bool RES>
static int conv_launch_plain(const ConvParams& p, hipStream_t stream) {
  using G = GemmGeom<TT, WN>;
  auto kern = conv3d_k3_kernel<TT, WN, RES, false>;
  static thread_local int attr_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev != attr_dev) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) {
      ltxk_set_error("ltxk_conv3d_k3_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return LTXK_ELAUNCH;
    }
    attr_dev = dev;
  }
  hipLaunchKernelGGL(kern, dim3(p.RT * p.CT), dim3(GEMM_THREADS), G::LDS_BYTES, stream, p);
  LTXK_CHECK_LAUNCH("ltxk_conv3d_k3_bf16");
  return LTXK_OK;
}

template <int TT, int WN, bool RES>
static int conv_launch(const ConvParams& p0, hipStream_t stream, float* workspace, size_t workspace_bytes) {
  using G = GemmGeom<TT, WN>;
  ConvParams p = p0;
  p.RT = (p.M + G::BM - 1) / G::BM;
  p.CT = (p.Cout + G::BN - 1) / G::BN;
  const int tiles = p.RT * p.CT, nk = p.ntaps * p.cpb;
  // split K when the tile grid leaves most of the 256 CUs idle (the 1024/512-channel stages of the decoder
  // have only 1280 / 9216 voxels): S slices of >= 16 K-steps, fp32 slabs in the caller's workspace.
  int S = 1;
  if (workspace && tiles <= 128) {
    S = 256 / tiles;
    if (S > nk / 16) S = nk / 16;
    const size_t per = (size_t)p.M * p.Cout * sizeof(float);
    if ((size_t)S * per > workspace_bytes) S = (int)(workspace_bytes / per);
    if (S > 16) S = 16;
  }
  if (S >= 2) {
    auto kern = conv3d_k3_kernel<TT, WN, false, true>;
    static thread_local int attr_dev_s = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev != attr_dev_s) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
      if (e != hipSuccess) { ltxk_set_error("ltxk_conv3d_k3_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e)); return LTXK_ELAUNCH; }
      attr_dev_s = dev;
    }
    p.slab = workspace; p.S = S; p.kper = (nk + S - 1) / S;
    p.S = (nk + p.kper - 1) / p.kper;                      // drop empty trailing slices
    hipLaunchKernelGGL(kern, dim3(tiles * p.S), dim3(GEMM_THREADS), G::LDS_BYTES, stream, p);
    LTXK_CHECK_LAUNCH("ltxk_conv3d_k3_bf16(split-K)");
    const size_t total = (size_t)p.M * p.Cout / 4;
    hipLaunchKernelGGL(conv_splitk_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       (const float*)workspace, p.bias, p.resid, p.out, p.M, p.Cout, p.S);
    LTXK_CHECK_LAUNCH("ltxk_conv3d_k3_bf16(finalize)");
    return LTXK_OK;
  }
  p.slab = nullptr; p.S = 1; p.kper = nk;
  // A short last round takes as long as a whole one: the 128-channel stage at 33x128x128 voxels is 2112 tiles of 256 rows =
  // 8.25 rounds of 256 CUs, the 512-channel stage 7.2, the 256-channel stage 27.2.  The rows past the last whole round run as
  // a second launch of lower tiles (128 instead of 256 rows, 96 instead of 160): more workgroups, less work each - a
  // fraction of a round instead of one.  Every output row is still the same K-ordered sum: same bits.
  if constexpr ((TT == 4 && WN == 2) || (TT == 5 && WN == 4)) {
    constexpr int TTT = TT == 4 ? 2 : 3;                          // tail tile height in 16-row MFMA blocks per wave row
    using GT = GemmGeom<TTT, WN>;
    const int tail_env = LTXK_AB_INT("LTXK_CONV_TAIL", 1);        // 0: off (A/B build only)
    const int rem_rt = (tiles % 256) / p.CT;                      // whole row tiles past the last whole round
    const int rt_main = p.RT - rem_rt;
    const long tail_rows = (long)p.M - (long)rt_main * G::BM;
    const long tail_tiles = (tail_rows + GT::BM - 1) / GT::BM * p.CT;
    if (tail_env && tiles > 256 && rem_rt > 0 && rt_main > 0 && (long)rt_main * p.CT % 256 < p.CT && tail_tiles <= 256) {
      ConvParams pm = p;
      pm.RT = rt_main;
      int rc = conv_launch_plain<TT, WN, RES>(pm, stream);
      if (rc != LTXK_OK) return rc;
      ConvParams pt = p;
      pt.m_base = rt_main * G::BM;
      pt.RT = (int)((tail_rows + GT::BM - 1) / GT::BM);
      return conv_launch_plain<TTT, WN, RES>(pt, stream);
    }
  }
  return conv_launch_plain<TT, WN, RES>(p, stream);
}

}  // namespace ltxk

extern "C" int ltxk_conv3d_k3_bf16(const ltxk_conv3d_args* a, void* stream) {
  using namespace ltxk;
  LTXK_CHECK_ARG(a != nullptr, "ltxk_conv3d_k3_bf16: null args");
  LTXK_CHECK_ARG(a->x && a->w && a->bias && (a->out || a->act_out) && a->zero_page, "ltxk_conv3d_k3_bf16: null x/w/bias/out/zero_page");
  LTXK_CHECK_ARG(a->B > 0 && a->D > 0 && a->H >= 2 && a->W >= 2, "ltxk_conv3d_k3_bf16: bad volume %dx%dx%dx%d", a->B, a->D, a->H, a->W);
  LTXK_CHECK_ARG(a->Cin % 64 == 0 && a->Cin > 0, "ltxk_conv3d_k3_bf16: Cin=%d must be a multiple of 64 (pad channels with zeros)", a->Cin);
  LTXK_CHECK_ARG(a->Cout % 8 == 0 && a->Cout > 0, "ltxk_conv3d_k3_bf16: Cout=%d must be a multiple of 8", a->Cout);
  LTXK_CHECK_ARG((((uintptr_t)a->x | (uintptr_t)a->w | (uintptr_t)a->zero_page) & 15) == 0 && ((uintptr_t)a->out & 7) == 0,
                 "ltxk_conv3d_k3_bf16: misaligned pointer");
  const long long M = (long long)a->B * a->D * a->H * a->W;
  LTXK_CHECK_ARG(M < (1ll << 31) && M * a->Cin * 2 < (1ll << 32), "ltxk_conv3d_k3_bf16: input volume must be < 4 GiB (32-bit tile offsets)");
  ConvParams p;
  p.x = (const bf16*)a->x; p.w = (const bf16*)a->w; p.bias = (const bf16*)a->bias; p.out = (bf16*)a->out;
  p.resid = (const bf16*)a->resid; p.zero = (const bf16*)a->zero_page;
  p.B = a->B; p.D = a->D; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout;
  p.causal = a->causal; p.pad_mode = a->pad_mode; p.M = (int)M; p.cpb = a->Cin / 64; p.RT = p.CT = 0;
  LTXK_CHECK_ARG(a->taps_d == 0 || a->taps_d == 3 || a->taps_d == 1, "ltxk_conv3d_k3_bf16: taps_d must be 3 (or 0) or 1, got %d", a->taps_d);
  p.ntaps = a->taps_d == 1 ? 9 : 27; p.kd0 = a->taps_d == 1 ? 1 : 0;
  p.act_out = (bf16*)a->act_out; p.act_scale = (const bf16*)a->act_scale; p.act_shift = (const bf16*)a->act_shift;
  p.act_eps = a->act_eps; p.act_silu = a->act_silu; p.rows_per_batch = a->D * a->H * a->W;
  p.xcd_order = LTXK_AB_INT("LTXK_CONV_XCD", 1);
  p.m_base = 0;
  if (a->act_out) {
    LTXK_CHECK_ARG(a->Cout == 128 || a->Cout == 256, "ltxk_conv3d_k3_bf16: the fused norm/activation output needs Cout == 128 or 256 (got %d)", a->Cout);
    LTXK_CHECK_ARG((a->act_scale == nullptr) == (a->act_shift == nullptr), "ltxk_conv3d_k3_bf16: act_scale and act_shift must both be set or both NULL");
    LTXK_CHECK_ARG(((uintptr_t)a->act_out & 15) == 0 && (((uintptr_t)a->act_scale | (uintptr_t)a->act_shift) & 7) == 0, "ltxk_conv3d_k3_bf16: misaligned act_* pointer");
  }
  hipStream_t st = (hipStream_t)stream;
  const bool res = a->resid != nullptr;
  float* ws = a->act_out ? nullptr : (float*)a->workspace;          // the fused row statistic lives in the tile: no split-K
  const size_t wsb = ws ? (size_t)a->workspace_bytes : 0;
  LTXK_CHECK_ARG(((uintptr_t)ws & 15) == 0, "ltxk_conv3d_k3_bf16: workspace must be 16-byte aligned");
  // kw-reuse form (see conv3d_k3_kw_kernel): the 27-tap convolution on volumes at least 64 voxels wide whose tile grid
  // fills the chip without split-K
  {
    const long tiles_kw = ((M + 255) / 256) * ((a->Cout + KW_BN - 1) / KW_BN);
    const int kw_mode = LTXK_AB_INT("LTXK_CONV_KW", 1);
    // (A/B build: 2 = wherever legal, 0 = never.  By default only where one 128-column tile spans Cout: with several
    // column tiles the A panel is re-fetched per column tile and the per-tap kernel's 160x256 tile moves fewer bytes.)
    const bool kw_pays = kw_mode == 2 || a->Cout <= KW_BN;
    if (kw_mode && kw_pays && p.ntaps == 27 && a->W >= 64 && a->Cin % 32 == 0 && !a->act_out && a->out && (tiles_kw > 128 || !ws)) {
      return res ? conv_launch_kw_all<true>(p, st) : conv_launch_kw_all<false>(p, st);
    }
  }
  if (a->Cout <= 128) {       // 256x128 tile: no wasted MFMA columns on the 128-channel stage
    return res ? conv_launch<4, 2, true>(p, st, ws, wsb) : conv_launch<4, 2, false>(p, st, ws, wsb);
  }
  return res ? conv_launch<5, 4, true>(p, st, ws, wsb) : conv_launch<5, 4, false>(p, st, ws, wsb);
}

// Dense bf16 GEMM with fused epilogues: ltxk_gemm_bf16 (include/ltxk.h).
// Replaces nn.Linear at attention.py:123-126,142; feed_forward.py:35-40; adaln.py:46,134-138;
// text_projection.py:22-25; ltx.py:130,455 — and the residual/gate algebra of
// transformer.py:254,257,347 as epilogues.
#include "gemm_core.h"
#include <stdlib.h>
#include <utility>

#ifndef LTXK_DEFER_GROUPS
#define LTXK_DEFER_GROUPS 2   // MFMA groups of a K-step run after the next barrier (gemm_core.h MmaPipe)
#endif
#ifndef LTXK_STAGGER
#define LTXK_STAGGER 0   // measured neutral (profiles/r01 notes); kept for A/B
#endif

#ifdef LTXK_DIAG
// Diagnostic build only (make diag -> libltxk_diag.so): per-workgroup phase stamps (s_memrealtime, 100 MHz) written to a
// buffer of their own; the product library contains none of this.
__device__ unsigned long long* ltxk_gemm_stamps = nullptr;
extern "C" int ltxk_diag_set_gemm_stamps(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(ltxk_gemm_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#define LTXK_STAMP(slot)                                                                              \
  do {                                                                                                \
    if (ltxk_gemm_stamps && threadIdx.x == 0) ltxk_gemm_stamps[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define LTXK_STAMP(slot) do { } while (0)
#endif

// The counted vmcnt waits below assume a FIXED number of loads per lane between the DMA pieces.  `load(c ? a : dummy)`
// whose value is only used when c holds gets unfolded by the optimiser into a load under `if (c)`: with bias == NULL a
// LoRA merge then issued four loads fewer than its first wait allows in flight and raced its first stage.  The source
// pointer is therefore made opaque first (an empty asm: no instruction, no wait), which leaves ONE unconditional load.
#define LTXK_VLOAD(T, ptr) (*opaque_gptr<T>((const T*)(ptr)))

namespace ltxk {


struct GemmParams {
  const bf16* A;
  const bf16* W;
  const bf16* bias;
  bf16* out;
  const bf16* resid;
  const bf16* gate;
  const int32_t* gate_row;
  int M, N, K, lda, ldo, ldr, gate_stride;
  int RT, CT;
  int T;  // tokens per batch for the transposed output
  float alpha;
  int wide;  // 1: rows are written as whole 128-byte lines through an LDS image (needs ldo % 8 == 0, out 16-B aligned)
  // split output: column tiles with n0 >= n_split are written TRANSPOSED (bias only) to out2[(b*(N-n_split) + n-n_split)*ldo2 + t]
  bf16* out2;
  int n_split, ldo2;
  // per-row sums of squares of the stored bf16 outputs, one fp32 per 64-column block: sumsq[m*sumsq_ld + n/64]
  float* sumsq;
  int sumsq_ld;
  // split-K (gemm_stream_kernel): blockIdx.y = K slice; slice s covers K-steps [s*ksteps, min(K/64, (s+1)*ksteps)) and stores its
  // fp32 accumulators to part[(s*M + m)*N + n]; splitk_epilogue_kernel sums the slices in order and applies the epilogue
  float* part;
  int ksteps;
};

// NT = 16-column MFMA tiles per wave: 4 -> 256-column workgroup tiles (64 per wave), 2 -> 128-column ones (32 per wave).
template <int TT, int NT, int EPI, bool TRANS>
__device__ __forceinline__ void gemm_tile(const GemmParams& p, char* smem, int m0, int n0, int kbase, int nk, int wave, int lane) {
  using G = GemmGeom<TT, 4, NT>;
  constexpr int GEMM_W_STAGE_BYTES = G::W_STAGE_BYTES;
  constexpr int WPW = G::W_PER_WAVE;          // W pieces per wave per stage (4 / 2)
  constexpr int WC = 16 * NT;                 // columns per wave
  const int wm = wave >> 2, wn = wave & 3;

  // ---- loader: per-lane source pointers (row clamped, 16-byte chunk pre-swizzled) ----
  const int lrow = lane >> 3;
  const int chunk = (lane & 7) ^ lrow;
  const bf16* wptr[WPW];
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    int r = n0 + (wave * WPW + i) * 8 + lrow;
    r = r < p.N ? r : p.N - 1;
    wptr[i] = p.W + (size_t)r * p.K + chunk * 8;
  }
  // Every wave issues exactly WPW W pieces + MAXA A pieces per stage, so one constant vmcnt retires a
  // stage.  A_PIECES is not a multiple of 8 (20 at BM=160): waves past A_REM own one piece fewer and
  // re-issue their last piece (same source, same LDS bytes: benign) to keep the count uniform.
  const int nA = G::A_BASE + (wave < G::A_REM ? 1 : 0);
  const int a0 = wave * G::A_BASE + (wave < G::A_REM ? wave : G::A_REM);
  const bf16* aptr[G::MAXA];
  int adst[G::MAXA];
#pragma unroll
  for (int i = 0; i < G::MAXA; ++i) {
    const int pi = nA > 0 ? a0 + (i < nA ? i : nA - 1) : G::A_PIECES - 1;   // no own piece: re-issue the tile's last one
    int r = m0 + pi * 8 + lrow;
    r = r < p.M ? r : p.M - 1;
    aptr[i] = p.A + (size_t)r * p.lda + chunk * 8;
    adst[i] = GEMM_W_STAGE_BYTES + (pi < G::A_PIECES ? pi : G::A_PIECES - 1) * 1024;
  }
  constexpr int PER_STAGE = WPW + G::MAXA;
  static_assert(PER_STAGE <= 7, "vmcnt immediates below assume <= 7 pieces per stage");

  // i-th LDS-DMA piece of this wave for K-step kt (of this launch slice) into ring slot s (pieces 0..WPW-1 = W, then A)
  auto issue_piece = [&](int i, int kt, int s) __attribute__((always_inline)) {
    char* base = smem + s * G::STAGE_BYTES;
    const int ko = (kbase + kt) * GEMM_BK;
    if (i < WPW) {
      glds16(wptr[i < WPW ? i : 0] + ko, base + (wave * WPW + i) * 1024);
    } else if (i < PER_STAGE) {
      glds16(aptr[i - WPW < G::MAXA ? i - WPW : 0] + ko, base + adst[i - WPW < G::MAXA ? i - WPW : 0]);
    }
  };

  constexpr bool HAS_RES = !TRANS && (EPI == LTXK_EPI_BIAS_GATE_RES || EPI == LTXK_EPI_BIAS_RES || EPI == LTXK_EPI_SCALE_RES);
  constexpr bool HAS_GATE = !TRANS && EPI == LTXK_EPI_BIAS_GATE_RES;
  f32x4 acc[TT][NT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[tt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  LTXK_STAMP(0);
  // Epilogue operands (bias, residual tile, gate rows) are requested up front and arrive under the main loop instead
  // of as a dependent-load chain (gate_row -> gate -> arithmetic) and a 21 MB residual burst when the loop ends.  They
  // are issued BEHIND the first two stages' DMA pieces - vmcnt retires in issue order, so the first K-steps wait with a
  // counted vmcnt that leaves these loads in flight (issued in front, the first barrier waited for all of them: 4-5 us
  // per launch by in-kernel stamps).  The number of loads per lane is a compile-time constant (dummy sources stand in
  // for absent operands), as the counted waits need.  (The output may alias the residual: each element is read and
  // written by this workgroup only.)
  int grow[HAS_GATE ? TT : 1];
  if constexpr (HAS_GATE) {
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      int m = m0 + wm * TT * 16 + tt * 16 + (lane & 15);
      m = m < p.M ? m : p.M - 1;
      grow[tt] = LTXK_VLOAD(int32_t, p.gate_row ? p.gate_row + m : (const int32_t*)p.W);      // (dummy source keeps the load count fixed)
    }
  }
#pragma unroll
  for (int i = 0; i < PER_STAGE; ++i) issue_piece(i, 0, 0);
#pragma unroll
  for (int i = 0; i < PER_STAGE; ++i) issue_piece(i, nk > 1 ? 1 : 0, 1);
  bf16x4 rres[HAS_RES ? TT : 1][NT], bpre[NT], gpre[HAS_GATE ? TT : 1][NT];
  bf16 bpre_t[NT];
  // The residual tile (21 MB per launch over the chip) is NOT requested here: read as one burst in the prologue it holds
  // up the first stages' arrival by ~4 us (the fabric serves it at ~5 TB/s; in-kernel stamps).  It trickles in during
  // K-steps 2 .. 1+TT, one 16-row band (4 loads per lane) per step; bands a short K (a LoRA merge: 1-2 steps) never
  // reaches are read after the loop.  (One writer per register on every path: with a second, "all bands up front"
  // path for short K, hipcc guarded the in-loop loads against the other path's possibly pending ones with
  // vmcnt(16) ... vmcnt(0) - a drain of the DMA pipeline in each of the steps.)
  auto load_res_band = [&](auto tt_c) __attribute__((always_inline)) {
    constexpr int tt = decltype(tt_c)::value;
    if constexpr (HAS_RES && tt < TT) {
      const int nq = (lane >> 4) * 4;
      int m = m0 + wm * TT * 16 + tt * 16 + (lane & 15);
      m = m < p.M ? m : p.M - 1;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        int n = n0 + wn * WC + nt * 16 + nq;
        n = n < p.N ? n : p.N - 4;
        rres[tt][nt] = LTXK_VLOAD(bf16x4, p.resid + (size_t)m * p.ldr + n);
      }
    }
  };
  constexpr int NB = NT;                                          // bias loads per lane, issued behind stage 1
  constexpr int NX2 = HAS_GATE ? TT * NT : 0;                     // gate values, issued after the first barrier
  {
    const int nq = (lane >> 4) * 4;
    if constexpr (TRANS) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        int n = n0 + wn * WC + nt * 16 + (lane & 15);
        n = n < p.N ? n : p.N - 1;
        bpre_t[nt] = LTXK_VLOAD(bf16, p.bias ? p.bias + n : p.W);
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        int n = n0 + wn * WC + nt * 16 + nq;
        n = n < p.N ? n : p.N - 4;
        bpre[nt] = LTXK_VLOAD(bf16x4, p.bias ? p.bias + n : p.W);
      }
    }
  }
  auto issue_gate = [&]() __attribute__((always_inline)) {
    if constexpr (HAS_GATE) {
      const int nq = (lane >> 4) * 4;
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const int gr = p.gate_row ? grow[tt] : 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          int n = n0 + wn * WC + nt * 16 + nq;
          n = n < p.N ? n : p.N - 4;
          gpre[tt][nt] = LTXK_VLOAD(bf16x4, p.gate + (size_t)gr * p.gate_stride + n);
        }
      }
    }
  };
  // K-steps.  Stage kt is waited for with a counted vmcnt that leaves in flight everything issued behind its pieces
  // (vmcnt retires in issue order):
  //   kt = 0: stage 1 + the bias loads;  kt = 1: those + the gate loads + stage 2;
  //   kt >= 2: stage kt+1 + the residual band requested at the start of step kt-1.
  // The first PEEL steps are unrolled with a compile-time kt: the epilogue-operand loads then sit in straight-line code
  // and land directly in their final registers.  (Issued from inside the loop - a switch on kt - hipcc loaded them into
  // temporaries and put `s_waitcnt vmcnt(0)` + copies INTO the loop: 72 -> 90 us at K=4096.)
  constexpr int PEEL = HAS_RES ? 3 + TT : 2;
  int s = 0;
  auto kstep = [&](auto& pipe, int kt, auto kc) __attribute__((always_inline)) {
    constexpr int KC = decltype(kc)::value;                       // compile-time kt in the peeled steps, -1 in the loop
    if constexpr (KC == 0) {
      wait_keep_and_barrier<PER_STAGE + NB>();
      LTXK_STAMP(7);
      issue_gate();
      LTXK_STAMP(1);
    } else if constexpr (KC == 1) {
      wait_keep_and_barrier<PER_STAGE + NB + NX2>();
    } else if constexpr (HAS_RES && KC >= 3 && KC < 3 + TT) {
      wait_keep_and_barrier<PER_STAGE + NT>();
    } else if constexpr (KC == -3) {
      wait_keep_and_barrier<0>();                                   // last K-step: nothing younger than its own stage is in flight
    } else {
      wait_keep_and_barrier<PER_STAGE>();
    }
    if constexpr (HAS_RES && KC >= 2 && KC < 2 + TT) load_res_band(IntC<(KC >= 2 ? KC - 2 : 0)>{});
    if (kt == 8) LTXK_STAMP(2);
    int s2 = s + 2;
    s2 = s2 >= 3 ? s2 - 3 : s2;
    const int kt2 = kt + 2 < nk ? kt + 2 : nk - 1;   // short K (no peeled tail): harmless re-load of the last stage into a free slot
    // KC = -2 / -3: the last two K-steps of a launch long enough to have them peeled request nothing (re-loading the last
    // stage there kept 2 x 7 pieces in flight past the loop: the epilogue began with a wait for them)
    auto issue = [&](int i) __attribute__((always_inline)) {
      if constexpr (KC != -2 && KC != -3) issue_piece(i, kt2, s2);
    };
    if constexpr (TT >= 2) {
      pipe.step(smem + s * G::STAGE_BYTES, wm, wn, lane, acc, issue);
    } else {
      mma_stage_pipelined<TT, 4, TRANS, NT>(smem + s * G::STAGE_BYTES, wm, wn, lane, acc, issue);
    }
    s = s + 1 == 3 ? 0 : s + 1;
  };
  auto kloop = [&](auto& pipe) __attribute__((always_inline)) {
    if constexpr (TT >= 2) pipe.init();
    [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
      ((I < nk ? kstep(pipe, I, IntC<I>{}) : (void)0), ...);
    }(std::make_integer_sequence<int, PEEL>{});
    const bool tail = nk >= PEEL + 2;
    const int nmain = tail ? nk - 2 : nk;
    for (int kt = PEEL; kt < nmain; ++kt) kstep(pipe, kt, IntC<-1>{});
    if (tail) {
      kstep(pipe, nk - 2, IntC<-2>{});
      kstep(pipe, nk - 1, IntC<-3>{});
    }
    if constexpr (TT >= 2) pipe.finish(acc);
  };
  if constexpr (TT >= 2) {
#if LTXK_STAGGER
    if (wave >= 4) {                  // SIMD partners of waves 0-3 run half a K-step out of phase
      MmaPipe<TT, 4, TRANS, TT, NT> pipe;
      kloop(pipe);
    } else
#endif
    {
      MmaPipe<TT, 4, TRANS, (LTXK_DEFER_GROUPS <= TT ? LTXK_DEFER_GROUPS : TT), NT> pipe;
      kloop(pipe);
    }
  } else {
    int dummy = 0;
    kloop(dummy);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // Every prefetched value is "used" here, whatever the epilogue does with it: a load whose value is dead (the bias under
  // EPI_SCALE_RES, which overwrites the biased value) is otherwise deleted - four loads fewer than the first waits allow
  // in flight, i.e. a LoRA merge that raced its first stage.
  if constexpr (TRANS) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(bpre_t[nt]));
  } else {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(bpre[nt]));
  }
  if constexpr (HAS_GATE) {
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(gpre[tt][nt]));
  }
  if constexpr (HAS_RES) {
    if (nk < 2 + TT) {                 // short K: the bands whose step never ran
      if (nk <= 2) load_res_band(IntC<0>{});
      if (nk <= 3) load_res_band(IntC<1>{});
      if (nk <= 4) load_res_band(IntC<2>{});
      if (nk <= 5) load_res_band(IntC<3>{});
      if (nk <= 6) load_res_band(IntC<4>{});
    }
  }
  LTXK_STAMP(3);

  // ---- epilogue ----
  if constexpr (!TRANS) {
    // acc[tt][nt][j]: token = lane&15, n = 4*(lane>>4) + j.  Written directly that is 16 rows x 32 bytes per store
    // instruction; with p.wide each wave instead transposes its (16*TT x WC) block through a private LDS image
    // (rows of WC*2 bytes, 16-byte chunks XOR-swizzled by row) and stores whole rows of it per instruction.
    constexpr int RB = WC * 2, CPR = RB / 16;       // staged row bytes (128 / 64), 16-byte chunks per row (8 / 4)
    const int nq = (lane >> 4) * 4;
    char* stg = smem + wave * (TT * 16 * RB);
    const bool want_ss = p.sumsq != nullptr;
    // NT == 2: a 64-column block of the row statistic spans the waves wn = 2c (columns 0-31) and 2c+1 (32-63).  The even wave
    // hands its lane-wise partial sums through LDS and the odd wave continues them with its own 8 values per lane: the same
    // 16 sequential additions per lane, then the same two exchanges, as one wave of the 256-column tile - the same bits.
    const bool pair_hi = NT == 2 && (wn & 1);
    float* xss = (float*)(smem + 8 * TT * 16 * RB) + ((wm * 2 + (wn >> 1)) * TT) * 64;
    bf16x4 okeep[NT == 2 ? TT : 1][NT];
    if (p.wide || (NT == 2 && want_ss)) __syncthreads();                   // every wave has read its last fragments from the ring
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int m = m0 + wm * TT * 16 + tt * 16 + (lane & 15);
      if (m >= p.M) continue;
      float ss = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + wn * WC + nt * 16 + nq;
        if (n >= p.N) continue;
        float y[4];
        if (p.bias) {
          const bf16x4 b = bpre[nt];
#pragma unroll
          for (int j = 0; j < 4; ++j) y[j] = rbf(acc[tt][nt][j] + (float)b[j]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) y[j] = rbf(acc[tt][nt][j]);
        }
        if constexpr (EPI == LTXK_EPI_BIAS_GELU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) y[j] = gelu_tanh_f(y[j]);
        } else if constexpr (EPI == LTXK_EPI_BIAS_SILU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) y[j] = silu_f(y[j]);
        } else if constexpr (EPI == LTXK_EPI_BIAS_GATE_RES) {
          const bf16x4 g = gpre[tt][nt];
          const bf16x4 r = rres[tt][nt];
#pragma unroll
          for (int j = 0; j < 4; ++j) y[j] = (float)r[j] + rbf(y[j] * (float)g[j]);
        } else if constexpr (EPI == LTXK_EPI_BIAS_RES) {
          const bf16x4 r = rres[tt][nt];
#pragma unroll
          for (int j = 0; j < 4; ++j) y[j] = (float)r[j] + y[j];
        } else if constexpr (EPI == LTXK_EPI_SCALE_RES) {
          const bf16x4 r = rres[tt][nt];
#pragma unroll
          for (int j = 0; j < 4; ++j) y[j] = (float)r[j] + rbf(p.alpha * acc[tt][nt][j]);
        }
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)y[j];
        if (want_ss) {                  // (uniform branch: the row statistic costs ~3 VALU per element, a quarter of a GELU epilogue)
          if constexpr (NT == 2) {
            if (pair_hi) okeep[tt][nt] = o;
          }
          if (!pair_hi) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float f = (float)o[j];
              ss += f * f;
            }
          }
        }
        if (p.wide) {
          const int r = tt * 16 + (lane & 15), cg = lane >> 4;
          *(bf16x4*)(stg + r * RB + (((nt * 2 + (cg >> 1)) ^ (r & (CPR - 1))) << 4) + (cg & 1) * 8) = o;
        } else {
          *(bf16x4*)(p.out + (size_t)m * p.ldo + n) = o;
        }
      }
      if (want_ss) {
        if constexpr (NT == 4) {
          // the wave's 64 columns of row m: 16 values per lane, then the four lanes sharing (lane & 15); fixed order
          ss = lane_xor16_sum(ss);
          ss = lane_xor32_sum(ss);
          // (a wave whose 64-column block lies past N - the last column tile when N % 256 != 0 - has nothing to report:
          // its slot would be the next row's first partial)
          if (lane < 16 && n0 + wn * 64 < p.N) p.sumsq[(size_t)m * p.sumsq_ld + ((n0 + wn * 64) >> 6)] = ss;
        } else if (!pair_hi) {
          xss[tt * 64 + lane] = ss;
        }
      }
    }
    if constexpr (NT == 2) {
      if (want_ss) {
        __syncthreads();
        if (pair_hi) {
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) {
            const int m = m0 + wm * TT * 16 + tt * 16 + (lane & 15);
            if (m >= p.M || n0 + wn * WC >= p.N) continue;
            float ss = xss[tt * 64 + lane];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float f = (float)okeep[tt][nt][j];
                ss += f * f;
              }
            ss = lane_xor16_sum(ss);
            ss = lane_xor32_sum(ss);
            if (lane < 16) p.sumsq[(size_t)m * p.sumsq_ld + ((n0 + wn * WC) >> 6)] = ss;
          }
        }
      }
    }
    if (p.wide) {
      constexpr int RPI = 64 / CPR;                 // rows per store instruction (8 / 16)
      const int c = lane & (CPR - 1);
      const int n = n0 + wn * WC + c * 8;
#pragma unroll
      for (int i = 0; i < TT * 16 / RPI; ++i) {
        const int r = i * RPI + lane / CPR;
        const int m = m0 + wm * TT * 16 + r;
        const bf16x8 v = *(const bf16x8*)(stg + r * RB + ((c ^ (r & (CPR - 1))) << 4));
        if (m < p.M && n < p.N) *(bf16x8*)(p.out + (size_t)m * p.ldo + n) = v;
      }
    }
  } else {
    // acc[tt][nt][j]: n = lane&15, token = 4*(lane>>4) + j ; out[(b*N + n)*ldo + t]
    // (split output: the transposed block has N - n_split rows per batch, lives in out2 and is indexed from n_split)
    const int tq = (lane >> 4) * 4;
    bf16* const tout = p.n_split ? p.out2 : p.out;
    const int tld = p.n_split ? p.ldo2 : p.ldo;
    const int tN = p.N - p.n_split;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + wn * WC + nt * 16 + (lane & 15);
      if (n >= p.N) continue;
      const float b = p.bias ? (float)bpre_t[nt] : 0.f;
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const int m = m0 + wm * TT * 16 + tt * 16 + tq;
        if (m >= p.M) continue;
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)(acc[tt][nt][j] + b);
        const int bidx = m / p.T, t = m - bidx * p.T;
        bf16* dst = tout + ((size_t)bidx * tN + (n - p.n_split)) * tld + t;
        if ((p.T & 3) == 0 && m + 3 < p.M) {
          *(bf16x4*)dst = o;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int mj = m + j;
            if (mj < p.M) {
              const int bj = mj / p.T, tj = mj - bj * p.T;
              tout[((size_t)bj * tN + (n - p.n_split)) * tld + tj] = o[j];
            }
          }
        }
      }
    }
  }
}

// MODE 0: every tile row-major with epilogue EPI; 1: every tile transposed (V^T); 2: split output - tiles with
// n0 < n_split row-major (EPI), the rest transposed (one launch for q|k|v, or for the text k|v pair)
template <int TT, int NT, int EPI, int MODE>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_bf16_kernel(GemmParams p) {
  using G = GemmGeom<TT, 4, NT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  LTXK_STAMP(4);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int rt, ct;
  map_tile(blockIdx.x, p.RT, p.CT, rt, ct);
  const int m0 = rt * G::BM, n0 = ct * G::BN;
  const int kbase = 0, nk = p.K / GEMM_BK;
  if constexpr (MODE == 0) {
    gemm_tile<TT, NT, EPI, false>(p, smem, m0, n0, kbase, nk, wave, lane);
  } else if constexpr (MODE == 1) {
    gemm_tile<TT, NT, LTXK_EPI_BIAS, true>(p, smem, m0, n0, kbase, nk, wave, lane);
  } else {
    if (n0 < p.n_split) gemm_tile<TT, NT, EPI, false>(p, smem, m0, n0, kbase, nk, wave, lane);
    else gemm_tile<TT, NT, LTXK_EPI_BIAS, true>(p, smem, m0, n0, kbase, nk, wave, lane);
  }
#ifdef LTXK_DIAG
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  LTXK_STAMP(5);
  if (ltxk_gemm_stamps && threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    ltxk_gemm_stamps[(size_t)blockIdx.x * 8 + 6] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
}

// One K slice of a small-M (weight-streaming) launch on the 128-column tile: nothing but the main loop and the fp32 tile store,
// with the LDS ring as deep as the CU's 160 KiB allow (6 stages of 24.5 KiB at 64 rows against the 3 of gemm_tile): such a
// launch is a stream of weight bytes, a workgroup's rate is (stages in flight) x (stage bytes) / (memory latency), and with three
// stages a 64 x 128 tile pulled ~50 GB/s - 160 of them 1 TB/s of a 32-MB panel (profiles/r04_small_m_ab.log).  Every wave
// issues the same number of LDS-DMA pieces per K-step (a clamped re-load of the last stage into a free slot once the real
// ones run out), so one constant counted vmcnt retires a stage.
template <int TT> struct StreamGeom {
  using G = GemmGeom<TT, 4, 2>;
  static constexpr int D = (160 * 1024) / G::STAGE_BYTES < 6 ? (160 * 1024) / G::STAGE_BYTES : 6;
  static constexpr int LDS = D * G::STAGE_BYTES;
  static_assert(D >= 3, "ring too shallow");
};

template <int TT>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_stream_kernel(GemmParams p) {
  using G = GemmGeom<TT, 4, 2>;
  constexpr int D = StreamGeom<TT>::D, NT = 2, WPW = G::W_PER_WAVE, PER_STAGE = WPW + G::MAXA;
  static_assert(PER_STAGE * (D - 2) <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int ct = blockIdx.x % p.CT, rt = blockIdx.x / p.CT;
  const int m0 = rt * G::BM, n0 = ct * G::BN;
  const int nk_all = p.K / GEMM_BK;
  const int kbase = (int)blockIdx.y * p.ksteps;
  const int nk = nk_all - kbase < p.ksteps ? nk_all - kbase : p.ksteps;
  const int lrow = lane >> 3, chunk = (lane & 7) ^ lrow;
  const bf16* wptr[WPW];
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    int r = n0 + (wave * WPW + i) * 8 + lrow;
    r = r < p.N ? r : p.N - 1;
    wptr[i] = p.W + (size_t)r * p.K + chunk * 8;
  }
  const int nA = G::A_BASE + (wave < G::A_REM ? 1 : 0);
  const int a0 = wave * G::A_BASE + (wave < G::A_REM ? wave : G::A_REM);
  const bf16* aptr[G::MAXA];
  int adst[G::MAXA];
#pragma unroll
  for (int i = 0; i < G::MAXA; ++i) {
    const int pi = nA > 0 ? a0 + (i < nA ? i : nA - 1) : G::A_PIECES - 1;
    int r = m0 + pi * 8 + lrow;
    r = r < p.M ? r : p.M - 1;
    aptr[i] = p.A + (size_t)r * p.lda + chunk * 8;
    adst[i] = G::W_STAGE_BYTES + pi * 1024;
  }
  auto issue_piece = [&](int i, int kt, int slot) __attribute__((always_inline)) {
    char* base = smem + slot * G::STAGE_BYTES;
    const int ko = (kbase + (kt < nk ? kt : nk - 1)) * GEMM_BK;
    if (i < WPW) glds16(wptr[i < WPW ? i : 0] + ko, base + (wave * WPW + i) * 1024);
    else if (i < PER_STAGE) glds16(aptr[i - WPW < G::MAXA ? i - WPW : 0] + ko, base + adst[i - WPW < G::MAXA ? i - WPW : 0]);
  };
  f32x4 acc[TT][NT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[tt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int st = 0; st < D - 1; ++st)
#pragma unroll
    for (int i = 0; i < PER_STAGE; ++i) issue_piece(i, st, st);
  auto loop = [&](auto& pipe) __attribute__((always_inline)) {
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
      wait_keep_and_barrier<PER_STAGE * (D - 2)>();            // stage kt has landed; D-2 younger stages stay in flight
      const int pre = slot == 0 ? D - 1 : slot - 1;            // slot of stage kt + D - 1 = the one consumed in step kt - 1
      auto issue = [&](int i) __attribute__((always_inline)) { issue_piece(i, kt + D - 1, pre); };
      if constexpr (TT >= 2) pipe.step(smem + slot * G::STAGE_BYTES, wm, wn, lane, acc, issue);
      else mma_stage_pipelined<TT, 4, false, NT>(smem + slot * G::STAGE_BYTES, wm, wn, lane, acc, issue);
      slot = slot + 1 == D ? 0 : slot + 1;
    }
  };
  if constexpr (TT >= 2) {
    MmaPipe<TT, 4, false, 2, NT> pipe;
    pipe.init();
    loop(pipe);
    pipe.finish(acc);
  } else {
    int dummy = 0;
    loop(dummy);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the clamped re-loads must not outlive the workgroup's LDS
  const int nq = (lane >> 4) * 4;
  float* dst = p.part + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int m = m0 + wm * TT * 16 + tt * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + wn * 32 + nt * 16 + nq;
      if (n < p.N) *(f32x4*)(dst + (size_t)m * p.N + n) = acc[tt][nt];
    }
  }
}

template <int TT>
static int launch_stream(const GemmParams& p, int ksplit, hipStream_t stream) {
  auto kern = gemm_stream_kernel<TT>;
  static thread_local int attr_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev != attr_dev) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, StreamGeom<TT>::LDS);
    if (e != hipSuccess) {
      ltxk_set_error("ltxk_gemm_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return LTXK_ELAUNCH;
    }
    attr_dev = dev;
  }
  hipLaunchKernelGGL(kern, dim3(p.RT * p.CT, ksplit), dim3(GEMM_THREADS), StreamGeom<TT>::LDS, stream, p);
  LTXK_CHECK_LAUNCH("ltxk_gemm_bf16 (split-K slices)");
  return LTXK_OK;
}

// Second launch of a split-K GEMM: sums the S fp32 slices in slice order (deterministic) and applies the epilogue of
// gemm_tile - the same formulas and rounding points, one rounding of the fp32 sum.  One thread = 4 consecutive columns of
// one row; a 64-column block of the row statistic is 16 consecutive lanes.  `trans` = every column transposed (V^T).
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(GemmParams p, int S, int epi, int trans) {
  const int n4 = p.N >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = idx < (long)p.M * n4;
  const int m = live ? (int)(idx / n4) : 0, n = live ? (int)(idx - (long)m * n4) * 4 : 0;
  const size_t slab = (size_t)p.M * p.N;
  const float* src = p.part + (size_t)m * p.N + n;
  f32x4 a = *(const f32x4*)src;
  int sl = 1;
  for (; sl + 4 <= S; sl += 4) {                 // four independent loads in flight, added in slice order
    const f32x4 b0 = *(const f32x4*)(src + (size_t)sl * slab), b1 = *(const f32x4*)(src + (size_t)(sl + 1) * slab);
    const f32x4 b2 = *(const f32x4*)(src + (size_t)(sl + 2) * slab), b3 = *(const f32x4*)(src + (size_t)(sl + 3) * slab);
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = ((a[j] + b0[j]) + b1[j]) + b2[j] + b3[j];
  }
  for (; sl < S; ++sl) {
    const f32x4 b = *(const f32x4*)(src + (size_t)sl * slab);
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] += b[j];
  }
  bf16x4 bias = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
  if (p.bias) bias = *(const bf16x4*)(p.bias + n);
  if (trans || (p.n_split && n >= p.n_split)) {
    if (!live) return;
    bf16* const tout = p.n_split ? p.out2 : p.out;
    const int tld = p.n_split ? p.ldo2 : p.ldo, tN = p.N - p.n_split;
    const int bidx = m / p.T, t = m - bidx * p.T;
#pragma unroll
    for (int j = 0; j < 4; ++j) tout[((size_t)bidx * tN + (n + j - p.n_split)) * tld + t] = (bf16)(a[j] + (float)bias[j]);
    return;
  }
  float y[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) y[j] = p.bias ? rbf(a[j] + (float)bias[j]) : rbf(a[j]);
  if (epi == LTXK_EPI_BIAS_GELU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = gelu_tanh_f(y[j]);
  } else if (epi == LTXK_EPI_BIAS_SILU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = silu_f(y[j]);
  } else if (epi == LTXK_EPI_BIAS_GATE_RES || epi == LTXK_EPI_BIAS_RES || epi == LTXK_EPI_SCALE_RES) {
    const bf16x4 r = *(const bf16x4*)(p.resid + (size_t)m * p.ldr + n);
    if (epi == LTXK_EPI_BIAS_GATE_RES) {
      const bf16x4 g = *(const bf16x4*)(p.gate + (size_t)(p.gate_row ? p.gate_row[m] : 0) * p.gate_stride + n);
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = (float)r[j] + rbf(y[j] * (float)g[j]);
    } else if (epi == LTXK_EPI_BIAS_RES) {
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = (float)r[j] + y[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = (float)r[j] + rbf(p.alpha * a[j]);
    }
  }
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (bf16)y[j];
  if (live) *(bf16x4*)(p.out + (size_t)m * p.ldo + n) = o;
  if (p.sumsq) {                                   // (N % 64 == 0: a 64-column block = 16 aligned consecutive lanes of one row)
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float f = (float)o[j];
      ss += f * f;
    }
    ss = group_sum<16>(ss);
    if (live && (threadIdx.x & 15) == 0) p.sumsq[(size_t)m * p.sumsq_ld + (n >> 6)] = ss;
  }
}

template <int TT, int NT, int EPI, int MODE>
static int launch(const GemmParams& p, hipStream_t stream) {
  using G = GemmGeom<TT, 4, NT>;
  auto kern = gemm_bf16_kernel<TT, NT, EPI, MODE>;
  static thread_local int attr_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev != attr_dev) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       G::LDS_BYTES);
    if (e != hipSuccess) {
      ltxk_set_error("ltxk_gemm_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return LTXK_ELAUNCH;
    }
    attr_dev = dev;
  }
  hipLaunchKernelGGL(kern, dim3(p.RT * p.CT), dim3(GEMM_THREADS), G::LDS_BYTES, stream, p);
  LTXK_CHECK_LAUNCH("ltxk_gemm_bf16");
  return LTXK_OK;
}

template <int TT, int NT>
static int dispatch_epi(const GemmParams& p, int epi, bool trans, hipStream_t stream) {
  if (p.n_split) return launch<TT, NT, LTXK_EPI_BIAS, 2>(p, stream);
  if (trans) return launch<TT, NT, LTXK_EPI_BIAS, 1>(p, stream);
  switch (epi) {
    case LTXK_EPI_BIAS: return launch<TT, NT, LTXK_EPI_BIAS, 0>(p, stream);
    case LTXK_EPI_BIAS_GELU: return launch<TT, NT, LTXK_EPI_BIAS_GELU, 0>(p, stream);
    case LTXK_EPI_BIAS_SILU: return launch<TT, NT, LTXK_EPI_BIAS_SILU, 0>(p, stream);
    case LTXK_EPI_BIAS_GATE_RES: return launch<TT, NT, LTXK_EPI_BIAS_GATE_RES, 0>(p, stream);
    case LTXK_EPI_BIAS_RES: return launch<TT, NT, LTXK_EPI_BIAS_RES, 0>(p, stream);
    case LTXK_EPI_SCALE_RES: return launch<TT, NT, LTXK_EPI_SCALE_RES, 0>(p, stream);
  }
  ltxk_set_error("ltxk_gemm_bf16: unknown epilogue %d", epi);
  return LTXK_EINVAL;
}

template <int NT>
static int dispatch_tt(const GemmParams& p, int tt, int epi, bool trans, hipStream_t st) {
  switch (tt) {
    case 5: return dispatch_epi<5, NT>(p, epi, trans, st);
    case 4: return dispatch_epi<4, NT>(p, epi, trans, st);
    case 3: return dispatch_epi<3, NT>(p, epi, trans, st);
    case 2: return dispatch_epi<2, NT>(p, epi, trans, st);
    default: return dispatch_epi<1, NT>(p, epi, trans, st);
  }
}

// split-K slices on the 128-column tile (the weight-streaming form for small M): fp32 partial tiles, then the epilogue launch
static int launch_partial(const GemmParams& p, int tt, int ksplit, hipStream_t st) {
  switch (tt) {
    case 5: return launch_stream<5>(p, ksplit, st);
    case 4: return launch_stream<4>(p, ksplit, st);
    case 3: return launch_stream<3>(p, ksplit, st);
    case 2: return launch_stream<2>(p, ksplit, st);
    default: return launch_stream<1>(p, ksplit, st);
  }
}

// Tile choice: row-tile height TT (32*TT rows) and column width (256 or 128), minimising
//   (rounds over 256 CUs) x (cost of one tile) ,  cost = rows + fixed per-tile overhead for the 256-column tile,
//   W128 x that for the 128-column one (half the MFMAs per K-step, but 0.7 instead of 0.45 fragment reads and 1.6x the
//   LDS-DMA pieces per MFMA: measured, profiles/r04_gemm_128col_ab.log).
// (Round 3, tried and removed: "whole rounds first" - the rows of the whole 256-CU rounds as 160-row tiles, the remaining rows
// as a second launch at their own best height.  The model below prices it 5-9 % cheaper at M=3328 / 5184 / 6656, N=4096;
// measured -0.5...-2.3 % at M=3328 / 6656 and +3...4 % at M=5184 (profiles/r03_gemm_rounds_ab.log): a part-filled last round
// costs less than the model's whole round - the busy CUs run at a higher clock - and the second launch pays its own fill,
// epilogue and boundary.)
struct TileChoice { int tt, nt; long cost; };
static long tile_cost(int M, int N, int tt, int nt) {
  const int bm = 32 * tt, bn = 64 * nt;
  const long RT = (M + bm - 1) / bm, CT = (N + bn - 1) / bn;
  const long rounds = (RT * CT + 255) / 256;
  const long per_tile = nt == 4 ? 100L * (bm + 48) : (long)LTXK_AB_INT("LTXK_GEMM_W128", 70) * (bm + 40);
  return rounds * per_tile;
}
static TileChoice pick_tile(int M, int N, int nt_force) {
  TileChoice best = {5, 4, -1};
  for (int nt = 4; nt >= 2; nt -= 2) {
    if (nt_force && nt != nt_force) continue;
    for (int tt = 5; tt >= 1; --tt) {
      const long c = tile_cost(M, N, tt, nt);
      if (best.cost < 0 || c < best.cost) best = {tt, nt, c};
    }
  }
  return best;
}

}  // namespace ltxk

// ---- 320 x 256 tile: FF1 (M=2560, N=16384), self-attention q|k (N=8192), the text k|v pair (M=2048) --------------------
// What keeps the matrix pipe of the 160x256 kernel at ~60 % busy is the CU's one LDS: every 1-KiB fragment read (4 cycles
// at 256 B/clk) and every 1-KiB LDS-DMA piece (8 cycles at 128 B/clk) is LDS time the MFMAs overlap only in part, and on
// random data the chip holds ~1.8 GHz instead of 2.4 on top (scripts/mfma_power_probe.hip, DESIGN.md §5b: that kernel's
// instruction mix sustains 1.25 PF in a barrier-free synthetic loop, the kernel itself 1.1-1.2; the mix of a 320x256 tile
// sustains 1.44).  Doubling the tile takes 31 % off the LDS-DMA bytes per MFMA, 22 % off the L2 -> fabric bytes (8x4 patches
// of 320-row tiles) and, with 80x128 per wave, 28 % off the fragment reads.  The price is registers: 160 accumulator
// registers per lane at two waves per SIMD leave room for ONE set of operand fragments, so a wave reads, waits and
// multiplies (the SIMD's other wave covers the wait), refilling each fragment register as soon as its last MFMA of the
// K-sub-step has issued; and LDS holds two stages, not three, so a stage's LDS-DMA pieces are all issued in the first MFMA
// rows of the step before it.  The accumulators live in VGPRs (gfx950's register file is unified; a kernel that names AGPRs
// gets its 256-register budget split 128 / 128 by hipcc, which 160 accumulators do not fit): the MFMAs are inline asm,
// accumulating in place.  Only launches that fill whole rounds use it (big_tile_pays below).  Same bits as the 160x256 kernel.
namespace ltxk {

// RB = 16-row MFMA blocks per wave: 5 -> the 320 x 256 tile (80 x 128 per wave); 4 -> a 256 x 256 tile (64 x 128 per wave) for
// launches whose row count is a multiple of 256 but not of 320 (the text k|v pair: M = B*1024; round 3).
constexpr int BIG_BN = 256;
constexpr int BIG_STAGE_W = BIG_BN * 128;
template <int RB> struct BigGeom {
  static constexpr int BM = 64 * RB;
  static constexpr int STAGE = (BIG_BN + BM) * 128, LDS = 2 * STAGE;
  static_assert(LDS <= 160 * 1024, "two stages must fit the CU's LDS");
};

#ifndef LTXK_BIG_PPR
#define LTXK_BIG_PPR 3   // LDS-DMA pieces issued per MFMA row of sub-step 0
#endif
#ifndef LTXK_BIG_HELD
#define LTXK_BIG_HELD 1  // W columns of sub-step 1 held back past the next barrier (5 MFMAs each)
#endif
#define LTXK_MFMA_V(acc, w, a) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a))

// TRANS: the tile is written transposed per batch (V^T) - operand roles swapped so that a lane holds 4 consecutive tokens
template <int EPI, bool TRANS, int RB>
__device__ __forceinline__ void gemm_big_tile(const GemmParams& p, char* smem, int m0, int n0, int wave, int lane) {
  constexpr int BIG_STAGE = BigGeom<RB>::STAGE;
  const int wm = wave >> 1, wn = wave & 1;                       // 4 x 2 waves of (16*RB) x 128

  // loader: 32 W pieces + 8*RB A pieces of 1 KiB (8 rows x 128 B) per stage; wave w owns W pieces 4w..4w+3 and A pieces
  // RB*w..RB*w+RB-1.  Source = wave-uniform base (SGPR) + per-lane offset; the 16-byte chunk is pre-swizzled by the row.
  const int lrow = lane >> 3;
  const int chunk = (lane & 7) ^ lrow;
  const unsigned w_lane = (unsigned)((lrow * p.K + chunk * 8) * 2);
  const uint64_t w_base = (uint64_t)(p.W + (size_t)(n0 + wave * 32) * p.K);
  unsigned a_lane[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    int r = m0 + (wave * RB + i) * 8 + lrow;
    r = r < p.M ? r : p.M - 1;
    a_lane[i] = (unsigned)(((size_t)(r - m0) * p.lda + chunk * 8) * 2);
  }
  const uint64_t a_base = (uint64_t)(p.A + (size_t)m0 * p.lda);
  auto issue_piece = [&](int i, int kt, int slot) __attribute__((always_inline)) {
    char* base = smem + slot * BIG_STAGE;
    const unsigned ko = (unsigned)(kt * GEMM_BK * 2);
    if (i < 4) glds16_s(w_base + (uint64_t)i * 8 * p.K * 2 + ko, w_lane, base + (wave * 4 + i) * 1024);
    else if (i < 4 + RB) glds16_s(a_base + ko, a_lane[i - 4 < RB ? i - 4 : 0], base + BIG_STAGE_W + (wave * RB + (i - 4)) * 1024);
  };

  f32x4 acc[RB][8];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / GEMM_BK;
#pragma unroll
  for (int i = 0; i < 4 + RB; ++i) issue_piece(i, 0, 0);

  const int frow = (lane & 15) * 128;
  const int koff0 = ((lane >> 4) ^ (lane & 7)) << 4;
  const int koff1 = ((4 + (lane >> 4)) ^ (lane & 7)) << 4;
  bf16x8 wf[8], af[RB];
  auto mma = [](f32x4& c, const bf16x8& w, const bf16x8& a) __attribute__((always_inline)) {
    if constexpr (TRANS) LTXK_MFMA_V(c, a, w);
    else LTXK_MFMA_V(c, w, a);
  };
  // The last W column(s) of every K-step's second half (5 MFMAs per wave each) are held back until AFTER the next step's barrier:
  // its operands are in registers, so it runs while the new stage's first fragment reads are in flight - the LDS cold start
  // after each barrier (all 8 waves reading at once) then does not idle the matrix pipe.  Step 0 multiplies zeros.
  constexpr int HB = LTXK_BIG_HELD;                                // held-back W columns: 8-HB .. 7
#pragma unroll
  for (int j = 8 - HB; j < 8; ++j)
#pragma unroll
    for (int q = 0; q < 8; ++q) wf[j][q] = (bf16)0.f;
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int q = 0; q < 8; ++q) af[i][q] = (bf16)0.f;
  auto kstep = [&](int kt, auto last_c) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_c)::value != 0;    // the last K-step requests nothing
    // stage kt has landed in every wave's view, and every wave has finished reading the other slot (stage kt-1)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int slot = kt & 1;
    const char* wb = smem + slot * BIG_STAGE + (wn * 128) * 128 + frow;
    const char* ab = smem + slot * BIG_STAGE + BIG_STAGE_W + (wm * 16 * RB) * 128 + frow;
    const int kt1 = kt + 1;
    // K-sub-step 0 fragments.  The W fragments whose registers are free first, then the held-back columns, each of their
    // last MFMAs freeing one A fragment register for the new stage, then the remaining W fragments.
#pragma unroll
    for (int j = 0; j < 8 - HB; ++j) {
      wf[j] = *(const bf16x8*)(wb + j * 2048 + koff0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
#pragma unroll
      for (int j = 8 - HB; j < 8; ++j) mma(acc[i][j], wf[j], af[i]);
      af[i] = *(const bf16x8*)(ab + i * 2048 + koff0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 8 - HB; j < 8; ++j) wf[j] = *(const bf16x8*)(wb + j * 2048 + koff0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < RB; ++i) {
#pragma unroll
        for (int j = 0; j < (ks == 0 ? 8 : 8 - HB); ++j) {
          mma(acc[i][j], wf[j], af[i]);
          // the last row of sub-step 0 frees the W fragments one by one: refill each for sub-step 1 at once
          if (ks == 0 && i == RB - 1) wf[j] = *(const bf16x8*)(wb + j * 2048 + koff1);
        }
        if (ks == 0) {
          af[i] = *(const bf16x8*)(ab + i * 2048 + koff1);        // row i done: its A fragment register is free
          // next stage's LDS-DMA: all 4 + RB pieces in the first three MFMA rows, so they have most of a K-step to land
          if constexpr (!LAST) {
#pragma unroll
            for (int q = 0; q < LTXK_BIG_PPR; ++q) issue_piece(i * LTXK_BIG_PPR + q, kt1, slot ^ 1);   // (no-op past piece 8)
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  for (int kt = 0; kt + 1 < nk; ++kt) kstep(kt, IntC<0>{});
  kstep(nk - 1, IntC<1>{});
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 8 - HB; j < 8; ++j) mma(acc[i][j], wf[j], af[i]);
  // hipcc does not see the asm MFMAs as matrix instructions and pads no hazard: let the last ones retire before the
  // accumulators are read (the operands tie the wait to the registers written last)
  // (every accumulator the last MFMA rows wrote appears exactly once as an operand: the last two rows whole, the held-back
  // columns of every row - an lvalue repeated among the read-write operands of one asm statement is unspecified)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (i >= RB - 2 || j >= 8 - HB) asm volatile("" : "+v"(acc[i][j]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if constexpr (!TRANS) {
    // ---- epilogue: acc[i][j][q]: row = lane & 15, column = 4 * (lane >> 4) + q; direct 8-byte stores ----
    const int nq = (lane >> 4) * 4;
    const bool want_ss = p.sumsq != nullptr;
    bf16x4 bpre[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int n = n0 + wn * 128 + j * 16 + nq;
      bpre[j] = p.bias ? *(const bf16x4*)(p.bias + n) : bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int m = m0 + wm * 16 * RB + i * 16 + (lane & 15);
      float ss[2] = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (m >= p.M) continue;
        const int n = n0 + wn * 128 + j * 16 + nq;
        float y[4];
        if (p.bias) {
#pragma unroll
          for (int q = 0; q < 4; ++q) y[q] = rbf(acc[i][j][q] + (float)bpre[j][q]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) y[q] = rbf(acc[i][j][q]);
        }
        if constexpr (EPI == LTXK_EPI_BIAS_GELU) {
#pragma unroll
          for (int q = 0; q < 4; ++q) y[q] = gelu_tanh_f(y[q]);
        } else if constexpr (EPI == LTXK_EPI_BIAS_SILU) {
#pragma unroll
          for (int q = 0; q < 4; ++q) y[q] = silu_f(y[q]);
        }
        bf16x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (bf16)y[q];
        if (want_ss) {                  // the 160x256 kernel's order: 16 values per lane and 64-column block, then the 4 lanes
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float f = (float)o[q];
            ss[j >> 2] += f * f;
          }
        }
        *(bf16x4*)(p.out + (size_t)m * p.ldo + n) = o;
      }
      if (want_ss) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float v = ss[b];
          v = lane_xor16_sum(v);
          v = lane_xor32_sum(v);
          if (lane < 16 && m < p.M) p.sumsq[(size_t)m * p.sumsq_ld + ((n0 + wn * 128) >> 6) + b] = v;
        }
      }
    }
  } else {
    // acc[i][j][q]: n = lane & 15, token = 4 * (lane >> 4) + q; out[(b * tN + n - n_split) * ld + t]  (as the 160x256 kernel)
    const int tq = (lane >> 4) * 4;
    bf16* const tout = p.n_split ? p.out2 : p.out;
    const int tld = p.n_split ? p.ldo2 : p.ldo;
    const int tN = p.N - p.n_split;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int n = n0 + wn * 128 + j * 16 + (lane & 15);
      const float b = p.bias ? (float)p.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int m = m0 + wm * 16 * RB + i * 16 + tq;
        if (m >= p.M) continue;
        bf16x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (bf16)(acc[i][j][q] + b);
        const int bidx = m / p.T, t = m - bidx * p.T;
        bf16* dst = tout + ((size_t)bidx * tN + (n - p.n_split)) * tld + t;
        if ((p.T & 3) == 0 && m + 3 < p.M) {
          *(bf16x4*)dst = o;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int mq = m + q;
            if (mq < p.M) {
              const int bq = mq / p.T, tqq = mq - bq * p.T;
              tout[((size_t)bq * tN + (n - p.n_split)) * tld + tqq] = o[q];
            }
          }
        }
      }
    }
  }
}

// MODE as in gemm_bf16_kernel: 0 row-major with EPI, 1 transposed, 2 split at n_split
template <int EPI, int MODE, int RB>
__global__ __launch_bounds__(512) void gemm_bf16_big_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int rt, ct;
  map_tile(blockIdx.x, p.RT, p.CT, rt, ct);
  const int m0 = rt * BigGeom<RB>::BM, n0 = ct * BIG_BN;
  if constexpr (MODE == 0) {
    gemm_big_tile<EPI, false, RB>(p, smem, m0, n0, wave, lane);
  } else if constexpr (MODE == 1) {
    gemm_big_tile<LTXK_EPI_BIAS, true, RB>(p, smem, m0, n0, wave, lane);
  } else {
    if (n0 < p.n_split) gemm_big_tile<EPI, false, RB>(p, smem, m0, n0, wave, lane);
    else gemm_big_tile<LTXK_EPI_BIAS, true, RB>(p, smem, m0, n0, wave, lane);
  }
}

template <int EPI, int MODE, int RB>
static int launch_big_rb(const GemmParams& p, hipStream_t stream) {
  constexpr int BIG_LDS = BigGeom<RB>::LDS;
  auto kern = gemm_bf16_big_kernel<EPI, MODE, RB>;
  static thread_local int attr_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev != attr_dev) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS);
    if (e != hipSuccess) {
      ltxk_set_error("ltxk_gemm_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return LTXK_ELAUNCH;
    }
    attr_dev = dev;
  }
  hipLaunchKernelGGL(kern, dim3(p.RT * p.CT), dim3(512), BIG_LDS, stream, p);
  LTXK_CHECK_LAUNCH("ltxk_gemm_bf16");
  return LTXK_OK;
}

template <int EPI, int MODE>
static int launch_big(const GemmParams& p, hipStream_t stream, int rb) {
  return rb == 4 ? launch_big_rb<EPI, MODE, 4>(p, stream) : launch_big_rb<EPI, MODE, 5>(p, stream);
}

// (FF2 - N=4096, K=16384: 128 big tiles - was also tried on this tile as split-K pairs, two workgroups per tile with one K
// half each, the first to finish parking its fp32 partial tile in a workspace, the second adding it and running the
// gate + residual epilogue: 416 us against 302 us with agent-scope release / acquire fences around the hand-over (each is
// a whole-L2 write-back / invalidate on this chip), and still 320 us with the fences taken out for the measurement; at
// K=4096 116 against 90 us.  The meeting costs more than the bigger tile saves.  Removed.)
// The big tile pays when it still fills whole 256-CU rounds: rounds x 2 (tile area) x 0.9 (measured gain) against the
// 160-row rounds.  M=2560: N=16384 -> 2 rounds against 4 (taken); N=12288 -> 2 against 3, N=4096 -> 1 against 1 (not).
// Returns the rows-per-wave block count of the big tile to use (5: 320 x 256, 4: 256 x 256) or 0 for the 160-row tiles:
// cost = rounds of 256 CUs x tile rows x per-row factor (1.0 for the 160-row tile, 0.9 measured for the 320-row tile, 0.93 for
// the 256-row one, whose 64 x 128 wave tile reads 0.375 fragments per MFMA against 0.325).  M=2560: N=16384 -> 320 (2 rounds
// against 4), N=8192 -> 320 (1 against 2), N=4096 -> 160 (1 against 1); M=2048 (text k|v), N=8192 -> 256 (exactly one round of
// 8 x 32 tiles; the 320-row tile leaves 32 CUs idle and a 128-row remainder tile).
static int big_tile_choice(int M, int N) {
  const long CT = N / BIG_BN;
  auto tiles = [&](int bm) { return ((long)(M + bm - 1) / bm) * CT; };
  auto rounds = [&](int bm) { return (tiles(bm) + 255) / 256; };
  const long cost_s = rounds(160) * 160 * 100, cost_5 = rounds(320) * 320 * 90, cost_4 = rounds(256) * 256 * 93;
  // the 256-row tile only where it fills whole rounds with whole tiles (M=6656, N=16384: 6.5 rounds of it measured 763 us
  // against 728 us for 5.25 rounds of the 320-row tile - a part-filled round costs less than a whole one, and the model
  // above cannot tell by how much; profiles/r03_gemm_256tile_ab.log)
  if (M % 256 == 0 && tiles(256) % 256 == 0 && cost_4 < cost_s && cost_4 < cost_5) return 4;
  if (cost_5 < cost_s) return 5;
  return 0;
}

}  // namespace ltxk

extern "C" int ltxk_gemm_bf16(const ltxk_gemm_args* a, void* stream) {
  using namespace ltxk;
  LTXK_CHECK_ARG(a != nullptr, "ltxk_gemm_bf16: null args");
  LTXK_CHECK_ARG(a->A && a->W && a->out, "ltxk_gemm_bf16: null A/W/out");
  LTXK_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "ltxk_gemm_bf16: bad dims M=%d N=%d K=%d", a->M, a->N, a->K);
  LTXK_CHECK_ARG(a->K % GEMM_BK == 0, "ltxk_gemm_bf16: K=%d must be a multiple of %d", a->K, GEMM_BK);
  LTXK_CHECK_ARG(a->N % 8 == 0, "ltxk_gemm_bf16: N=%d must be a multiple of 8", a->N);
  LTXK_CHECK_ARG(a->lda >= a->K && a->lda % 8 == 0, "ltxk_gemm_bf16: lda=%d (K=%d) must be >=K, multiple of 8", a->lda, a->K);
  LTXK_CHECK_ARG(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->W & 15) == 0 && ((uintptr_t)a->out & 7) == 0,
                 "ltxk_gemm_bf16: A/W must be 16-byte aligned, out 8-byte aligned");
  const bool split = a->n_split > 0;
  const bool trans = a->out_tokens_per_batch > 0 && !split;
  if (split) {
    LTXK_CHECK_ARG(a->epilogue == LTXK_EPI_BIAS, "ltxk_gemm_bf16: split output supports EPI_BIAS only");
    LTXK_CHECK_ARG(a->n_split % 256 == 0 && a->n_split < a->N, "ltxk_gemm_bf16: n_split=%d must be a multiple of 256 below N=%d", a->n_split, a->N);
    LTXK_CHECK_ARG(a->out2 != nullptr && ((uintptr_t)a->out2 & 7) == 0, "ltxk_gemm_bf16: split output needs an 8-byte aligned out2");
    LTXK_CHECK_ARG(a->out_tokens_per_batch > 0 && a->M % a->out_tokens_per_batch == 0, "ltxk_gemm_bf16: split output needs out_tokens_per_batch dividing M=%d", a->M);
    LTXK_CHECK_ARG(a->ldo2 >= a->out_tokens_per_batch && a->ldo2 % 4 == 0, "ltxk_gemm_bf16: transposed ldo2=%d", a->ldo2);
    LTXK_CHECK_ARG(a->ldo >= a->n_split && a->ldo % 4 == 0, "ltxk_gemm_bf16: ldo=%d (n_split=%d)", a->ldo, a->n_split);
  } else if (trans) {
    LTXK_CHECK_ARG(a->epilogue == LTXK_EPI_BIAS, "ltxk_gemm_bf16: transposed output supports EPI_BIAS only");
    LTXK_CHECK_ARG(a->M % a->out_tokens_per_batch == 0, "ltxk_gemm_bf16: M=%d not a multiple of tokens/batch=%d", a->M, a->out_tokens_per_batch);
    LTXK_CHECK_ARG(a->ldo >= a->out_tokens_per_batch && a->ldo % 4 == 0, "ltxk_gemm_bf16: transposed ldo=%d", a->ldo);
  } else {
    LTXK_CHECK_ARG(a->ldo >= a->N && a->ldo % 4 == 0, "ltxk_gemm_bf16: ldo=%d (N=%d)", a->ldo, a->N);
  }
  if (a->sumsq) {
    LTXK_CHECK_ARG(!trans && a->N % 64 == 0 && a->sumsq_ld >= (split ? a->n_split : a->N) / 64 && ((uintptr_t)a->sumsq & 3) == 0,
                   "ltxk_gemm_bf16: sumsq needs a row-major output, N %% 64 == 0 and sumsq_ld >= columns/64");
  }
  if (a->epilogue == LTXK_EPI_BIAS_GATE_RES || a->epilogue == LTXK_EPI_BIAS_RES || a->epilogue == LTXK_EPI_SCALE_RES) {
    LTXK_CHECK_ARG(a->resid != nullptr && a->ldr >= a->N && a->ldr % 4 == 0, "ltxk_gemm_bf16: residual epilogue needs resid/ldr");
  }
  if (a->epilogue == LTXK_EPI_BIAS_GATE_RES) {
    LTXK_CHECK_ARG(a->gate != nullptr && a->gate_stride % 4 == 0, "ltxk_gemm_bf16: gate epilogue needs gate");
  }
  GemmParams p;
  p.A = (const bf16*)a->A; p.W = (const bf16*)a->W; p.bias = (const bf16*)a->bias;
  p.out = (bf16*)a->out; p.resid = (const bf16*)a->resid; p.gate = (const bf16*)a->gate;
  p.gate_row = a->gate_row;
  p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldo = a->ldo; p.ldr = a->ldr;
  p.gate_stride = a->gate_stride; p.T = (trans || split) ? a->out_tokens_per_batch : 1; p.alpha = a->alpha;
  p.out2 = (bf16*)a->out2; p.n_split = split ? a->n_split : 0; p.ldo2 = a->ldo2;
  p.sumsq = a->sumsq; p.sumsq_ld = a->sumsq_ld;
  const int wide_env = LTXK_AB_INT("LTXK_GEMM_WIDE", 1);
  // not for the GELU epilogue: its direct stores already issue under the activation arithmetic, and staging
  // them behind it measured 2 % slower on FF1
  p.wide = (!trans && wide_env && a->epilogue != LTXK_EPI_BIAS_GELU && a->ldo % 8 == 0 && ((uintptr_t)a->out & 15) == 0) ? 1 : 0;
  const int tt_env = LTXK_AB_INT("LTXK_GEMM_TT", 0);
  const int nt_env = LTXK_AB_INT("LTXK_GEMM_NT", 0);                  // A/B build: 2 / 4 forces the 128- / 256-column tile
  const bool nt2_legal = !split || a->n_split % 128 == 0;
  TileChoice tc = pick_tile(a->M, a->N, nt_env == 2 && nt2_legal ? 2 : (nt_env == 4 || !nt2_legal ? 4 : 0));
  if (tt_env >= 1 && tt_env <= 5) tc.tt = tt_env;
  const int tt = tc.tt;
  const int bm = 32 * tt;
  p.RT = (a->M + bm - 1) / bm;
  p.CT = (a->N + 64 * tc.nt - 1) / (64 * tc.nt);
  p.part = nullptr; p.ksteps = a->K / GEMM_BK;
  hipStream_t st = (hipStream_t)stream;
  const int big_env = LTXK_AB_INT("LTXK_GEMM_BIG", 1);
  const bool big_legal = a->N % BIG_BN == 0 && a->K <= (1 << 20) &&
                         (a->epilogue == LTXK_EPI_BIAS || ((a->epilogue == LTXK_EPI_BIAS_GELU || a->epilogue == LTXK_EPI_BIAS_SILU) && !a->sumsq));
  // A/B build: LTXK_GEMM_BIG=0 never, 2 / 3 the 320-row / 256-row tile whenever legal (tests compare the tiles bit for bit)
  const int rb = !big_legal || tt_env != 0 || nt_env != 0 || big_env == 0 ? 0 : (big_env == 2 ? 5 : (big_env == 3 ? 4 : big_tile_choice(a->M, a->N)));
  if (rb) {
    const int bm = 64 * rb;
    p.RT = (a->M + bm - 1) / bm;
    p.CT = a->N / BIG_BN;
    if (split) return launch_big<LTXK_EPI_BIAS, 2>(p, st, rb);
    if (trans) return launch_big<LTXK_EPI_BIAS, 1>(p, st, rb);
    switch (a->epilogue) {
      case LTXK_EPI_BIAS: return launch_big<LTXK_EPI_BIAS, 0>(p, st, rb);
      case LTXK_EPI_BIAS_GELU: return launch_big<LTXK_EPI_BIAS_GELU, 0>(p, st, rb);
      default: return launch_big<LTXK_EPI_BIAS_SILU, 0>(p, st, rb);
    }
  }
  // Small M: the launch is a weight stream, and a 32..160-row tiling gives it far fewer workgroups than the chip has CUs (M=64,
  // N=4096: 16 tiles of 256 columns - 16 CUs pulling 32 MB).  Split-K on the 128-column tile: every K slice is a workgroup of its
  // own that parks its fp32 accumulators in the caller's workspace, and splitk_epilogue_kernel sums the slices in order and
  // applies the epilogue (deterministic; fp32 sum of slice sums instead of one running sum: another summation order, one rounding).
  {
    const int ks_env = LTXK_AB_INT("LTXK_GEMM_KSPLIT", 0);           // A/B build: -1 never, n >= 2 forces n slices where legal
    const int nk = a->K / GEMM_BK;
    int S = 0;
    if (ks_env >= 0 && rb == 0 && a->workspace != nullptr && nt2_legal && a->N % 4 == 0 && ((uintptr_t)a->workspace & 15) == 0 &&
        (a->M <= LTXK_AB_INT("LTXK_GEMM_KSPLIT_MAXM", 640) || ks_env >= 2)) {
      // the tallest row tile that covers M (every row tile re-streams its W slab), then as many K slices as fill the 256 CUs
      // (one workgroup each: the deep ring takes most of a CU's LDS) - but never more fp32 slice traffic (S x M x N x 8 bytes,
      // written and read back) than the weight panel itself (N x K x 2 bytes), and at least 8 K-steps per slice
      const int tt2 = a->M >= 160 ? 5 : (a->M + 31) / 32;
      const long tiles = ((long)(a->M + 32 * tt2 - 1) / (32 * tt2)) * ((a->N + 127) / 128);
      const long have = ((long)p.RT * p.CT);
      if (ks_env >= 2) S = ks_env;
      else if (tiles <= 160 && have <= 240 && (long)a->N * a->K >= (1L << 21)) {     // a real weight stream (>= 4 MB) the tiling cannot spread well (have: its workgroups unsplit - up to 240 small LDS-bound tiles, M=640 FF2)
        S = (int)((256 + tiles / 2) / tiles);
        const int cap = a->K / (4 * a->M);
        if (S > cap) S = cap;
      }
      if (S > nk / 8 && ks_env < 2) S = nk / 8;
      if (S > nk) S = nk;
      const long per = (long)a->M * a->N * 4;
      if (S > 1 && (long)S * per > a->workspace_bytes) S = (int)(a->workspace_bytes / per);
      if (S >= 2) {
        p.RT = (a->M + 32 * tt2 - 1) / (32 * tt2);
        p.CT = (a->N + 127) / 128;
        p.ksteps = (nk + S - 1) / S;
        S = (nk + p.ksteps - 1) / p.ksteps;                            // no empty slice
        p.part = (float*)a->workspace;
        const int rc = launch_partial(p, tt2, S, st);
        if (rc != LTXK_OK) return rc;
        const long threads = (long)a->M * (a->N / 4);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, p, S, a->epilogue, trans ? 1 : 0);
        LTXK_CHECK_LAUNCH("ltxk_gemm_bf16 (split-K epilogue)");
        return LTXK_OK;
      }
    }
  }
  return tc.nt == 2 ? dispatch_tt<2>(p, tt, a->epilogue, trans, st) : dispatch_tt<4>(p, tt, a->epilogue, trans, st);
}

// HBM-bound row / elementwise kernels of the DiT step (include/ltxk.h).  All loads/stores are
// 8- or 16-byte vectors; row reductions are wave64 shuffles (one wave per row); every bf16
// rounding point of the reference's op chain is reproduced.
#include "common.h"
#include <math.h>
#include <stdlib.h>

namespace ltxk {

constexpr int ROWS_PER_BLOCK = 4;  // one wave per row, 4 waves per workgroup
constexpr int MAX_CHUNKS = 16;     // D <= 16*512 = 8192

// ---------------------------------------------------------------------------------------
// rms_norm / layer_norm (no affine) + AdaLN modulation
// ---------------------------------------------------------------------------------------
// One wave per row.  Tried and measured slower in the step (11.5-14 us -> 18.5 us per launch at M=2560): walking
// several rows per wave with the next row's loads in flight, and requesting the modulation rows up front.
template <bool LAYERNORM>
__global__ __launch_bounds__(256) void norm_modulate_kernel(
    const bf16* __restrict__ x, bf16* __restrict__ y, int M, int D, float eps,
    const bf16* __restrict__ scale, const bf16* __restrict__ shift, int mod_stride,
    const int32_t* __restrict__ mod_row) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= M) return;
  const bf16* xr = x + (size_t)row * D;
  const int nch = D >> 9;  // chunks of 512 elements (64 lanes x 8)
  bf16x8 v[MAX_CHUNKS];
  float sum = 0.f, sq = 0.f;
#pragma unroll
  for (int i = 0; i < MAX_CHUNKS; ++i) {
    if (i < nch) {
      v[i] = *(const bf16x8*)(xr + i * 512 + lane * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)v[i][j];
        sum += f;
        sq += f * f;
      }
    }
  }
  float mean = 0.f, rstd;
  if constexpr (LAYERNORM) {
    mean = wave_sum(sum) / (float)D;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_CHUNKS; ++i)
      if (i < nch) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = (float)v[i][j] - mean;
          var += d * d;
        }
      }
    var = wave_sum(var) / (float)D;
    rstd = rsqrtf(var + eps);
  } else {
    rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
  }
  const size_t mrow = scale ? (size_t)(mod_row ? mod_row[row] : 0) * mod_stride : 0;
  bf16* yr = y + (size_t)row * D;
#pragma unroll
  for (int i = 0; i < MAX_CHUNKS; ++i) {
    if (i < nch) {
      const int col = i * 512 + lane * 8;
      bf16x8 o;
      if (scale) {
        const bf16x8 sc = *(const bf16x8*)(scale + mrow + col);
        const bf16x8 sh = *(const bf16x8*)(shift + mrow + col);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float n = rbf(((float)v[i][j] - mean) * rstd);
          const float one_p = rbf(1.0f + (float)sc[j]);
          o[j] = (bf16)(rbf(n * one_p) + (float)sh[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)(((float)v[i][j] - mean) * rstd);
      }
      *(bf16x8*)(yr + col) = o;
    }
  }
}

// rms_norm + modulation when the row's sum of squares is already known (the producing GEMM's `sumsq` output: NP fp32
// partials per row).  With the statistic precomputed nothing ties a row to one wave: a row is cut into WPR pieces of
// 64*EPL elements, one per wave, each a short independent chain {1 partial load + x loads in flight together ->
// 6-step wave sum -> a dozen VALU ops per element -> store}.  The wave-per-row kernel above keeps a 4096-element row
// in one wave (8 KB of loads, then a reduction that waits for all of them, then 64 elements of arithmetic per lane):
// at M=2560 that is 10 waves per CU each running a long serial chain, i.e. latency-bound at ~3 TB/s.
// ONE_PLUS: `scale` already holds bf16(1 + scale) (ltxk_ada_combine's one_plus_mask), saving an add and a rounding
// per element.
template <int CH, bool ONE_PLUS>     // CH = 16-byte chunks per lane
__global__ __launch_bounds__(256) void norm_scale_kernel(
    const bf16* __restrict__ x, bf16* __restrict__ y, int M, int D, float eps, const float* __restrict__ sumsq, int ss_ld, int NP,
    const bf16* __restrict__ scale, const bf16* __restrict__ shift, int mod_stride, const int32_t* __restrict__ mod_row) {
  const int lane = threadIdx.x & 63;
  const int wpr = D / (512 * CH);                                  // waves per row
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int row = item / wpr, piece = item - row * wpr;
  if (row >= M) return;
  const int col0 = piece * (512 * CH) + lane * 8;
  const bf16* xr = x + (size_t)row * D + col0;
  // Every load of the wave is requested before the reduction, the token -> row index first (it is the oldest load, so the
  // modulation loads that depend on it wait for it alone): one memory latency in the chain - two with a row map - not three.
  const size_t mrow = scale ? (size_t)(mod_row ? mod_row[row] : 0) * mod_stride : 0;
  bf16x8 v[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) v[i] = *(const bf16x8*)(xr + i * 512);
  const float ss0 = lane < NP ? sumsq[(size_t)row * ss_ld + lane] : 0.f;
  bf16x8 scv[CH], shv[CH];
  if (scale) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      scv[i] = *(const bf16x8*)(scale + mrow + col0 + i * 512);
      shv[i] = *(const bf16x8*)(shift + mrow + col0 + i * 512);
    }
  }
  float ss = ss0;
  for (int i = lane + 64; i < NP; i += 64) ss += sumsq[(size_t)row * ss_ld + i];
  const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
  bf16* yr = y + (size_t)row * D + col0;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    bf16x8 o;
    if (scale) {
      const bf16x8 sc = scv[i];
      const bf16x8 sh = shv[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float n = rbf((float)v[i][j] * rstd);
        const float one_p = ONE_PLUS ? (float)sc[j] : rbf(1.0f + (float)sc[j]);
        o[j] = (bf16)(rbf(n * one_p) + (float)sh[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)v[i][j] * rstd);
    }
    *(bf16x8*)(yr + i * 512) = o;
  }
}

// ---------------------------------------------------------------------------------------
// q/k RMSNorm (full inner dim, learned weight) + SPLIT RoPE, in place.
// One wave per (row, segment).  Lane map: a wave pass covers 8 heads; 8 lanes per head; lane holds
// x1 = [j0, j0+8) of the head's first half and x2 = the same offsets of the second half (the rotation
// partners), so every access is a 16-byte vector and each 8-lane group touches whole 128-byte lines.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qknorm_rope_kernel(
    bf16* __restrict__ buf, int ld, int M, int nseg, int D, const bf16* __restrict__ weight,
    const float* __restrict__ cosb, const float* __restrict__ sinb, int T, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (item >= M * nseg) return;
  const int row = item / nseg, sgi = item - row * nseg;
  const int t = row % T;
  const int npass = (H + 7) >> 3;    // 8 heads per pass (lanes past the last head idle), dh = 128
  const int hl = lane >> 3;          // head within pass
  const int j0 = (lane & 7) * 8;     // offset within the 64-wide half
  constexpr int MAXP = 4;            // H <= 32
  bf16* xr = buf + (size_t)row * ld + (size_t)sgi * D;
  const bf16* wr = weight + (size_t)sgi * D;
  bf16x8 a[MAXP], b[MAXP];
  float sq = 0.f;
#pragma unroll
  for (int ps = 0; ps < MAXP; ++ps)
    if (ps < npass && ps * 8 + hl < H) {
      const int base = (ps * 8 + hl) * 128 + j0;
      a[ps] = *(const bf16x8*)(xr + base);
      b[ps] = *(const bf16x8*)(xr + base + 64);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float fa = (float)a[ps][j], fb = (float)b[ps][j];
        sq += fa * fa + fb * fb;
      }
    }
  const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
#pragma unroll
  for (int ps = 0; ps < MAXP; ++ps)
    if (ps < npass && ps * 8 + hl < H) {
      const int base = (ps * 8 + hl) * 128 + j0;
      const bf16x8 wa = *(const bf16x8*)(wr + base);
      const bf16x8 wb = *(const bf16x8*)(wr + base + 64);
      f32x4 c0, c1, s0, s1;
      if (cosb) {
        const size_t off = ((size_t)(ps * 8 + hl) * T + t) * 64 + j0;
        c0 = *(const f32x4*)(cosb + off); c1 = *(const f32x4*)(cosb + off + 4);
        s0 = *(const f32x4*)(sinb + off); s1 = *(const f32x4*)(sinb + off + 4);
      }
      bf16x8 oa, ob;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x1 = rbf((float)a[ps][j] * rstd * (float)wa[j]);
        const float x2 = rbf((float)b[ps][j] * rstd * (float)wb[j]);
        if (cosb) {
          const float c = j < 4 ? c0[j & 3] : c1[j & 3], sn = j < 4 ? s0[j & 3] : s1[j & 3];
          oa[j] = (bf16)(x1 * c - sn * x2);
          ob[j] = (bf16)(x2 * c + sn * x1);
        } else {
          oa[j] = (bf16)x1;
          ob[j] = (bf16)x2;
        }
      }
      *(bf16x8*)(xr + base) = oa;
      *(bf16x8*)(xr + base + 64) = ob;
    }
}

// The same with the row's sum of squares precomputed (GEMM `sumsq` output, NP partials per segment): one wave per
// (row, segment, pass of 8 heads) - 4x the waves of the kernel above at H=32, each 2 + 2 row loads deep.
__global__ __launch_bounds__(256) void qknorm_rope_ss_kernel(
    bf16* __restrict__ buf, int ld, int M, int nseg, int D, const bf16* __restrict__ weight,
    const float* __restrict__ cosb, const float* __restrict__ sinb, int T, int H, float eps,
    const float* __restrict__ sumsq, int ss_ld, int NP) {
  const int lane = threadIdx.x & 63;
  const int npass = (H + 7) >> 3;
  const int item = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (item >= M * nseg * npass) return;
  const int ps = item % npass;
  const int rs = item / npass;
  const int row = rs / nseg, sgi = rs - row * nseg;
  const int t = row % T;
  const int hl = lane >> 3, j0 = (lane & 7) * 8;
  const int head = ps * 8 + hl;
  bf16* xr = buf + (size_t)row * ld + (size_t)sgi * D;
  const bf16* wr = weight + (size_t)sgi * D;
  const bool act = head < H;
  const int base = (act ? head : 0) * 128 + j0;
  bf16x8 a, b;
  if (act) {
    a = *(const bf16x8*)(xr + base);
    b = *(const bf16x8*)(xr + base + 64);
  }
  float ss = 0.f;
  for (int i = lane; i < NP; i += 64) ss += sumsq[(size_t)row * ss_ld + sgi * NP + i];
  // weights and the rotation table are requested before the reduction (they do not depend on it)
  bf16x8 wa, wb;
  f32x4 c0, c1, s0, s1;
  if (act) {
    wa = *(const bf16x8*)(wr + base);
    wb = *(const bf16x8*)(wr + base + 64);
    if (cosb) {
      const size_t off = ((size_t)head * T + t) * 64 + j0;
      c0 = *(const f32x4*)(cosb + off); c1 = *(const f32x4*)(cosb + off + 4);
      s0 = *(const f32x4*)(sinb + off); s1 = *(const f32x4*)(sinb + off + 4);
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
  if (!act) return;
  bf16x8 oa, ob;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x1 = rbf((float)a[j] * rstd * (float)wa[j]);
    const float x2 = rbf((float)b[j] * rstd * (float)wb[j]);
    if (cosb) {
      const float c = j < 4 ? c0[j & 3] : c1[j & 3], sn = j < 4 ? s0[j & 3] : s1[j & 3];
      oa[j] = (bf16)(x1 * c - sn * x2);
      ob[j] = (bf16)(x2 * c + sn * x1);
    } else {
      oa[j] = (bf16)x1;
      ob[j] = (bf16)x2;
    }
  }
  *(bf16x8*)(xr + base) = oa;
  *(bf16x8*)(xr + base + 64) = ob;
}

// ---------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------
__global__ void timestep_embed_kernel(const bf16* __restrict__ t, bf16* __restrict__ out, int U, int dim, float mult) {
  const int half = dim >> 1;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= U * half) return;
  const int u = idx / half, i = idx - u * half;
  const float freq = expf(-9.210340371976184f * (float)i / (float)half);  // -ln(1e4)
  const float arg = rbf((float)t[u] * mult) * freq;   // timestep*1000 stays bf16 (ltx.py:68)
  out[(size_t)u * dim + i] = (bf16)cosf(arg);          // flip_sin_to_cos: cos first
  out[(size_t)u * dim + half + i] = (bf16)sinf(arg);
}

// RoPE table (SPLIT layout): rope.py:419-529.  f < pad: cos=1,sin=0 (front pad, rope.py:504-509);
// else idx=(f-pad)/3, dim=(f-pad)%3; angle = ((mid/max_pos[dim])*2-1) * freq[idx].
__global__ void rope_table_kernel(const float* __restrict__ pos, const float* __restrict__ freq,
                                  float* __restrict__ cosb, float* __restrict__ sinb, int T, int H,
                                  int half_dim, int pad, float mp0, float mp1, float mp2) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * half_dim) return;
  const int t = idx / half_dim, f = idx - t * half_dim;
  float c = 1.f, s = 0.f;
  if (f >= pad) {
    const int fi = (f - pad) / 3, d = (f - pad) - fi * 3;
    const float st = pos[((size_t)d * T + t) * 2], en = pos[((size_t)d * T + t) * 2 + 1];
    const float mid = (st + en) / 2.0f;
    const float mp = d == 0 ? mp0 : (d == 1 ? mp1 : mp2);
    const float frac = __fdiv_rn(mid, mp);
    const float ang = (frac * 2.0f - 1.0f) * freq[fi];
    c = cosf(ang);
    s = sinf(ang);
  }
  const int per_head = half_dim / H;
  const int h = f / per_head, j = f - h * per_head;
  const size_t o = ((size_t)h * T + t) * per_head + j;
  cosb[o] = c;
  sinb[o] = s;
}

__global__ void ada_combine_kernel(const bf16* __restrict__ table, const bf16* __restrict__ ada,
                                   bf16* __restrict__ out, int L, int U, int K, int D, unsigned one_plus_mask) {
  // out[l,u,k,d] = bf16(table[l,k,d] + ada[u,k,d]) (+1, rounded again, for the k in one_plus_mask); 8 elements per thread
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t kd8 = (size_t)K * D / 8;
  const size_t total = (size_t)L * U * kd8;
  if (idx >= total) return;
  const size_t e = idx % kd8;
  const size_t lu = idx / kd8;
  const int u = (int)(lu % U), l = (int)(lu / U);
  const bf16x8 a = *(const bf16x8*)(table + ((size_t)l * kd8 + e) * 8);
  const bf16x8 b = *(const bf16x8*)(ada + ((size_t)u * kd8 + e) * 8);
  const bool op = (one_plus_mask >> (unsigned)((e * 8) / D)) & 1u;
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = rbf((float)a[j] + (float)b[j]);
    o[j] = (bf16)(op ? 1.0f + v : v);
  }
  *(bf16x8*)(out + idx * 8) = o;
}

__global__ void silu_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int64_t n8) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n8) return;
  const bf16x8 v = *(const bf16x8*)(x + idx * 8);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float f = (float)v[j];
    o[j] = (bf16)(f / (1.0f + expf(-f)));
  }
  *(bf16x8*)(y + idx * 8) = o;
}

// (B,C,S) -> (rep*B,S,C): thread = (b, s, 8-channel group); lanes run along s (coalesced reads).
__global__ void latent_to_tokens_kernel(const bf16* __restrict__ lat, bf16* __restrict__ tok,
                                        int B, int C, int S, int rep) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  const int cg = blockIdx.y, b = blockIdx.z;
  if (s >= S) return;
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = lat[((size_t)b * C + cg * 8 + j) * S + s];
  for (int r = 0; r < rep; ++r)
    *(bf16x8*)(tok + (((size_t)(r * B + b)) * S + s) * C + cg * 8) = v;
}

__global__ void cfg_euler_kernel(const bf16* __restrict__ vp, const bf16* __restrict__ vn,
                                 const bf16* __restrict__ lat, bf16* __restrict__ out,
                                 const bf16* __restrict__ clean, const float* __restrict__ mask,
                                 int B, int C, int S, float cfg, float sigma, float sigma_next,
                                 const float* __restrict__ sig_dev, int flags) {
  if (sig_dev) {            // graph replay: the two scalars live in device memory
    sigma = sig_dev[0];
    sigma_next = sig_dev[1];
  }
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  const int cg = blockIdx.y, b = blockIdx.z;
  if (s >= S) return;
  const size_t tokoff = ((size_t)b * S + s) * C + cg * 8;
  const bf16x8 p = *(const bf16x8*)(vp + tokoff);
  bf16x8 n = p;
  if (vn) n = *(const bf16x8*)(vn + tokoff);
  float m = 1.f;
  if (mask) m = mask[(size_t)b * S + s];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const size_t li = ((size_t)b * C + cg * 8 + j) * S + s;
    float v = (float)p[j];
    if (vn) v = rbf(v + rbf((cfg - 1.0f) * rbf(v - (float)n[j])));
    const float x = (float)lat[li];
    float x0 = rbf(x - sigma * v);
    if (mask) x0 = rbf(rbf(x0 * m) + rbf((float)clean[li] * rbf(1.0f - m)));
    float o = x0;
    if (flags & LTXK_STEP_BF16_EULER) {
      // fp32_euler=False (generate.py:748): every op of x0 + s'*(x - x0)/s materialises a bf16 array
      o = x0 + rbf(__fdiv_rn(rbf(sigma_next * rbf(x - x0)), sigma));
    } else if (sigma_next > 0.f) {
      const float t1 = x - x0;
      const float t2 = sigma_next * t1;
      o = x0 + __fdiv_rn(t2, sigma);
    }
    out[li] = (bf16)o;
  }
}

__global__ void euler_kernel(const bf16* __restrict__ x, const bf16* __restrict__ x0, bf16* __restrict__ out,
                             int64_t n8, float sigma, float sigma_next) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n8) return;
  const bf16x8 a = *(const bf16x8*)(x + idx * 8);
  const bf16x8 d = *(const bf16x8*)(x0 + idx * 8);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float df = (float)d[j];
    o[j] = (bf16)(df + __fdiv_rn(sigma_next * ((float)a[j] - df), sigma));
  }
  *(bf16x8*)(out + idx * 8) = o;
}

// One wave.  Row *step of the per-step scalar tables -> the buffers the step's kernels read; *step += 1.
__global__ void step_scalars_kernel(const bf16* __restrict__ ts_all, const float* __restrict__ sig_all,
                                    int32_t* __restrict__ step, bf16* __restrict__ ts, float* __restrict__ sig,
                                    int U, int n_steps) {
  int s = *step;
  s = s < n_steps ? s : n_steps - 1;
  __syncthreads();
  for (int i = threadIdx.x; i < U; i += blockDim.x) ts[i] = ts_all[(size_t)s * U + i];
  if (threadIdx.x < 2) sig[threadIdx.x] = sig_all[2 * s + threadIdx.x];
  if (threadIdx.x == 0) *step = s + 1;
}

}  // namespace ltxk

using namespace ltxk;

extern "C" int ltxk_step_scalars(const void* ts_all, const float* sig_all, int32_t* step, void* ts, float* sig,
                                 int32_t U, int32_t n_steps, void* stream) {
  LTXK_CHECK_ARG(ts_all && sig_all && step && ts && sig && U > 0 && n_steps > 0, "ltxk_step_scalars: bad arguments");
  hipLaunchKernelGGL(step_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16*)ts_all, sig_all, step,
                     (bf16*)ts, sig, U, n_steps);
  LTXK_CHECK_LAUNCH("ltxk_step_scalars");
  return LTXK_OK;
}

extern "C" int ltxk_euler_step(const void* latent, const void* denoised, void* out, int64_t n,
                               float sigma, float sigma_next, void* stream) {
  LTXK_CHECK_ARG(latent && denoised && out && n > 0 && n % 8 == 0, "ltxk_euler_step: n must be a positive multiple of 8");
  LTXK_CHECK_ARG(sigma > 0.f, "ltxk_euler_step: sigma must be > 0");
  const int64_t n8 = n / 8;
  hipLaunchKernelGGL(euler_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)latent, (const bf16*)denoised, (bf16*)out, n8, sigma, sigma_next);
  LTXK_CHECK_LAUNCH("ltxk_euler_step");
  return LTXK_OK;
}

template <int CH>
static void norm_scale_launch(bool one_plus, dim3 grid, hipStream_t st, const bf16* x, bf16* y, int M, int D, float eps,
                              const float* ss, int ss_ld, int np, const bf16* sc, const bf16* sh, int ms, const int32_t* mr) {
  if (one_plus) hipLaunchKernelGGL((norm_scale_kernel<CH, true>), grid, dim3(256), 0, st, x, y, M, D, eps, ss, ss_ld, np, sc, sh, ms, mr);
  else hipLaunchKernelGGL((norm_scale_kernel<CH, false>), grid, dim3(256), 0, st, x, y, M, D, eps, ss, ss_ld, np, sc, sh, ms, mr);
}

extern "C" int ltxk_rmsnorm_modulate_ss(const void* x, void* y, int32_t M, int32_t D, float eps, const float* sumsq,
                                        int32_t sumsq_ld, int32_t sumsq_n, const void* scale, const void* shift,
                                        int32_t mod_stride, const int32_t* mod_row, int32_t flags, void* stream) {
  const char* name = "ltxk_rmsnorm_modulate_ss";
  LTXK_CHECK_ARG(x && y && sumsq && M > 0, "%s: null/empty input", name);
  LTXK_CHECK_ARG(D % 512 == 0 && D > 0, "%s: D=%d must be a multiple of 512", name, D);
  LTXK_CHECK_ARG(sumsq_n > 0 && sumsq_ld >= sumsq_n, "%s: bad sumsq_n/sumsq_ld", name);
  LTXK_CHECK_ARG((scale == nullptr) == (shift == nullptr), "%s: scale and shift must both be set or both NULL", name);
  LTXK_CHECK_ARG(!scale || mod_stride % 8 == 0, "%s: mod_stride must be a multiple of 8", name);
  // 16-byte chunks per lane: 2 (rows cut into 1024-element pieces) when D allows, else 1
  const int ch_env = LTXK_AB_INT("LTXK_NORM_CH", 0);
  int ch = (D % 1024 == 0) ? 2 : 1;
  if ((ch_env == 1 || ch_env == 2 || ch_env == 4 || ch_env == 8) && D % (512 * ch_env) == 0) ch = ch_env;
  const int wpr = D / (512 * ch);
  const dim3 grid((unsigned)(((long long)M * wpr + 3) / 4));
  const bool op = (flags & LTXK_NORM_SCALE_IS_ONE_PLUS) != 0;
  hipStream_t st = (hipStream_t)stream;
  switch (ch) {
    case 1: norm_scale_launch<1>(op, grid, st, (const bf16*)x, (bf16*)y, M, D, eps, sumsq, sumsq_ld, sumsq_n, (const bf16*)scale, (const bf16*)shift, mod_stride, mod_row); break;
    case 2: norm_scale_launch<2>(op, grid, st, (const bf16*)x, (bf16*)y, M, D, eps, sumsq, sumsq_ld, sumsq_n, (const bf16*)scale, (const bf16*)shift, mod_stride, mod_row); break;
    case 4: norm_scale_launch<4>(op, grid, st, (const bf16*)x, (bf16*)y, M, D, eps, sumsq, sumsq_ld, sumsq_n, (const bf16*)scale, (const bf16*)shift, mod_stride, mod_row); break;
    default: norm_scale_launch<8>(op, grid, st, (const bf16*)x, (bf16*)y, M, D, eps, sumsq, sumsq_ld, sumsq_n, (const bf16*)scale, (const bf16*)shift, mod_stride, mod_row); break;
  }
  LTXK_CHECK_LAUNCH(name);
  return LTXK_OK;
}

static int norm_modulate_launch(bool ln, const void* x, void* y, int32_t M, int32_t D, float eps,
                                const void* scale, const void* shift, int32_t mod_stride,
                                const int32_t* mod_row, void* stream, const char* name) {
  LTXK_CHECK_ARG(x && y && M > 0, "%s: null/empty input", name);
  LTXK_CHECK_ARG(D % 512 == 0 && D <= 512 * MAX_CHUNKS, "%s: D=%d must be a multiple of 512, <= %d", name, D, 512 * MAX_CHUNKS);
  LTXK_CHECK_ARG((scale == nullptr) == (shift == nullptr), "%s: scale and shift must both be set or both NULL", name);
  LTXK_CHECK_ARG(!scale || mod_stride % 8 == 0, "%s: mod_stride must be a multiple of 8", name);
  const dim3 grid((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
  if (ln)
    hipLaunchKernelGGL(norm_modulate_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, M, D, eps,
                       (const bf16*)scale, (const bf16*)shift, mod_stride, mod_row);
  else
    hipLaunchKernelGGL(norm_modulate_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, M, D, eps,
                       (const bf16*)scale, (const bf16*)shift, mod_stride, mod_row);
  LTXK_CHECK_LAUNCH(name);
  return LTXK_OK;
}

extern "C" int ltxk_rmsnorm_modulate(const void* x, void* y, int32_t M, int32_t D, float eps,
                                     const void* scale, const void* shift, int32_t mod_stride,
                                     const int32_t* mod_row, void* stream) {
  return norm_modulate_launch(false, x, y, M, D, eps, scale, shift, mod_stride, mod_row, stream, "ltxk_rmsnorm_modulate");
}

extern "C" int ltxk_layernorm_modulate(const void* x, void* y, int32_t M, int32_t D, float eps,
                                       const void* scale, const void* shift, int32_t mod_stride,
                                       const int32_t* mod_row, void* stream) {
  return norm_modulate_launch(true, x, y, M, D, eps, scale, shift, mod_stride, mod_row, stream, "ltxk_layernorm_modulate");
}

extern "C" int ltxk_qknorm_rope_ss(void* buf, int32_t ld, int32_t M, int32_t nseg, int32_t D,
                                   const void* weight, const float* cos, const float* sin,
                                   int32_t T, int32_t H, float eps, const float* sumsq, int32_t sumsq_ld, void* stream) {
  LTXK_CHECK_ARG(buf && weight && sumsq && M > 0 && nseg > 0, "ltxk_qknorm_rope_ss: null/empty input");
  LTXK_CHECK_ARG(D == H * 128 && H % 4 == 0, "ltxk_qknorm_rope_ss: need D == H*128, H %% 4 == 0 (D=%d H=%d)", D, H);
  LTXK_CHECK_ARG(ld >= nseg * D && ld % 8 == 0 && ((uintptr_t)buf & 15) == 0, "ltxk_qknorm_rope_ss: ld=%d must be >= nseg*D and a multiple of 8, buf 16-byte aligned", ld);
  LTXK_CHECK_ARG((cos == nullptr) == (sin == nullptr), "ltxk_qknorm_rope_ss: cos and sin must both be set or both NULL");
  LTXK_CHECK_ARG(T > 0 && sumsq_ld >= nseg * (D / 64), "ltxk_qknorm_rope_ss: T must be > 0, sumsq_ld >= nseg*D/64");
  const int npass = (H + 7) / 8;
  const long long items = (long long)M * nseg * npass;
  hipLaunchKernelGGL(qknorm_rope_ss_kernel, dim3((unsigned)((items + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream,
                     (bf16*)buf, ld, M, nseg, D, (const bf16*)weight, cos, sin, T, H, eps, sumsq, sumsq_ld, D / 64);
  LTXK_CHECK_LAUNCH("ltxk_qknorm_rope_ss");
  return LTXK_OK;
}

extern "C" int ltxk_qknorm_rope(void* buf, int32_t ld, int32_t M, int32_t nseg, int32_t D,
                                const void* weight, const float* cos, const float* sin,
                                int32_t T, int32_t H, float eps, void* stream) {
  LTXK_CHECK_ARG(buf && weight && M > 0 && nseg > 0, "ltxk_qknorm_rope: null/empty input");
  LTXK_CHECK_ARG(D == H * 128 && H % 4 == 0 && H <= 32, "ltxk_qknorm_rope: need D == H*128, H %% 4 == 0, H <= 32 (D=%d H=%d)", D, H);
  LTXK_CHECK_ARG(ld >= nseg * D && ld % 8 == 0 && ((uintptr_t)buf & 15) == 0, "ltxk_qknorm_rope: ld=%d must be >= nseg*D and a multiple of 8, buf 16-byte aligned", ld);
  LTXK_CHECK_ARG((cos == nullptr) == (sin == nullptr), "ltxk_qknorm_rope: cos and sin must both be set or both NULL");
  LTXK_CHECK_ARG(T > 0, "ltxk_qknorm_rope: T must be > 0");
  const dim3 grid((M * nseg + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
  hipLaunchKernelGGL(qknorm_rope_kernel, grid, dim3(256), 0, (hipStream_t)stream, (bf16*)buf, ld, M, nseg, D,
                     (const bf16*)weight, cos, sin, T, H, eps);
  LTXK_CHECK_LAUNCH("ltxk_qknorm_rope");
  return LTXK_OK;
}

extern "C" int ltxk_rope_table(const float* positions, const float* freq, float* cos, float* sin,
                               int32_t T, int32_t H, int32_t dim, int32_t n_freq, const float* max_pos,
                               void* stream) {
  LTXK_CHECK_ARG(positions && freq && cos && sin && max_pos, "ltxk_rope_table: null pointer");
  LTXK_CHECK_ARG(T > 0 && H > 0 && dim % (2 * H) == 0, "ltxk_rope_table: bad dims");
  const int half_dim = dim / 2;
  const int pad = half_dim - 3 * n_freq;
  LTXK_CHECK_ARG(pad >= 0, "ltxk_rope_table: 3*n_freq exceeds dim/2");
  const int total = T * half_dim;
  hipLaunchKernelGGL(rope_table_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     positions, freq, cos, sin, T, H, half_dim, pad, max_pos[0], max_pos[1], max_pos[2]);
  LTXK_CHECK_LAUNCH("ltxk_rope_table");
  return LTXK_OK;
}

extern "C" int ltxk_timestep_embed(const void* t, void* out, int32_t U, int32_t dim, float mult, void* stream) {
  LTXK_CHECK_ARG(t && out && U > 0 && dim > 0 && dim % 2 == 0, "ltxk_timestep_embed: bad arguments");
  const int total = U * (dim / 2);
  hipLaunchKernelGGL(timestep_embed_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)t, (bf16*)out, U, dim, mult);
  LTXK_CHECK_LAUNCH("ltxk_timestep_embed");
  return LTXK_OK;
}

extern "C" int ltxk_ada_combine(const void* table, const void* ada, void* out, int32_t L, int32_t U,
                                int32_t K, int32_t D, uint32_t one_plus_mask, void* stream) {
  LTXK_CHECK_ARG(table && ada && out && L > 0 && U > 0 && K > 0 && K <= 32 && D > 0 && D % 8 == 0, "ltxk_ada_combine: bad arguments");
  const size_t total = (size_t)L * U * K * D / 8;
  hipLaunchKernelGGL(ada_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)table, (const bf16*)ada, (bf16*)out, L, U, K, D, one_plus_mask);
  LTXK_CHECK_LAUNCH("ltxk_ada_combine");
  return LTXK_OK;
}

extern "C" int ltxk_silu(const void* x, void* y, int64_t n, void* stream) {
  LTXK_CHECK_ARG(x && y && n > 0 && n % 8 == 0, "ltxk_silu: n must be a positive multiple of 8");
  const int64_t n8 = n / 8;
  hipLaunchKernelGGL(silu_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)x, (bf16*)y, n8);
  LTXK_CHECK_LAUNCH("ltxk_silu");
  return LTXK_OK;
}

extern "C" int ltxk_latent_to_tokens(const void* latent, void* tokens, int32_t B, int32_t C, int32_t S,
                                     int32_t rep, void* stream) {
  LTXK_CHECK_ARG(latent && tokens && B > 0 && S > 0 && C > 0 && C % 8 == 0 && rep >= 1, "ltxk_latent_to_tokens: bad arguments");
  hipLaunchKernelGGL(latent_to_tokens_kernel, dim3((S + 63) / 64, C / 8, B), dim3(64), 0, (hipStream_t)stream,
                     (const bf16*)latent, (bf16*)tokens, B, C, S, rep);
  LTXK_CHECK_LAUNCH("ltxk_latent_to_tokens");
  return LTXK_OK;
}

static int cfg_euler_launch(const void* v_pos, const void* v_neg, const void* latent, void* out, const void* clean,
                            const float* mask, int32_t B, int32_t C, int32_t S, float cfg_scale, float sigma,
                            float sigma_next, const float* sig_dev, int32_t flags, void* stream, const char* name) {
  LTXK_CHECK_ARG(v_pos && latent && out && B > 0 && S > 0 && C > 0 && C % 8 == 0, "%s: bad arguments", name);
  LTXK_CHECK_ARG((clean == nullptr) == (mask == nullptr), "%s: clean and mask must both be set or both NULL", name);
  LTXK_CHECK_ARG(sig_dev != nullptr || sigma > 0.f, "%s: sigma must be > 0", name);
  hipLaunchKernelGGL(cfg_euler_kernel, dim3((S + 63) / 64, C / 8, B), dim3(64), 0, (hipStream_t)stream,
                     (const bf16*)v_pos, (const bf16*)v_neg, (const bf16*)latent, (bf16*)out, (const bf16*)clean, mask,
                     B, C, S, cfg_scale, sigma, sigma_next, sig_dev, (int)flags);
  LTXK_CHECK_LAUNCH(name);
  return LTXK_OK;
}

extern "C" int ltxk_cfg_euler_step(const void* v_pos, const void* v_neg, const void* latent, void* out,
                                   const void* clean, const float* mask, int32_t B, int32_t C, int32_t S,
                                   float cfg_scale, float sigma, float sigma_next, int32_t flags, void* stream) {
  return cfg_euler_launch(v_pos, v_neg, latent, out, clean, mask, B, C, S, cfg_scale, sigma, sigma_next, nullptr, flags, stream,
                          "ltxk_cfg_euler_step");
}

extern "C" int ltxk_cfg_euler_step_dev(const void* v_pos, const void* v_neg, const void* latent, void* out,
                                       const void* clean, const float* mask, int32_t B, int32_t C, int32_t S,
                                       float cfg_scale, const float* sigmas_dev, int32_t flags, void* stream) {
  LTXK_CHECK_ARG(sigmas_dev != nullptr, "ltxk_cfg_euler_step_dev: null sigmas_dev");
  return cfg_euler_launch(v_pos, v_neg, latent, out, clean, mask, B, C, S, cfg_scale, 1.f, 0.f, sigmas_dev, flags, stream,
                          "ltxk_cfg_euler_step_dev");
}

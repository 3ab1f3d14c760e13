// HBM-bound kernels of the video VAE (channels-last volumes, rows = voxels): pixel-norm +
// AdaLN + SiLU, depth-to-space / space-to-depth rearrangements with their residual paths,
// latent (de)normalise + layout change, (un)patchify, uint8 conversion.  Every bf16 rounding
// point of the reference's op-by-op bf16 chain is reproduced (see oracle/vae.py).
#include "common.h"
#include <math.h>

namespace ltxk {

// ---------------------------------------------------------------------------------------
// pixel_norm [+ (1+scale)+shift] [+ SiLU]   decoder.py:136-180,415-437; utils.py:477-483
//   q = bf16(x^2); m = bf16(mean_c q); e = bf16(m+eps); s = bf16(sqrt e); y = bf16(x/s)
// LPR lanes per row (8 channels per lane per pass), 64/LPR rows per wave.
// ---------------------------------------------------------------------------------------
template <int LPR, int PASSES>
__global__ __launch_bounds__(256) void pixelnorm_act_kernel(
    const bf16* __restrict__ x, bf16* __restrict__ y, int64_t V, int C, float eps,
    const bf16* __restrict__ scale, const bf16* __restrict__ shift, int64_t rows_per_batch, int apply_silu) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t row = wave_id * RPW + lane / LPR;
  const int sub = lane % LPR;
  const bool live = row < V;
  const int64_t rr = live ? row : V - 1;
  const bf16* xr = x + rr * C;
  bf16x8 v[PASSES];
  float sq = 0.f;
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    v[ps] = *(const bf16x8*)(xr + (ps * LPR + sub) * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = (float)v[ps][j];
      sq += rbf(f * f);
    }
  }
  sq = group_sum<LPR>(sq);                       // permlane / DPP butterfly, same bits as the __shfl_xor one (common.h)
  const float m = rbf(sq / (float)C);
  const float sd = rbf(sqrtf(rbf(m + eps)));
  // x / sd for the whole row: reciprocal once, then one Newton step per element (q = x*r; q += (x - q*sd)*r).
  // For normal-range operands that is the fp32 quotient up to a rare last-bit difference, which the bf16
  // rounding that follows absorbs; the IEEE division sequence the compiler emits costs ~10 VALU ops per element
  // and made this kernel VALU-bound (1.43 -> 0.90 ms per 33x512x512 decode).
  const float rsd = 1.0f / sd;
  const int64_t bidx = scale ? rr / rows_per_batch : 0;
  if (!live) return;
  bf16* yr = y + row * C;
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int col = (ps * LPR + sub) * 8;
    bf16x8 o;
    bf16x8 sc, sh;
    if (scale) {
      sc = *(const bf16x8*)(scale + bidx * C + col);
      sh = *(const bf16x8*)(shift + bidx * C + col);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xv = (float)v[ps][j];
      const float q0 = xv * rsd;
      float t = rbf(__builtin_fmaf(__builtin_fmaf(-q0, sd, xv), rsd, q0));
      if (scale) t = rbf(rbf(t * rbf(1.0f + (float)sc[j])) + (float)sh[j]);
      if (apply_silu) t = silu_f(t);          // v_exp_f32 + v_rcp_f32 (1 ulp each), far below the bf16 it is rounded to
      o[j] = (bf16)t;
    }
    *(bf16x8*)(yr + col) = o;
  }
}

// ---------------------------------------------------------------------------------------
// depth-to-space (2,2,2) of the conv output + tiled residual of the input, first frame dropped
// (sampling.py:143-197).  conv: (B,D,H,W,8*Co); xin: (B,D,H,W,Ci) or NULL; out: (B,2D-1,2H,2W,Co)
//   out[b,2d+st-1,2h+sh,2w+sw,c] = bf16(conv[v, c*8+s] + xin[v, (c % (Ci/8))*8 + s]), s=st*4+sh*2+sw
// One workgroup per input voxel; thread c handles channel c for all 8 sub-positions.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void d2s_add_kernel(const bf16* __restrict__ conv, const bf16* __restrict__ xin,
                                                      bf16* __restrict__ out, int B, int D, int H, int W, int Co, int Ci) {
  const int64_t v = blockIdx.x;
  int64_t r = v;
  const int w = r % W; r /= W;
  const int h = r % H; r /= H;
  const int d = r % D;
  const int b = r / D;
  const int Do = 2 * D - 1, Ho = 2 * H, Wo = 2 * W;
  const int cr = Ci / 8;
  for (int c = threadIdx.x; c < Co; c += blockDim.x) {
    const bf16x8 a = *(const bf16x8*)(conv + (v * Co + c) * 8);
    bf16x8 rsd;
    if (xin) rsd = *(const bf16x8*)(xin + v * Ci + (c % cr) * 8);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int st = s >> 2, sh = (s >> 1) & 1, sw = s & 1;
      const int dd = 2 * d + st - 1;
      if (dd < 0) continue;
      float val = (float)a[s];
      if (xin) val = val + (float)rsd[s];
      out[((((int64_t)b * Do + dd) * Ho + 2 * h + sh) * Wo + 2 * w + sw) * Co + c] = (bf16)val;
    }
  }
}

// ---------------------------------------------------------------------------------------
// latents (B,C,S) channels-first -> (B,S,C) channels-last with fp32 per-channel x*std+mean
// (decoder.py:349-355); dir<0: (B,S,C) -> (B,C,S) with (x-mean)/std (ops.py:94-109).
// ---------------------------------------------------------------------------------------
__global__ void latent_denorm_cl_kernel(const bf16* __restrict__ lat, const bf16* __restrict__ noise, float noise_scale,
                                        const bf16* __restrict__ mean, const bf16* __restrict__ stdv,
                                        bf16* __restrict__ out, int B, int C, int64_t S) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cg = blockIdx.y, b = blockIdx.z;
  if (s >= S) return;
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    const int64_t li = ((int64_t)b * C + c) * S + s;
    float x = (float)lat[li];
    // timestep-conditioned decode: sample = noise*s + (1-s)*sample (decoder.py:381-385), bf16 op by op
    if (noise) x = rbf(rbf((float)noise[li] * noise_scale) + rbf((1.0f - noise_scale) * x));
    o[j] = (bf16)(x * (float)stdv[c] + (float)mean[c]);
  }
  *(bf16x8*)(out + ((int64_t)b * S + s) * C + cg * 8) = o;
}

__global__ void latent_norm_cf_kernel(const bf16* __restrict__ x, int ldx, const bf16* __restrict__ mean,
                                      const bf16* __restrict__ stdv, bf16* __restrict__ out, int B, int C, int64_t S) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cg = blockIdx.y, b = blockIdx.z;
  if (s >= S) return;
  const bf16x8 v = *(const bf16x8*)(x + ((int64_t)b * S + s) * ldx + cg * 8);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    out[((int64_t)b * C + c) * S + s] = (bf16)__fdiv_rn((float)v[j] - (float)mean[c], (float)stdv[c]);
  }
}

// ---------------------------------------------------------------------------------------
// unpatchify (ops.py:47-80): x (B,D,H,W,C*P*P) channels-last, channel = (c, p_w, p_h) ->
// video (B,C,D,H*P,W*P) channels-first.   patchify (ops.py:9-44) is the inverse, with the
// channel axis zero-padded to Cpad for the first convolution.
// ---------------------------------------------------------------------------------------
// One thread = the P output pixels of one patch row: P strided 2-byte reads inside one voxel's channel row, ONE P*2-byte store
// (P = 4: 8 bytes; consecutive threads write consecutive pixels - whole lines per wave; one element per thread ran at 0.9 TB/s).
__global__ void unpatchify_cf_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, int B, int D, int H, int W,
                                     int C, int P) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int Ho = H * P;
  const int64_t total = (int64_t)B * C * D * Ho * W;
  if (idx >= total) return;
  int64_t r = idx;
  const int w = r % W; r /= W;
  const int yo = r % Ho; r /= Ho;
  const int d = r % D; r /= D;
  const int c = r % C;
  const int b = r / C;
  const bf16* src = x + ((((int64_t)b * D + d) * H + yo / P) * W + w) * (C * P * P) + c * P * P + (yo % P);
  bf16* dst = out + idx * P;
  if (P == 4) {
    bf16x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = src[j * 4];
    *(bf16x4*)dst = v;
  } else {
    for (int j = 0; j < P; ++j) dst[j] = src[j * P];
  }
}

__global__ void patchify_cl_kernel(const bf16* __restrict__ vid, bf16* __restrict__ out, int B, int C, int D, int H, int W,
                                   int P, int Cpad) {
  // out (B,D,H/P,W/P,Cpad); channel (c,p_w,p_h) = vid[b,c,d,h*P+p_h,w*P+p_w]; pad channels = 0
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int Hp = H / P, Wp = W / P;
  const int64_t total = (int64_t)B * D * Hp * Wp * Cpad;
  if (idx >= total) return;
  int64_t r = idx;
  const int ch = r % Cpad; r /= Cpad;
  const int w = r % Wp; r /= Wp;
  const int h = r % Hp; r /= Hp;
  const int d = r % D;
  const int b = r / D;
  bf16 val = (bf16)0.f;
  if (ch < C * P * P) {
    const int ph = ch % P, pw = (ch / P) % P, c = ch / (P * P);
    val = vid[((((int64_t)b * C + c) * D + d) * H + h * P + ph) * W + w * P + pw];
  }
  out[idx] = val;
}

// video (B,C,F,H,W) bf16 -> uint8 (B,F,H,W,C): generate.py:3894-3898 in bf16, op by op.
__global__ void to_uint8_kernel(const bf16* __restrict__ x, uint8_t* __restrict__ out, int B, int C, int F, int H, int W) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * F * H * W * C;
  if (idx >= total) return;
  int64_t r = idx;
  const int c = r % C; r /= C;
  const int w = r % W; r /= W;
  const int h = r % H; r /= H;
  const int f = r % F;
  const int b = r / F;
  float v = (float)x[((((int64_t)b * C + c) * F + f) * H + h) * W + w];
  v = rbf(rbf(v + 1.0f) / 2.0f);
  v = fminf(fmaxf(v, 0.f), 1.f);
  v = rbf(v * 255.0f);
  out[idx] = (uint8_t)v;
}

// ---------------------------------------------------------------------------------------
// space-to-depth (encoder, sampling.py:53-103).  xin (B,D,H,W,C) channels-last.
// s2d of the conv branch: conv (B,Dp,Hp,Wp,Cc) computed on the (temporally front-padded) input,
// out[b,d,h,w, (c,p1,p2,p3)] = conv[b, d*st+p1, h*sh+p2, w*sw+p3, c]
//                            + mean_{g<G} s2d(xpad)[.., co*G+g]          (group-mean skip)
// where xpad = input with the first frame duplicated when st==2 (sampling.py:78-81).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void s2d_skip_kernel(const bf16* __restrict__ conv, const bf16* __restrict__ xpad,
                                                       bf16* __restrict__ out, int B, int Dp, int Hp, int Wp,
                                                       int Cc, int Cx, int st, int sh, int sw, int G) {
  // one workgroup per output voxel; thread per output channel
  const int Do = Dp / st, Ho = Hp / sh, Wo = Wp / sw;
  const int mult = st * sh * sw;
  const int Co = Cc * mult;
  int64_t r = blockIdx.x;
  const int w = r % Wo; r /= Wo;
  const int h = r % Ho; r /= Ho;
  const int d = r % Do;
  const int b = r / Do;
  for (int co = threadIdx.x; co < Co; co += blockDim.x) {
    // conv branch channel co = (c, p1, p2, p3)
    int q = co;
    const int p3 = q % sw; q /= sw;
    const int p2 = q % sh; q /= sh;
    const int p1 = q % st;
    const int c = q / st;
    const int64_t vsrc = ((((int64_t)b * Dp + d * st + p1) * Hp + h * sh + p2) * Wp + w * sw + p3);
    const float cv = (float)conv[vsrc * Cc + c];
    // skip: mean over G consecutive s2d channels co*G .. co*G+G-1 of the input (fp32 sum, bf16 result)
    float acc = 0.f;
    for (int g = 0; g < G; ++g) {
      int k = co * G + g;
      const int k3 = k % sw; k /= sw;
      const int k2 = k % sh; k /= sh;
      const int k1 = k % st;
      const int kc = k / st;
      const int64_t vs = ((((int64_t)b * Dp + d * st + k1) * Hp + h * sh + k2) * Wp + w * sw + k3);
      acc += (float)xpad[vs * Cx + kc];
    }
    const float sk = rbf(acc / (float)G);
    out[(int64_t)blockIdx.x * Co + co] = (bf16)(cv + sk);
  }
}

// ---------------------------------------------------------------------------------------
// GroupNorm3d (upsampler.py:65-98): fp32 mean/var over (voxels, C/G) per (batch, group), affine,
// cast to bf16; then the ResBlock tail: [+ residual] [SiLU] (upsampler.py:160-174).
// One workgroup per (batch, group); the volume is latent-sized (<= 9x24x24 voxels), so three
// sweeps over it (mean, centred variance, apply) stay in L2.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void groupnorm_act_kernel(const bf16* __restrict__ x, bf16* __restrict__ out,
                                                            const bf16* __restrict__ gamma, const bf16* __restrict__ beta,
                                                            const bf16* __restrict__ resid, int64_t V, int C, int G,
                                                            float eps, int apply_silu) {
  __shared__ float red[8];
  const int b = blockIdx.x / G, g = blockIdx.x % G;
  const int cg = C / G;
  const bf16* xb = x + (int64_t)b * V * C + g * cg;
  const int64_t n = V * cg;
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
  };
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += (float)xb[(i / cg) * C + (i % cg)];
  const float mean = block_sum(s) / (float)n;
  float q = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const float d = (float)xb[(i / cg) * C + (i % cg)] - mean;
    q += d * d;
  }
  const float var = block_sum(q) / (float)n;
  const float sd = sqrtf(var + eps);
  bf16* ob = out + (int64_t)b * V * C + g * cg;
  const bf16* rb = resid ? resid + (int64_t)b * V * C + g * cg : nullptr;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const int c = (int)(i % cg);
    const int64_t off = (i / cg) * C + c;
    float y = rbf(__fdiv_rn((float)xb[off] - mean, sd) * (float)gamma[g * cg + c] + (float)beta[g * cg + c]);
    if (rb) y = rbf(y + (float)rb[off]);
    if (apply_silu) y = y / (1.0f + expf(-y));
    ob[off] = (bf16)y;
  }
}

// ---------------------------------------------------------------------------------------
// tiled-decode blending (tiling.py:399-447,492-509): out += tile*mask, wsum += mask with
// mask = mt[t]*mh[h]*mw[w] (fp32 accumulators); finalize: out/max(wsum,1e-8) -> bf16.
// ---------------------------------------------------------------------------------------
__global__ void tile_blend_accum_kernel(const bf16* __restrict__ tile, int Tt, int Th, int Tw, int at, int ah, int aw,
                                        const float* __restrict__ mt, const float* __restrict__ mh,
                                        const float* __restrict__ mw, float* __restrict__ acc, float* __restrict__ wsum,
                                        int B, int C, int F, int H, int W, int t0, int h0, int w0) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * at * ah * aw;
  if (idx >= total) return;
  int64_t r = idx;
  const int x = r % aw; r /= aw;
  const int y = r % ah; r /= ah;
  const int t = r % at;
  const int b = r / at;
  const float m = mt[t] * mh[y] * mw[x];
  const int64_t opos = ((int64_t)(t0 + t) * H + (h0 + y)) * W + (w0 + x);
  for (int c = 0; c < C; ++c) {
    const float v = (float)tile[((((int64_t)b * C + c) * Tt + t) * Th + y) * Tw + x];
    acc[((int64_t)b * C + c) * F * H * W + opos] += v * m;
  }
  wsum[(int64_t)b * F * H * W + opos] += m;
}

__global__ void tile_blend_finalize_kernel(const float* __restrict__ acc, const float* __restrict__ wsum,
                                           bf16* __restrict__ out, int B, int C, int64_t S) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * C * S) return;
  const int64_t s = idx % S;
  const int64_t b = idx / (S * C);
  out[idx] = (bf16)__fdiv_rn(acc[idx], fmaxf(wsum[b * S + s], 1e-8f));
}

// ---- area resize of conditioning frames: cv2.resize(..., interpolation=cv2.INTER_AREA) on float frames, DOWNscaling
// (prepare_video_for_encoding, mlx_video/utils.py:699-705; OpenCV's resizeArea_ / computeResizeAreaTab).  Separable:
// along each axis destination pixel d covers the source interval [d*scale, (d+1)*scale); a source pixel's weight is the
// length of its overlap with it divided by min(scale, ssize - d*scale).  One thread per output pixel.
struct AreaSpan { int s1, s2; float wl, wm, wr; };     // sources s1-1 (weight wl, if > 0), s1 .. s2-1 (wm each), s2 (wr, if > 0)
__device__ __forceinline__ AreaSpan area_span(int d, double scale, int ssize) {
  const double f1 = d * scale, f2 = f1 + scale;
  const double cell = fmin(scale, (double)ssize - f1);
  int s1 = (int)ceil(f1), s2 = (int)floor(f2);
  s2 = s2 < ssize - 1 ? s2 : ssize - 1;
  s1 = s1 < s2 ? s1 : s2;
  AreaSpan a;
  a.s1 = s1; a.s2 = s2;
  a.wl = (s1 - f1 > 1e-3) ? (float)((s1 - f1) / cell) : 0.f;
  a.wm = (float)(1.0 / cell);
  a.wr = (f2 - s2 > 1e-3) ? (float)(fmin(fmin(f2 - s2, 1.0), cell) / cell) : 0.f;
  return a;
}

template <class T>
__global__ void resize_area_kernel(const T* __restrict__ x, bf16* __restrict__ out, int64_t planes, int H, int W, int OH, int OW,
                                   double sy, double sx) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= planes * OH * OW) return;
  const int ox = (int)(idx % OW);
  const int oy = (int)((idx / OW) % OH);
  const int64_t pl = idx / ((int64_t)OW * OH);
  const AreaSpan ax = area_span(ox, sx, W), ay = area_span(oy, sy, H);
  const T* src = x + pl * (int64_t)H * W;
  auto hrow = [&](int y) {                            // horizontal pass of one source row (cv2: buf[dx] += S[sx] * alpha)
    const T* r = src + (int64_t)y * W;
    float acc = 0.f;
    if (ax.wl > 0.f) acc += (float)r[ax.s1 - 1] * ax.wl;
    for (int s = ax.s1; s < ax.s2; ++s) acc += (float)r[s] * ax.wm;
    if (ax.wr > 0.f) acc += (float)r[ax.s2] * ax.wr;
    return acc;
  };
  float sum = 0.f;                                    // vertical pass (cv2: sum[dx] += beta * buf[dx])
  if (ay.wl > 0.f) sum += ay.wl * hrow(ay.s1 - 1);
  for (int s = ay.s1; s < ay.s2; ++s) sum += ay.wm * hrow(s);
  if (ay.wr > 0.f) sum += ay.wr * hrow(ay.s2);
  out[idx] = (bf16)sum;
}

}  // namespace ltxk

extern "C" int ltxk_resize_area(const void* x, int32_t x_is_f32, void* out, int64_t planes, int32_t H, int32_t W,
                                int32_t OH, int32_t OW, void* stream) {
  using namespace ltxk;
  LTXK_CHECK_ARG(x && out && planes > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "ltxk_resize_area: bad arguments");
  LTXK_CHECK_ARG(OH <= H && OW <= W, "ltxk_resize_area: %dx%d -> %dx%d is not a downscale (INTER_AREA enlarges by a different rule)", H, W, OH, OW);
  const int64_t total = planes * OH * OW;
  const dim3 grid((unsigned)((total + 255) / 256));
  const double sy = (double)H / OH, sx = (double)W / OW;
  if (x_is_f32) hipLaunchKernelGGL(resize_area_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (bf16*)out, planes, H, W, OH, OW, sy, sx);
  else hipLaunchKernelGGL(resize_area_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)out, planes, H, W, OH, OW, sy, sx);
  LTXK_CHECK_LAUNCH("ltxk_resize_area");
  return LTXK_OK;
}

namespace ltxk {
}  // namespace ltxk

using namespace ltxk;

extern "C" int ltxk_groupnorm_act(const void* x, void* out, const void* gamma, const void* beta, const void* resid,
                                  int32_t B, int64_t V, int32_t C, int32_t G, float eps, int32_t apply_silu, void* stream) {
  LTXK_CHECK_ARG(x && out && gamma && beta && B > 0 && V > 0 && C > 0 && G > 0 && C % G == 0, "ltxk_groupnorm_act: bad arguments");
  hipLaunchKernelGGL(groupnorm_act_kernel, dim3(B * G), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)out,
                     (const bf16*)gamma, (const bf16*)beta, (const bf16*)resid, V, C, G, eps, apply_silu);
  LTXK_CHECK_LAUNCH("ltxk_groupnorm_act");
  return LTXK_OK;
}

extern "C" int ltxk_tile_blend_accum(const void* tile, int32_t Tt, int32_t Th, int32_t Tw, int32_t at, int32_t ah,
                                     int32_t aw, const float* mt, const float* mh, const float* mw, float* acc,
                                     float* wsum, int32_t B, int32_t C, int32_t F, int32_t H, int32_t W, int32_t t0,
                                     int32_t h0, int32_t w0, void* stream) {
  LTXK_CHECK_ARG(tile && mt && mh && mw && acc && wsum, "ltxk_tile_blend_accum: null pointer");
  LTXK_CHECK_ARG(at > 0 && ah > 0 && aw > 0 && at <= Tt && ah <= Th && aw <= Tw, "ltxk_tile_blend_accum: bad tile extent");
  LTXK_CHECK_ARG(t0 >= 0 && h0 >= 0 && w0 >= 0 && t0 + at <= F && h0 + ah <= H && w0 + aw <= W,
                 "ltxk_tile_blend_accum: tile [%d+%d,%d+%d,%d+%d] outside the output volume %dx%dx%d", t0, at, h0, ah, w0, aw, F, H, W);
  const int64_t total = (int64_t)B * at * ah * aw;
  hipLaunchKernelGGL(tile_blend_accum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)tile, Tt, Th, Tw, at, ah, aw, mt, mh, mw, acc, wsum, B, C, F, H, W, t0, h0, w0);
  LTXK_CHECK_LAUNCH("ltxk_tile_blend_accum");
  return LTXK_OK;
}

extern "C" int ltxk_tile_blend_finalize(const float* acc, const float* wsum, void* out, int32_t B, int32_t C,
                                        int64_t S, void* stream) {
  LTXK_CHECK_ARG(acc && wsum && out && B > 0 && C > 0 && S > 0, "ltxk_tile_blend_finalize: bad arguments");
  const int64_t total = (int64_t)B * C * S;
  hipLaunchKernelGGL(tile_blend_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     acc, wsum, (bf16*)out, B, C, S);
  LTXK_CHECK_LAUNCH("ltxk_tile_blend_finalize");
  return LTXK_OK;
}

extern "C" int ltxk_pixelnorm_act(const void* x, void* y, int64_t V, int32_t C, float eps, const void* scale,
                                  const void* shift, int64_t rows_per_batch, int32_t apply_silu, void* stream) {
  LTXK_CHECK_ARG(x && y && V > 0, "ltxk_pixelnorm_act: null/empty input");
  LTXK_CHECK_ARG((scale == nullptr) == (shift == nullptr), "ltxk_pixelnorm_act: scale and shift must both be set or both NULL");
  LTXK_CHECK_ARG(!scale || rows_per_batch > 0, "ltxk_pixelnorm_act: rows_per_batch must be > 0 with modulation");
  hipStream_t st = (hipStream_t)stream;
#define PN_LAUNCH(LPR, PASSES)                                                                               \
  {                                                                                                          \
    const int64_t rows_per_block = 4 * (64 / LPR);                                                           \
    hipLaunchKernelGGL((pixelnorm_act_kernel<LPR, PASSES>), dim3((unsigned)((V + rows_per_block - 1) / rows_per_block)), \
                       dim3(256), 0, st, (const bf16*)x, (bf16*)y, V, C, eps, (const bf16*)scale, (const bf16*)shift, \
                       rows_per_batch, apply_silu);                                                          \
  }
  switch (C) {
    case 64: PN_LAUNCH(8, 1) break;
    case 128: PN_LAUNCH(16, 1) break;
    case 256: PN_LAUNCH(32, 1) break;
    case 512: PN_LAUNCH(64, 1) break;
    case 1024: PN_LAUNCH(64, 2) break;
    case 2048: PN_LAUNCH(64, 4) break;
    default:
      ltxk_set_error("ltxk_pixelnorm_act: unsupported channel count %d (64..2048 powers of two)", C);
      return LTXK_EINVAL;
  }
#undef PN_LAUNCH
  LTXK_CHECK_LAUNCH("ltxk_pixelnorm_act");
  return LTXK_OK;
}

extern "C" int ltxk_d2s_add(const void* conv, const void* xin, void* out, int32_t B, int32_t D, int32_t H, int32_t W,
                            int32_t Co, int32_t Ci, void* stream) {
  LTXK_CHECK_ARG(conv && out && B > 0 && D > 0 && H > 0 && W > 0 && Co > 0, "ltxk_d2s_add: bad arguments");
  LTXK_CHECK_ARG(!xin || (Ci % 8 == 0 && Ci > 0), "ltxk_d2s_add: Ci must be a multiple of 8");
  const int64_t V = (int64_t)B * D * H * W;
  hipLaunchKernelGGL(d2s_add_kernel, dim3((unsigned)V), dim3(Co < 256 ? (Co + 63) / 64 * 64 : 256), 0, (hipStream_t)stream,
                     (const bf16*)conv, (const bf16*)xin, (bf16*)out, B, D, H, W, Co, Ci);
  LTXK_CHECK_LAUNCH("ltxk_d2s_add");
  return LTXK_OK;
}

extern "C" int ltxk_latent_denorm_cl(const void* latent, const void* noise, float noise_scale, const void* mean,
                                     const void* std, void* out, int32_t B, int32_t C, int64_t S, void* stream) {
  LTXK_CHECK_ARG(latent && mean && std && out && B > 0 && C > 0 && C % 8 == 0 && S > 0, "ltxk_latent_denorm_cl: bad arguments");
  hipLaunchKernelGGL(latent_denorm_cl_kernel, dim3((unsigned)((S + 63) / 64), C / 8, B), dim3(64), 0, (hipStream_t)stream,
                     (const bf16*)latent, (const bf16*)noise, noise_scale, (const bf16*)mean, (const bf16*)std, (bf16*)out, B, C, S);
  LTXK_CHECK_LAUNCH("ltxk_latent_denorm_cl");
  return LTXK_OK;
}

extern "C" int ltxk_latent_norm_cf(const void* x, int32_t ldx, const void* mean, const void* std, void* out,
                                   int32_t B, int32_t C, int64_t S, void* stream) {
  LTXK_CHECK_ARG(x && mean && std && out && B > 0 && C > 0 && C % 8 == 0 && S > 0 && ldx >= C && ldx % 8 == 0,
                 "ltxk_latent_norm_cf: bad arguments");
  hipLaunchKernelGGL(latent_norm_cf_kernel, dim3((unsigned)((S + 63) / 64), C / 8, B), dim3(64), 0, (hipStream_t)stream,
                     (const bf16*)x, ldx, (const bf16*)mean, (const bf16*)std, (bf16*)out, B, C, S);
  LTXK_CHECK_LAUNCH("ltxk_latent_norm_cf");
  return LTXK_OK;
}

extern "C" int ltxk_unpatchify_cf(const void* x, void* out, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C,
                                  int32_t P, void* stream) {
  LTXK_CHECK_ARG(x && out && B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && P > 0, "ltxk_unpatchify_cf: bad arguments");
  LTXK_CHECK_ARG(P != 4 || ((uintptr_t)out & 7) == 0, "ltxk_unpatchify_cf: out must be 8-byte aligned");
  const int64_t total = (int64_t)B * C * D * H * P * W;
  hipLaunchKernelGGL(unpatchify_cf_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)x, (bf16*)out, B, D, H, W, C, P);
  LTXK_CHECK_LAUNCH("ltxk_unpatchify_cf");
  return LTXK_OK;
}

extern "C" int ltxk_patchify_cl(const void* video, void* out, int32_t B, int32_t C, int32_t D, int32_t H, int32_t W,
                                int32_t P, int32_t Cpad, void* stream) {
  LTXK_CHECK_ARG(video && out && B > 0 && C > 0 && D > 0 && P > 0 && H % P == 0 && W % P == 0 && Cpad >= C * P * P,
                 "ltxk_patchify_cl: bad arguments");
  const int64_t total = (int64_t)B * D * (H / P) * (W / P) * Cpad;
  hipLaunchKernelGGL(patchify_cl_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)video, (bf16*)out, B, C, D, H, W, P, Cpad);
  LTXK_CHECK_LAUNCH("ltxk_patchify_cl");
  return LTXK_OK;
}

extern "C" int ltxk_to_uint8(const void* x, void* out, int32_t B, int32_t C, int32_t F, int32_t H, int32_t W, void* stream) {
  LTXK_CHECK_ARG(x && out && B > 0 && C > 0 && F > 0 && H > 0 && W > 0, "ltxk_to_uint8: bad arguments");
  const int64_t total = (int64_t)B * C * F * H * W;
  hipLaunchKernelGGL(to_uint8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)x, (uint8_t*)out, B, C, F, H, W);
  LTXK_CHECK_LAUNCH("ltxk_to_uint8");
  return LTXK_OK;
}

extern "C" int ltxk_s2d_skip(const void* conv, const void* xpad, void* out, int32_t B, int32_t Dp, int32_t Hp,
                             int32_t Wp, int32_t Cc, int32_t Cx, int32_t st, int32_t sh, int32_t sw, int32_t G,
                             void* stream) {
  LTXK_CHECK_ARG(conv && xpad && out && B > 0 && Dp > 0 && Hp > 0 && Wp > 0 && Cc > 0 && Cx > 0 && G > 0, "ltxk_s2d_skip: bad arguments");
  LTXK_CHECK_ARG(st >= 1 && sh >= 1 && sw >= 1 && Dp % st == 0 && Hp % sh == 0 && Wp % sw == 0, "ltxk_s2d_skip: dims not divisible by stride");
  LTXK_CHECK_ARG(Cc * G == Cx, "ltxk_s2d_skip: Cc*G must equal Cx");
  const int64_t V = (int64_t)B * (Dp / st) * (Hp / sh) * (Wp / sw);
  hipLaunchKernelGGL(s2d_skip_kernel, dim3((unsigned)V), dim3(256), 0, (hipStream_t)stream, (const bf16*)conv,
                     (const bf16*)xpad, (bf16*)out, B, Dp, Hp, Wp, Cc, Cx, st, sh, sw, G);
  LTXK_CHECK_LAUNCH("ltxk_s2d_skip");
  return LTXK_OK;
}

/*
 * ltxk.h — C ABI of libltxk.so: the MI355X (gfx950) kernels behind the LTX-2 denoise step
 * (DiT forward + step algebra) and the causal-3D-conv video VAE.
 *
 * The reference (CharafChnioune/mlx-video) has no FFI of its own; its seams are Python
 * callables that hand every numeric op to MLX built-ins.  Each entry point below replaces
 * one of those MLX call sites (file:line relative to the reference root).  Host code that
 * sits where LTXModel.__call__ (mlx_video/models/ltx/ltx.py:459), LTX2VideoDecoder.__call__
 * (video_vae/decoder.py:361) and VideoEncoder.__call__ (video_vae/video_vae.py:321) sit
 * calls these through ctypes.
 *
 * Conventions
 *   - the caller owns every buffer; the library never allocates, frees or retains memory;
 *   - all pointers are device pointers unless noted; activations/weights are bf16,
 *     side tables fp32, indices int32;
 *   - token tensors are row-major (tokens, dim); volumes are channels-last (B,D,H,W,C);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream), re-entrant across streams, and keeps no global mutable state;
 *   - returns 0 on success, a negative LTXK_E* code on error; ltxk_last_error() returns a
 *     thread-local message.  Nothing throws across the ABI.
 */
#ifndef LTXK_H
#define LTXK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LTXK_VERSION 100

#define LTXK_OK 0
#define LTXK_EINVAL (-1)   /* bad argument (shape / alignment / null pointer) */
#define LTXK_ELAUNCH (-2)  /* HIP launch error */

int ltxk_version(void);
const char* ltxk_last_error(void);

/* ---------------------------------------------------------------------------------------
 * GEMM with fused epilogue: replaces nn.Linear (mlx x@W.T+b) at attention.py:123-126,142,
 * feed_forward.py:35-40, adaln.py:46,134-138, text_projection.py, ltx.py:130,455.
 * out = epi(A[M,K] @ W[N,K]^T + bias[N]); fp32 accumulate on MFMA; bf16 rounding is applied
 * at every point where the reference materialises a bf16 array.
 * ------------------------------------------------------------------------------------- */
enum {
  LTXK_EPI_BIAS = 0,          /* y = bf16(acc + b)                                        */
  LTXK_EPI_BIAS_GELU = 1,     /* bf16(gelu_tanh(y))          feed_forward.py:12           */
  LTXK_EPI_BIAS_SILU = 2,     /* bf16(y*sigmoid(y))          adaln.py:136                 */
  LTXK_EPI_BIAS_GATE_RES = 3, /* bf16(res + bf16(y*gate))    transformer.py:254,347       */
  LTXK_EPI_BIAS_RES = 4       /* bf16(res + y)               transformer.py:257           */
};

typedef struct ltxk_gemm_args {
  const void* A;        /* (M,K) bf16, row stride lda elements                            */
  const void* W;        /* (N,K) bf16, row-major (nn.Linear weight layout)                */
  const void* bias;     /* (N) bf16 or NULL                                               */
  void* out;            /* (M,N) bf16 row stride ldo; or see out_tokens_per_batch         */
  const void* resid;    /* (M,N) bf16 row stride ldr, for *_RES epilogues                 */
  const void* gate;     /* gate value for row m, col n: gate[gate_row[m]*gate_stride + n] */
  const int32_t* gate_row; /* (M) int32 or NULL (=> row 0)                                */
  int32_t M, N, K;
  int32_t lda, ldo, ldr, gate_stride;
  int32_t epilogue;
  /* >0: write the output transposed per batch: row m = b*T + t goes to
   * out[(b*N + n)*ldo + t]   (used for V^T so that attention reads V k-contiguous)       */
  int32_t out_tokens_per_batch;
} ltxk_gemm_args;

int ltxk_gemm_bf16(const ltxk_gemm_args* args, void* stream);

/* ---------------------------------------------------------------------------------------
 * Fused attention: replaces mx.fast.scaled_dot_product_attention (attention.py:47) incl. the
 * (B,T,H*dh)<->(B,H,T,dh) reshapes (attention.py:24-33,50-51).  dh must be 128.  No mask
 * (context_mask=None on this path, generate.py:800).
 *   q  : (B,Tq,H*128) bf16 row stride ldq        k : (B,Tk,H*128) bf16 row stride ldk
 *   vt : (B,H*128,ldvt) bf16 = V transposed, ldvt >= Tk, multiple of 8, pad columns finite
 *   out: (B,Tq,H*128) bf16 row stride ldo
 * ------------------------------------------------------------------------------------- */
int ltxk_flash_attn_bf16(const void* q, int32_t ldq, const void* k, int32_t ldk,
                         const void* vt, int32_t ldvt, void* out, int32_t ldo,
                         int32_t B, int32_t H, int32_t Tq, int32_t Tk, float scale,
                         void* stream);

/* ---------------------------------------------------------------------------------------
 * rms_norm (weight 1) + AdaLN modulation: utils.py:398-400 + transformer.py:253,258,346.
 *   y = bf16(bf16(bf16(rms(x)) * bf16(1+scale)) + shift); scale/shift NULL => plain rms_norm.
 *   scale/shift value for row m, col d: p[mod_row[m]*mod_stride + d].
 * ------------------------------------------------------------------------------------- */
int ltxk_rmsnorm_modulate(const void* x, void* y, int32_t M, int32_t D, float eps,
                          const void* scale, const void* shift, int32_t mod_stride,
                          const int32_t* mod_row, void* stream);

/* LayerNorm(affine=False) + modulation of the output head: ltx.py:432-457.               */
int ltxk_layernorm_modulate(const void* x, void* y, int32_t M, int32_t D, float eps,
                            const void* scale, const void* shift, int32_t mod_stride,
                            const int32_t* mod_row, void* stream);

/* ---------------------------------------------------------------------------------------
 * q/k RMSNorm over the whole inner dim (all heads jointly, learned weight) + SPLIT RoPE:
 * attention.py:96-97,129-136 + rope.py:109-172.  In place on `nseg` column segments of
 * width D of each row of buf (self-attn: q|k packed => nseg=2).
 *   weight: (nseg,D) bf16.  cos/sin: (H,T,dh/2) fp32 or NULL (no rope: cross-attention).
 *   row m belongs to token t = m % T.
 * ------------------------------------------------------------------------------------- */
int ltxk_qknorm_rope(void* buf, int32_t ld, int32_t M, int32_t nseg, int32_t D,
                     const void* weight, const float* cos, const float* sin,
                     int32_t T, int32_t H, float eps, void* stream);

/* Sinusoidal timestep projection: utils.py:486-526 (flip_sin_to_cos, shift 0), applied to
 * bf16(t*mult) (ltx.py:68: timestep*timestep_scale_multiplier stays in the model dtype).
 * t: (U) bf16 timesteps; out: (U,dim) bf16.                                              */
int ltxk_timestep_embed(const void* t, void* out, int32_t U, int32_t dim, float mult, void* stream);

/* SPLIT-layout RoPE table, fp32: rope.py:419-529 (_precompute_freqs_cis_double_precision with
 * use_middle_indices_grid).  positions: (3,T,2) fp32 [start,end); freq: (n_freq) fp32 =
 * theta^linspace(0,1,n_freq)*pi/2 (host table, rope.py:449-450); max_pos: 3 HOST floats.
 * cos/sin out: (H,T,dim/2/H) fp32; the first dim/2-3*n_freq entries are cos=1,sin=0.     */
int ltxk_rope_table(const float* positions, const float* freq, float* cos, float* sin,
                    int32_t T, int32_t H, int32_t dim, int32_t n_freq, const float* max_pos,
                    void* stream);

/* out[l,u,k,:] = bf16(table[l,k,:] + ada[u,k,:]): transformer.py:135-177, ltx.py:440-447. */
int ltxk_ada_combine(const void* table, const void* ada, void* out, int32_t L, int32_t U,
                     int32_t K, int32_t D, void* stream);

/* Elementwise bf16 SiLU (adaln.py:45).                                                   */
int ltxk_silu(const void* x, void* y, int64_t n, void* stream);

/* (B,C,S) channels-first latent -> (B,S,C) tokens, optionally replicated `rep` times on the
 * batch axis (cfg_batch): generate.py:1236,1239-1241.  Bit-exact index map.              */
int ltxk_latent_to_tokens(const void* latent, void* tokens, int32_t B, int32_t C, int32_t S,
                          int32_t rep, void* stream);

/* One denoise-step tail: CFG combine + token->latent transpose + x0 + mask blend + Euler:
 * generate.py:1255,1283-1301 (compiled: 1160-1174), utils.py:404-440, latent.py:180-196.
 *   v_pos/v_neg: (B,S,C) bf16 velocities (v_neg NULL => no CFG)
 *   latent/out : (B,C,S) bf16;  clean: (B,C,S) bf16 or NULL;  mask: (B,S) float or NULL
 *   x0  = bf16(x - sigma*v);  x0 = x0*m + clean*(1-m);  out = bf16(x0 + sigma_next*(x-x0)/sigma)
 *   sigma_next <= 0 => out = x0.                                                         */
int ltxk_cfg_euler_step(const void* v_pos, const void* v_neg, const void* latent, void* out,
                        const void* clean, const float* mask, int32_t B, int32_t C, int32_t S,
                        float cfg_scale, float sigma, float sigma_next, void* stream);

/* Euler update alone (eager path, generate.py:1293-1301, with un-rounded float sigmas):
 * out = bf16(x0 + sigma_next*(x - x0)/sigma) in fp32; n elements, any layout.             */
int ltxk_euler_step(const void* latent, const void* denoised, void* out, int64_t n,
                    float sigma, float sigma_next, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LTXK_H */

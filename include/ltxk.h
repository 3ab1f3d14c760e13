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

#define LTXK_VERSION 400

#define LTXK_OK 0
#define LTXK_EINVAL (-1)   /* bad argument (shape / alignment / null pointer) */
#define LTXK_ELAUNCH (-2)  /* HIP launch error */

int ltxk_version(void);
const char* ltxk_last_error(void);
/* sizeof the argument structs in THIS build: 0 = ltxk_gemm_args, 1 = ltxk_conv3d_args, 2 = ltxk_attn_args; anything else returns -1;
 * lets a foreign-language binding verify its struct layout before the first call.        */
int ltxk_abi_sizeof(int which);

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
  LTXK_EPI_BIAS_RES = 4,      /* bf16(res + y)               transformer.py:257           */
  LTXK_EPI_SCALE_RES = 5      /* bf16(res + bf16(alpha*acc)) LoRA merge W += s*(B@A), lora.py:94-127 */
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
  float alpha;          /* LTXK_EPI_SCALE_RES only                                        */
  /* Split output (n_split > 0, a multiple of 256; EPI_BIAS): columns [0,n_split) go row-major to `out` (ldo), columns
   * [n_split,N) transposed per batch to out2[(b*(N-n_split) + n-n_split)*ldo2 + t] with T = out_tokens_per_batch.
   * One launch then produces q|k row-major and V^T (attention.py:123-125 as one GEMM over the packed to_q|to_k|to_v
   * panel), or the text-side k and V^T.                                                   */
  void* out2;
  int32_t n_split, ldo2;
  /* Optional (row-major columns only): sumsq[m*sumsq_ld + n/64] = sum over the 64-column block of the squares of the
   * bf16 values stored for row m (fp32, fixed summation order).  Lets the consumer of this output (rms_norm,
   * utils.py:398-400; q/k RMSNorm, attention.py:129-131) take its row statistic without re-reading the row.   */
  float* sumsq;
  int32_t sumsq_ld;
  /* Optional caller-owned scratch (16-byte aligned, fp32), reused by every call on the stream.  With it, a launch whose
   * row count gives the chip too few workgroups to stream the weight panel (small M: low-resolution / distilled stage-1
   * steps, ltx.py:459-506 at config 1's geometry) runs as split-K: every K slice parks its fp32 accumulator tile here
   * (slices x M x N x 4 bytes; fewer slices if the scratch is smaller) and a second launch sums the slices in slice order
   * and applies the epilogue - deterministic, one rounding of the fp32 sum.  NULL: never split.                        */
  void* workspace;
  int64_t workspace_bytes;
} ltxk_gemm_args;

int ltxk_gemm_bf16(const ltxk_gemm_args* args, void* stream);

/* ---------------------------------------------------------------------------------------
 * Fused attention: replaces mx.fast.scaled_dot_product_attention (attention.py:47) incl. the
 * (B,T,H*dh)<->(B,H,T,dh) reshapes (attention.py:24-33,50-51).  dh must be 128.  No mask
 * (context_mask=None on this path, generate.py:800).
 *   q  : (B,Tq,H*128) bf16 row stride ldq        k : (B,Tk,H*128) bf16 row stride ldk
 *   vt : (B,H*128,ldvt) bf16 = V transposed, ldvt >= Tk rounded up to 64, multiple of 8, pad columns finite
 *   out: (B,Tq,H*128) bf16 row stride ldo
 * q, k, vt, out 16-byte aligned; ldq, ldk, ldo multiples of 8; scale > 0.
 * Rounding points (oracle/dit.py::sdpa, "flash" policy): S = q.k^T in fp32; P = exp2(fma(S, scale*log2 e, -M)) with M an
 * INTEGER row offset, so bf16(P) does not depend on the tiling; l = sum P in fp32; O = bf16((bf16(P) @ V) / l).
 * ------------------------------------------------------------------------------------- */
enum {
  /* Launches of fewer than 1.25 rounds of 128-row query tiles: the tiles of the short last round of workgroups are normally
   * split over two workgroups that halve the keys and merge (O, M, l): faster, but those rows then sum their keys in another
   * order than in a launch whose grid has no short round.  With this flag the result for a (batch, head) does not depend on
   * how many share a launch.  Larger launches run a mixed grid of 192- and 128-row tiles that never splits keys: they have
   * the flag's bits with or without it.                                                                                  */
  LTXK_ATTN_NO_TAIL_SPLIT = 1
};
typedef struct ltxk_attn_args {
  const void* q; const void* k; const void* vt; void* out;
  int32_t ldq, ldk, ldvt, ldo;
  int32_t B, H, Tq, Tk;
  float scale;
  /* Optional fused query preparation (attention.py:129-136).  With q_sumsq set, `q` holds the RAW to_q projection
   * and the kernel applies, to its Q fragments in registers, q_norm (RMSNorm over all H*128 channels jointly, learned
   * weight) and - if cos/sin are given - the SPLIT rotation of rope.py:109-172, with the rounding points of
   * ltxk_qknorm_rope.  q_sumsq: (B*Tq, q_sumsq_ld) fp32, the first q_sumsq_n = H*128/64 entries of a row are the sums
   * of squares of its 64-column blocks (the `sumsq` output of ltxk_gemm_bf16).  cos/sin: (H,Tq,64) fp32.           */
  const float* q_sumsq;
  int32_t q_sumsq_ld, q_sumsq_n;
  const void* q_norm_weight;   /* (H*128) bf16 */
  const float* cos;
  const float* sin;
  float eps;
  int32_t flags;               /* LTXK_ATTN_* */
} ltxk_attn_args;

int ltxk_flash_attn(const ltxk_attn_args* args, void* stream);

/* The same without query preparation (positional form kept for existing bindings).       */
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

/* The same when the rows' sums of squares are already known: sumsq (M, sumsq_ld) fp32, row statistic = the sum of
 * the first sumsq_n entries (ltxk_gemm_bf16's `sumsq` output for the GEMM that produced x).  The row is then not
 * re-read for its statistic and is split over several waves.  flags & LTXK_NORM_SCALE_IS_ONE_PLUS: `scale` already
 * holds bf16(1+scale) (ltxk_ada_combine's one_plus_mask).                                 */
enum { LTXK_NORM_SCALE_IS_ONE_PLUS = 1 };
int ltxk_rmsnorm_modulate_ss(const void* x, void* y, int32_t M, int32_t D, float eps, const float* sumsq,
                             int32_t sumsq_ld, int32_t sumsq_n, const void* scale, const void* shift,
                             int32_t mod_stride, const int32_t* mod_row, int32_t flags, void* stream);

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

/* The same with the per-row sums of squares precomputed: sumsq[m*sumsq_ld + seg*(D/64) + i], i < D/64
 * (ltxk_gemm_bf16's `sumsq` output of the projection that wrote buf).                     */
int ltxk_qknorm_rope_ss(void* buf, int32_t ld, int32_t M, int32_t nseg, int32_t D,
                        const void* weight, const float* cos, const float* sin,
                        int32_t T, int32_t H, float eps, const float* sumsq, int32_t sumsq_ld, void* stream);

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

/* out[l,u,k,:] = bf16(table[l,k,:] + ada[u,k,:]): transformer.py:135-177, ltx.py:440-447.  For the k whose bit is set
 * in one_plus_mask the stored value is bf16(1 + that) - the `(1 + scale)` factor of transformer.py:253,346, which is
 * the same for every token that shares the row.                                           */
int ltxk_ada_combine(const void* table, const void* ada, void* out, int32_t L, int32_t U,
                     int32_t K, int32_t D, uint32_t one_plus_mask, void* stream);

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
 *   sigma_next <= 0 => out = x0.
 *   flags & LTXK_STEP_BF16_EULER: the Euler update runs op by op in bf16, the reference's
 *   fp32_euler=False / LTX_FP32_EULER=0 compiled step (generate.py:741-748):
 *   out = bf16(x0 + bf16(bf16(sigma_next * bf16(x - x0)) / sigma)).                        */
enum { LTXK_STEP_BF16_EULER = 1 };
int ltxk_cfg_euler_step(const void* v_pos, const void* v_neg, const void* latent, void* out,
                        const void* clean, const float* mask, int32_t B, int32_t C, int32_t S,
                        float cfg_scale, float sigma, float sigma_next, int32_t flags, void* stream);

/* Same step tail with {sigma, sigma_next} read from DEVICE memory (2 floats): lets one captured
 * hipGraph of the whole denoise step be replayed for every step of the schedule.           */
int ltxk_cfg_euler_step_dev(const void* v_pos, const void* v_neg, const void* latent, void* out,
                            const void* clean, const float* mask, int32_t B, int32_t C, int32_t S,
                            float cfg_scale, const float* sigmas_dev, int32_t flags, void* stream);

/* Per-step scalars of a replayed step graph: with s = min(*step, n_steps-1), copies ts_all[s,:] (U bf16
 * timestep values = bf16(sigma_s)*mask, generate.py:1084,1237) to ts and sig_all[s,:] ({sigma, sigma_next}
 * fp32) to sig, then stores *step = s+1.  As the first node of a captured denoise step it lets the whole
 * schedule run as graph replays with no host->device traffic between steps.               */
int ltxk_step_scalars(const void* ts_all, const float* sig_all, int32_t* step, void* ts, float* sig,
                      int32_t U, int32_t n_steps, void* stream);

/* Euler update alone (eager path, generate.py:1293-1301, with un-rounded float sigmas):
 * out = bf16(x0 + sigma_next*(x - x0)/sigma) in fp32; n elements, any layout.             */
int ltxk_euler_step(const void* latent, const void* denoised, void* out, int64_t n,
                    float sigma, float sigma_next, void* stream);

/* ---------------------------------------------------------------------------------------
 * Video VAE (volumes are channels-last (B,D,H,W,C) bf16; a row = one voxel)
 * ------------------------------------------------------------------------------------- */
enum { LTXK_PAD_ZEROS = 0, LTXK_PAD_REFLECT = 1 };

typedef struct ltxk_conv3d_args {
  const void* x;         /* (B,D,H,W,Cin) bf16, Cin % 64 == 0                              */
  const void* w;         /* (Cout,3,3,3,Cin) bf16 (MLX layout, decoder.py:708-710)         */
  const void* bias;      /* (Cout) bf16                                                    */
  void* out;             /* (B,D,H,W,Cout) bf16                                            */
  const void* resid;     /* optional (B,D,H,W,Cout): out = bf16(conv + resid) (decoder.py:180) */
  const void* zero_page; /* >= 128 zero bytes in device memory (zero-padding source)       */
  int32_t B, D, H, W, Cin, Cout;
  int32_t causal;        /* temporal halo: 1 = 2x first frame; 0 = first + last (convolution.py:126-137);
                          * 2 = zeros on both sides (plain Conv3d padding=1 of the latent upsampler, upsampler.py:6-62) */
  int32_t pad_mode;      /* spatial halo: LTXK_PAD_ZEROS | LTXK_PAD_REFLECT (convolution.py:143-157) */
  /* optional caller-owned fp32 scratch for split-K on small volumes (the decoder's 1024/512-channel stages
   * have too few voxels to fill 256 CUs): S*M*Cout*4 bytes are used if they fit; NULL disables split-K.
   * Partial slabs are written with plain stores and summed in slice order, so results are deterministic. */
  void* workspace;
  int64_t workspace_bytes;
  /* temporal taps: 0 or 3 = the 3x3x3 kernel; 1 = a per-frame 3x3 kernel, w = (Cout,3,3,Cin) (the latent
   * upsampler's nn.Conv2d applied frame by frame, upsampler.py:64-99): only the centre temporal tap exists,
   * K = 9*Cin instead of 27*Cin.                                                                        */
  int32_t taps_d;
  /* Optional fused PixelNorm (+ AdaLN modulation) + SiLU of the OUTPUT row (decoder.py:136-180: the pixel_norm / scale /
   * shift / SiLU that follows every conv of a res block; utils.py:477-483):
   *   act_out[v,:] = silu?( modulate?( pixel_norm(y[v,:], act_eps) ) ),  y = this conv's bf16 output row (bias, residual),
   * with the rounding points of ltxk_pixelnorm_act.  Needs Cout == 128 or 256 (the tile then holds whole voxel rows; the
   * row statistic is reduced inside the tile).  act_out == NULL: off.  `out` may be NULL when act_out is set (a conv whose
   * raw output nothing else reads).  act_scale / act_shift: (B,Cout) bf16 or both NULL; act_rows_per_batch = D*H*W. */
  void* act_out;
  const void* act_scale;
  const void* act_shift;
  float act_eps;
  int32_t act_silu;
} ltxk_conv3d_args;

/* nn.Conv3d 3x3x3 stride 1 inside CausalConv3d (convolution.py:78-222) as implicit GEMM.   */
int ltxk_conv3d_k3_bf16(const ltxk_conv3d_args* args, void* stream);

/* pixel_norm over channels [+ (1+scale)+shift per (batch,channel)] [+ SiLU]: decoder.py:136-180,
 * 415-437; utils.py:477-483.  x,y: (V,C) rows = voxels, C in {64..2048, power of two}.
 * scale/shift: (B,C) bf16 or NULL; rows_per_batch = D*H*W.                                 */
int ltxk_pixelnorm_act(const void* x, void* y, int64_t V, int32_t C, float eps, const void* scale,
                       const void* shift, int64_t rows_per_batch, int32_t apply_silu, void* stream);

/* DepthToSpaceUpsample tail (sampling.py:143-197): conv (B,D,H,W,8*Co) -> out (B,2D-1,2H,2W,Co),
 * channel (c,st,sh,sw), first output frame dropped, plus the residual d2s(xin) tiled over
 * channels (xin (B,D,H,W,Ci) or NULL).                                                     */
int ltxk_d2s_add(const void* conv, const void* xin, void* out, int32_t B, int32_t D, int32_t H, int32_t W,
                 int32_t Co, int32_t Ci, void* stream);

/* SpaceToDepthDownsample tail (sampling.py:53-103): conv (B,Dp,Hp,Wp,Cc) and the (front-frame
 * duplicated) input xpad (B,Dp,Hp,Wp,Cx) -> out (B,Dp/st,Hp/sh,Wp/sw,Cc*st*sh*sw) =
 * s2d(conv) + group-mean_G(s2d(xpad)), G = Cx/Cc.                                          */
int ltxk_s2d_skip(const void* conv, const void* xpad, void* out, int32_t B, int32_t Dp, int32_t Hp,
                  int32_t Wp, int32_t Cc, int32_t Cx, int32_t st, int32_t sh, int32_t sw, int32_t G,
                  void* stream);

/* latents (B,C,S) channels-first -> (B,S,C) channels-last with fp32 x*std+mean (decoder.py:349-355);
 * noise != NULL: the timestep-conditioned decoder's blend noise*s + (1-s)*x first (decoder.py:381-385). */
int ltxk_latent_denorm_cl(const void* latent, const void* noise, float noise_scale, const void* mean,
                          const void* std, void* out, int32_t B, int32_t C, int64_t S, void* stream);

/* (B,S,ldx>=C) channels-last -> (B,C,S) channels-first with fp32 (x-mean)/std (ops.py:94-109). */
int ltxk_latent_norm_cf(const void* x, int32_t ldx, const void* mean, const void* std, void* out,
                        int32_t B, int32_t C, int64_t S, void* stream);

/* conv_out (B,D,H,W,C*P*P) channels-last -> unpatchify (ops.py:47-80, channel order (c,p_w,p_h))
 * -> video (B,C,D,H*P,W*P) channels-first.  Bit-exact index map.                           */
int ltxk_unpatchify_cf(const void* x, void* out, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C,
                       int32_t P, void* stream);

/* video (B,C,D,H,W) -> patchify (ops.py:9-44) -> (B,D,H/P,W/P,Cpad) channels-last, channels
 * beyond C*P*P zero (pads 48 -> 64 for the first encoder convolution).                     */
int ltxk_patchify_cl(const void* video, void* out, int32_t B, int32_t C, int32_t D, int32_t H, int32_t W,
                     int32_t P, int32_t Cpad, void* stream);

/* video (B,C,F,H,W) bf16 in [-1,1] -> uint8 frames (B,F,H,W,C): generate.py:3894-3898.     */
int ltxk_to_uint8(const void* x, void* out, int32_t B, int32_t C, int32_t F, int32_t H, int32_t W,
                  void* stream);

/* Area resize of conditioning frames, DOWNscaling only: cv2.resize(frame, (OW,OH), interpolation=cv2.INTER_AREA) on float
 * frames (prepare_video_for_encoding, utils.py:699-705; OpenCV resizeArea_: destination pixel d averages the source
 * interval [d*scale,(d+1)*scale) with fractional end weights, horizontal pass then vertical pass in fp32).
 * x: `planes` contiguous (H,W) planes, fp32 (x_is_f32 != 0) or bf16; out: planes x (OH,OW) bf16.  Linear, so it may be
 * applied to frames already mapped to [-1,1].                                               */
int ltxk_resize_area(const void* x, int32_t x_is_f32, void* out, int64_t planes, int32_t H, int32_t W,
                     int32_t OH, int32_t OW, void* stream);

/* GroupNorm over (D*H*W, C/G) per (batch, group) in fp32 + affine [+ residual] [+ SiLU]:
 * upsampler.py:65-98,160-174.  x,out (B,V,C) bf16 channels-last, gamma/beta (C) bf16.
 *   y = bf16((x-mean)/sqrt(var+eps)*gamma+beta); if resid: y = bf16(y+resid); if silu: bf16(silu(y)) */
int ltxk_groupnorm_act(const void* x, void* out, const void* gamma, const void* beta, const void* resid,
                       int32_t B, int64_t V, int32_t C, int32_t G, float eps, int32_t apply_silu,
                       void* stream);

/* Tiled-decode blending (tiling.py:399-447): acc[b,c,t0+t,h0+y,w0+x] += tile[b,c,t,y,x]*m,
 * wsum[b,t0+t,h0+y,w0+x] += m with m = mt[t]*mh[y]*mw[x]; tile (B,C,Tt,Th,Tw) bf16 of which the
 * leading (at,ah,aw) box is used; acc (B,C,F,H,W) / wsum (B,F,H,W) fp32.                    */
int ltxk_tile_blend_accum(const void* tile, int32_t Tt, int32_t Th, int32_t Tw, int32_t at, int32_t ah,
                          int32_t aw, const float* mt, const float* mh, const float* mw, float* acc,
                          float* wsum, int32_t B, int32_t C, int32_t F, int32_t H, int32_t W, int32_t t0,
                          int32_t h0, int32_t w0, void* stream);

/* out = bf16(acc / max(wsum, 1e-8)) (tiling.py:492-509); S = F*H*W.                         */
int ltxk_tile_blend_finalize(const float* acc, const float* wsum, void* out, int32_t B, int32_t C,
                             int64_t S, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LTXK_H */

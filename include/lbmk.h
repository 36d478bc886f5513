/*
 * lbmk — C-ABI of the MI355X (gfx950) kernels behind the LBM bridge step.
 *
 * The reference (nauyihsnehs/LBM) has no native layer: its hot path runs
 * through diffusers modules and torch ops.  Each entry point below names the
 * reference call site (file:line, relative to the reference tree) whose
 * arithmetic it replaces.  Every function
 *   - takes raw device pointers, sizes and a hipStream_t (passed as void*),
 *   - never allocates, frees or synchronises (the one exception exists in diagnostic builds only, see LBMK_DIAG),
 *   - returns LBMK_OK or an LBMK_ERR_* code (nothing is launched on error).
 * Activations are bf16 (raw uint16 bit patterns), channels-last (NHWC):
 * a [B][H][W][C] tensor is also the [B*H*W][C] token matrix.
 */
#ifndef LBMK_H
#define LBMK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LBMK_OK 0
#define LBMK_ERR_ARG 1      /* unsupported shape / misaligned pointer / bad enum */
#define LBMK_ERR_LAUNCH 2   /* hipLaunch reported an error */

#define LBMK_ACT_NONE 0
#define LBMK_ACT_SILU 1
#define LBMK_ACT_GEGLU 2    /* packed weights interleave 16 value / 16 gate columns */
#define LBMK_ACT_RELU 3

int lbmk_version(void);
const char* lbmk_last_error(void);

/* ------------------------------------------------------------------------
 * Implicit GEMM on bf16 MFMA:  out[m][n] = act(alpha * sum_k A(m,k) * w[n][k]
 *                                           + bias[n] + rowvec[b(m)][n] + residual[m][n])
 * A(m,k) is gathered from an NHWC tensor: m -> (b, oy, ox), k -> (kh, kw, c); KH x KW is 1x1, 2x2 or 3x3.
 * Replaces torch conv2d / linear under diffusers' UNet2DConditionModel and
 * AutoencoderKL as called from src/lbm/models/unets/unet.py:127-140 and
 * src/lbm/models/vae/autoencoderKL.py:64,116,134.  A linear layer is the
 * KH=KW=IH=IW=OH=OW=1 case (M rows of K channels).
 * ---------------------------------------------------------------------- */
typedef struct lbmk_igemm_desc {
  const void* a;        /* bf16 input, pixel p at a + (p*a_cs + a_coff) elements      */
  const void* w;        /* bf16 weights [N] rows of stride Kp, k = (kh*KW + kw)*Cin + c */
  void* out;            /* [M][ldo] bf16, or f32 when out_f32                          */
  const float* bias;    /* [N] (or [M] when bias_per_m), may be NULL                   */
  const void* rowvec;   /* bf16 [B][rowvec_ld], added to every row of image b, or NULL */
  const void* residual; /* bf16 [M][ldr], or NULL                                      */
  int32_t M, N, K, Kp;  /* K = KH*KW*Cin; Kp = row stride of w (multiple of 8, >= K)    */
  int32_t Cin, a_cs, a_coff;
  int32_t IH, IW, OH, OW;
  int32_t KH, KW, stride, pad_t, pad_l;
  int32_t up2;          /* 1: input is read through a nearest-neighbour x2 upsample    */
  int32_t ldo, ldr, rowvec_ld;
  int32_t act;          /* LBMK_ACT_*                                                  */
  int32_t out_f32;
  int32_t bias_per_m;
  float alpha;
  int32_t batch;        /* independent problems (grid.z); strides below in elements    */
  int64_t sa, sw, so, sr;
  int32_t tile;         /* -1 auto (a cost model over the production tiles), else index into the tile table (see
                           lbmk_igemm_tiles).  Every tile sums a layer's products in the same order, so the choice never
                           changes a bit of the result.  Tile 15 (384x128, persistent, one staged input slab per filter
                           row shared by its three taps) takes 3x3 stride-1 pad-1 same-size convolutions with
                           Cin % 64 == 0 and batch 1 only.  Tiles 17 / 18 / 19 (256x256 / 256x160 / 256x128, 8 waves in
                           two groups that alternate between reading fragments and issuing MFMAs) take 1x1 and 2x2 / 3x3
                           stride-1 same-size layers with Cin % 64 == 0.  Environment, read once per process, for A/B
                           runs: LBMK_NO_ROWA=1 (never tile 15), LBMK_NO_PERSIST=1 (tile 15 with one workgroup per
                           tile), LBMK_NO_PP=1 (never tiles 17 / 18) */
  int32_t k_order;      /* 0: k = tap*Cin + c;  1 (Cin % 64 == 0): k = (c/64)*KH*KW*64 + tap*64 + c%64,
                           i.e. the 9 taps of one 64-channel chunk are consecutive K tiles (L1 reuse) */
  float* gn_partial;    /* NULL, or GroupNorm partial sums of the (bf16-rounded) output, written by the
                           epilogue: [image][group][strip of 64 output rows][sum, sum of squares] f32,
                           lbmk_igemm_gn_partial_floats() floats; consumed by lbmk_groupnorm_from_partials,
                           which then skips its own statistics pass over the tensor                   */
  int32_t gn_groups;    /* groups G of the consuming GroupNorm: N % G == 0, N / G in {4, 8, 16},
                           OH*OW % 64 == 0, Cin % 64 == 0, bf16 output, no GEGLU                        */
  int32_t batch_inner;  /* 0 / 1: one batch level.  > 1: problem z = outer * batch_inner + inner, the inner
                           index advancing a / w / out by sa2 / sw2 / so2 (attention heads inside images:
                           outer stride = one image, inner stride = head_dim columns); no residual         */
  int64_t sa2, sw2, so2;
  int32_t in_f32;       /* fp32 mode: a, w, rowvec, residual and out are f32 (out_f32 must be 1) and the products run on
                           v_mfma_f32_32x32x2_f32 (exact f32).  Any M / N; Cin, Kp, a_cs, a_coff multiples of 4; any
                           KH x KW; k_order 0; LBMK_ACT_NONE / SILU / RELU; no gn_partial; tile is ignored            */
  int32_t out_up2;      /* 0: output row m is pixel m.  1 + 2*py + px: pixel (b, y, x) of the OH x OW grid is written to
                           pixel (b, 2y + py, 2x + px) of a [B][2*OH][2*OW][ldo] tensor.  A nearest-x2 upsample followed
                           by a 3x3 / pad-1 convolution (diffusers Upsample2D, called from the decoders behind
                           src/lbm/models/vae/autoencoderKL.py:116 and src/lbm/models/unets/unet.py:127-140) is four such
                           2x2 convolutions of the low-resolution input, one per output parity, with the taps that fall
                           on the same source pixel summed (pad_t = 1 - py, pad_l = 1 - px): 4/9 of the products.
                           bf16 only; no residual                                                                   */
  int32_t gn_nslab;     /* 0: gn_partial holds OH*OW/64 strips per (image, group).  > 0: it holds gn_nslab strips and */
  int32_t gn_strip_off; /* this launch writes strips gn_strip_off ... gn_strip_off + OH*OW/64 - 1 (the four parity
                           launches of one upsampling convolution share one statistics buffer)                      */
  int32_t smx_mode;     /* 0: off.  Two-pass row softmax over a scores GEMM (the single-head, head_dim-512 attention of the
                           VAE mid block, src/lbm/models/vae/autoencoderKL.py:64,116: torch SDPA there), without
                           ever storing the f32 scores: pass 1 writes only row maxima, pass 2 multiplies again and stores
                           the exponentials as bf16.
                           1: no output tile; smx_part[z][m][strip] = max over the strip's columns of alpha * acc.
                           2: out[m][n] = bf16(exp2(alpha * acc - smx_max[z][m])); smx_part[z][m][strip] = sum of the
                              stored values.  lbmk_igemm_smx_parts(N) strips per row (64 columns each; a strip wholly
                              past N holds -inf in pass 1 and 0 in pass 2): reduce over all of them.  1x1, K % 64 == 0,
                              no bias / residual / activation                                                     */
  const float* smx_max; /* pass 2: f32 [batch][M] row maxima (same alpha as pass 1)                               */
  float* smx_part;      /* f32 [batch][M][lbmk_igemm_smx_parts(N)]                                                */
  int32_t f32_chunks;   /* fp32 mode only.  0 / 1: one workgroup walks the whole K range of its tile.  > 1 (must equal    */
  float* f32_part;      /* lbmk_igemm_f32_chunks(K, N, OH * OW); batch 1): the chunks of K (a function of the layer, not of the row count) run as separate workgroups that
                           write raw products to f32_part[chunk][M][N] (f32_chunks * M * N floats), and a finishing pass
                           adds them in chunk order and applies alpha / bias / rowvec / residual / activation.  The
                           single-workgroup kernel adds its own chunk sums in the same order, so a layer
                           returns the same bits either way (the choice may depend on the batch size, the result
                           does not).  For the few-row, deep-K layers of a small batch (256 ... 1024 tokens against
                           K up to 23,040: 20 ... 80 tiles on 256 CUs)                                            */
} lbmk_igemm_desc;

int lbmk_igemm(const lbmk_igemm_desc* d, void* stream);
/* fp32 mode (the reference's torch_dtype=torch.float32 configuration, src/lbm/inference/utils.py:21-26, 81-82, whose
 * layers are torch's f32 convolutions / linears): number of K chunks of a layer (the f32_chunks a split launch must
 * carry), and whether splitting pays for this shape (few output tiles, several chunks): 0 = launch unsplit */
int lbmk_igemm_f32_chunks(int K, int N, int pixels_per_image /* OH * OW of the descriptor */);
int lbmk_igemm_f32_split_pays(int M, int N, int K, int pixels_per_image, int batch);
int lbmk_igemm_smx_parts(int N);
/* floats behind gn_partial for a descriptor's shape, or -1 when the shape cannot carry fused statistics */
int64_t lbmk_igemm_gn_partial_floats(int M, int N, int Cin, int OH, int OW, int gn_groups);
/* number of tile configurations; cfg i has block tile bm[i] x bn[i] */
int lbmk_igemm_tiles(int* bm, int* bn, int cap);
#ifdef LBMK_DIAG
/* Diagnostic builds only (make -C lbm_amd/csrc DIAG=1; absent from the product library; no reference counterpart):
 * tile ids 115 / 116 / 117 run the 512x128 / 256x256 16-wave tiles and the 256x256 ping-pong tile with in-kernel time stamps; this copies out 8 x uint64 per
 * workgroup (100 MHz wall-clock stamps at start, after set-up, first stage landed, K loop done, epilogue issued, stores
 * drained; shader-clock counter at the K loop's start and end), nblocks <= 65536.  Tile id 118 runs the shared-row
 * 384x128 tile and stamps the shader clock around the parts of one steady-state K-step instead (slots 0 ... 4: step
 * start, requests issued, MFMAs issued, own requests landed, barrier passed).  The one entry point that
 * synchronises the device.  Returns the hipError_t of the copy. */
int lbmk_debug_igemm_times(unsigned long long* host, int nblocks);
#endif

/* ------------------------------------------------------------------------
 * GroupNorm(+SiLU) over NHWC bf16.  Replaces torch.nn.GroupNorm(32, C, eps)
 * followed by SiLU inside diffusers' ResnetBlock2D / Transformer2DModel /
 * conv_norm_out (UNet eps 1e-5, src/lbm/inference/utils.py:132-133; VAE and
 * transformer norm eps 1e-6).
 * partial: float scratch of lbmk_groupnorm_scratch_floats(B, HW, C, G) floats.
 * ---------------------------------------------------------------------- */
int64_t lbmk_groupnorm_scratch_floats(int B, int HW, int C, int G);
int lbmk_groupnorm(const void* x, void* y, const float* gamma, const float* beta,
                   float* scratch, int B, int HW, int C, int G, float eps,
                   int silu, void* stream);

/* lbmk_groupnorm for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arithmetic and kernels as
 * lbmk_groupnorm; the slab partials and statistics live in a fixed device array inside the library
 * (lbmk_groupnorm_static_max_floats() floats = 1 MiB per GPU, held until the process ends) instead of a caller
 * workspace.  Shapes whose lbmk_groupnorm_scratch_floats() exceeds that return LBMK_ERR_ARG.  Calls no hipMalloc /
 * hipFree.  All calls on one GPU share the array: calls that overlap on one GPU (different streams) race. */
int64_t lbmk_groupnorm_static_max_floats(void);
int lbmk_groupnorm_static(const void* x, void* y, const float* gamma, const float* beta,
                          int B, int HW, int C, int G, float eps, int silu, void* stream);

/* Same normalisation, with the statistics pass replaced by partial sums a producer already wrote
 * (lbmk_igemm with gn_partial): partial is [B][G][nslab][2] f32; scratch holds 2*B*G floats. */
int lbmk_groupnorm_from_partials(const void* x, void* y, const float* gamma, const float* beta,
                                 const float* partial, int nslab, float* scratch, int B, int HW, int C,
                                 int G, float eps, int silu, void* stream);

/* LayerNorm over the last dim of [rows][C] bf16 (BasicTransformerBlock norm1/2/3, eps 1e-5). */
int lbmk_layernorm(const void* x, void* y, const float* gamma, const float* beta,
                   int rows, int C, float eps, void* stream);

/* ------------------------------------------------------------------------
 * Flash-style attention forward, softmax(q k^T * scale) v, no mask.
 * Replaces F.scaled_dot_product_attention inside diffusers' AttnProcessor2_0
 * (self-attention attn1/attn2 of BasicTransformerBlock; the reference passes
 * encoder_hidden_states=None, src/lbm/models/unets/unet.py:98,132).
 *   q  : bf16 [B][Nq][ldq], head h at columns h*D .. h*D+D-1 (+ q_off)
 *   k  : bf16 [B][Nk][ldk] likewise
 *   vt : bf16 [B][H*D][ldvt]  (V transposed: row = channel, column = key)
 *   o  : bf16 [B][Nq][ldo]
 * Supported D: 8, 16, 32, 40, 64, 80, 160; keys at or past Nk are masked, ldvt >= Nk rounded up to 8.
 *   lse : optional f32 [B][H][Nq], receives log2(sum_j exp(scale * q_i.k_j)) per row (what the
 *         backward pass recomputes the probabilities from); NULL to skip.
 * ---------------------------------------------------------------------- */
int lbmk_flash_attn_fwd(const void* q, const void* k, const void* vt, void* o,
                        int B, int H, int Nq, int Nk, int D,
                        int ldq, int ldk, int ldvt, int ldo,
                        int64_t sq, int64_t sk, int64_t svt, int64_t so,
                        float scale, float* lse, void* stream);

/* Attention backward without materialised scores (torch autograd through
 * F.scaled_dot_product_attention in the reference's training step,
 * src/lbm/trainer/trainer.py:99-118 -> unet forward).  D = 40, 64, 80 or 160; Nq and Nk multiples of 64.
 *   q, dq : bf16 [B][Nq][ld]   k, v, dk, dv : bf16 [B][Nk][ld]   o, dout : bf16 [B][Nq][ld_o]   (head h at
 *           columns h*D ..; q / k / v may be column windows of one packed projection output, ld its width)
 *   qt, dout_t : bf16 [B][H*D][ld_t_q] channel-major copies of q and dout;  kt : [B][H*D][ld_t_k]
 *   lse    : f32 [B][H][Nq] from lbmk_flash_attn_fwd
 *   delta  : f32 [B][H][Nq] scratch, receives rowsum(dout * o)
 * Batch strides (elements): sq (q, dq), sk (k, v, dk, dv), so (o, dout), sqt (qt), sdot (dout_t), skt (kt). */
int lbmk_flash_attn_bwd(const void* q, const void* qt, const void* k, const void* kt,
                        const void* v, const void* o, const void* dout, const void* dout_t,
                        const float* lse, float* delta, void* dq, void* dk, void* dv,
                        int B, int H, int Nq, int Nk, int D, int ld, int ld_o, int ld_t_q, int ld_t_k,
                        int64_t sq, int64_t sk, int64_t so, int64_t sqt, int64_t sdot, int64_t skt,
                        float scale, void* stream);

/* Row softmax: f32 scores [rows][ld_in] -> bf16 probabilities [rows][ld_out]
 * (unfused attention for the VAE mid-block: 1 head, head_dim 512). */
int lbmk_softmax_rows(const float* s, void* p, int64_t rows, int cols,
                      int64_t ld_in, int64_t ld_out, void* stream);

/* ------------------------------------------------------------------------
 * Bridge arithmetic (src/lbm/models/lbm/lbm_model.py).
 * All tensors are contiguous [B][per_sample] with per-sample sigma.
 * dtype codes: 0 = bf16, 1 = f32; sigma is f32 [B].  The reference multiplies by an
 * f32 sigma tensor, so its interpolant is f32 even for bf16 latents (out_dtype 1).
 * ---------------------------------------------------------------------- */
/* x_t = s*z_src + (1-s)*z + sigma_b*sqrt(s*(1-s))*eps ; rows with copy_src[b]!=0 get z_src
 * (lbm_model.py:138-148). */
int lbmk_bridge_interp(const void* z_src, const void* z, const void* eps,
                       const float* sigma, const uint8_t* copy_src, void* out,
                       int B, int64_t per_sample, float bridge_sigma, int in_dtype,
                       int out_dtype, void* stream);
/* x' = cast(x + (s_next - s)*pred) ; if eps: x' = cast(x' + sigma_b*sqrt(s_next*(1-s_next))*eps)
 * (scheduler.step + re-noise, lbm_model.py:346-351). */
int lbmk_euler_step(const void* x, const void* pred, const void* eps, void* out,
                    int64_t n, float s, float s_next, float bridge_sigma,
                    int dtype, void* stream);
/* per-sample mean over per_sample elements of ((pred-tgt)*mask)^2 (l2) or abs (l1)
 * with tgt = z_src - z  (lbm_model.py:159,186-195). mask may be NULL. out: f32 [B]. */
int lbmk_latent_loss(const void* pred, const void* z_src, const void* z,
                     const uint8_t* mask, float* out, int B, int64_t per_sample,
                     int l1, int pred_dtype, int z_dtype, void* stream);
/* out[b][i] = x[b][i] + a[b]*y[b][i], f32 out: predicted x0 = x_t - sigma*prediction
 * (lbm_model.py:160,293-298). */
int lbmk_axpy_rows(const void* x, const void* y, const float* a, float* out, int B,
                   int64_t per_sample, int x_dtype, int y_dtype, void* stream);

/* ------------------------------------------------------------------------
 * Layout / glue kernels.
 * ---------------------------------------------------------------------- */
/* NCHW (bf16 or f32, in_dtype 0/1) -> NHWC bf16 with channels zero-padded to Cp. */
int lbmk_nchw_to_nhwc(const void* x, void* y, int B, int C, int H, int W, int Cp,
                      int in_dtype, void* stream);
/* NHWC bf16 (channel stride Cs, first C channels) -> NCHW (bf16 or f32). */
int lbmk_nhwc_to_nchw(const void* x, void* y, int B, int C, int H, int W, int Cs,
                      int out_dtype, void* stream);
/* copy C channels of src (stride src_cs, offset src_off) into dst (stride dst_cs, offset dst_off);
 * channel counts and offsets are multiples of 8 (skip-connection concat). */
int lbmk_copy_channels(const void* src, void* dst, int64_t pixels, int C,
                       int src_cs, int src_off, int dst_cs, int dst_off, void* stream);
/* DiagonalGaussian sample + latent scaling (autoencoderKL.py:64-67):
 * moments NHWC bf16 [B][HW][ms] (first 2*Cz channels = mean | logvar),
 * z[b][c][hw] = ((mean + exp(0.5*clamp(logvar,-30,20))*eps) - shift)*scale ; eps NCHW; z NCHW.
 * eps / z dtype: 0 bf16, 1 f32. */
int lbmk_posterior_sample(const void* moments, const void* eps, void* z,
                          int B, int Cz, int HW, int ms, float shift, float scale,
                          int dtype, void* stream);
/* sinusoidal timestep embedding, cos first (flip_sin_to_cos=True, freq_shift=0):
 * out bf16 [B][dim]. */
int lbmk_timestep_embedding(const float* t, void* out, int B, int dim, void* stream);
/* y = silu(x), bf16, n multiple of 8 */
int lbmk_silu(const void* x, void* y, int64_t n, void* stream);
/* y = x*a + b elementwise on bf16 (n multiple of 8) — latent un-scaling before decode. */
int lbmk_scale_shift(const void* x, void* y, int64_t n, float a, float b, void* stream);
/* Gaussian-weighted tile accumulation (src/lbm/models/utils.py:206-258):
 * acc[b][c][y0+y][x0+x] += tile[b][c][y][x]*wy[y]*wx[x]; wsum[y0+y][x0+x] += wy[y]*wx[x]
 * tile: NHWC bf16 [B][th_full][tw_full][Cs] (cropped to th x tw); acc f32 NCHW [B][C][H][W]; */
int lbmk_tile_accumulate(const void* tile, float* acc, float* wsum,
                         const float* wy, const float* wx,
                         int B, int C, int Cs, int th_full, int tw_full, int th, int tw,
                         int H, int W, int y0, int x0, void* stream);
/* out = acc / wsum  (f32 NCHW -> out dtype 0 bf16 / 1 f32) */
int lbmk_tile_finalize(const float* acc, const float* wsum, void* out,
                       int B, int C, int H, int W, int out_dtype, void* stream);

/* ------------------------------------------------------------------------
 * Backward (training) path: the reference back-propagates through diffusers' UNet with torch
 * autograd (Lightning's training_step, examples/training/train_base.py:55-70 around
 * LBMModel.forward, src/lbm/models/lbm/lbm_model.py:96-184).  Here every gradient GEMM is an
 * lbmk_igemm launch on re-laid-out operands produced by the kernels below:
 *   dX = dY . W          igemm with the transposed (conv: flipped) packed weight
 *   dW = dY^T . col(X)   igemm over dY^T [N][M] and col(X)^T [K][M] (reduction over rows)
 * ---------------------------------------------------------------------- */
/* out[z][c][r] = in[z][r][c], bf16; R and C multiples of 8; batch matrices s_in / s_out elements apart */
int lbmk_transpose(const void* in, void* out, int64_t R, int C, int64_t ld_in, int64_t ld_out,
                   int batch, int64_t s_in, int64_t s_out, void* stream);
/* transposed im2col of a convolution input (same geometry fields as lbmk_igemm_desc):
 * out[(kh*KW+kw)*Cin + c][p] = x[pixel read by tap (kh,kw) of output pixel p][c], 0 outside the image;
 * out is [KH*KW*Cin][ld_out] bf16, ld_out >= B*OH*OW */
int lbmk_im2col_t(const void* x, void* out, int B, int IH, int IW, int OH, int OW, int KH, int KW,
                  int stride, int pad_t, int pad_l, int up2, int Cin, int a_cs, int a_coff,
                  int64_t ld_out, void* stream);
/* input gradient of a strided / upsampled convolution from dcol = dY . Wpacked ([B*OH*OW][ld_col] bf16):
 * dx[q][c] = sum over taps and over output pixels p whose tap reads q of dcol[p][tap*Cin + c]; dx NHWC bf16 */
int lbmk_col2im(const void* dcol, void* dx, int B, int IH, int IW, int OH, int OW, int KH, int KW,
                int stride, int pad_t, int pad_l, int up2, int Cin, int64_t ld_col, int dx_cs, void* stream);
/* Convolution parameter bf16 [N][Cin][KH][KW] -> GEMM operand bf16 [n_p][Kp] in lbmk_igemm's K order (k_order as in
 * lbmk_igemm_desc; channel count padded to 8, zero fill).  swap_flip = 1 packs the input-gradient filter: rows are
 * the input channels, K runs over (mirrored tap, output channel).  Done once per optimizer step per layer. */
int lbmk_pack_conv_weight(const void* w, void* out, int N, int Cin, int KH, int KW, int n_p, int Kp,
                          int k_order, int swap_flip, void* stream);

/* Weight-gradient GEMM on untransposed operands (dW = dY^T X of torch autograd's linear backward in the reference's
 * training step): out[n][k] = sum_m a[m][n] * b[m][k], a bf16 [M][lda] (N columns), b bf16 [M][ldb] (K columns).
 * splits > 1 cuts the m range; the kernel then writes f32 partials out[split][N][ldo] for lbmk_sum_partials
 * (out_f32 must be 1).  splits == 1: out is the result, bf16 or f32, [N][ldo].  N, K multiples of 8. */
int lbmk_gemm_tn(const void* a, const void* b, void* out, int M, int N, int K, int lda, int ldb, int ldo,
                 int splits, int out_f32, void* stream);

/* Input gradient of a linear layer on the UNTRANSPOSED weight (torch autograd of nn.Linear in diffusers' Attention /
 * FeedForward under the reference's trainer, src/lbm/trainer/trainer.py:217-240: dX = dY . W):
 * out[m][k] = sum_n a[m][n] * b[n][k];  a = dY bf16 [M][lda], b = W bf16 [N][ldb] (the parameter as stored), out bf16
 * [M][ldo] (splits == 1) or f32 partials [splits][M][ldo] over equal chunks of n, to be summed by lbmk_sum_partials.
 * N, K multiples of 8.  No transposed copy of the weight is made (the weights change every optimisation step). */
int lbmk_gemm_nn(const void* a, const void* b, void* out, int M, int N, int K, int lda, int ldb, int ldo,
                 int splits, int out_f32, void* stream);

/* out[i] = sum_s partial[s][i]: the split-K partial products of a weight-gradient GEMM (f32 [nseg][n],
 * summed in a fixed order) -> bf16 (out_dtype 0) or f32 (1).  Part of the dW = dY^T X path above. */
int lbmk_sum_partials(const float* partial, void* out, int64_t n, int nseg, int out_dtype, void* stream);

/* Finish of a split-K linear layer of the training step (torch.nn.functional.linear forward / input gradient under the
 * reference's trainer, src/lbm/trainer/trainer.py:60-90, on the few-row token matrices of a 512x512 batch): the K
 * range is cut into nseg chunks that run as ONE batched lbmk_igemm launch with f32 outputs part[s][M][N] (chunk s reads
 * channels s*K/nseg ... of a and columns s*K/nseg ... of w: sa = sw = K/nseg, so = seg_stride), and this kernel writes
 * out[m][n] = bf16(sum_s part[s][m][n] + bias[n] + residual[m][n]), chunks added in order.  N % 8 == 0. */
int lbmk_splitk_finish(const float* part, int nseg, int64_t seg_stride, const float* bias, const void* residual,
                       int ldr, void* out, int ldo, int M, int N, void* stream);

/* out[g][n] = sum of x[m][n] over the rows_per_group consecutive rows of group g (bias gradient: one group;
 * time-embedding gradient: one group per image).  x bf16 [rows][ld]; out bf16 (0) / f32 (1) [rows/rows_per_group][N] */
int64_t lbmk_colsum_scratch_floats(int64_t rows, int64_t rows_per_group, int N);
int lbmk_colsum(const void* x, void* out, float* scratch, int64_t rows, int64_t rows_per_group, int N,
                int64_t ld, int out_dtype, void* stream);

/* GroupNorm(+SiLU) backward (torch autograd of nn.GroupNorm + SiLU in diffusers' ResnetBlock2D).
 * stats: [B][G][mean, rstd] f32 as left by the forward pass; dgamma / dbeta f32 [C] (may be NULL). */
int64_t lbmk_groupnorm_bwd_scratch_floats(int B, int HW, int C, int G);
int lbmk_groupnorm_bwd(const void* x, const void* dy, void* dx, const float* stats, const float* gamma,
                       const float* beta, float* dgamma, float* dbeta, float* scratch, int B, int HW, int C,
                       int G, int silu, void* stream);
/* The same with (a) dgamma / dbeta written in the parameters' own dtype (param_f32: 1 f32, 0 bf16), e.g. straight into
 * the parameters' slots of a flat gradient buffer, and (b) `add` (bf16, x's shape, may be NULL): the gradient that reached
 * the SAME input tensor through its other consumer -- the residual link around the block, torch autograd's accumulation
 * of two gradients of one tensor -- added to dx in this pass instead of by a separate elementwise kernel:
 * dx = bf16(bf16(dx_norm) + add). */
int lbmk_groupnorm_bwd_ex(const void* x, const void* dy, void* dx, const float* stats, const float* gamma,
                          const float* beta, void* dgamma, void* dbeta, int param_f32, const void* add, float* scratch,
                          int B, int HW, int C, int G, int silu, void* stream);
/* LayerNorm backward; dgamma_dbeta f32 [2][C] (row 0 dgamma, row 1 dbeta), may be NULL; C <= 2048 */
int64_t lbmk_layernorm_bwd_scratch_floats(int rows, int C);
int lbmk_layernorm_bwd(const void* x, const void* dy, void* dx, const float* gamma, float* dgamma_dbeta,
                       float* scratch, int rows, int C, float eps, void* stream);
/* LayerNorm backward with separate dgamma / dbeta destinations in the parameters' dtype and the fused residual-gradient
 * add: see lbmk_groupnorm_bwd_ex (torch autograd of nn.LayerNorm inside diffusers' BasicTransformerBlock, whose input also
 * feeds the residual sum).  dgamma and dbeta are both NULL or both set. */
int lbmk_layernorm_bwd_ex(const void* x, const void* dy, void* dx, const float* gamma, void* dgamma, void* dbeta,
                          int param_f32, const void* add, float* scratch, int rows, int C, float eps, void* stream);
/* dx = dy * silu'(x), bf16, n % 8 == 0 */
int lbmk_silu_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream);
/* GEGLU (diffusers GEGLU: hidden, gate = proj(x).chunk(2, -1); hidden * gelu(gate)): h bf16 [rows][2F] */
int lbmk_geglu_fwd(const void* h, void* y, int64_t rows, int F, void* stream);
int lbmk_geglu_bwd(const void* h, const void* dy, void* dh, int64_t rows, int F, void* stream);
/* softmax backward on rows: dS = scale * P * (dP - rowsum(dP * P)); P bf16, dP f32, dS bf16 */
int lbmk_softmax_bwd_rows(const void* p, const float* dp, void* ds, int64_t rows, int cols, float scale, void* stream);
/* gradient of lbmk_latent_loss w.r.t. pred given gout f32 [B] (lbm_model.py:186-195) */
int lbmk_latent_loss_bwd(const void* pred, const void* z_src, const void* z, const uint8_t* mask,
                         const float* gout, void* dpred, int B, int64_t per_sample, int l1, int pred_dtype,
                         int z_dtype, void* stream);

/* Pixel reconstruction loss, l2 / l1 variants (lbm_model.py:197-249): out[b] = mean over the sample of
 * ((c(dec) - target) * mask)^2 or |.|.  clamp = 1: c clips to [-1, 1] and the gradient flows inside [-1, 1] only (the
 * reference clamps the VAE-decoded prediction, lbm_model.py:233-234); clamp = 0: c is the identity (no VAE, :235-236). */
int lbmk_pixel_loss(const void* dec, const void* target, const uint8_t* mask, float* out, int B,
                    int64_t per_sample, int l1, int clamp, int dec_dtype, int tgt_dtype, void* stream);
int lbmk_pixel_loss_bwd(const void* dec, const void* target, const uint8_t* mask, const float* gout,
                        void* ddec, int B, int64_t per_sample, int l1, int clamp, int dec_dtype, int tgt_dtype,
                        void* stream);

/* One AdamW step (torch.optim.AdamW semantics, the optimizer named by the reference's training configs,
 * src/lbm/trainer/trainer.py:118-121) over a flat range: p, g bf16 [n]; m, v f32 [n]; step counts from 1;
 * grad_scale multiplies g first (1/world_size after a summing all-reduce). */
int lbmk_adamw(void* p, const void* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
               float eps, float weight_decay, int step, float grad_scale, void* stream);

/* ------------------------------------------------------------------------
 * LPIPS perceptual pixel loss (lpips.LPIPS(net="vgg"): src/lbm/models/lbm/lbm_model.py:83, 248-249).  The
 * VGG16 convolutions are lbmk_igemm launches with LBMK_ACT_RELU; these are the pieces around them.
 * Activations NHWC bf16; reductions in a fixed order.
 * ---------------------------------------------------------------------- */
/* x NCHW [B][3][H][W] (x_dtype 0 bf16 / 1 f32), mask f32 [B][H][W] or NULL -> y NHWC bf16 [B][H][W][8]:
 * ((clamp ? clip(x,-1,1) : x) * mask - shift3[c]) * inv_scale3[c], channels 3..7 zero (the ScalingLayer). */
int lbmk_lpips_prep(const void* x, const float* mask, void* y, int B, int H, int W, int x_dtype, int clamp,
                    const float* shift3, const float* inv_scale3, void* stream);
int lbmk_lpips_prep_bwd(const void* x, const float* mask, const void* dy, void* dx, int B, int H, int W,
                        int x_dtype, int clamp, const float* inv_scale3, void* stream);
/* dx = dy where y > 0 (y = relu output), n elements, n % 8 == 0 */
int lbmk_relu_bwd(const void* y, const void* dy, void* dx, int64_t n, void* stream);
/* 2x2 stride-2 max pooling, x [B][H][W][C] -> y [B][H/2][W/2][C]; backward routes dy to the first maximum */
int lbmk_maxpool2(const void* x, void* y, int B, int H, int W, int C, void* stream);
int lbmk_maxpool2_bwd(const void* x, const void* dy, void* dx, int B, int H, int W, int C, void* stream);
/* per-layer head: partial[chunk][b] = sum over a chunk of 256 pixels of sum_c w[c] * keep * (n0_c - n1_c)^2,
 * n = f / (|f|_2 + 1e-10) over channels (normalize_tensor + NetLinLayer + spatial sum); keep: optional
 * dropout mask, uint8 [B][HW][C], kept elements count twice (nn.Dropout(0.5) of a NetLinLayer in train mode).
 * lbmk_sum_partials over the chunks, divided by HW, gives the layer's distance per image.  C in 64..512. */
int64_t lbmk_lpips_head_partial_floats(int B, int HW, int C);
int lbmk_lpips_head(const void* f0, const void* f1, const float* w, const uint8_t* keep, float* partial,
                    int B, int HW, int C, void* stream);
/* df0 = gradient of (gout[b] / HW) * that sum with respect to f0 */
int lbmk_lpips_head_bwd(const void* f0, const void* f1, const float* w, const uint8_t* keep, const float* gout,
                        void* df0, int B, int HW, int C, void* stream);

/* ------------------------------------------------------------------------
 * fp32 mode: the model held and evaluated in f32 (get_model(..., torch_dtype=torch.float32),
 * src/lbm/inference/utils.py:21-26, 81-82 — the reference's CPU-runnable configuration).  Same operators as above
 * on f32 channels-last activations; statistics in f64, library exp / erf / sin / cos.  lbmk_igemm takes f32
 * operands through lbmk_igemm_desc.in_f32.  Attention runs unfused: lbmk_igemm (scores, alpha = scale) ->
 * lbmk_softmax_rows_f32 -> lbmk_igemm (probabilities x V^T).
 * ---------------------------------------------------------------------- */
int64_t lbmk_groupnorm_f32_scratch_floats(int B, int G);
int lbmk_groupnorm_f32(const void* x, void* y, const float* gamma, const float* beta, float* scratch,
                       int B, int HW, int C, int G, float eps, int silu, void* stream);     /* C % 4 == 0 */
int lbmk_layernorm_f32(const void* x, void* y, const float* gamma, const float* beta,
                       int rows, int C, float eps, void* stream);                            /* C <= 2048 */
int lbmk_softmax_rows_f32(const float* s, float* p, int64_t rows, int cols, int64_t ld_in, int64_t ld_out, void* stream);
/* GEGLU after the projection: y[r][c] = h[r][c] * gelu_erf(h[r][F + c]), h [rows][2F] (the value half comes first) */
int lbmk_geglu_f32(const float* h, float* y, int64_t rows, int F, void* stream);
/* NCHW (bf16 / f32, in_dtype 0 / 1) -> NHWC f32, channels zero-padded to Cp (multiple of 4) */
int lbmk_nchw_to_nhwc_f32(const void* x, float* y, int B, int C, int H, int W, int Cp, int in_dtype, void* stream);
/* NHWC f32 (channel stride Cs) -> NCHW bf16 / f32 (out_dtype 0 / 1) */
int lbmk_nhwc_to_nchw_f32(const float* x, void* y, int B, int C, int H, int W, int Cs, int out_dtype, void* stream);
int lbmk_copy_channels_f32(const float* src, float* dst, int64_t pixels, int C, int src_cs, int src_off,
                           int dst_cs, int dst_off, void* stream);                           /* multiples of 4 */
int lbmk_posterior_sample_f32(const float* moments, const float* eps, float* z, int B, int Cz, int HW, int ms,
                              float shift, float scale, void* stream);
int lbmk_timestep_embedding_f32(const float* t, float* out, int B, int dim, void* stream);
int lbmk_silu_f32(const float* x, float* y, int64_t n, void* stream);
int lbmk_scale_shift_f32(const float* x, float* y, int64_t n, float a, float b, void* stream);
int lbmk_tile_accumulate_f32(const float* tile, float* acc, float* wsum, const float* wy, const float* wx,
                             int B, int C, int Cs, int th_full, int tw_full, int th, int tw,
                             int H, int W, int y0, int x0, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LBMK_H */

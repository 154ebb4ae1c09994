/* libcvaegan_hip.so -- C ABI of the MI355X (gfx950) kernels behind the CVAE train step.
 *
 * The reference (YuliangXiu/CVAE-GAN) has no FFI layer: every operation below
 * replaces a torch.nn call made by models/networks.py or models/cvae.py, cited per
 * entry point.  The Python host (cvae_gan_amd/) binds these with ctypes.
 *
 * Conventions
 *  - Plain pointers and sizes only.  Every pointer is DEVICE memory owned by the
 *    caller; the library never allocates or frees device memory, never
 *    synchronises, and enqueues only on the hipStream_t it is given.
 *  - Return value 0 = enqueued; negative = rejected before any launch
 *    (cvg_last_error() describes why).  No C++ exception crosses the boundary.
 *  - dtype: CVG_F32 computes with exact-f32 MFMA (v_mfma_f32_32x32x2_f32);
 *    CVG_BF16 stores activations / weight shadows as bf16 and accumulates in f32
 *    (v_mfma_f32_32x32x16_bf16).  Statistics, losses, master weights, gradients
 *    and optimiser state are f32 in both modes.
 *  - Activations are NHWC ("pixel rows x channels"), channel counts multiples of 8.
 */
#ifndef CVAEGAN_H
#define CVAEGAN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cvg_stream;            /* a hipStream_t */
enum { CVG_F32 = 0, CVG_BF16 = 1 };
enum { CVG_PAD_ZERO = 0, CVG_PAD_REFLECT = 1 };
enum { CVG_FORM_GATHER = 0, CVG_FORM_S2_TRANSPOSED = 1 };
enum { CVG_CONV_ALONE = 1 };
enum { CVG_WS_HEADER = 40960 };      /* floats at the start of a reduction workspace: arrival counters, see cvg_bn_stats */

int         cvg_version(void);
const char* cvg_last_error(void);

/* ---- implicit-GEMM convolution geometry ----------------------------------------
 * One descriptor covers every dense contraction of the step: Conv2d forward,
 * its input gradient (a convolution of dY with re-laid-out weights), Linear layers
 * (H = W = 1, one tap) and the weight gradient.  GEMM row m = (b, oy, ox) over
 * OH x OW "output" pixels per image; GEMM column k = tap * Cin + c.
 * Gathered input coordinate of (m, tap = ty * KW + tx):
 *     iy = oy * stride + ty * tstep_y + off_y,   ix likewise,
 * out-of-range coordinates read 0 (CVG_PAD_ZERO) or mirror (CVG_PAD_REFLECT).
 * Output element (m, n) is stored at
 *     y + b*y_img_pitch + (oy*oy_mul + oy_off)*y_row_pitch
 *       + (ox*ox_mul + ox_off)*y_pix_pitch + n          (pitches in elements).    */
typedef struct cvg_conv_desc {
    int32_t dtype;
    int32_t B, H, W;                 /* gathered tensor: images, rows, cols        */
    int32_t Cin;                     /* k-extent per tap, multiple of 8; a power of
                                        two when ntaps > 1                         */
    int32_t x_img_pitch, x_row_pitch, x_pix_pitch;   /* elements                   */
    int32_t OH, OW;                  /* GEMM-M pixels per image                    */
    int32_t N;                       /* output channels stored (multiple of 4)     */
    int32_t KW, ntaps;               /* taps per tap-row, total taps               */
    int32_t stride, tstep_y, tstep_x, off_y, off_x;
    int32_t pad_mode;
    int32_t y_img_pitch, y_row_pitch, y_pix_pitch;
    int32_t oy_mul, oy_off, ox_mul, ox_off;
    int32_t w_ld;                    /* weight row length = K padded to 64 (bf16) / 32 (f32) */
    int32_t w_rows;                  /* weight rows allocated (N padded to 128)    */
    int32_t co_launches;             /* weight gradient only: launches of this kind the caller runs side
                                        by side on different streams (0 or 1: one).  Each then aims for
                                        1/co_launches of the CUs, i.e. that fraction of the split count
                                        and of the partial-slab traffic                              */
    int32_t form;                    /* CVG_FORM_GATHER: the geometry above.
                                        CVG_FORM_S2_TRANSPOSED: the INPUT GRADIENT of a 3x3, stride-2,
                                        zero-padding-1 convolution (ConvResBlock's down conv,
                                        networks.py:209-210) in one launch: x = dY [B][H][W][Cin] (Cin = that
                                        convolution's output channels, a multiple of 32), y = dX
                                        [B][2H][2W][N] through the y pitches; kernel tap (kh, kw) adds
                                        dY[a + (kh == 0)][b + (kw == 0)] * W[.][n][kh][kw] to
                                        dX[2a + (kh != 1)][2b + (kw != 1)][n] (rows / columns of dY beyond H / W
                                        read 0).  KW = 3, ntaps = 9, weights packed by cvg_weight_pack*
                                        with k = tap * Cin + c; stride, tstep, off, o*_mul / o*_off are
                                        ignored; no bias / cbias / res, gate scales the output.  Only where
                                        cvg_conv_fwd_kernel() names "patch_dgrad_s2".                  */
    int32_t flags;                   /* CVG_CONV_ALONE: no other launch runs beside this one (a forward pass on
                                        one stream): the forward kernel may then keep one workgroup resident
                                        per CU and let it walk over several image groups                  */
} cvg_conv_desc;

/* Conv2d / Linear forward and input-gradient.
 * Replaces nn.Conv2d at models/networks.py:173, :246-249, :279-280, :148 and
 * nn.Linear at :153, :156, :187, :190 (forward), and their autograd input
 * gradients (models/cvae.py:152).
 *   x      gathered activations (dtype)
 *   w      weights [w_rows][w_ld] (dtype), row n = output channel, zero padded
 *   y      output (dtype)
 *   bias   f32 [N] or NULL
 *   cbias  f32 [n_class][N] per-class bias or NULL; labels int32 [B], every entry in
 *          [0, n_class): the kernels index tables with them unchecked (here and in
 *          cvg_label_bias_bwd, cvg_reparam_kl_fwd, cvg_gan_c_head); the caller validates
 *          (the encoder's one-hot label planes folded into the stem)
 *   res    tensor added to the output, addressed like y, or NULL
 *   gate   if non-NULL, res is scaled by (gate > 0 ? 1 : 0.2) elementwise
 *          (LeakyReLU derivative read from the saved activation); without res the
 *          OUTPUT is scaled instead (layers on the gather kernel only, see
 *          cvg_conv_fwd_kernel(); a bias is added before the scaling, split-K is off)
 *   ws     f32 workspace of cvg_conv_fwd_workspace() bytes (split-K partials)    */
size_t cvg_conv_fwd_workspace(const cvg_conv_desc* d);
int    cvg_conv_fwd(const cvg_conv_desc* d, const void* x, const void* w, void* y,
                    const float* bias, const float* cbias, const int32_t* labels,
                    const void* res, const void* gate, void* ws, size_t ws_bytes,
                    cvg_stream s);

/* Up to four convolutions of the same input over the same output grid in ONE launch (same
 * dtype, N, B, OH, OW; each with its own taps, packed weights and output offsets): the four
 * output-parity classes of the input gradient of the stride-2 convolutions
 * (ConvResBlock's conv3x3 stride 2, networks.py:198-221).  No bias / residual epilogue.
 * descs points to n consecutive descriptors, w to n packed-weight pointers.
 * gate (optional, laid out as y): y = conv * (gate > 0 ? 1 : 0.2), the derivative of the
 * LeakyReLU in front of the stride-2 convolution (networks.py:220), so that y is the gradient
 * of the res block's pre-activation sum and no consumer has to read gate again.          */
int cvg_conv_fwd_group(int32_t n, const cvg_conv_desc* descs, const void* x,
                       const void* const* w, void* y, const void* gate, cvg_stream s);

/* Weight gradient:  dWg[n][k] = sum_m dY[m][n] * gather(x)[m][k], then scattered to
 * the canonical (reference-layout) f32 gradient through `map`.
 * Replaces the weight-gradient half of autograd for the layers listed above.
 *   dy     [M][dy_ld] (dtype), dense rows
 *   grad   canonical f32 gradient tensor (OIHW conv / [out,in] linear)
 *   accumulate != 0 adds into grad instead of overwriting                          */
typedef struct cvg_wmap {
    /* canonical index of GEMM-layout element (n, k):
     *   (n / nd)*na + (n % nd)*nb + tap(k / kd)*ka + (k % kd)*kb
     * tap(j) = j (tap_mode 0), ntaps-1-j (1) or tap_table[j] (2; -1 = structural zero).
     * Elements with n >= n_valid, (k % kd) >= kc_valid or k / kd >= ntaps are zero, and so are
     * those with (n % nd) >= nr_valid when nr_valid > 0 (padding rows inside every group of nd). */
    int32_t nd, na, nb, kd, ka, kb;
    int32_t n_valid, kc_valid, ntaps, tap_mode;
    int32_t nr_valid, reserved;
    int16_t tap_table[64];
} cvg_wmap;

size_t cvg_conv_wgrad_workspace(const cvg_conv_desc* d);
/* Name of the kernel family the library picks for this descriptor (profiling labels).     */
const char* cvg_conv_fwd_kernel(const cvg_conv_desc* d);
const char* cvg_conv_wgrad_kernel(const cvg_conv_desc* d);
int    cvg_conv_wgrad(const cvg_conv_desc* d, const void* x, const void* dy, int32_t dy_ld,
                      const cvg_wmap* map, float* grad, int32_t accumulate,
                      void* ws, size_t ws_bytes, cvg_stream s);

/* The same with dY addressed through the descriptor's output pitches:
 * dY[b][oy][ox][n] = dy[b*y_img_pitch + oy*y_row_pitch + ox*y_pix_pitch + n], n < N.  A pixel
 * pitch below N makes the N columns an overlapping window over neighbouring pixels: with the
 * tail's dpre stored as 4-channel pixels, 8-pixel windows turn the 7x7 tail's weight gradient
 * (nn.Conv2d(64, 3, 7), networks.py:146-148) into 7 row taps with 32 "output channels" (kw, n).
 * Only descriptors the patch-staged kernels accept.                                        */
int    cvg_conv_wgrad_strided(const cvg_conv_desc* d, const void* x, const void* dy,
                              const cvg_wmap* map, float* grad, int32_t accumulate,
                              void* ws, size_t ws_bytes, cvg_stream s);

/* Canonical f32 weights -> the layout the kernels chosen for `d` consume, in d->dtype:
 * row-major [w_rows][w_ld] (k = tap*Cin + c, zero padded) for the gather kernel, MFMA
 * fragment order for the patch-staged kernel.  packed holds w_rows * w_ld elements.
 * Runs once per optimiser step per layout.                                          */
int cvg_weight_pack(const cvg_conv_desc* d, const cvg_wmap* map, const float* canonical,
                    void* packed, cvg_stream s);

/* The same for every layout of a network in ONE launch.  cvg_weight_pack_job() fills a
 * job record on the host (block0 = first 256-thread block of the job, the running sum of
 * the blocks of the jobs before it) and returns the job's block count (< 0: error); the
 * caller copies the records to device memory once and calls cvg_weight_pack_batch() after
 * every optimiser step (the AdamW update of cvae.py:157 invalidates the packed copies). */
typedef struct cvg_pack_job {
    cvg_wmap map;
    const float* canonical;
    void* packed;
    int64_t n_groups;            /* 8-element groups of the packed buffer               */
    int32_t block0, mode;        /* mode 0: row-major [rows][ld], 1: fragment order, 2: row-major
                                    with a tiled thread order for transposed layouts      */
    int32_t rows, ld, nchunks, ntaps, Cin, dtype;
} cvg_pack_job;
int cvg_weight_pack_job(const cvg_conv_desc* d, const cvg_wmap* map, const float* canonical,
                        void* packed, int32_t block0, cvg_pack_job* job);
int cvg_weight_pack_batch(const cvg_pack_job* jobs_dev, int32_t n_jobs, int32_t total_blocks,
                          cvg_stream s);

/* Stem label fold (encoder.forward's torch.cat of label planes, networks.py:86,
 * cvae.py:129): cbias[c][n] = sum over the 49 taps of W[n][3 + c][kh][kw].          */
int cvg_label_bias(const float* w_stem, float* cbias, int32_t cout, int32_t y_dim,
                   cvg_stream s);
/* ... and its gradient: dW[n][3+c][*][*] = sum_{b: label[b]==c} S[b][n] where
 * S[b][n] = sum_pixels dY[b][.][n]; db[n] = sum_b S[b][n].  ws: f32 [B][cout].      */
int cvg_label_bias_bwd(const void* dy, int32_t dtype, const int32_t* labels, int32_t B,
                       int32_t pixels, int32_t cout, int32_t y_dim, float* dw_stem,
                       float* dbias, float* ws, cvg_stream s);

/* ---- HBM-bound kernels ---------------------------------------------------------- */

/* Loader output on the device: uint8 images [B][P][P][3] (channel order as stored, BGR for
 * cv2.imread) -> f32 NCHW [B,3,P,P] = (v/255 - .5)/.5, i.e. ToTensor + Normalize(.5, .5) of
 * datasets.py:14-18 after the H2D copy (a quarter of the bytes of the f32 batch).          */
int cvg_image_u8_to_f32(const uint8_t* hwc, float* nchw, int32_t B, int32_t P, cvg_stream s);

/* images NCHW f32 [B,3,P,P] -> reflection-padded NHWC4 [B][P+6][P+8][4] (dtype),
 * channel 3 and the two extra columns are zero (nn.ReflectionPad2d(3), networks.py:172). */
int cvg_pack_image(const float* x, void* xpad, int32_t B, int32_t P, int32_t dtype, cvg_stream s);

/* Per-channel batch statistics of rows[M][C] (BatchNorm2d / BatchNorm1d training
 * mode, networks.py:174, :235, :238, :188).  Writes stats = {mean[C], invstd[C]} and,
 * if running_mean != NULL, updates running_mean / running_var (momentum 0.1,
 * unbiased variance) and increments *num_batches_tracked (int64).
 * ws: f32 workspace of 2 * cvg_bn_stats_blocks(M, C) * C floats, 8-byte aligned: CVG_WS_HEADER arrival counters,
 * then the block partials and the partials of groups of 16 blocks.  The totals are formed in the same launch: the
 * last block of a group to arrive sums the group, the last group sums the groups, each in index order, so the
 * result does not depend on the arrival order.  The counters must be ZERO before the first call that uses a
 * workspace (fill it with zeros once); every call leaves them at zero, so one workspace serves any sequence of
 * calls of cvg_bn_stats / cvg_bn_act_bwd / cvg_colsum on one stream, whatever their shapes.  Calls that may overlap
 * (different streams) need workspaces of their own.
 * The sums are taken about row 0 (sum of (x - x0), sum of (x - x0)^2), so the variance does
 * not lose digits when |mean| >> std or M is a handful of rows; during the call the invstd half of
 * `stats` holds x0.                                                                  */
int32_t cvg_bn_stats_blocks(int32_t M, int32_t C);
int cvg_bn_stats(const void* x, int32_t dtype, int32_t M, int32_t C, float* stats,
                 float* running_mean, float* running_var, int64_t* num_batches_tracked,
                 float* ws, cvg_stream s);
/* The same statistics without the pass over x: a convolution that the patch kernel takes can emit, from
 * its epilogue, one row of column sums and sums of squares of the values it stores per pixel tile
 * (cvg_conv_fwd_stats: partials f32 [2][tiles][N], tiles = cvg_conv_stats_partials(d), 0 = this layer
 * runs on the gather kernel); cvg_bn_stats_finalize turns such rows (or cvg_bn_stats's workspace) into
 * mean / invstd and the running-statistics update.  Fixed summation orders throughout.  Replaces
 * nn.Conv2d + the statistics half of nn.BatchNorm2d at models/networks.py:173-174, :234-238.       */
int32_t cvg_conv_stats_partials(const cvg_conv_desc* d);
int cvg_conv_fwd_stats(const cvg_conv_desc* d, const void* x, const void* w, void* y, const float* bias,
                       const float* cbias, const int32_t* labels, float* partials, cvg_stream s);
int cvg_bn_stats_finalize(const float* partials, int32_t nblk, int32_t M, int32_t C, float* stats,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked,
                          cvg_stream s);
/* cvg_bn_stats for callers that cannot allocate or pass device scratch memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the
 * reference.  Same kernels, arguments and results as cvg_bn_stats without `ws`: the block partials
 * go to a 4.7 MiB array that is part of the library's device image (one copy per GPU the library runs
 * on, held until the process ends), so the call allocates nothing.  Limit: C <= 1024 (larger C is
 * rejected with an error, use cvg_bn_stats).  All calls on one GPU share that array: calls whose
 * kernels may overlap on the same GPU (different streams) race and must be ordered by the caller.
 * Nothing needs clearing between calls: every partial that is read was written by the same call, and the
 * arrival counters start at zero in the library's image and are left at zero by every call.       */
int cvg_bn_stats_static(const void* x, int32_t dtype, int32_t M, int32_t C, float* stats,
                        float* running_mean, float* running_var, int64_t* num_batches_tracked,
                        cvg_stream s);
/* y = LeakyReLU_0.2( gamma * (x - mean) * invstd + beta  [+ res] ).
 * eval mode: pass stats built from the running statistics (cvg_bn_eval_stats).   */
int cvg_bn_act_fwd(const void* x, const void* res, void* y, int32_t dtype, int32_t M, int32_t C,
                   const float* stats, const float* gamma, const float* beta, cvg_stream s);
int cvg_bn_eval_stats(const float* running_mean, const float* running_var, int32_t C,
                      float* stats, cvg_stream s);
/* Inference (module.eval(), models/cvae.py:167 and :209): BatchNorm with running statistics is one
 * multiply-add per channel, scale = gamma / sqrt(running_var + eps), shift = beta - running_mean *
 * scale, and is applied in the epilogue of the convolution in front of it:
 *   y = LeakyReLU_0.2( (conv(x, w) + bias + cbias[label]) * scale[n] + shift[n] + res )
 * (res: the res block's skip tensor, addressed like y, or NULL), replacing nn.Conv2d +
 * nn.BatchNorm2d + nn.LeakyReLU at models/networks.py:173-175 and :233-243 without the separate
 * normalisation pass.  Only for layers the patch kernel takes (cvg_conv_fwd_kernel() names
 * "patch_fwd..."); others keep cvg_conv_fwd + cvg_bn_eval_stats + cvg_bn_act_fwd.               */
int cvg_bn_eval_scale_shift(const float* running_mean, const float* running_var, const float* gamma,
                            const float* beta, int32_t C, float* scale, float* shift, cvg_stream s);
int cvg_conv_bn_act_fwd(const cvg_conv_desc* d, const void* x, const void* w, void* y,
                        const float* bias, const float* cbias, const int32_t* labels,
                        const float* scale, const float* shift, const void* res, cvg_stream s);
/* Backward of the same: g = dy * (y > 0 ? 1 : 0.2);
 * dgamma = sum g*xhat, dbeta = sum g, dx = gamma*invstd*(g - dbeta/M - xhat*dgamma/M).
 * The residual branch's gradient is g itself; consumers recompute it from (dy, y)
 * through cvg_conv_fwd's res/gate arguments.  ws as for cvg_bn_stats (counters included).
 * y may be NULL for a BatchNorm without a residual input: the sign of the activation is then
 * recomputed from x (gamma * xhat + beta, the expression cvg_bn_act_fwd evaluates), which
 * saves one of the three tensor reads of both passes; beta is only read in that case.
 * y = NULL and beta = NULL: dy already carries the activation's derivative (g = dy).     */
int cvg_bn_act_bwd(const void* dy, const void* y, const void* x, void* dx, int32_t dtype,
                   int32_t M, int32_t C, const float* stats, const float* gamma, const float* beta,
                   float* dgamma, float* dbeta, float* ws, cvg_stream s);

/* Column sums of rows[M][ld] over the first C columns -> out[C] (bias gradients).
 * out_map_d/a/b: out index of column c = (c / d)*a + (c % d)*b.  ws as for cvg_bn_stats(M, C). */
int cvg_colsum(const void* x, int32_t dtype, int32_t M, int32_t C, int32_t ld, float* out,
               int32_t map_d, int32_t map_a, int32_t map_b, float* ws, cvg_stream s);

/* out[i] = in[(i / d)*a + (i % d)*b]  (f32 gather; bias permutation for NHWC).   */
int cvg_permute_f32(const float* in, float* out, int32_t n, int32_t d, int32_t a, int32_t b,
                    cvg_stream s);

/* CVAE_T._sample_latent + latent_loss (networks.py:24-47, cvae.py:137), fused.
 *   h      [B][h_ld] (dtype) encoder head output: mean in columns [0, z), log sigma in [z, 2z);
 *          h_ld >= 2z (rows padded to the 8-element vector width when 2z is not a multiple of 8)
 *   eps    f32 [B][z]
 *   labels int32 [B]
 *   zin    [B][zin_ld] (dtype): z, then the one-hot label, then zeros (decoder input,
 *          networks.py:64)
 *   z_mean, z_sigma  f32 [B][z]
 *   kl     f32 scalar: mean(mu^2 + sigma^2 - log sigma^2 - 1) / batch_size         */
int cvg_reparam_kl_fwd(const void* h, int32_t dtype, int32_t h_ld, const float* eps,
                       const int32_t* labels, int32_t B, int32_t z, int32_t y_dim, int32_t zin_ld, void* zin,
                       float* z_mean, float* z_sigma, float* kl, float batch_size,
                       cvg_stream s);
/* dh[:, :z] = dz + g_mean + kl_scale*2*mu/(B*z*batch)
 * dh[:, z:] = (dz*eps + g_sigma)*sigma + kl_scale*(2*sigma^2 - 2)/(B*z*batch)
 * dz is read from dzin[B][zin_ld] (dtype); g_mean / g_sigma (f32, may be NULL) are
 * gradients that arrive through the exposed z_mean / z_sigma tensors.  dh is [B][h_ld];
 * columns from 2z on are left as they are (the caller keeps them zero).             */
int cvg_reparam_kl_bwd(const void* dzin, int32_t dtype, int32_t zin_ld, const float* eps,
                       const float* z_mean, const float* z_sigma, const float* g_mean,
                       const float* g_sigma, float kl_scale, float batch_size, int32_t B,
                       int32_t z, void* dh, int32_t h_ld, cvg_stream s);

/* Class indices as the kernels use them: out[i] = labels[i] kept inside [0, n_class); *bad is set to
 * 1.0f when any index was outside (it is never cleared here: the caller reads and clears it when it
 * next synchronises).  The reference indexes a one-hot table with the label on the host
 * (models/cvae.py:124-130), where a bad index raises at once; on the device the error is deferred. */
int cvg_labels_prepare(const int32_t* labels, int32_t* out, int32_t n, int32_t n_class,
                       float* bad, cvg_stream s);

/* LeakyReLU(0.2) followed by Dropout(0.5) with an explicit keep mask (uint8, 1 = keep,
 * kept values x2) -- networks.py:154-155.  mask == NULL: eval mode (identity dropout). */
int cvg_lrelu_dropout_fwd(const void* x, const uint8_t* mask, void* y, int32_t dtype,
                          int64_t n, cvg_stream s);
int cvg_lrelu_dropout_bwd(const void* dy, const void* y, const uint8_t* mask, void* dx,
                          int32_t dtype, int64_t n, cvg_stream s);
/* Device-side keep mask: counter-based hash of (seed, index), p_keep = 0.5.        */
int cvg_dropout_mask(uint8_t* mask, int64_t n, uint64_t seed, cvg_stream s);

/* Interpolate(x2, bilinear, align_corners=False) + ReflectionPad2d(1)
 * (networks.py:277-278): x [B][h][w][C] -> u [B][2h+2][2w+2][C]; and the transpose. */
int cvg_upsample_pad_fwd(const void* x, void* u, int32_t dtype, int32_t B, int32_t h, int32_t w,
                         int32_t C, cvg_stream s);
int cvg_upsample_pad_bwd(const void* du, void* dx, int32_t dtype, int32_t B, int32_t h, int32_t w,
                         int32_t C, cvg_stream s);

/* Backward of deconv3x3 (Interpolate x2 + ReflectionPad2d(1) + Conv2d 3x3, networks.py:274-281) at LOW
 * resolution.  The up-sampling and padding are one linear spatial operator A (U[u] = sum_q A[u, q] x[q]) that does
 * not mix channels, so with Z_t[q] = sum_p A[p + t, q] dY[p] for the nine kernel taps t:
 *     dX[q] = sum_t W_t^T Z_t[q]          dW_t = sum_q Z_t[q] (x) x[q]
 * two plain GEMMs over the low-resolution pixels with K = 9 N (cvg_conv_fwd / cvg_conv_wgrad with one-tap
 * descriptors): a quarter of the multiplications of the same gradients taken on the up-sampled grid, and no
 * gradient of the up-sampled tensor.  This call builds z [B][h][w][9][N] (dtype, tap = ky * 3 + kx) from
 * dy [B][2h][2w][N]; N a multiple of 8.                                                                      */
int cvg_upsample_taps_bwd(const void* dy, void* z, int32_t dtype, int32_t B, int32_t h, int32_t w,
                          int32_t N, cvg_stream s);
/* ... and its forward the same way: y[p] = bias + sum_t sum_q A[p + t, q] V_t[q] with V_t[q] = W_t x[q].  The
 * channel mixing is ONE one-tap GEMM over the low-resolution pixels with 9 N output columns (cvg_conv_fwd; a
 * quarter of the multiplications of the convolution over the up-sampled tensor, which is never built); this call
 * applies the spatial operator: v [B][h][w][9][N] -> y [B][2h][2w][N] (dtype), bias f32 [N] or NULL.          */
int cvg_upsample_taps_fwd(const void* v, const float* bias, void* y, int32_t dtype, int32_t B, int32_t h,
                          int32_t w, int32_t N, cvg_stream s);

/* Transpose of ReflectionPad2d(pad): dxp [B][H+2p][W+2p][C] -> dx [B][H][W][C].     */
int cvg_reflect_fold(const void* dxp, void* dx, int32_t dtype, int32_t B, int32_t H, int32_t W,
                     int32_t C, int32_t pad, cvg_stream s);

/* nn.Tanh on the tail conv output (networks.py:149): pre [B*P*P][4] (dtype) ->
 * dec NCHW f32 [B][3][P][P]; backward d_dec NCHW f32 -> dpre [B*P*P][8] (dtype).
 * dpre_win (may be NULL) receives the same gradient as the interior [6..6+P) x [6..6+P)
 * of a 4-channel image [B][P+12][P+14][4] whose border the caller keeps at zero: the
 * layout the tail's input-gradient convolution reads as 7 row taps of 8-pixel windows. */
int cvg_tanh_fwd(const void* pre, float* dec, int32_t dtype, int32_t B, int32_t P, cvg_stream s);
/* Tail conv (nn.Conv2d(64, 3, 7) after ReflectionPad2d(3), networks.py:146-148) finished from
 * column partials + nn.Tanh: v [B][P][P+6][32] (dtype) holds, for padded column x' and kernel
 * column kw, the sum over the 7 kernel rows and all input channels at v[..][x'][kw*4 + n] (a
 * 7-row-tap convolution with 32 "output channels"); pre[b][y][x][n] = bias[n] +
 * sum_kw v[b][y][x + kw][kw*4 + n] -> pre [B*P*P][4] (dtype) and dec = tanh(pre), NCHW f32.
 * bias: the canonical 3 floats.                                                             */
int cvg_tail_sum_tanh(const void* v, const float* bias, void* pre, float* dec, int32_t dtype,
                      int32_t B, int32_t P, cvg_stream s);
int cvg_tanh_bwd(const float* d_dec, const float* dec, void* dpre, void* dpre_win, int32_t dtype,
                 int32_t B, int32_t P, cvg_stream s);
/* Gradient with respect to the padded 3-channel input of a 7x7 convolution (the stem, nn.Conv2d(3 + y, 64, 7) after
 * ReflectionPad2d(3), networks.py:168-170, as a critic's trunk sees it) finished from column partials: v [B][rows][P][32]
 * (dtype) holds, for column j of the incoming gradient and kernel column kw, the sum over the 7 kernel rows and all
 * output channels at v[..][j][kw*4 + c] (a 7-row-tap convolution with 32 "output channels", a seventh of the
 * multiplications of the 49-tap form with 3 real output rows of 32);
 * out[b][y][x][c] = sum_kw v[b][y][x - kw][kw*4 + c], x in [0, P + 6) -> out [B*rows*(P+6)][8] (dtype), channels 3.. zero. */
int cvg_window_col_sum(const void* v, void* out, int32_t dtype, int32_t B, int32_t rows, int32_t P, cvg_stream s);

/* Measurement aid with no counterpart in the reference (for bench.py and tools): `workgroups` workgroups of four waves
 * each issue iters * 32 independent v_mfma_f32_32x32x16_bf16 from registers, no memory traffic:
 * workgroups * 4 * iters * 32 * 32768 FLOPs.  out: workgroups * 256 floats (written so that the loop is kept).  The
 * rate it reaches is what the matrix pipes deliver at the clock the GPU sustains under matrix load. */
int cvg_mfma_rate_probe(float* out, int32_t workgroups, int32_t iters, cvg_stream s);
/* Reconstruction loss of models/cvae.py:139: ll = 50 * mean((dec - x)^2) / batch_size,
 * and d_dec = scale * 100 * (dec - x) / (numel * batch_size).                       */
int cvg_mse_fwd_bwd(const float* dec, const float* x, float* d_dec, float* ll, int64_t numel,
                    float batch_size, float scale, float* ws, cvg_stream s);

/* torch.optim.AdamW(betas=(0.5, 0.999), eps=1e-8, weight_decay=0.01) over one flat
 * f32 arena (models/cvae.py:67, :157).  step counts from 1.  The hyper-parameters are doubles, as
 * the Python floats torch receives: 1 - lr * weight_decay, 1 - beta, lr / (1 - beta1^t) and
 * sqrt(1 - beta2^t) are evaluated in double and rounded to f32 once, where torch rounds them
 * (1 - 0.999 evaluated in f32 is off by 1.3e-5 relative, and with it the second moment).   */
int cvg_adamw(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
              double beta2, double eps, double weight_decay, int32_t step, cvg_stream s);

/* The same with the step count kept on the device: *step_counter (int64) is incremented
 * and bc[2] = { lr / (1 - beta1^t), sqrt(1 - beta2^t) } is written by a one-thread kernel before
 * the update, so a captured HIP graph of the step replays a correct optimiser.            */
int cvg_adamw_dev(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                  double beta2, double eps, double weight_decay, int64_t* step_counter, float* bc,
                  cvg_stream s);
/* Keep mask seeded by a device counter that advances on every call (graph-replay safe). */
int cvg_dropout_mask_dev(uint8_t* mask, int64_t n, int64_t* seed_counter, cvg_stream s);

/* f32 <-> dtype copies of dense buffers (module-boundary conversions).             */
int cvg_cast(const void* in, int32_t in_dtype, void* out, int32_t out_dtype, int64_t n,
             cvg_stream s);
/* NHWC (dtype) [B][HW][C] <-> NCHW f32 [B][C][HW] (test / export helper).           */
int cvg_nhwc_to_nchw(const void* in, int32_t dtype, float* out, int32_t B, int32_t HW, int32_t C,
                     int32_t ld, cvg_stream s);
int cvg_nchw_to_nhwc(const float* in, void* out, int32_t dtype, int32_t B, int32_t HW, int32_t C,
                     int32_t ld, cvg_stream s);

/* ---- adversarial extension (CVAE-GAN, arXiv 1703.10155, named by the reference's README.md:3;
 * the reference itself holds no discriminator / classifier code).  Discriminator and classifier
 * are the encoder trunk (ResNetEncoder, networks.py:164-196, without label planes) followed by
 * one linear head on its F-dimensional output h [M][ld] (dtype).  ws: float scratch.           */
/* Discriminator head + BCEWithLogits: logit = <h_row, w> + b, target 1 for rows < n_real, 0 after;
 * loss = mean over the M rows; dh = grad_scale * dloss/dh (padding columns 0), dw [F], db [1].
 * ws >= 2*M floats.                                                                              */
int cvg_gan_d_head(const void* h, int32_t dtype, int32_t ld, const float* w, const float* b,
                   int32_t M, int32_t F, int32_t n_real, float grad_scale,
                   float* dw, float* db, float* loss, void* dh, float* ws, cvg_stream s);
/* Classifier head + softmax cross-entropy over the first n_real rows (mean): W [y][F], b [y],
 * labels int32 [>= n_real]; rows >= n_real get dh = 0.  ws >= M*y + M floats.                    */
int cvg_gan_c_head(const void* h, int32_t dtype, int32_t ld, const float* W, const float* b,
                   const int32_t* labels, int32_t M, int32_t F, int32_t y, int32_t n_real,
                   float grad_scale, float* dW, float* db, float* loss, void* dh, float* ws,
                   cvg_stream s);
/* Mean feature matching: loss = 1/2 |mean(h[r0:r1]) - mean(h[f0:f1])|^2 with the first mean a
 * constant target; dh rows [f0, f1) = grad_scale * (mean_f - mean_r) / (f1 - f0), other rows 0. */
int cvg_gan_feature_match(const void* h, int32_t dtype, int32_t ld, int32_t M, int32_t F,
                          int32_t r0, int32_t r1, int32_t f0, int32_t f1, float grad_scale,
                          float* loss, void* dh, cvg_stream s);
/* Per-class feature centres (the paper's L_GC).  labels_real int32 [r1 - r0] and labels_fake int32 [f1 - f0]
 * are the classes of the rows [r0, r1) (real images) and [f0, f1) (generated images) of h.  centres f32
 * [n_class][F] and seen int32 [n_class] are state kept by the caller across steps (start at zero): the centre
 * of a class is the moving average (momentum) of the mean feature of its real rows, taken over as is the first
 * time the class appears.  loss = sum over classes with generated rows of 1/2 |mean_fake_c - centre_c|^2; the
 * gradient grad_scale * (mean_fake_c - centre_c) / n_fake_c is ADDED to the generated rows of dh (other rows
 * untouched, so it composes with cvg_gan_feature_match, which writes every row).  ws >= n_class floats.     */
int cvg_gan_class_feature_match(const void* h, int32_t dtype, int32_t ld, int32_t F, const int32_t* labels_real,
                                const int32_t* labels_fake, int32_t r0, int32_t r1, int32_t f0, int32_t f1,
                                int32_t n_class, float* centres, int32_t* seen, float momentum, float grad_scale,
                                float* loss, void* dh, float* ws, cvg_stream s);
/* Pairwise feature reconstruction (the feature terms of the paper's L_G): loss = mean over i < n of
 * 1/2 |h[f0 + i] - h[r0 + i]|^2 with row r0 + i (the real image) a constant target; the gradient
 * grad_scale * (h[f0 + i] - h[r0 + i]) / n is ADDED to row f0 + i of dh.  ws >= n floats.                   */
int cvg_gan_pair_feature_match(const void* h, int32_t dtype, int32_t ld, int32_t F, int32_t r0, int32_t f0,
                               int32_t n, float grad_scale, float* loss, void* dh, float* ws, cvg_stream s);
/* y += alpha * x over n floats (gradient arenas of two passes over shared weights, image
 * gradients arriving from several critics).                                                     */
int cvg_axpy_f32(float* y, const float* x, float alpha, int64_t n, cvg_stream s);

#ifdef __cplusplus
}
#endif
#endif /* CVAEGAN_H */

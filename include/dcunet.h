/* dcunet.h -- C ABI of libdcunet.so: the UNet2DS hot path as gfx950 HIP kernels.
 *
 * The reference (alexklibisz/deep-calcium) has no FFI for this path: every op
 * below is a Keras-2.0.6/TF-1.2.1 layer instantiated by unet() at
 * /root/reference/deepcalcium/models/neurons/unet_2d_summary.py:123-224 and run
 * through model.fit_generator (:429) / model.predict (:87, :588, :592).  Each
 * entry point names the reference call site whose arithmetic it replaces; the
 * ctypes binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - All pointers are DEVICE pointers (plain pointers, no torch types); the one
 *     exception, dc_crop_augment's `items_host`, says so in its name.
 *   - Activations NHWC fp32; conv kernels HWIO (3,3,Cin,Cout); transposed-conv
 *     kernels (2,2,Cout,Cin)  -- Keras `channels_last` layouts.
 *   - "ld" arguments are pixel strides in floats: a tensor argument `p, ld`
 *     addresses channel c of pixel i at p[i*ld + c] (lets producers write into
 *     channel slices of a concat buffer; pass ld == C for a dense tensor).
 *   - The library never allocates, frees or retains DEVICE memory; workspaces
 *     are caller-allocated (sizes from the *_ws_floats helpers).  There is no
 *     hipMalloc / hipFree / hipStreamSynchronize anywhere in the library.
 *   - Every launch is asynchronous on `stream` (a hipStream_t passed as void*;
 *     NULL = the legacy default stream).  The caller sets the device.
 *   - Return value: 0 OK, -1 invalid argument/shape/alignment, -2 HIP runtime
 *     error, -3 unsupported shape.  dc_last_error() returns a thread-local message.
 *   - Channel counts must be multiples of 4 (except the 1-channel network input
 *     and the 2-logit head) and powers of two where noted.
 */
#ifndef DCUNET_H
#define DCUNET_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* dc_stream_t;

/* ABI revision: dc_version() of the loaded library must EQUAL the DC_ABI_VERSION of the header the caller was built /
 * bound against (argument lists change between revisions; the Python binding refuses a mismatch). */
#define DC_ABI_VERSION 107
int dc_version(void);
const char* dc_last_error(void);

/* ---- weight re-layout for the implicit-GEMM kernels -------------------------
 * dst[tap][k/4][n][k%4] = src[(flip ? taps-1-tap : tap)*s_tap + k*s_k + n*s_n].
 *   conv3x3 fwd  : taps 9, K=Cin,  Ncols=Cout, s_tap=Cin*Cout, s_k=Cout, s_n=1,   flip 0
 *   conv3x3 dgrad: taps 9, K=Cout, Ncols=Cin,  s_tap=Cin*Cout, s_k=1,    s_n=Cout, flip 1
 *   convT   fwd  : taps 1, K=Cin,  Ncols=4*Cout, s_tap=0,      s_k=1,    s_n=Cin,  flip 0
 *   convT   dgrad: taps 4, K=Cout, Ncols=Cin,  s_tap=Cout*Cin, s_k=Cin,  s_n=1,    flip 0   */
int dc_pack_weights(const float* src, float* dst, int taps, int K, int Ncols,
                    long s_tap, long s_k, long s_n, int flip, dc_stream_t stream);

/* ---- Conv2D(nf,(3,3),'same')  unet_2d_summary.py:164-165 ---------------------
 * z = conv(x, w) + bias ; optional per-channel (sum, sumsq) partials of z for
 * BatchNorm (stats != NULL: double[tiles][Cout][2], tiles = dc_conv3x3_tiles(); formed per tile from shifted sums
 * and Chan merges, so the variance survives |mean| >> sigma -- csrc/common.h DcMoments);
 * optional fused inference epilogue y = relu?(z*scale + shift).
 * wp = dc_pack_weights(conv3x3 fwd form).  x: [N,H,W,Cin] dense, z: [N,H,W,Cout] with pixel stride z_ld. */
int dc_conv3x3_tiles(int N, int H, int W, int Cout);
int dc_conv3x3_fwd(const float* x, const float* wp, const float* bias, float* z, long z_ld, double* stats,
                   const float* scale, const float* shift, int relu,
                   int N, int H, int W, int Cin, int Cout, dc_stream_t stream);
/* First layer (Cin == 1): x is the (N,H,W) image itself, w is the plain HWIO (3,3,1,Cout) kernel.
 * stats: double[dc_conv3x3_c1_tiles()][Cout][2].  out_absmax (nullable, float[Cout], zeroed by the caller): max |output|
 * per channel is folded in (atomic max) -- the range-guard bound of the next f16x3 layer in inference. */
int dc_conv3x3_c1_tiles(int N, int H, int W, int Cout);
int dc_conv3x3_c1_fwd(const float* x, const float* w, const float* bias, float* z, long z_ld, double* stats,
                      const float* scale, const float* shift, int relu, float* out_absmax, long out_absmax_ld,
                      int N, int H, int W, int Cout, dc_stream_t stream);
/* dx = conv3x3_transpose(dz): wp = dc_pack_weights(conv3x3 dgrad form). dz: [N,H,W,Cout], dx: [N,H,W,Cin]. */
int dc_conv3x3_dgrad(const float* dz, const float* wp, float* dx,
                     int N, int H, int W, int Cin, int Cout, dc_stream_t stream);
/* dW (HWIO) = sum_p x[p+tap] (x) dz[p].  ws: float[dc_conv3x3_wgrad_ws_floats()] scratch (split-K slabs,
 * reduced in a fixed order => bit-reproducible). */
long dc_conv3x3_wgrad_ws_floats(int N, int H, int W, int Cin, int Cout);
int dc_conv3x3_wgrad(const float* x, const float* dz, float* dw, float* ws,
                     int N, int H, int W, int Cin, int Cout, dc_stream_t stream);

/* ---- split-fp16 ("f16x3") variants: same contraction on the fp16 matrix cores with fp32-grade accuracy ----
 * Every fp32 operand is split exactly into hi + lo fp16 (22-bit significand) and hi*hi + hi*lo + lo*hi is
 * accumulated in fp32: 3 fp16 MFMAs replace 8 fp32 ones.  wp16 = dc_pack_weights_f16x3 (same forms/strides as
 * dc_pack_weights; dc_pack_weights_f16x3_floats() floats of storage).
 * fp16's exponent range is covered by exact POWER-OF-TWO operand scales, undone in the epilogue:
 *   - weights: dc_pack_weights_f16x3 brings max|w| into [2^10, 2^11) and stores the scale in a 16-byte trailer of wp16;
 *   - gradient operands: in_scale / dz_scale, a nullable DEVICE scalar from dc_bn_bwd_apply_finalize;
 *   - activation operands: in_abound / x_abound, a nullable per-channel magnitude bound (float[Cin]) written by the
 *     layer that produced the tensor -- dc_bn_stats_finalize_affine / dc_bn_relu_drop_fwd in training
 *     (|gamma|*sqrt(count)+|beta|, valid for ANY data), the out_absmax of the producing kernel in inference; the
 *     consumer scales so that the largest bound lands in [2^14, 2^15): no activation can reach fp16 inf, tiny
 *     ones keep their low bits.  NULL = scale 1.
 *   Measured bounds (inference) live in DC_ABOUND_SLOTS = 8 replicas of the per-channel array, `*_ld` floats apart: a
 *   producer folds max |output| into replica (workgroup id mod 8) with an atomic max (out_absmax, out_absmax_ld; the
 *   caller zeroes all replicas first), the consumer takes the max over the replicas (in_abound_ld > 0).
 *   in_abound_ld == 0: a single array (the training-mode bound).
 *   out_absmax_ld == -1 (optimistic inference): nothing is measured; out_absmax[0] is set to 1 if any output exceeds
 *   32768 (or is NaN) -- the caller then repeats the forward pass with measured bounds.  A BatchNorm network's
 *   activations are O(1), so this path normally runs with no guard traffic at all.
 *   Narrow launches (images of at most 16 x 16 pixels whose grid would leave most of the chip idle: the 256- / 512-channel
 *   layers of a 128^2 / 96^2 training window) are split over K into 2-4 slabs and combined by a second launch in a fixed
 *   order (bit-reproducible; dc_conv3x3_tiles() rows unchanged).  The slabs live in the CALLER's `splitk_ws`
 *   (float[dc_conv3x3_splitk_ws_floats()] / float[dc_convT2x2_dgrad_splitk_ws_floats()]; 0 floats: this shape never
 *   splits); splitk_ws == NULL: the launch is not split (same result up to summation order).  One workspace serves the
 *   launches of ONE stream: launches on different streams that may run concurrently need their own. */
#define DC_ABOUND_SLOTS 8
long dc_pack_weights_f16x3_floats(int taps, int K, int Ncols);
int dc_pack_weights_f16x3(const float* src, void* dst, int taps, int K, int Ncols,
                          long s_tap, long s_k, long s_n, int flip, dc_stream_t stream);
/* every layer's dc_pack_weights_f16x3 in ONE launch.  jobs_dev: device array of (njobs + 1) x 10 longs,
 *   { src pointer, dst pointer, taps, K, Ncols, s_tap, s_k, s_n, flip, first block of the job };
 * the trailing sentinel entry only carries the grid size (= total_blocks) in its last field. */
int dc_pack_weights_f16x3_batch(const long* jobs_dev, int njobs, int total_blocks, dc_stream_t stream);
/* floats of split-K scratch a conv3x3 launch of this shape may use: forward incl. the _bnin variant (dgrad == 0) or plain
 * data gradient (dgrad != 0). */
long dc_conv3x3_splitk_ws_floats(int N, int H, int W, int Cin, int Cout, int dgrad);
long dc_convT2x2_dgrad_splitk_ws_floats(int N, int H, int W, int Cin, int Cout);
/* BatchNorm-partial rows of a split-fp16 conv3x3 forward.  stats_rows == 0 (or dc_conv3x3_tiles()): one row per pixel tile,
 * double[dc_conv3x3_tiles()][Cout][2].  stats_rows == dc_conv3x3_stats_rows(): where the persistent role-split kernel serves the
 * launch, every (workgroup, consumer set) Chan-merges the tiles it computes and writes ONE row -- at most 2 x #CUs rows instead
 * of up to 8 192, so the finalize launch that follows reads KBs instead of MBs (86 -> < 10 us at 512^2 x 16); the per-channel
 * statistics they finalize to are the same up to fp32 rounding of the merges.  For launches the role-split kernel does not
 * serve the query returns dc_conv3x3_tiles().  Needs the current device's CU count: call it on the GPU box. */
int dc_conv3x3_stats_rows(int N, int H, int W, int Cin, int Cout);
int dc_conv3x3_fwd_f16x3(const float* x, const void* wp16, const float* bias, float* z, long z_ld, double* stats, int stats_rows,
                         const float* scale, const float* shift, int relu, const float* in_abound, long in_abound_ld,
                         float* out_absmax, long out_absmax_ld, float* splitk_ws, int N, int H, int W, int Cin, int Cout,
                         dc_stream_t stream);
/* data gradients: the power-of-two scale of dz comes EITHER as a device scalar (dz_scale, from dc_pow2_scale_from_absmax /
 * dc_bn_bwd_apply_finalize; nullable = 1) OR as the per-block max |dz| array dc_bn_bwd_apply wrote (dz_absmax[dz_absmax_n]):
 * every workgroup then derives the same power of two itself (target 1024), and no finalize launch sits between the
 * BatchNorm-backward apply pass and the data gradient.  Pass at most one of the two. */
int dc_conv3x3_dgrad_f16x3(const float* dz, const void* wp16, float* dx, const float* dz_scale,
                           const float* dz_absmax, int dz_absmax_n, float* splitk_ws,
                           int N, int H, int W, int Cin, int Cout, dc_stream_t stream);
int dc_convT2x2_fwd_f16x3(const float* x, const void* wp16, const float* bias, float* z, long z_ld, double* stats,
                          const float* scale, const float* shift, int relu, const float* in_abound, long in_abound_ld,
                          float* out_absmax, long out_absmax_ld, int N, int H, int W, int Cin, int Cout, dc_stream_t stream);
int dc_convT2x2_dgrad_f16x3(const float* dz, const void* wp16, float* dx, const float* dz_scale,
                            const float* dz_absmax, int dz_absmax_n, float* splitk_ws,
                            int N, int H, int W, int Cin, int Cout, dc_stream_t stream);
/* BatchNorm-partial rows (`tiles`) of the split-fp16 conv-transpose forward: it runs wider column blocks than the fp32 kernel
 * (256 / 128 columns per workgroup: each staged input tile feeds 4x / 2x the MFMAs) on correspondingly flatter pixel tiles,
 * so its stats buffer is double[dc_convT2x2_f16x3_tiles()][4*Cout][2] -- NOT dc_convT2x2_tiles(), which sizes dc_convT2x2_fwd's. */
int dc_convT2x2_f16x3_tiles(int N, int H, int W, int Cout);
/* weight gradients: same workspace (dc_*_wgrad_ws_floats) and fixed-order slab reduction as the fp32 entry points;
 * dz_scale = device scalar from dc_pow2_scale_from_absmax (nullable). */
int dc_conv3x3_wgrad_f16x3(const float* x, const float* dz, float* dw, float* ws, const float* dz_scale,
                           const float* x_abound, int N, int H, int W, int Cin, int Cout, dc_stream_t stream);
int dc_convT2x2_wgrad_f16x3(const float* x, const float* dz, float* dw, float* ws, const float* dz_scale,
                            const float* x_abound, int N, int H, int W, int Cin, int Cout, dc_stream_t stream);

/* ---- BatchNorm + ReLU applied ON LOAD ("bnin"): the layer input is a NON-materialised activation -------------
 * Conv2D -> BatchNormalization -> Activation('relu') (unet_2d_summary.py:164-167) feeding the next Conv2D /
 * Conv2DTranspose / head with no Dropout or pooling in between never needs its activation in HBM: the consumer
 * takes the producer's pre-BN tensor z_in plus the per-channel training-mode affine
 *     in_scale[c] = gamma*invstd,  in_shift[c] = beta - mean*in_scale[c]      (dc_bn_stats_finalize_affine)
 * and forms relu(fmaf(z, in_scale, in_shift)) while staging the operand; zero padding stays zero.  Saves one HBM
 * write + one read of the activation per such layer (dc_bn_relu_drop_fwd is skipped).  Other arguments exactly as
 * in the entry point without _bnin. */
/* abound (nullable, float[C]): the activation's per-channel magnitude bound |gamma|*sqrt(count) + |beta| for the
 * consumers' fp16 range guard. */
int dc_bn_stats_finalize_affine(const double* partial, int parts, int groups, int C, double count, float eps,
                                float momentum, float* mean, float* invstd, float* moving_mean, float* moving_var,
                                const float* gamma, const float* beta, float* scale, float* shift, float* abound,
                                dc_stream_t stream);
int dc_conv3x3_fwd_bnin_f16x3(const float* z_in, const float* in_scale, const float* in_shift, const float* in_abound,
                              const void* wp16, const float* bias, float* z, long z_ld, double* stats, int stats_rows, const float* scale,
                              const float* shift, int relu, float* splitk_ws, int N, int H, int W, int Cin, int Cout,
                              dc_stream_t stream);
int dc_convT2x2_fwd_bnin_f16x3(const float* z_in, const float* in_scale, const float* in_shift, const float* in_abound,
                               const void* wp16, const float* bias, float* z, long z_ld, double* stats, const float* scale,
                               const float* shift, int relu, int N, int H, int W, int Cin, int Cout, dc_stream_t stream);
int dc_conv3x3_wgrad_bnin_f16x3(const float* z_in, const float* in_scale, const float* in_shift, const float* in_abound,
                                const float* dz, float* dw, float* ws, const float* dz_scale, int N, int H, int W,
                                int Cin, int Cout, dc_stream_t stream);
int dc_convT2x2_wgrad_bnin_f16x3(const float* z_in, const float* in_scale, const float* in_shift, const float* in_abound,
                                 const float* dz, float* dw, float* ws, const float* dz_scale, int N, int H, int W,
                                 int Cin, int Cout, dc_stream_t stream);
int dc_head_fwd_bnin(const float* z_in, const float* in_scale, const float* in_shift, const float* kh, const float* bh,
                     const uint8_t* y, float* p, float* partial, long pixels, int C, dc_stream_t stream);
int dc_head_bwd_bnin(const float* z_in, const float* in_scale, const float* in_shift, const float* p, const uint8_t* y,
                     const float* kh, float* da, float* partial, int loss_kind, const double* sums, long pixels, int C,
                     dc_stream_t stream);


/* ---- Conv2DTranspose(nf, 2, strides=2)  unet_2d_summary.py:156-157 -----------
 * x: [N,H,W,Cin] -> z: [N,2H,2W,Cout].  wp = dc_pack_weights(convT fwd form).
 * stats: double[dc_convT2x2_tiles()][4*Cout][2] (finalize with groups = 4). */
int dc_convT2x2_tiles(int N, int H, int W, int Cout);
int dc_convT2x2_fwd(const float* x, const float* wp, const float* bias, float* z, long z_ld, double* stats,
                    const float* scale, const float* shift, int relu,
                    int N, int H, int W, int Cin, int Cout, dc_stream_t stream);
int dc_convT2x2_dgrad(const float* dz, const float* wp, float* dx,
                      int N, int H, int W, int Cin, int Cout, dc_stream_t stream);
long dc_convT2x2_wgrad_ws_floats(int N, int H, int W, int Cin, int Cout);
int dc_convT2x2_wgrad(const float* x, const float* dz, float* dw, float* ws,
                      int N, int H, int W, int Cin, int Cout, dc_stream_t stream);

/* ---- BatchNormalization + Activation('relu') + Dropout  :158-159,:166-167,:179 ---
 * finalize: partial (sum,sumsq)[parts][groups*C][2] -> mean[C], invstd[C] = 1/sqrt(var_biased+eps);
 * if momentum >= 0: moving <- moving*momentum + batch*(1-momentum) (biased variance, Keras 2.0.6). */
int dc_bn_stats_finalize(const double* partial, int parts, int groups, int C, double count, float eps,
                         float momentum, float* mean, float* invstd, float* moving_mean, float* moving_var,
                         dc_stream_t stream);
/* inference fold: scale = gamma/sqrt(mvar+eps); shift = beta + (bias - mmean)*scale */
int dc_bn_fold(const float* gamma, const float* beta, const float* mmean, const float* mvar, const float* bias,
               float eps, float* scale, float* shift, int C, dc_stream_t stream);
/* a = dropout(relu(gamma*(z-mean)*invstd + beta)).  z dense [pixels][C]; out strided (out_ld).
 * Dropout: keep >= 1 -> none; mask != NULL -> explicit uint8 {0,1} [pixels][C]; else counter-based RNG(seed).
 * abound (nullable, float[C]): also writes the activation's per-channel magnitude bound
 * (|gamma|*sqrt(count) + |beta|)/keep for the consumers' fp16 range guard; count = elements per channel the
 * statistics were taken over (pixels; world*pixels under 'sync' data parallelism). */
int dc_bn_relu_drop_fwd(const float* z, const float* mean, const float* invstd, const float* gamma,
                        const float* beta, const uint8_t* mask, float keep, uint64_t seed,
                        float* out, long out_ld, long pixels, int C, double count, float* abound, dc_stream_t stream);
/* dc_bn_relu_drop_fwd of a block that feeds MaxPooling2D((2,2)) (unet_2d_summary.py:166-170) AND that pooling, in one
 * pass over z [N,H,W,C]: writes the activation (out, strided: it is also the skip connection), pooled [N,H/2,W/2,C] and
 * idx (nullable uint8, argmax 0..3 in row-major window order, first max on ties).  Bit-identical to the two separate
 * calls; one tensor read less. */
int dc_bn_relu_drop_pool_fwd(const float* z, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, const uint8_t* mask, float keep, uint64_t seed, float* out, long out_ld,
                             float* pooled, uint8_t* idx, int N, int H, int W, int C, double count, float* abound,
                             dc_stream_t stream);
/* backward, pass 1: partial[blocks][C][2] = (sum dy, sum dy*xhat) with dy = da*relu'(.)*dropmask/keep.
 * blocks = dc_bn_bwd_blocks(pixels, C).  amax_partial (nullable) [blocks][C] = max |dy| per block and channel
 * (what dc_bn_bwd_finalize_dzin bounds |dz| with; every pass-1 producer below takes the same argument). */
int dc_bn_bwd_blocks(long pixels, int C);
int dc_bn_bwd_reduce(const float* da, long da_ld, const float* z, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, const uint8_t* mask, float keep, uint64_t seed,
                     float* partial, float* amax_partial, long pixels, int C, dc_stream_t stream);
/* backward, pass 2: dz = gamma*invstd*(dy - dbeta/M - xhat*dgamma/M); dbias_partial[blocks][C] = sum dz;
 * absmax_partial (nullable) [blocks] = max |dz| per block, for dc_pow2_scale_from_absmax. */
int dc_bn_bwd_apply(const float* da, long da_ld, const float* z, const float* mean, const float* invstd,
                    const float* gamma, const float* beta, const uint8_t* mask, float keep, uint64_t seed,
                    const float* dgamma, const float* dbeta, float* dz, float* dbias_partial, float* absmax_partial,
                    long pixels, int C, dc_stream_t stream);
/* scale[0] = 2^floor(log2(target / max_i partial[i])) (1 for an all-zero tensor): the exact power-of-two input scale
 * of the f16x3 gradient contractions. */
int dc_pow2_scale_from_absmax(const float* partial, int n, float target, float* scale, dc_stream_t stream);
/* what follows dc_bn_bwd_apply, in ONE launch: dbias[c] = sum over blocks of dbias_partial[b][c] (fixed order) and,
 * when absmax_partial / scale are given, scale[0] as dc_pow2_scale_from_absmax(absmax_partial, blocks, target). */
int dc_bn_bwd_apply_finalize(const float* dbias_partial, const float* absmax_partial, int blocks, int C, float target,
                             float* dbias, float* scale, dc_stream_t stream);

/* ---- inference: Conv2D -> (folded BN) -> ReLU whose output also feeds MaxPooling2D((2,2)) ----------------------------
 * dc_conv3x3_fwd_f16x3 (optimistic range flag out_flag, nullable) that also writes pool [N,H/2,W/2,Cout] = what
 * dc_maxpool2x2_fwd gives on z.  dc_conv3x3_fwd_pool_blocks() == 0: shape not served, use the two kernels. */
int dc_conv3x3_fwd_pool_blocks(int N, int H, int W, int Cin, int Cout);
int dc_conv3x3_fwd_pool_f16x3(const float* x, const void* wp16, const float* bias, float* z, long z_ld,
                              const float* scale, const float* shift, int relu, float* out_flag, float* pool,
                              int N, int H, int W, int Cin, int Cout, dc_stream_t stream);

/* ---- BatchNorm-backward sums fused into the kernel that PRODUCES da ("bnred") -------------------------------
 * The kernel that writes da of a BatchNorm layer also emits that layer's pass-1 partial sums
 * bn_partial[blocks][C][2] = (sum dy, sum dy*xhat), dy = da * [relu gate] * dropout, so dc_bn_bwd_reduce (and its
 * re-read of da and z) is skipped; dc_bn_bwd_finalize(bn_partial, blocks, ...) follows as usual.
 *   dc_head_bwd_bnin_bnred : blocks = dc_head_blocks(pixels); BN layer = the head's (non-materialised) input layer
 *   dc_maxpool2x2_bwd_bnred: blocks = dc_maxpool2x2_bwd_blocks(); BN layer = the pooled layer (z dense [N,H,W,C]) */
/*   dc_conv3x3_dgrad_bnred_f16x3: dc_conv3x3_dgrad_f16x3 whose output dx IS the `da` of the BatchNorm layer in front
 *   (dense [N,H,W,Cin], no dropout, pre-BN tensor z of the same shape): rows = dc_conv3x3_dgrad_bnred_blocks(...) partial
 *   rows of bn_partial[rows][Cin][2]; rows == 0 -> shape not served, use dc_conv3x3_dgrad_f16x3 + dc_bn_bwd_reduce. */
/* Kernel routing query (no launch): > 0 (the pixel-tile count) when a conv3x3 launch of this shape -- forward
 * (dgrad == 0; with_stats != 0: BatchNorm partials requested) or data gradient (dgrad != 0) -- runs the persistent
 * role-split kernel `igemm_pp_kernel` (64-column instantiation <2,2,*> when the GEMM column count, Cout forward / Cin
 * data gradient, exceeds 32; <4,1,*> otherwise); 0: one of the 256-thread kernels.  Depends on shape and DC_IGEMM_PP only. */
int dc_conv3x3_pp_blocks(int N, int H, int W, int Cin, int Cout, int dgrad, int with_stats);
int dc_conv3x3_dgrad_bnred_blocks(int N, int H, int W, int Cin, int Cout);
int dc_conv3x3_dgrad_bnred_f16x3(const float* dz, const void* wp16, float* dx, const float* dz_scale,
                                 const float* dz_absmax, int dz_absmax_n, const float* z,
                                 const float* mean, const float* invstd, const float* gamma, const float* beta,
                                 float* bn_partial, float* amax_partial, int N, int H, int W, int Cin, int Cout,
                                 dc_stream_t stream);
int dc_head_bwd_bnin_bnred(const float* z_in, const float* in_scale, const float* in_shift, const float* p,
                           const uint8_t* y, const float* kh, float* da, float* partial, int loss_kind,
                           const double* sums, const float* bn_mean, const float* bn_invstd, float* bn_partial,
                           float* amax_partial, long pixels, int C, dc_stream_t stream);
/*   dc_convT2x2_dgrad_bnred_f16x3: dc_convT2x2_dgrad_f16x3 whose output dx IS the `da` of the block in front of the up-convolution
 *   (dense [N,H,W,Cin], no Dropout): rows = dc_convT2x2_dgrad_bnred_blocks(); 0 -> not served (W <= 16), two-pass path. */
int dc_convT2x2_dgrad_bnred_blocks(int N, int H, int W, int Cin, int Cout);
int dc_convT2x2_dgrad_bnred_f16x3(const float* dz, const void* wp16, float* dx, const float* dz_scale,
                                  const float* dz_absmax, int dz_absmax_n, const float* z,
                                  const float* mean, const float* invstd, const float* gamma, const float* beta,
                                  float* bn_partial, float* amax_partial, int N, int H, int W, int Cin, int Cout,
                                  dc_stream_t stream);
int dc_maxpool2x2_bwd_blocks(int N, int H, int W, int C);
int dc_maxpool2x2_bwd_bnred(const float* dy, const uint8_t* idx, const float* skip, long skip_ld, float* dx,
                            const float* z, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, const uint8_t* mask, float keep, uint64_t seed, float* bn_partial,
                            float* amax_partial, int N, int H, int W, int C, dc_stream_t stream);

/* ---- BatchNorm backward WITHOUT the apply pass: "dz on load" (unet_2d_summary.py:163-167, Dropout-free blocks) ------
 * dz = gamma*invstd*(dy - dbeta/M - xhat*dgamma/M) is a per-channel affine of (dy, z): the data- and weight-gradient
 * kernels of the block form it while they stage their operands, so dz is never written and dc_bn_bwd_apply (read da,
 * read z, write dz) disappears:  dy = [fmaf(z, sc, sh) > 0] * da ;  dz = fmaf(A, dy, fmaf(D, z - mu, E)).
 *   dc_bn_bwd_finalize_dzin = dc_bn_bwd_finalize (partial != NULL; partial == NULL: dgamma / dbeta are INPUTS, the
 *     all-reduced sums of 'sync' BatchNorm, count = global elements per channel) + the table
 *     dz_coef[DC_DZ_COEF_ROWS][C] = { sc, sh, mu, A, D, E, bound } with bound_c >= max |dz_c| for ANY data
 *     (= |A| (max|dy_c| + |dbeta|/M + sqrt(M) |dgamma|/M); amax_partial[P][C] = per-row max |dy_c| from the pass-1
 *     producer): the consumers' fp16 range guard brings the largest bound into [2^14, 2^15).
 *   dc_conv3x3_dgrad_dzin_f16x3: dx = conv3x3_transpose(dz) from (da, z, dz_coef) (all dense [N,H,W,Cout]); role-split
 *     kernel only: dc_conv3x3_dgrad_dzin_blocks() == 0 -> shape not served, use dc_bn_bwd_apply + dc_conv3x3_dgrad_f16x3.
 *     red_z != NULL: dx IS the `da` of the BatchNorm layer in front (as dc_conv3x3_dgrad_bnred_f16x3): also emits
 *     bn_partial[rows][Cin][2] and amax_partial[rows][Cin], rows = the returned block count.
 *     dz_out != NULL (round 5): the dz the producers form is also WRITTEN, once per pixel (dense [N,H,W,Cout], fp32, unscaled),
 *     for a plain weight gradient (dc_conv3x3_wgrad_f16x3 / _bnin, dz_scale = dc_pow2_scale_from_absmax over row 6 of dz_coef):
 *     the BatchNorm-backward apply pass of the block disappears (1 tensor write instead of 2 reads + 1 write) and only the data
 *     gradient pays for forming dz -- the dz-on-load weight gradient is the slower of the two dz-on-load kernels.
 *   dc_conv3x3_wgrad_dzin_f16x3: dW = sum_p x[p+tap] (x) dz[p]; x as dc_conv3x3_wgrad_f16x3 (in_scale / in_shift NULL)
 *     or the producer's pre-BN tensor with BN + ReLU on load (as dc_conv3x3_wgrad_bnin_f16x3).  Cin == 1 (first layer):
 *     x is the (N,H,W) image.
 * The conv bias in front of a BatchNorm has the exact gradient sum_p dz = 0: dc_bn_bwd_finalize_dzin writes that 0 to
 * dbias[C] (nullable; the apply pass summed rounding noise there). */
#define DC_DZ_COEF_ROWS 7
int dc_bn_bwd_finalize_dzin(const float* partial, const float* amax_partial, int P, int C, const float* mean,
                            const float* invstd, const float* gamma, const float* beta, double count,
                            float* dgamma, float* dbeta, float* dz_coef, float* dbias, dc_stream_t stream);
int dc_conv3x3_dgrad_dzin_blocks(int N, int H, int W, int Cin, int Cout);
int dc_conv3x3_dgrad_dzin_f16x3(const float* da, const float* z, const float* dz_coef, const void* wp16, float* dx, float* dz_out,
                                const float* red_z, const float* red_mean, const float* red_invstd,
                                const float* red_gamma, const float* red_beta, float* bn_partial, float* amax_partial,
                                int N, int H, int W, int Cin, int Cout, dc_stream_t stream);
int dc_conv3x3_wgrad_dzin_f16x3(const float* x, const float* in_scale, const float* in_shift, const float* x_abound,
                                const float* da, const float* z, const float* dz_coef, float* dw, float* ws,
                                int N, int H, int W, int Cin, int Cout, dc_stream_t stream);

/* Joint backward of a conv block with 32 output channels at a 512^2-class resolution (the HBM-bound blocks of the network:
 * e0b / d0b, 32 -> 32, and d0a, 64 -> 32): ONE kernel stages dz (formed on load, as above) and the block input x once per
 * tile and produces dx, dW and -- 32 -> 32 only, red_z != NULL -- the pass-1 sums / max |dy| of the layer in front: every
 * tensor of the block is read once (4 tensor passes instead of 7; when red_z is x itself, the usual case, its values reach the
 * sums through LDS).  Arguments as dc_conv3x3_dgrad_dzin_f16x3 + dc_conv3x3_wgrad_dzin_f16x3 (wp16 = the data-gradient form
 * of the kernel); the 64 -> 32 variant takes a materialised x (in_scale NULL) and no red_z.
 * dc_conv3x3_bwd_joint_blocks() = rows of bn_partial / amax_partial (0: shape not served -- needs Cout == 32, Cin 32 or 64,
 * W >= 32: use the two separate kernels); ws: float[dc_conv3x3_bwd_joint_ws_floats()] (one dW slab per workgroup, reduced
 * in a fixed order => bit-reproducible). */
int dc_conv3x3_bwd_joint_blocks(int N, int H, int W, int Cin, int Cout);
long dc_conv3x3_bwd_joint_ws_floats(int N, int H, int W, int Cin, int Cout);
int dc_conv3x3_bwd_joint_f16x3(const float* x, const float* in_scale, const float* in_shift, const float* x_abound,
                               const float* da, const float* z, const float* dz_coef, const void* wp16, float* dx,
                               const float* red_z, const float* red_mean, const float* red_invstd, const float* red_gamma,
                               const float* red_beta, float* bn_partial, float* amax_partial, float* dw, float* ws,
                               int N, int H, int W, int Cin, int Cout, dc_stream_t stream);

/* ---- synchronised BatchNorm for batch-sharded data parallelism ('sync' mode) ---------------------
 * The per-channel sums leave the device between two launches so the host can all-reduce them over the ranks:
 *   forward : dc_bn_stats_reduce (conv partials -> double sums[C][2]) | all-reduce(sum) | dc_bn_stats_finalize_sums with
 *             count = GLOBAL elements per channel (scale/shift nullable: the training affine of the _bnin consumers);
 *   backward: dc_bn_bwd_finalize | all-reduce(sum) of (dgamma, dbeta) | dc_bn_bwd_apply_count with the same count.
 * With one rank (or count == pixels) the results equal the local entry points'. */
int dc_bn_stats_reduce(const double* partial, int parts, int groups, int C, double* sums, dc_stream_t stream);
/* partial rows folded parts -> chunks (fixed order, whole rows read coalesced): out = double[chunks][groups * C][2], same layout
 * as `partial`; dc_bn_stats_finalize* / dc_bn_stats_reduce then take (out, chunks, groups).  For launches with thousands of tile
 * rows (the 512^2-class layers), where the finalize's strided 16-byte reads cost more than this extra launch. */
int dc_bn_stats_rows_fold(const double* partial, int parts, int groups, int C, int chunks, double* out, dc_stream_t stream);
int dc_bn_stats_finalize_sums(const double* sums, int C, double count, float eps, float momentum, float* mean,
                              float* invstd, float* moving_mean, float* moving_var, const float* gamma,
                              const float* beta, float* scale, float* shift, float* abound, dc_stream_t stream);
int dc_bn_bwd_apply_count(const float* da, long da_ld, const float* z, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, const uint8_t* mask, float keep, uint64_t seed,
                          const float* dgamma, const float* dbeta, float* dz, float* dbias_partial,
                          float* absmax_partial, long pixels, double count, int C, dc_stream_t stream);

/* ---- MaxPooling2D(2, strides=2)  :176 ----------------------------------------
 * in strided [N,H,W,C] (in_ld), out dense [N,H/2,W/2,C], idx (nullable) uint8 in {0..3}: FIRST max in
 * row-major window order (bit-exact contract). */
int dc_maxpool2x2_fwd(const float* in, long in_ld, float* out, uint8_t* idx,
                      int N, int H, int W, int C, dc_stream_t stream);
/* dx[N,H,W,C] dense = route(dy, idx) + (skip ? skip (strided, skip_ld) : 0) */
int dc_maxpool2x2_bwd(const float* dy, const uint8_t* idx, const float* skip, long skip_ld, float* dx,
                      int N, int H, int W, int C, dc_stream_t stream);

/* ---- UpSampling2D() + Dropout  :160-161,:198 (the upsampling_or_transpose != 'transpose' branch) -------------
 * nearest 2x of in [N,H,W,C] (dense) into out [N,2H,2W,C] (pixel stride out_ld), dropout applied to the
 * up-sampled tensor (mask uint8 [N,2H,2W,C] or counter RNG(seed); keep >= 1: none).
 * abound_in / abound_out (nullable pair, float[C]): range-guard bound of the source / of the up-sampled tensor
 * (= abound_in / keep); abound_in_ld > 0: all DC_ABOUND_SLOTS replicas (inference), *_ld floats apart. */
int dc_upsample2x_drop_fwd(const float* in, float* out, long out_ld, const uint8_t* mask, float keep, uint64_t seed,
                           const float* abound_in, long abound_in_ld, float* abound_out, long abound_out_ld,
                           int N, int H, int W, int C, dc_stream_t stream);
/* din[N,H,W,C] (dense) = sum over each 2x2 block of dout (pixel stride dout_ld) * dropout factor */
int dc_upsample2x_drop_bwd(const float* dout, long dout_ld, const uint8_t* mask, float keep, uint64_t seed, float* din,
                           int N, int H, int W, int C, dc_stream_t stream);

/* ---- Conv2D(2,1,softmax) + Lambda(x[...,-1]) + loss + metrics  :221-222,:372-380,:398-399 ------------------
 * p = softmax(a.Kh + bh)[...,1].  y != NULL: also partial[blocks][DC_HEAD_SUMS = 12] =
 *   {bce_sum, sum round(p)*y, sum round(p), sum clip(y-round(p),0,1), sum y, sum y*p, sum p*p, sum y*y,
 *    sum p, weighted_bce_sum, 0, 0}
 * (Keras clip/logit BCE form; round half to even; weighted BCE of utils/neurons.py:13-29).
 * blocks = dc_head_blocks(pixels); reduce with dc_reduce_partials_f64(partial, blocks, 12, sums). */
int dc_head_blocks(long pixels);
int dc_head_fwd(const float* a, const float* kh, const float* bh, const uint8_t* y, float* p, float* partial,
                long pixels, int C, dc_stream_t stream);
/* da = dlogits.Kh^T; partial[blocks][C+4] = (sum a[c]*s ..., sum s, pad) with s = dL/dlogit1 = -dL/dlogit0.
 * loss_kind: 0 binary_crossentropy (mean), 1 weighted_binary_crossentropy (mean), 2 dice_loss, 3 dicesq_loss
 * (unet_2d_summary.py:372-377); kinds 2,3 read the forward's reduced sums (device double[12]). */
int dc_head_bwd(const float* a, const float* p, const uint8_t* y, const float* kh, float* da, float* partial,
                int loss_kind, const double* sums, long pixels, int C, dc_stream_t stream);
/* head gradient from partial: dkh[c][0] = -S_c, dkh[c][1] = S_c, dbh = (-S, S) */
/* Training with a per-pixel loss (loss_kind 0 binary_crossentropy / 1 weighted_binary_crossentropy): dc_head_fwd and
 * dc_head_bwd(_bnin)(_bnred) in ONE pass over the head's input -- p, the 12 metric partials, da, the head's
 * weight-gradient partials (grad_partial: [blocks][C+4], for dc_head_grad_finalize) and, when bn_partial is given, the
 * producing BatchNorm layer's backward sums ([blocks][C][2]; needs in_scale / in_shift / bn_mean / bn_invstd).
 * in_scale / in_shift NULL: `a` is a materialised activation.  The dice losses need the global sums before their
 * backward: DC_EUNSUP. */
int dc_head_fwd_bwd(const float* a, const float* in_scale, const float* in_shift, const float* kh, const float* bh,
                    const uint8_t* y, float* p, float* partial, float* da, float* grad_partial, int loss_kind,
                    const float* bn_mean, const float* bn_invstd, float* bn_partial, float* amax_partial, long pixels,
                    int C, dc_stream_t stream);
int dc_head_grad_finalize(const float* partial, int blocks, int C, float* dkh, float* dbh, dc_stream_t stream);

/* ---- generic deterministic reductions ------------------------------------------
 * out[l] = scale * sum_p in[p*L + l]  (double accumulation, fixed order). tmp: float[32*L] scratch. */
int dc_reduce_partials(const float* in, int P, long L, float scale, float* out, float* tmp, dc_stream_t stream);
int dc_reduce_partials_f64(const float* in, int P, int L, double* out, dc_stream_t stream);
/* out[c] = sum_p in[(p*C + c)*2 + j] -> dbeta (j=0), dgamma (j=1) */
int dc_bn_bwd_finalize(const float* partial, int P, int C, float* dgamma, float* dbeta, dc_stream_t stream);

/* ---- Adam(0.002), Keras 2.0.6 form  :335 ------------------------------------------
 * g' = g*gscale; m = b1*m+(1-b1)*g'; v = b2*v+(1-b2)*g'^2; p -= lr_t*m/(sqrt(v)+eps), flat over n floats. */
int dc_adam_step_flat(float* p, const float* g, float* m, float* v, long n, float lr_t, float b1, float b2,
                      float eps, float gscale, dc_stream_t stream);

/* ---- 8x test-time augmentation of predict()  unet_2d_summary.py:585-595, utils/neurons.py:112-137 ---------------
 * The augmentations are pixel permutations; maps / invmaps (int32 [K][H*W]) are made on the host by applying the table's
 * own functions to an index image.  dc_gather_maps: out[k][p] = in[maps[k][p]] (n = H*W elements per copy).
 * dc_tta_merge: m = sum_k preds[k][invmaps[k][p]] / K in double, table order (numpy's float64 accumulation), cropped to
 * hs x ws; mask (uint8 [hs][ws]) = m > threshold; mean_out (nullable float [hs][ws]) = m. */
int dc_gather_maps(const float* in, const int* maps, float* out, int K, long n, dc_stream_t stream);
int dc_tta_merge(const float* preds, const int* invmaps, int K, int H, int W, int hs, int ws, double threshold,
                 uint8_t* mask, float* mean_out, dc_stream_t stream);

/* ---- training-batch assembly on the device: UNet2DSummary._batch_gen  unet_2d_summary.py:434-530 -----------------
 * S (float32) / M (uint8): the normalised summary images / flattened masks of ALL datasets, concatenated into one device
 * buffer each (src_elems elements; same layout for both), uploaded once per fit().  The host draws the reference's random
 * stream (:479-527) and describes item b by four longs
 *   items_host[b] = { element offset of the crop origin (y0, x0) in S / M, row stride of that image, eh << 32 | ew, d4 }:
 * the eh x ew crop is zero-filled to the (hw, hw) window (:505-521) and permuted by the composition of the drawn flips /
 * rot90s (:524-527), one of 8 dihedral maps:  out[i][j] = win[r][c], (r, c) = (d4 & 1) ? (j, i) : (i, j),
 * r = (d4 & 2) ? hw-1-r : r, c = (d4 & 4) ? hw-1-c : c.  x: float32 [B][hw][hw], y: uint8 [B][hw][hw].
 * items_host is the ONE HOST pointer of this ABI: the items are validated against src_elems and travel in the kernel
 * argument block (chunks of DC_CROP_MAX_ITEMS) -- nothing is staged, copied asynchronously or retained. */
#define DC_CROP_MAX_ITEMS 64
int dc_crop_augment(const float* S, const uint8_t* M, long src_elems, const long* items_host, int B, int hw,
                    float* x, uint8_t* y, dc_stream_t stream);
/* out[n][r][c] = p[n][y0 + r][x0 + c] > 0.5 (uint8): numpy's `mp[y0:y1, x0:x1].round()` of _ValidationMetricsCB
 * (unet_2d_summary.py:90-91; half-to-even: 0.5 -> 0) on N probability maps of H x W. */
int dc_round_window_u8(const float* p, int N, int H, int W, int y0, int y1, int x0, int x1, uint8_t* out,
                       dc_stream_t stream);

/* ---- host-side Neurofinder scoring of the validation callback  unet_2d_summary.py:90-91 -> datasets/nf.py:153-174 -----
 * HOST pointers, no device work, no stream: the per-mask scoring _ValidationMetricsCB runs 6 x n times per epoch, as one native
 * call (the GIL is released around it, so scoring threads run beside the validation forwards).  truth / pred: uint8 [H][W]
 * (non-zero = foreground).  8-connected components numbered in raster order of their first pixel (skimage.measure.label's
 * default on 2-D input); neurofinder.centers with `threshold` (5 in the reference's nf_mask_metrics) and neurofinder.shapes
 * (threshold inf) restated as in deep_calcium_amd/nf_metrics.py, bit-identical to it (tests/test_nf_matching.py).
 * counts[4] = { regions in truth, regions in pred, centre hits, matched pairs np }; inc[k] / exc[k] (k < np <= cap) =
 * |A & B| / |A| and |A & B| / |B| of the k-th matched pair in the order of the truth's regions (the caller takes the means:
 * recall = hits / counts[0], precision = hits / counts[1]).
 * ws: dc_host_nf_ws_bytes(H, W) bytes of HOST scratch owned by the caller (one per scoring thread; nothing is allocated). */
long dc_host_nf_ws_bytes(int H, int W);
int dc_host_nf_pairs(const uint8_t* truth, const uint8_t* pred, int H, int W, double threshold, int* counts,
                     double* inc, double* exc, int cap, void* ws);
int dc_host_label8(const uint8_t* mask, int H, int W, int* labels, int* n, void* ws);

/* misc */
int dc_fill(float* p, long n, float value, dc_stream_t stream);
/* p[i] *= s: the 1/G of the BatchNorm moving-statistics average over G data-parallel ranks (moving stats are
 * "averaged ... at checkpoint"; the sum itself is the RCCL all-reduce) */
int dc_scale_flat(float* p, long n, float s, dc_stream_t stream);
/* timing helper: run `fn`-independent HIP event timing is done by the caller via hipEvent* through ctypes:
 * these wrap hipEventCreate/Record/Synchronize/ElapsedTime on the given stream (so bench.py measures on the
 * stream the kernels are launched on). */
int dc_event_create(void** ev);
/* measurement hook (bench.py's per-kernel timer): the NEXT conv3x3 weight-gradient / joint-backward entry point called from THIS host
 * thread records ev_before right in front of its matrix kernel and ev_after right behind it, on the launch stream -- i.e. without
 * the slab-reduction launches that follow inside the same entry point (what rocprofv3 reports for that kernel symbol).  One shot. */
int dc_bracket_next_launch(void* ev_before, void* ev_after);
/* the kernel symbol (as rocprofv3 prints it, template arguments included) a conv3x3 weight-gradient launch of this shape runs:
 * dzin != 0 for dc_conv3x3_wgrad_dzin_f16x3, else dc_conv3x3_wgrad_f16x3 / _bnin.  Routing only, no launch. */
const char* dc_conv3x3_wgrad_kernel_name(int N, int H, int W, int Cin, int Cout, int dzin);
/* an event for stream ordering on ONE device only (hipEventDisableTiming | hipEventDisableSystemFence): what the engine's two-stream
 * backward hands between its streams.  Not for host synchronisation (no system-scope fence at the marker). */
int dc_event_create_sync(void** ev);
/* the same WITH the marker's system-scope fence (hipEventDisableTiming only): for the hand-offs whose consumer is not a queue of
 * this device -- the events in front of and behind a collective (a peer GPU reads / writes the buffer over xGMI). */
int dc_event_create_fenced(void** ev);
/* hipStreamWaitEvent(stream, ev): work queued on `stream` after this call waits for the work `ev` was last recorded behind */
int dc_stream_wait_event(dc_stream_t stream, void* ev);
int dc_event_record(void* ev, dc_stream_t stream);
int dc_event_elapsed_ms(void* start, void* stop, float* ms);
int dc_event_destroy(void* ev);

/* ---- collectives: the data-parallel gradient exchange (SURVEY 8b "call librccl.so ... directly via ctypes or a 3-function
 * dc_comm_* wrapper", 8e: one ncclAllReduce(sum, fp32) over the flat gradient buffer, optionally in buckets) ---------------
 * No reference counterpart: the reference trains on ONE device (README.md:23 pins CUDA_VISIBLE_DEVICES="0"; its step is the
 * model.fit_generator -> train_on_batch of unet_2d_summary.py:429-430); what is exchanged here is that step's flat gradient.
 * RCCL (librccl.so.1) is resolved with dlopen at the first dc_comm_* call -- the copy the process already holds (PyTorch-ROCm's)
 * when there is one; every other entry point works without it (DC_EUNSUP + message from these when it cannot be loaded).
 *   dc_comm_unique_id: HOST buffer of DC_COMM_ID_BYTES, filled by ONE rank and handed to the others out of band (the binding
 *     sends it through the torch.distributed store that rendezvoused the ranks);
 *   dc_comm_init_rank: collective over the nranks callers (one per GPU; binds the caller's current device), returns the handle;
 *   dc_comm_all_reduce_sum (fp32: the flat gradient) / _f64 (the step's loss / metric sums): buf[0..n) <- sum over ranks, in place,
 *     asynchronous on `stream` (a launch like any other: it can be recorded on a tape -- the backward of a data-parallel step
 *     replays as ONE tape, collectives included);
 *   dc_comm_group_start / _end: ncclGroupStart / ncclGroupEnd around several of them.
 * The caller orders the buffer's producers in front of the call with events made by dc_event_create_fenced. */
#define DC_COMM_ID_BYTES 128
int dc_comm_unique_id(void* id_host);
int dc_comm_init_rank(void** comm, const void* id_host, int nranks, int rank);
int dc_comm_all_reduce_sum(void* comm, float* buf, long n, dc_stream_t stream);
int dc_comm_all_reduce_sum_f64(void* comm, double* buf, long n, dc_stream_t stream);
int dc_comm_group_start(void);
int dc_comm_group_end(void);
int dc_comm_destroy(void* comm);

/* ---- launch tape: a step's enqueue sequence, recorded once, replayed from C --------------------------------------------
 * The reference's train loop calls train_on_batch once per step (model.fit_generator, unet_2d_summary.py:429-430); here a
 * step is ~230 asynchronous launches whose sequence is FIXED once the engine is warm -- same entry points, pointers and
 * shapes; only a few scalars (dropout seeds, Adam's lr_t, the batch pointers) change.  A tape holds such a sequence as
 * (entry point name, arguments) and replays it with ONE call: every operation goes through its entry point unchanged
 * (argument checks, status codes); `dc_tape_patch` marks the arguments that take a per-replay value.  HOST memory only; the
 * tape retains the recorded pointer VALUES (the caller keeps those buffers alive and in place, as for any launch).
 *   dc_tape_append: fn = the name of an int-returning launch entry point of this header (not: size / routing queries,
 *     dc_crop_augment and dc_host_* [host data read at call time], the event create / elapsed / destroy helpers);
 *     args8 = its arguments as 8-byte slots in declaration order: integers and pointers as longs, float / double
 *     arguments as the bit pattern of a double.
 *   dc_tape_patch(op, arg, slot): argument `arg` of operation `op` is values[slot] at replay (call in operation order).
 *   dc_tape_replay(first, count, values, nvalues): operations [first, first + count); stops at the first failing one. */
int dc_tape_create(void** tape);
int dc_tape_destroy(void* tape);
int dc_tape_append(void* tape, const char* fn, const long* args8, int nargs);
int dc_tape_patch(void* tape, int op, int arg, int slot);
int dc_tape_replay(void* tape, int first, int count, const long* values, int nvalues);
int dc_tape_len(void* tape);

#ifdef __cplusplus
}
#endif
#endif

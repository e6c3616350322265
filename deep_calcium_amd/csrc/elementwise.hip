// HBM-bound kernels of the UNet2DS path: BatchNorm statistics / apply / backward, ReLU, dropout,
// 2x2 max-pool (bit-exact first-max argmax), softmax head + Keras-form BCE + metric sums, Adam, reductions.
// Reference call sites: /root/reference/deepcalcium/models/neurons/unet_2d_summary.py:158-159,:166-167
// (BatchNormalization+relu), :176-216 (MaxPooling2D, Dropout), :221-222 (head), :398-399 (loss+metrics),
// :335 (Adam); op semantics SURVEY.md Appendix A.
// All of them are float4 (16 B/lane) streaming kernels with the channel quad fixed per thread
// (256 % (C/4) == 0), so per-channel parameters live in registers and per-channel sums need no atomics:
// partials per block are combined by a fixed-order second stage (bit-reproducible).
#include "common.h"

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// grid cap of the grid-stride BatchNorm passes: 1 280 = ONE resident round of the backward-apply kernel (5 workgroups per
// CU at its 90 VGPRs) -- no second, partly filled round beside the weight-gradient stream (2 048: -1.7 % end to end,
// 1 024: -0.8 %, 4 096: -3.9 %)
static const int kMaxBlocks = 1280;
// the BatchNorm-backward passes (reduce / apply) run BESIDE the weight-gradient kernel of the side stream: the same cap
// (measured flat between 768 and 1280 workgroups, -0.7 % at 512 and at 1792)
static const int kMaxBwdBlocks = kMaxBlocks;
__device__ __forceinline__ void bn_affine4(const f32x4& mu, const f32x4& is, const f32x4& ga, const f32x4& be, f32x4& sc,
                                           f32x4& sh) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float a, b;
    dc_bn_affine(mu[e], is[e], ga[e], be[e], a, b);
    sc[e] = a; sh[e] = b;
  }
}
__device__ __forceinline__ f32x4 fma4(const f32x4& a, const f32x4& b, const f32x4& c) {
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf(a[e], b[e], c[e]);
  return r;
}

static int chan_check(const char* fn, int C) {
  DC_REQUIRE(C >= 4 && C <= 1024 && dc_is_pow2(C), DC_EUNSUP, "%s: C=%d must be a power of two in [4,1024]", fn, C);
  return DC_OK;
}
static int ew_blocks(long pixels, int C, int cap = kMaxBlocks) {
  const int PPB = 256 / (C / 4);
  long b = (pixels + PPB - 1) / PPB;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}
static int ew_bwd_blocks(long pixels, int C) { return ew_blocks(pixels, C, kMaxBwdBlocks); }

// ------------------------------------------------------------------------------------------------
// BN statistics finalize: one block per channel, double accumulation over all partials.
// abound (nullable): |gamma|*sqrt(count) + |beta| >= max |gamma*xhat + beta| for ANY data (|xhat| <= sqrt(count-1)): the
// magnitude bound the f16x3 consumers' fp16 range guard reads (common.h dc_block_guard_scale).
__device__ __forceinline__ float bn_abound(float gamma, float beta, double count, float inv_keep) {
  return (fabsf(gamma) * (float)sqrt(count) + fabsf(beta)) * inv_keep;
}
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const double* __restrict__ partial, int parts, int groups,
                                                               int C, double count, float eps, float momentum,
                                                               float* mean, float* invstd, float* mmean, float* mvar,
                                                               const float* gamma, const float* beta, float* scale,
                                                               float* shift, float* abound) {
  __shared__ double sh1[256], sh2[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  const int Ct = groups * C;
  double s1 = 0.0, s2 = 0.0;
  for (int i = tid; i < parts * groups; i += 256) {
    const int pt = i / groups, g = i - pt * groups;
    const double* src = partial + ((long)pt * Ct + g * C + c) * 2;
    s1 += src[0];
    s2 += src[1];
  }
  sh1[tid] = s1; sh2[tid] = s2;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { sh1[tid] += sh1[tid + s]; sh2[tid] += sh2[tid + s]; }
    __syncthreads();
  }
  if (tid == 0) {
    const double mu = sh1[0] / count;
    double var = sh2[0] / count - mu * mu;  // population (biased) variance, Keras 2.0.6 / tf.nn.moments
    if (var < 0.0) var = 0.0;
    const float muf = (float)mu, isf = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = muf;
    invstd[c] = isf;
    if (scale) dc_bn_affine(muf, isf, gamma[c], beta[c], scale[c], shift[c]);
    if (abound) abound[c] = bn_abound(gamma[c], beta[c], count, 1.f);
    if (momentum >= 0.f && mmean && mvar) {
      mmean[c] = (float)((double)mmean[c] * momentum + mu * (1.0 - (double)momentum));
      mvar[c] = (float)((double)mvar[c] * momentum + var * (1.0 - (double)momentum));
    }
  }
}

extern "C" int dc_bn_stats_finalize(const double* partial, int parts, int groups, int C, double count, float eps,
                                    float momentum, float* mean, float* invstd, float* moving_mean, float* moving_var,
                                    dc_stream_t stream) {
  DC_REQUIRE(partial && mean && invstd, DC_EINVAL, "dc_bn_stats_finalize: null pointer");
  DC_REQUIRE(parts > 0 && groups > 0 && C > 0 && count > 0, DC_EINVAL, "dc_bn_stats_finalize: bad sizes");
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, parts, groups, C,
                     count, eps, momentum, mean, invstd, moving_mean, moving_var, (const float*)nullptr,
                     (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr);
  DC_CHECK_LAUNCH("dc_bn_stats_finalize");
  return DC_OK;
}

extern "C" int dc_bn_stats_finalize_affine(const double* partial, int parts, int groups, int C, double count, float eps,
                                           float momentum, float* mean, float* invstd, float* moving_mean,
                                           float* moving_var, const float* gamma, const float* beta, float* scale,
                                           float* shift, float* abound, dc_stream_t stream) {
  DC_REQUIRE(partial && mean && invstd && gamma && beta && scale && shift, DC_EINVAL,
             "dc_bn_stats_finalize_affine: null pointer");
  DC_REQUIRE(parts > 0 && groups > 0 && C > 0 && count > 0, DC_EINVAL, "dc_bn_stats_finalize_affine: bad sizes");
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, parts, groups, C,
                     count, eps, momentum, mean, invstd, moving_mean, moving_var, gamma, beta, scale, shift, abound);
  DC_CHECK_LAUNCH("dc_bn_stats_finalize_affine");
  return DC_OK;
}

// ---- partial rows folded parts -> chunks, in a fixed order.  The finalize launches above read one 16-byte piece per (row, channel),
// 16 * groups * C bytes apart: with the 16 384 tile rows of a 512^2 up-convolution that is 77 us of quarter-used sectors on the
// step's critical path.  Here a block reads WHOLE rows (coalesced) of its chunk of rows; the finalize then reads `chunks` rows.
__global__ __launch_bounds__(256) void bn_stats_rows_fold_kernel(const double* __restrict__ partial, int parts, int width,
                                                                int chunks, double* __restrict__ out) {
  const int b = blockIdx.x, col = blockIdx.y * 256 + threadIdx.x;
  if (col >= width) return;
  const int r0 = (int)((long)parts * b / chunks), r1 = (int)((long)parts * (b + 1) / chunks);
  const double* src = partial + (long)r0 * width + col;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int r = r0;
  for (; r + 4 <= r1; r += 4, src += 4L * width) {
    s0 += src[0];
    s1 += src[width];
    s2 += src[2L * width];
    s3 += src[3L * width];
  }
  for (; r < r1; ++r, src += width) s0 += src[0];
  out[(long)b * width + col] = (s0 + s1) + (s2 + s3);
}
extern "C" int dc_bn_stats_rows_fold(const double* partial, int parts, int groups, int C, int chunks, double* out,
                                     dc_stream_t stream) {
  DC_REQUIRE(partial && out && partial != out, DC_EINVAL, "dc_bn_stats_rows_fold: null or aliased pointer");
  DC_REQUIRE(parts > 0 && groups > 0 && C > 0 && chunks > 0 && chunks <= parts, DC_EINVAL,
             "dc_bn_stats_rows_fold: bad sizes (parts %d, groups %d, C %d, chunks %d)", parts, groups, C, chunks);
  const int width = groups * C * 2;
  hipLaunchKernelGGL(bn_stats_rows_fold_kernel, dim3(chunks, dc_cdiv(width, 256)), dim3(256), 0, (hipStream_t)stream, partial,
                     parts, width, chunks, out);
  DC_CHECK_LAUNCH("dc_bn_stats_rows_fold");
  return DC_OK;
}

// ---- synchronised BatchNorm (data-parallel 'sync' mode): the per-channel sums leave the device-local finalize so that
// they can be all-reduced over the ranks: reduce (partials -> double sums) | all-reduce | finalize from sums.
__global__ __launch_bounds__(256) void bn_stats_reduce_kernel(const double* __restrict__ partial, int parts, int groups,
                                                             int C, double* __restrict__ sums) {
  __shared__ double sh1[256], sh2[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  const int Ct = groups * C;
  double s1 = 0.0, s2 = 0.0;
  for (int i = tid; i < parts * groups; i += 256) {
    const int pt = i / groups, g = i - pt * groups;
    const double* src = partial + ((long)pt * Ct + g * C + c) * 2;
    s1 += src[0];
    s2 += src[1];
  }
  sh1[tid] = s1; sh2[tid] = s2;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { sh1[tid] += sh1[tid + s]; sh2[tid] += sh2[tid + s]; }
    __syncthreads();
  }
  if (tid == 0) { sums[2 * c] = sh1[0]; sums[2 * c + 1] = sh2[0]; }
}
extern "C" int dc_bn_stats_reduce(const double* partial, int parts, int groups, int C, double* sums, dc_stream_t stream) {
  DC_REQUIRE(partial && sums && parts > 0 && groups > 0 && C > 0, DC_EINVAL, "dc_bn_stats_reduce: bad arguments");
  hipLaunchKernelGGL(bn_stats_reduce_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, parts, groups, C, sums);
  DC_CHECK_LAUNCH("dc_bn_stats_reduce");
  return DC_OK;
}
__global__ void bn_stats_finalize_sums_kernel(const double* __restrict__ sums, int C, double count, float eps,
                                              float momentum, float* mean, float* invstd, float* mmean, float* mvar,
                                              const float* gamma, const float* beta, float* scale, float* shift,
                                              float* abound) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mu = sums[2 * c] / count;
  double var = sums[2 * c + 1] / count - mu * mu;
  if (var < 0.0) var = 0.0;
  const float muf = (float)mu, isf = (float)(1.0 / sqrt(var + (double)eps));
  mean[c] = muf;
  invstd[c] = isf;
  if (scale) dc_bn_affine(muf, isf, gamma[c], beta[c], scale[c], shift[c]);
  if (abound) abound[c] = bn_abound(gamma[c], beta[c], count, 1.f);
  if (momentum >= 0.f && mmean && mvar) {
    mmean[c] = (float)((double)mmean[c] * momentum + mu * (1.0 - (double)momentum));
    mvar[c] = (float)((double)mvar[c] * momentum + var * (1.0 - (double)momentum));
  }
}
extern "C" int dc_bn_stats_finalize_sums(const double* sums, int C, double count, float eps, float momentum, float* mean,
                                         float* invstd, float* moving_mean, float* moving_var, const float* gamma,
                                         const float* beta, float* scale, float* shift, float* abound,
                                         dc_stream_t stream) {
  DC_REQUIRE(sums && mean && invstd && C > 0 && count > 0, DC_EINVAL, "dc_bn_stats_finalize_sums: bad arguments");
  DC_REQUIRE(!scale || (gamma && beta && shift), DC_EINVAL, "dc_bn_stats_finalize_sums: scale needs gamma, beta and shift");
  DC_REQUIRE(!abound || (gamma && beta), DC_EINVAL, "dc_bn_stats_finalize_sums: abound needs gamma and beta");
  hipLaunchKernelGGL(bn_stats_finalize_sums_kernel, dim3(dc_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, C,
                     count, eps, momentum, mean, invstd, moving_mean, moving_var, gamma, beta, scale, shift, abound);
  DC_CHECK_LAUNCH("dc_bn_stats_finalize_sums");
  return DC_OK;
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* mmean, const float* mvar,
                               const float* bias, float eps, float* scale, float* shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float s = gamma[c] / sqrtf(mvar[c] + eps);
    scale[c] = s;
    shift[c] = beta[c] + ((bias ? bias[c] : 0.f) - mmean[c]) * s;
  }
}
extern "C" int dc_bn_fold(const float* gamma, const float* beta, const float* mmean, const float* mvar,
                          const float* bias, float eps, float* scale, float* shift, int C, dc_stream_t stream) {
  DC_REQUIRE(gamma && beta && mmean && mvar && scale && shift && C > 0, DC_EINVAL, "dc_bn_fold: bad arguments");
  hipLaunchKernelGGL(bn_fold_kernel, dim3(dc_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, mmean,
                     mvar, bias, eps, scale, shift, C);
  DC_CHECK_LAUNCH("dc_bn_fold");
  return DC_OK;
}

// ------------------------------------------------------------------------------------------------
struct BnParams {
  const float* da; long da_ld;
  const float* z;
  const float* mean; const float* invstd; const float* gamma; const float* beta;
  const uint8_t* mask; float keep; uint64_t seed;
  const float* dgamma; const float* dbeta;
  float* out; long out_ld;
  float* partial;
  float* absmax;   // bn_bwd_apply: per-block max |dz| (nullable)
  float* amax_c;   // pass-1 producers: per-(block, channel) max |dy| [blocks][C] (nullable) -- dc_bn_bwd_finalize_dzin
  long pixels; int C;
  double count;   // elements per channel the statistics were taken over (> pixels in 'sync' data parallelism)
};

// dropout keep-factor (0 or 1/keep) for the 4 channels of element quad `e4` (element index = pix*C + c)
__device__ __forceinline__ f32x4 drop_factor(const BnParams& p, long elem, float inv_keep) {
  f32x4 f;
  if (p.mask) {
    const uchar4 m = *reinterpret_cast<const uchar4*>(p.mask + elem);
    f[0] = m.x ? inv_keep : 0.f; f[1] = m.y ? inv_keep : 0.f; f[2] = m.z ? inv_keep : 0.f; f[3] = m.w ? inv_keep : 0.f;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = dc_keep_factor(p.seed, (uint64_t)(elem + e), p.keep, inv_keep);
  }
  return f;
}

__global__ __launch_bounds__(256) void bn_relu_drop_fwd_kernel(BnParams p) {
  const int C4 = p.C >> 2, PPB = 256 / C4;
  const int tid = threadIdx.x, q = tid % C4, pl = tid / C4;
  const f32x4 mu = ld4(p.mean + 4 * q), is = ld4(p.invstd + 4 * q), ga = ld4(p.gamma + 4 * q), be = ld4(p.beta + 4 * q);
  const bool drop = p.keep < 1.f;
  const float inv_keep = drop ? 1.f / p.keep : 1.f;
  f32x4 sc, sh;
  bn_affine4(mu, is, ga, be, sc, sh);
  if (p.partial && blockIdx.x == 0 && pl == 0) {       // p.partial doubles as the range-guard bound output here
#pragma unroll
    for (int e = 0; e < 4; ++e) p.partial[4 * q + e] = bn_abound(ga[e], be[e], p.count, inv_keep);
  }
  for (long pix = (long)blockIdx.x * PPB + pl; pix < p.pixels; pix += (long)gridDim.x * PPB) {
    const long elem = pix * p.C + 4 * q;
    const f32x4 z = ld4(p.z + elem);
    f32x4 y = fma4(z, sc, sh);
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
    if (drop) y *= drop_factor(p, elem, inv_keep);
    st4(p.out + pix * p.out_ld + 4 * q, y);
  }
}

extern "C" int dc_bn_relu_drop_fwd(const float* z, const float* mean, const float* invstd, const float* gamma,
                                   const float* beta, const uint8_t* mask, float keep, uint64_t seed, float* out,
                                   long out_ld, long pixels, int C, double count, float* abound, dc_stream_t stream) {
  DC_REQUIRE(z && mean && invstd && gamma && beta && out, DC_EINVAL, "dc_bn_relu_drop_fwd: null pointer");
  DC_REQUIRE(pixels > 0 && out_ld >= C && out_ld % 4 == 0 && keep > 0.f, DC_EINVAL, "dc_bn_relu_drop_fwd: bad sizes");
  int rc = chan_check("dc_bn_relu_drop_fwd", C);
  if (rc) return rc;
  BnParams p{};
  p.z = z; p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.beta = beta; p.mask = mask; p.keep = keep;
  p.seed = seed; p.out = out; p.out_ld = out_ld; p.pixels = pixels; p.C = C;
  p.partial = abound; p.count = count > 0 ? count : (double)pixels;
  hipLaunchKernelGGL(bn_relu_drop_fwd_kernel, dim3(ew_blocks(pixels, C)), dim3(256), 0, (hipStream_t)stream, p);
  DC_CHECK_LAUNCH("dc_bn_relu_drop_fwd");
  return DC_OK;
}

// BN + ReLU + dropout of a block that feeds MaxPooling2D, AND the pooling, in one pass: a thread takes one 2x2 window of
// 4 channels, forms the four activations exactly as bn_relu_drop_fwd_kernel does (same expressions, same dropout
// element indices), writes them (they are also the skip connection) and the first-max + argmax of the window exactly as
// maxpool_fwd_kernel does.  Saves re-reading the activation: 1 of the 4.3 tensor passes of the two separate kernels.
__global__ __launch_bounds__(256) void bn_relu_drop_pool_fwd_kernel(BnParams p, float* __restrict__ pooled,
                                                                    uint8_t* __restrict__ idx, int N, int H, int W) {
  const int C = p.C, C4 = C >> 2, h2 = H >> 1, w2 = W >> 1;
  const bool drop = p.keep < 1.f;
  const float inv_keep = drop ? 1.f / p.keep : 1.f;
  const long total = (long)N * h2 * w2 * C4;
  if (p.partial && blockIdx.x == 0 && threadIdx.x < C4) {
    const int q = threadIdx.x;
    const f32x4 ga = ld4(p.gamma + 4 * q), be = ld4(p.beta + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) p.partial[4 * q + e] = bn_abound(ga[e], be[e], p.count, inv_keep);
  }
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const int q = (int)(i % C4);
    long r = i / C4;
    const int x = (int)(r % w2); r /= w2;
    const int y = (int)(r % h2);
    const long n = r / h2;
    const f32x4 mu = ld4(p.mean + 4 * q), is = ld4(p.invstd + 4 * q), ga = ld4(p.gamma + 4 * q), be = ld4(p.beta + 4 * q);
    f32x4 sc, sh;
    bn_affine4(mu, is, ga, be, sc, sh);
    const long pix00 = (n * H + 2 * y) * W + 2 * x;
    const long pix[4] = {pix00, pix00 + 1, pix00 + W, pix00 + W + 1};
    f32x4 z[4], a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) z[k] = ld4(p.z + pix[k] * C + 4 * q);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f32x4 v = fma4(z[k], sc, sh);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      if (drop) v *= drop_factor(p, pix[k] * C + 4 * q, inv_keep);
      a[k] = v;
      st4(p.out + pix[k] * p.out_ld + 4 * q, v);
    }
    f32x4 m = a[0];
    uchar4 kq = make_uchar4(0, 0, 0, 0);
    uint8_t* kk = reinterpret_cast<uint8_t*>(&kq);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (a[1][e] > m[e]) { m[e] = a[1][e]; kk[e] = 1; }
      if (a[2][e] > m[e]) { m[e] = a[2][e]; kk[e] = 2; }
      if (a[3][e] > m[e]) { m[e] = a[3][e]; kk[e] = 3; }
    }
    st4(pooled + i * 4, m);
    if (idx) *reinterpret_cast<uchar4*>(idx + i * 4) = kq;
  }
}

extern "C" int dc_bn_relu_drop_pool_fwd(const float* z, const float* mean, const float* invstd, const float* gamma,
                                        const float* beta, const uint8_t* mask, float keep, uint64_t seed, float* out,
                                        long out_ld, float* pooled, uint8_t* idx, int N, int H, int W, int C, double count,
                                        float* abound, dc_stream_t stream) {
  DC_REQUIRE(z && mean && invstd && gamma && beta && out && pooled, DC_EINVAL, "dc_bn_relu_drop_pool_fwd: null pointer");
  DC_REQUIRE(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && out_ld >= C && out_ld % 4 == 0 && keep > 0.f, DC_EINVAL,
             "dc_bn_relu_drop_pool_fwd: bad sizes");
  int rc = chan_check("dc_bn_relu_drop_pool_fwd", C);
  if (rc) return rc;
  BnParams p{};
  const long pixels = (long)N * H * W;
  p.z = z; p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.beta = beta; p.mask = mask; p.keep = keep;
  p.seed = seed; p.out = out; p.out_ld = out_ld; p.pixels = pixels; p.C = C;
  p.partial = abound; p.count = count > 0 ? count : (double)pixels;
  const long total = pixels / 4 * (C / 4);
  long blocks = (total + 255) / 256;
  if (blocks > kMaxBlocks) blocks = kMaxBlocks;
  hipLaunchKernelGGL(bn_relu_drop_pool_fwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, pooled, idx, N, H, W);
  DC_CHECK_LAUNCH("dc_bn_relu_drop_pool_fwd");
  return DC_OK;
}

// dy = da * [y > 0] * dropfactor ; xhat = (z - mean) * invstd
__device__ __forceinline__ void bn_bwd_math(const BnParams& p, long pix, int q, const f32x4& z, const f32x4& da,
                                            const f32x4& mu, const f32x4& is, const f32x4& sc, const f32x4& sh,
                                            bool drop, float inv_keep, f32x4& dy, f32x4& xh) {
  xh = (z - mu) * is;
  const f32x4 y = fma4(z, sc, sh);      // the forward's own expression: identical ReLU gate
  dy = da;
  if (drop) dy *= drop_factor(p, pix * p.C + 4 * q, inv_keep);
#pragma unroll
  for (int e = 0; e < 4; ++e) dy[e] = (y[e] > 0.f) ? dy[e] : 0.f;
}
__device__ __forceinline__ void bn_bwd_elem(const BnParams& p, long pix, int q, const f32x4& mu, const f32x4& is,
                                            const f32x4& sc, const f32x4& sh, bool drop, float inv_keep, f32x4& dy,
                                            f32x4& xh) {
  const f32x4 z = ld4(p.z + pix * p.C + 4 * q);
  const f32x4 da = ld4(p.da + pix * p.da_ld + 4 * q);
  bn_bwd_math(p, pix, q, z, da, mu, is, sc, sh, drop, inv_keep, dy, xh);
}

__device__ __forceinline__ f32x4 absmax4(const f32x4& m, const f32x4& v) {
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = fmaxf(m[e], fabsf(v[e]));
  return r;
}
__device__ __forceinline__ f32x4 max4(const f32x4& a, const f32x4& b) {
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = fmaxf(a[e], b[e]);
  return r;
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BnParams p) {
  __shared__ f32x4 sm1[256], sm2[256];
  const int C4 = p.C >> 2, PPB = 256 / C4;
  const int tid = threadIdx.x, q = tid % C4, pl = tid / C4;
  const f32x4 mu = ld4(p.mean + 4 * q), is = ld4(p.invstd + 4 * q), ga = ld4(p.gamma + 4 * q), be = ld4(p.beta + 4 * q);
  const bool drop = p.keep < 1.f;
  const float inv_keep = drop ? 1.f / p.keep : 1.f;
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, am = {0.f, 0.f, 0.f, 0.f};
  f32x4 sc, sh;
  bn_affine4(mu, is, ga, be, sc, sh);
  auto finish = [&](long pix, const f32x4& z, const f32x4& da) {
    f32x4 dy, xh;
    bn_bwd_math(p, pix, q, z, da, mu, is, sc, sh, drop, inv_keep, dy, xh);
    s1 += dy;
    s2 += dy * xh;
    am = absmax4(am, dy);
  };
  const long stride = (long)gridDim.x * PPB;
  long pix = (long)blockIdx.x * PPB + pl;
  for (; pix + stride < p.pixels; pix += 2 * stride) {
    const long pb = pix + stride;
    const f32x4 z0 = ld4(p.z + pix * p.C + 4 * q), d0 = ld4(p.da + pix * p.da_ld + 4 * q);
    const f32x4 z1 = ld4(p.z + pb * p.C + 4 * q), d1 = ld4(p.da + pb * p.da_ld + 4 * q);
    finish(pix, z0, d0);
    finish(pb, z1, d1);
  }
  if (pix < p.pixels) finish(pix, ld4(p.z + pix * p.C + 4 * q), ld4(p.da + pix * p.da_ld + 4 * q));
  sm1[tid] = s1; sm2[tid] = s2;
  __syncthreads();
  if (pl == 0) {
    for (int k = 1; k < PPB; ++k) { s1 += sm1[k * C4 + q]; s2 += sm2[k * C4 + q]; }
    float* dst = p.partial + ((long)blockIdx.x * p.C + 4 * q) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) { dst[2 * e] = s1[e]; dst[2 * e + 1] = s2[e]; }
  }
  if (p.amax_c) {
    __syncthreads();
    sm1[tid] = am;
    __syncthreads();
    if (pl == 0) {
      for (int k = 1; k < PPB; ++k) am = max4(am, sm1[k * C4 + q]);
      st4(p.amax_c + (long)blockIdx.x * p.C + 4 * q, am);
    }
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnParams p) {
  __shared__ f32x4 sm1[256];
  const int C4 = p.C >> 2, PPB = 256 / C4;
  const int tid = threadIdx.x, q = tid % C4, pl = tid / C4;
  const f32x4 mu = ld4(p.mean + 4 * q), is = ld4(p.invstd + 4 * q), ga = ld4(p.gamma + 4 * q), be = ld4(p.beta + 4 * q);
  const float invM = (float)(1.0 / p.count);
  const f32x4 mdy = ld4(p.dbeta + 4 * q) * invM, mdyx = ld4(p.dgamma + 4 * q) * invM;
  const f32x4 gs = ga * is;
  const bool drop = p.keep < 1.f;
  const float inv_keep = drop ? 1.f / p.keep : 1.f;
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f};
  float amax = 0.f;
  f32x4 sc, sh;
  bn_affine4(mu, is, ga, be, sc, sh);
  auto finish = [&](long pix, const f32x4& z, const f32x4& da) {
    f32x4 dy, xh;
    bn_bwd_math(p, pix, q, z, da, mu, is, sc, sh, drop, inv_keep, dy, xh);
    const f32x4 dz = gs * (dy - mdy - xh * mdyx);
    s1 += dz;
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(dz[0]), fabsf(dz[1])), fmaxf(fabsf(dz[2]), fabsf(dz[3]))));
    st4(p.out + pix * p.out_ld + 4 * q, dz);
  };
  // two pixels per iteration: four 16-byte loads in flight per lane before the first use
  const long stride = (long)gridDim.x * PPB;
  long pix = (long)blockIdx.x * PPB + pl;
  for (; pix + stride < p.pixels; pix += 2 * stride) {
    const long pb = pix + stride;
    const f32x4 z0 = ld4(p.z + pix * p.C + 4 * q), d0 = ld4(p.da + pix * p.da_ld + 4 * q);
    const f32x4 z1 = ld4(p.z + pb * p.C + 4 * q), d1 = ld4(p.da + pb * p.da_ld + 4 * q);
    finish(pix, z0, d0);
    finish(pb, z1, d1);
  }
  if (pix < p.pixels) finish(pix, ld4(p.z + pix * p.C + 4 * q), ld4(p.da + pix * p.da_ld + 4 * q));
  if (p.partial) {
    sm1[tid] = s1;
    __syncthreads();
    if (pl == 0) {
      for (int k = 1; k < PPB; ++k) s1 += sm1[k * C4 + q];
      st4(p.partial + (long)blockIdx.x * p.C + 4 * q, s1);
    }
  }
  if (p.absmax) {
    __shared__ float smax[4];
    for (int s = 32; s > 0; s >>= 1) amax = fmaxf(amax, __shfl_xor(amax, s));
    if ((tid & 63) == 0) smax[tid >> 6] = amax;
    __syncthreads();
    if (tid == 0) p.absmax[blockIdx.x] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  }
}

// scale = 2^floor(log2(target / max_i partial[i]))  (1 if the tensor is all zero): a power of two, so applying and
// undoing it is exact in fp32.
__global__ void pow2_scale_kernel(const float* __restrict__ partial, int n, float target, float* __restrict__ scale) {
  __shared__ float sm[256];
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, partial[i]);
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) scale[0] = dc_pow2_from_absmax(sm[0], target);
}
extern "C" int dc_pow2_scale_from_absmax(const float* partial, int n, float target, float* scale, dc_stream_t stream) {
  DC_REQUIRE(partial && scale && n > 0 && target > 0.f, DC_EINVAL, "dc_pow2_scale_from_absmax: bad arguments");
  hipLaunchKernelGGL(pow2_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n, target, scale);
  DC_CHECK_LAUNCH("dc_pow2_scale_from_absmax");
  return DC_OK;
}

// One launch for what follows bn_bwd_apply: block c < C sums column c of dbias_partial[blocks][C] (fixed order, double
// accumulation -> bit-reproducible); block C turns absmax_partial[blocks] into the power-of-two scale (nullable).
__global__ __launch_bounds__(256) void bn_bwd_apply_finalize_kernel(const float* __restrict__ dbias_partial,
                                                                   const float* __restrict__ absmax_partial, int blocks,
                                                                   int C, float target, float* __restrict__ dbias,
                                                                   float* __restrict__ scale) {
  __shared__ double smd[256];
  __shared__ float smf[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  if (c < C) {
    double s = 0.0;
    for (int i = tid; i < blocks; i += 256) s += (double)dbias_partial[(long)i * C + c];
    smd[tid] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
      if (tid < k) smd[tid] += smd[tid + k];
      __syncthreads();
    }
    if (tid == 0) dbias[c] = (float)smd[0];
    return;
  }
  float m = 0.f;
  for (int i = tid; i < blocks; i += 256) m = fmaxf(m, absmax_partial[i]);
  smf[tid] = m;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (tid < k) smf[tid] = fmaxf(smf[tid], smf[tid + k]);
    __syncthreads();
  }
  if (tid == 0) scale[0] = dc_pow2_from_absmax(smf[0], target);
}
extern "C" int dc_bn_bwd_apply_finalize(const float* dbias_partial, const float* absmax_partial, int blocks, int C,
                                        float target, float* dbias, float* scale, dc_stream_t stream) {
  DC_REQUIRE(dbias_partial && dbias && blocks > 0 && C > 0, DC_EINVAL, "dc_bn_bwd_apply_finalize: bad arguments");
  DC_REQUIRE((absmax_partial == nullptr) == (scale == nullptr) && (!scale || target > 0.f), DC_EINVAL,
             "dc_bn_bwd_apply_finalize: absmax_partial, scale and target go together");
  hipLaunchKernelGGL(bn_bwd_apply_finalize_kernel, dim3(C + (scale ? 1 : 0)), dim3(256), 0, (hipStream_t)stream,
                     dbias_partial, absmax_partial, blocks, C, target, dbias, scale);
  DC_CHECK_LAUNCH("dc_bn_bwd_apply_finalize");
  return DC_OK;
}

extern "C" int dc_bn_bwd_blocks(long pixels, int C) { return ew_bwd_blocks(pixels, C); }

extern "C" int dc_bn_bwd_reduce(const float* da, long da_ld, const float* z, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, const uint8_t* mask, float keep, uint64_t seed,
                                float* partial, float* amax_partial, long pixels, int C, dc_stream_t stream) {
  DC_REQUIRE(da && z && mean && invstd && gamma && beta && partial, DC_EINVAL, "dc_bn_bwd_reduce: null pointer");
  DC_REQUIRE(pixels > 0 && da_ld >= C && da_ld % 4 == 0 && keep > 0.f, DC_EINVAL, "dc_bn_bwd_reduce: bad sizes");
  int rc = chan_check("dc_bn_bwd_reduce", C);
  if (rc) return rc;
  BnParams p{};
  p.da = da; p.da_ld = da_ld; p.z = z; p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.beta = beta;
  p.mask = mask; p.keep = keep; p.seed = seed; p.partial = partial; p.amax_c = amax_partial; p.pixels = pixels; p.C = C;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(ew_bwd_blocks(pixels, C)), dim3(256), 0, (hipStream_t)stream, p);
  DC_CHECK_LAUNCH("dc_bn_bwd_reduce");
  return DC_OK;
}

static int bn_bwd_apply_impl(const float* da, long da_ld, const float* z, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, const uint8_t* mask, float keep, uint64_t seed,
                             const float* dgamma, const float* dbeta, float* dz, float* dbias_partial,
                             float* absmax_partial, long pixels, double count, int C, dc_stream_t stream);
extern "C" int dc_bn_bwd_apply(const float* da, long da_ld, const float* z, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, const uint8_t* mask, float keep, uint64_t seed,
                               const float* dgamma, const float* dbeta, float* dz, float* dbias_partial,
                               float* absmax_partial, long pixels, int C, dc_stream_t stream) {
  return bn_bwd_apply_impl(da, da_ld, z, mean, invstd, gamma, beta, mask, keep, seed, dgamma, dbeta, dz, dbias_partial,
                           absmax_partial, pixels, (double)pixels, C, stream);
}
extern "C" int dc_bn_bwd_apply_count(const float* da, long da_ld, const float* z, const float* mean, const float* invstd,
                                     const float* gamma, const float* beta, const uint8_t* mask, float keep, uint64_t seed,
                                     const float* dgamma, const float* dbeta, float* dz, float* dbias_partial,
                                     float* absmax_partial, long pixels, double count, int C, dc_stream_t stream) {
  DC_REQUIRE(count >= (double)pixels, DC_EINVAL, "dc_bn_bwd_apply_count: count < pixels");
  return bn_bwd_apply_impl(da, da_ld, z, mean, invstd, gamma, beta, mask, keep, seed, dgamma, dbeta, dz, dbias_partial,
                           absmax_partial, pixels, count, C, stream);
}
static int bn_bwd_apply_impl(const float* da, long da_ld, const float* z, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, const uint8_t* mask, float keep, uint64_t seed,
                             const float* dgamma, const float* dbeta, float* dz, float* dbias_partial,
                             float* absmax_partial, long pixels, double count, int C, dc_stream_t stream) {
  DC_REQUIRE(da && z && mean && invstd && gamma && beta && dgamma && dbeta && dz, DC_EINVAL,
             "dc_bn_bwd_apply: null pointer");
  DC_REQUIRE(pixels > 0 && da_ld >= C && da_ld % 4 == 0 && keep > 0.f, DC_EINVAL, "dc_bn_bwd_apply: bad sizes");
  int rc = chan_check("dc_bn_bwd_apply", C);
  if (rc) return rc;
  BnParams p{};
  p.da = da; p.da_ld = da_ld; p.z = z; p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.beta = beta;
  p.mask = mask; p.keep = keep; p.seed = seed; p.dgamma = dgamma; p.dbeta = dbeta; p.out = dz; p.out_ld = C;
  p.partial = dbias_partial; p.absmax = absmax_partial; p.pixels = pixels; p.C = C; p.count = count;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_bwd_blocks(pixels, C)), dim3(256), 0, (hipStream_t)stream, p);
  DC_CHECK_LAUNCH("dc_bn_bwd_apply");
  return DC_OK;
}

// dbeta[c] = sum_p partial[p][c][0], dgamma[c] = sum_p partial[p][c][1]
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int P, int C,
                                                             float* dgamma, float* dbeta) {
  __shared__ double sh1[256], sh2[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int i = tid; i < P; i += 256) {
    s1 += (double)partial[((long)i * C + c) * 2];
    s2 += (double)partial[((long)i * C + c) * 2 + 1];
  }
  sh1[tid] = s1; sh2[tid] = s2;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { sh1[tid] += sh1[tid + s]; sh2[tid] += sh2[tid + s]; }
    __syncthreads();
  }
  if (tid == 0) { dbeta[c] = (float)sh1[0]; dgamma[c] = (float)sh2[0]; }
}
extern "C" int dc_bn_bwd_finalize(const float* partial, int P, int C, float* dgamma, float* dbeta, dc_stream_t stream) {
  DC_REQUIRE(partial && dgamma && dbeta && P > 0 && C > 0, DC_EINVAL, "dc_bn_bwd_finalize: bad arguments");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, P, C, dgamma, dbeta);
  DC_CHECK_LAUNCH("dc_bn_bwd_finalize");
  return DC_OK;
}

// "dz on load": instead of materialising dz = gamma*invstd*(dy - mean(dy) - xhat*mean(dy*xhat)) with dc_bn_bwd_apply, the
// data- and weight-gradient kernels of a Dropout-free block form it from (da, z) while they stage their operands:
//   dy = [fmaf(z, sc, sh) > 0] * da ;  dz = fmaf(A, dy, fmaf(D, z - mu, E))
// This launch is dc_bn_bwd_finalize (partial != NULL; with partial == NULL dgamma / dbeta are INPUTS: the all-reduced
// sums of 'sync' BatchNorm) plus the per-channel table those kernels read: dz_coef[DC_DZ_COEF_ROWS][C] =
//   0 sc, 1 sh (dc_bn_affine: the forward's gate expression), 2 mu, 3 A = gamma*invstd, 4 D = -A*invstd*dgamma/count,
//   5 E = -A*dbeta/count, 6 bound >= max |dz_c| = |A| (max|dy_c| + |dbeta|/count + sqrt(count) |dgamma|/count)
// (|xhat| <= sqrt(count) for ANY data).  The consumers' fp16 range guard scales by the power of two that brings the
// largest bound into [2^14, 2^15) (dc_block_guard_scale): no overflow whatever the data, and an element keeps its full
// 22-bit split down to 2^-12 of the bound.  amax_partial[P][C]: per-row max |dy_c| from the kernel that wrote da.
__global__ __launch_bounds__(256) void bn_bwd_finalize_dzin_kernel(const float* __restrict__ partial,
                                                                  const float* __restrict__ amax_partial, int P, int C,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, double count,
                                                                  float* dgamma, float* dbeta, float* __restrict__ coef,
                                                                  float* __restrict__ dbias) {
  __shared__ double sh1[256], sh2[256];
  __shared__ float shm[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  float am = 0.f;
  for (int i = tid; i < P; i += 256) {
    if (partial) {
      s1 += (double)partial[((long)i * C + c) * 2];
      s2 += (double)partial[((long)i * C + c) * 2 + 1];
    }
    am = fmaxf(am, amax_partial[(long)i * C + c]);
  }
  sh1[tid] = s1; sh2[tid] = s2; shm[tid] = am;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { sh1[tid] += sh1[tid + s]; sh2[tid] += sh2[tid + s]; shm[tid] = fmaxf(shm[tid], shm[tid + s]); }
    __syncthreads();
  }
  if (tid == 0) {
    float db, dg;
    if (partial) { db = (float)sh1[0]; dg = (float)sh2[0]; dbeta[c] = db; dgamma[c] = dg; }
    else { db = dbeta[c]; dg = dgamma[c]; }
    const float mu = mean[c], is = invstd[c], ga = gamma[c];
    const float invM = (float)(1.0 / count);
    const float mdy = db * invM, mdyx = dg * invM, gs = ga * is;      // dc_bn_bwd_apply's own constants
    float sc, sh;
    dc_bn_affine(mu, is, ga, beta[c], sc, sh);
    coef[c] = sc;
    coef[C + c] = sh;
    coef[2 * C + c] = mu;
    coef[3 * C + c] = gs;
    coef[4 * C + c] = -(gs * is) * mdyx;
    coef[5 * C + c] = -gs * mdy;
    coef[6 * C + c] = fabsf(gs) * (shm[0] + fabsf(mdy) + fabsf(mdyx) * (float)sqrt(count));
    if (dbias) dbias[c] = 0.f;      // sum_p dz_c = A (sum dy - dbeta - (dgamma/M) sum xhat) = 0 exactly
  }
}
extern "C" int dc_bn_bwd_finalize_dzin(const float* partial, const float* amax_partial, int P, int C, const float* mean,
                                       const float* invstd, const float* gamma, const float* beta, double count,
                                       float* dgamma, float* dbeta, float* dz_coef, float* dbias, dc_stream_t stream) {
  DC_REQUIRE(amax_partial && mean && invstd && gamma && beta && dgamma && dbeta && dz_coef && P > 0 && C > 0 && count >= 1.0,
             DC_EINVAL, "dc_bn_bwd_finalize_dzin: bad arguments");
  hipLaunchKernelGGL(bn_bwd_finalize_dzin_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, amax_partial, P, C,
                     mean, invstd, gamma, beta, count, dgamma, dbeta, dz_coef, dbias);
  DC_CHECK_LAUNCH("dc_bn_bwd_finalize_dzin");
  return DC_OK;
}

// ------------------------------------------------------------------------------------------------
// MaxPooling2D(2,2): first max in row-major window order (strict '>' keeps the earliest on ties).
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ in, long in_ld, float* __restrict__ out,
                                                         uint8_t* __restrict__ idx, int N, int H, int W, int C) {
  const int C4 = C >> 2, h2 = H >> 1, w2 = W >> 1;
  const long total = (long)N * h2 * w2 * C4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const int q = (int)(i % C4);
    long r = i / C4;
    const int x = (int)(r % w2); r /= w2;
    const int y = (int)(r % h2);
    const long n = r / h2;
    const float* base = in + ((n * H + 2 * y) * W + 2 * x) * in_ld + 4 * q;
    const f32x4 v0 = ld4(base), v1 = ld4(base + in_ld), v2 = ld4(base + (long)W * in_ld), v3 = ld4(base + (long)(W + 1) * in_ld);
    f32x4 m = v0;
    uchar4 k = make_uchar4(0, 0, 0, 0);
    uint8_t* kk = reinterpret_cast<uint8_t*>(&k);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (v1[e] > m[e]) { m[e] = v1[e]; kk[e] = 1; }
      if (v2[e] > m[e]) { m[e] = v2[e]; kk[e] = 2; }
      if (v3[e] > m[e]) { m[e] = v3[e]; kk[e] = 3; }
    }
    st4(out + i * 4, m);
    if (idx) *reinterpret_cast<uchar4*>(idx + i * 4) = k;
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                         const float* __restrict__ skip, long skip_ld,
                                                         float* __restrict__ dx, int N, int H, int W, int C) {
  const int C4 = C >> 2, h2 = H >> 1, w2 = W >> 1;
  const long total = (long)N * h2 * w2 * C4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const int q = (int)(i % C4);
    long r = i / C4;
    const int x = (int)(r % w2); r /= w2;
    const int y = (int)(r % h2);
    const long n = r / h2;
    const f32x4 g = ld4(dy + i * 4);
    const uchar4 k = *reinterpret_cast<const uchar4*>(idx + i * 4);
    const uint8_t* kk = reinterpret_cast<const uint8_t*>(&k);
    const long pix00 = (n * H + 2 * y) * W + 2 * x;
#pragma unroll
    for (int pos = 0; pos < 4; ++pos) {
      const long pix = pix00 + (pos >> 1) * (long)W + (pos & 1);
      f32x4 v = skip ? ld4(skip + pix * skip_ld + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += (kk[e] == pos) ? g[e] : 0.f;
      st4(dx + pix * C + 4 * q, v);
    }
  }
}

// maxpool backward + the BatchNorm-backward partial sums of the layer whose output was pooled: the routed gradient
// (+ skip gradient) is da of that layer, so (sum dy, sum dy*xhat) are accumulated while it is written and
// dc_bn_bwd_reduce's re-read of da is saved.  The channel quad of a thread is loop-invariant (grid stride is a
// multiple of C/4), partials land as dc_bn_bwd_finalize expects: bn.partial[block][C][2].
__global__ __launch_bounds__(256) void maxpool_bwd_bnred_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                               const float* __restrict__ skip, long skip_ld,
                                                               float* __restrict__ dx, int N, int H, int W, BnParams p) {
  __shared__ f32x4 sm1[256], sm2[256];
  const int C = p.C, C4 = C >> 2, h2 = H >> 1, w2 = W >> 1, PPB = 256 / C4;
  const int tid = threadIdx.x, q = tid % C4, pl = tid / C4;
  const f32x4 mu = ld4(p.mean + 4 * q), is = ld4(p.invstd + 4 * q), ga = ld4(p.gamma + 4 * q), be = ld4(p.beta + 4 * q);
  f32x4 sc, sh;
  bn_affine4(mu, is, ga, be, sc, sh);
  const bool drop = p.keep < 1.f;
  const float inv_keep = drop ? 1.f / p.keep : 1.f;
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, am = {0.f, 0.f, 0.f, 0.f};
  const long total = (long)N * h2 * w2 * C4;
  for (long i = blockIdx.x * 256L + tid; i < total; i += gridDim.x * 256L) {
    long r = i / C4;
    const int x = (int)(r % w2); r /= w2;
    const int y = (int)(r % h2);
    const long n = r / h2;
    const f32x4 g = ld4(dy + i * 4);
    const uchar4 k = *reinterpret_cast<const uchar4*>(idx + i * 4);
    const uint8_t* kk = reinterpret_cast<const uint8_t*>(&k);
    const long pix00 = (n * H + 2 * y) * W + 2 * x;
#pragma unroll
    for (int pos = 0; pos < 4; ++pos) {
      const long pix = pix00 + (pos >> 1) * (long)W + (pos & 1);
      f32x4 v = skip ? ld4(skip + pix * skip_ld + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += (kk[e] == pos) ? g[e] : 0.f;
      st4(dx + pix * C + 4 * q, v);
      const long elem = pix * C + 4 * q;
      const f32x4 z = ld4(p.z + elem);
      const f32x4 xh = (z - mu) * is;
      const f32x4 yv = fma4(z, sc, sh);
      f32x4 d = v;
      if (drop) d *= drop_factor(p, elem, inv_keep);
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = (yv[e] > 0.f) ? d[e] : 0.f;
      s1 += d;
      s2 += d * xh;
      am = absmax4(am, d);
    }
  }
  sm1[tid] = s1; sm2[tid] = s2;
  __syncthreads();
  if (pl == 0) {
    for (int k = 1; k < PPB; ++k) { s1 += sm1[k * C4 + q]; s2 += sm2[k * C4 + q]; }
    float* dst = p.partial + ((long)blockIdx.x * C + 4 * q) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) { dst[2 * e] = s1[e]; dst[2 * e + 1] = s2[e]; }
  }
  if (p.amax_c) {
    __syncthreads();
    sm1[tid] = am;
    __syncthreads();
    if (pl == 0) {
      for (int k = 1; k < PPB; ++k) am = max4(am, sm1[k * C4 + q]);
      st4(p.amax_c + (long)blockIdx.x * C + 4 * q, am);
    }
  }
}

static int pool_check(const char* fn, int N, int H, int W, int C) {
  DC_REQUIRE(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, DC_EINVAL, "%s: H and W must be positive and even", fn);
  DC_REQUIRE(C >= 4 && C % 4 == 0, DC_EUNSUP, "%s: C=%d must be a multiple of 4", fn, C);
  return DC_OK;
}
extern "C" int dc_maxpool2x2_fwd(const float* in, long in_ld, float* out, uint8_t* idx, int N, int H, int W, int C,
                                 dc_stream_t stream) {
  DC_REQUIRE(in && out && in_ld >= C && in_ld % 4 == 0, DC_EINVAL, "dc_maxpool2x2_fwd: bad arguments");
  int rc = pool_check("dc_maxpool2x2_fwd", N, H, W, C);
  if (rc) return rc;
  const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, in_ld, out, idx, N, H, W, C);
  DC_CHECK_LAUNCH("dc_maxpool2x2_fwd");
  return DC_OK;
}
extern "C" int dc_maxpool2x2_bwd(const float* dy, const uint8_t* idx, const float* skip, long skip_ld, float* dx, int N,
                                 int H, int W, int C, dc_stream_t stream) {
  DC_REQUIRE(dy && idx && dx && (!skip || (skip_ld >= C && skip_ld % 4 == 0)), DC_EINVAL, "dc_maxpool2x2_bwd: bad arguments");
  int rc = pool_check("dc_maxpool2x2_bwd", N, H, W, C);
  if (rc) return rc;
  const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, idx, skip, skip_ld, dx, N, H, W, C);
  DC_CHECK_LAUNCH("dc_maxpool2x2_bwd");
  return DC_OK;
}
static int pool_bwd_blocks(int N, int H, int W, int C) {
  const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
  return (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
}
extern "C" int dc_maxpool2x2_bwd_blocks(int N, int H, int W, int C) { return pool_bwd_blocks(N, H, W, C); }
extern "C" int dc_maxpool2x2_bwd_bnred(const float* dy, const uint8_t* idx, const float* skip, long skip_ld, float* dx,
                                       const float* z, const float* mean, const float* invstd, const float* gamma,
                                       const float* beta, const uint8_t* mask, float keep, uint64_t seed,
                                       float* bn_partial, float* amax_partial, int N, int H, int W, int C, dc_stream_t stream) {
  DC_REQUIRE(dy && idx && dx && z && mean && invstd && gamma && beta && bn_partial && keep > 0.f, DC_EINVAL,
             "dc_maxpool2x2_bwd_bnred: bad arguments");
  DC_REQUIRE(!skip || (skip_ld >= C && skip_ld % 4 == 0), DC_EINVAL, "dc_maxpool2x2_bwd_bnred: bad skip_ld");
  int rc = pool_check("dc_maxpool2x2_bwd_bnred", N, H, W, C);
  if (rc) return rc;
  rc = chan_check("dc_maxpool2x2_bwd_bnred", C);
  if (rc) return rc;
  BnParams p{};
  p.z = z; p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.beta = beta; p.mask = mask; p.keep = keep;
  p.seed = seed; p.partial = bn_partial; p.amax_c = amax_partial; p.C = C;
  hipLaunchKernelGGL(maxpool_bwd_bnred_kernel, dim3(pool_bwd_blocks(N, H, W, C)), dim3(256), 0, (hipStream_t)stream, dy,
                     idx, skip, skip_ld, dx, N, H, W, p);
  DC_CHECK_LAUNCH("dc_maxpool2x2_bwd_bnred");
  return DC_OK;
}

// ------------------------------------------------------------------------------------------------
// UpSampling2D() + Dropout  (the `upsampling_or_transpose != 'transpose'` branch, unet_2d_summary.py:160-161,:198):
// nearest-neighbour 2x, then dropout on the up-sampled tensor.  in dense [N,H,W,C] -> out strided [N,2H,2W,C].
// The dropout element index is that of the dense up-sampled tensor.
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          long out_ld, const uint8_t* __restrict__ mask, float keep,
                                                          uint64_t seed, int N, int H, int W, int C,
                                                          const float* __restrict__ ab_in, long ab_in_ld,
                                                          float* __restrict__ ab_out, long ab_out_ld) {
  const int C4 = C >> 2;
  const long total = (long)N * 4 * H * W * C4;
  const bool drop = keep < 1.f;
  const float inv_keep = drop ? 1.f / keep : 1.f;
  if (ab_out && blockIdx.x == 0)       // range-guard bound of the up-sampled tensor = the source's, times 1/keep
    for (int s = 0; s < (ab_in_ld > 0 ? DC_ABOUND_SLOTS : 1); ++s)
      for (int c = threadIdx.x; c < C; c += 256) ab_out[s * ab_out_ld + c] = ab_in[s * ab_in_ld + c] * inv_keep;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const int q = (int)(i % C4);
    long r = i / C4;
    const int x = (int)(r % (2 * W)); r /= (2 * W);
    const int y = (int)(r % (2 * H));
    const long n = r / (2 * H);
    f32x4 v = ld4(in + ((n * H + (y >> 1)) * W + (x >> 1)) * C + 4 * q);
    const long pix = (n * 2 * H + y) * 2 * W + x;
    if (drop) {
      const long elem = pix * C + 4 * q;
      if (mask) {
        const uchar4 m = *reinterpret_cast<const uchar4*>(mask + elem);
        v[0] *= m.x ? inv_keep : 0.f; v[1] *= m.y ? inv_keep : 0.f; v[2] *= m.z ? inv_keep : 0.f; v[3] *= m.w ? inv_keep : 0.f;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= dc_keep_factor(seed, (uint64_t)(elem + e), keep, inv_keep);
      }
    }
    st4(out + pix * out_ld + 4 * q, v);
  }
}

// din[n,i,j,c] = sum_{a,b} dout[n,2i+a,2j+b,c] * dropfactor ; dout strided (dout_ld), din dense
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ dout, long dout_ld,
                                                          const uint8_t* __restrict__ mask, float keep, uint64_t seed,
                                                          float* __restrict__ din, int N, int H, int W, int C) {
  const int C4 = C >> 2;
  const long total = (long)N * H * W * C4;
  const bool drop = keep < 1.f;
  const float inv_keep = drop ? 1.f / keep : 1.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const int q = (int)(i % C4);
    long r = i / C4;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const long n = r / H;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pos = 0; pos < 4; ++pos) {
      const long pix = (n * 2 * H + 2 * y + (pos >> 1)) * 2 * W + 2 * x + (pos & 1);
      f32x4 g = ld4(dout + pix * dout_ld + 4 * q);
      if (drop) {
        const long elem = pix * C + 4 * q;
        if (mask) {
          const uchar4 m = *reinterpret_cast<const uchar4*>(mask + elem);
          g[0] *= m.x ? inv_keep : 0.f; g[1] *= m.y ? inv_keep : 0.f; g[2] *= m.z ? inv_keep : 0.f; g[3] *= m.w ? inv_keep : 0.f;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] *= dc_keep_factor(seed, (uint64_t)(elem + e), keep, inv_keep);
        }
      }
      acc += g;
    }
    st4(din + i * 4, acc);
  }
}

extern "C" int dc_upsample2x_drop_fwd(const float* in, float* out, long out_ld, const uint8_t* mask, float keep,
                                      uint64_t seed, const float* abound_in, long abound_in_ld, float* abound_out,
                                      long abound_out_ld, int N, int H, int W, int C, dc_stream_t stream) {
  DC_REQUIRE((abound_in == nullptr) == (abound_out == nullptr), DC_EINVAL, "dc_upsample2x_drop_fwd: abound_in and abound_out go together");
  DC_REQUIRE(in && out && N > 0 && H > 0 && W > 0 && C >= 4 && C % 4 == 0 && out_ld >= C && out_ld % 4 == 0 && keep > 0.f,
             DC_EINVAL, "dc_upsample2x_drop_fwd: bad arguments");
  const long total = (long)N * 4 * H * W * (C / 4);
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(upsample_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, out_ld, mask, keep,
                     seed, N, H, W, C, abound_in, abound_in_ld, abound_out, abound_out_ld);
  DC_CHECK_LAUNCH("dc_upsample2x_drop_fwd");
  return DC_OK;
}
extern "C" int dc_upsample2x_drop_bwd(const float* dout, long dout_ld, const uint8_t* mask, float keep, uint64_t seed,
                                      float* din, int N, int H, int W, int C, dc_stream_t stream) {
  DC_REQUIRE(dout && din && N > 0 && H > 0 && W > 0 && C >= 4 && C % 4 == 0 && dout_ld >= C && dout_ld % 4 == 0 && keep > 0.f,
             DC_EINVAL, "dc_upsample2x_drop_bwd: bad arguments");
  const long total = (long)N * H * W * (C / 4);
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout, dout_ld, mask, keep,
                     seed, din, N, H, W, C);
  DC_CHECK_LAUNCH("dc_upsample2x_drop_bwd");
  return DC_OK;
}

// ------------------------------------------------------------------------------------------------
// Head: C/4 lanes per pixel, xor-shuffle the two partial logits, lane q==0 finishes the pixel.
#define DC_HEAD_SUMS 12   // bce, tp, sum round(p), fn, sum y, sum y*p, sum p^2, sum y^2, sum p, weighted-bce, 2 spare
__device__ __forceinline__ float round_half_even(float x) { return rintf(x); }  // default RN mode = half to even


// One pixel's contribution of a lane's 4 channels to a logit: an EXPLICIT fma chain, so that every head kernel (inference forward,
// training forward, fused forward + backward in both instantiations) rounds it identically whatever the compiler would contract.
__device__ __forceinline__ float head_dot4(const f32x4& v, const float (&k)[4]) {
  return __builtin_fmaf(v[3], k[3], __builtin_fmaf(v[2], k[2], __builtin_fmaf(v[1], k[1], v[0] * k[0])));
}
// C4T = 8 (C == 32): the 8 loads of a lane's pixel group are issued together (see head_fwd_bwd_kernel); 0 = generic.
template <int C4T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ a, const float* __restrict__ kh,
                                                      const float* __restrict__ bh, const uint8_t* __restrict__ y,
                                                      float* __restrict__ p, float* __restrict__ partial, long pixels,
                                                      int C, const float* __restrict__ in_sc,
                                                      const float* __restrict__ in_sh) {
  __shared__ float sm[256][DC_HEAD_SUMS];
  const int C4 = C4T ? C4T : (C >> 2), PPB = 256 / C4;
  const int tid = threadIdx.x, q = tid % C4, pl = tid / C4;
  float k0[4], k1[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { k0[e] = kh[(4 * q + e) * 2]; k1[e] = kh[(4 * q + e) * 2 + 1]; }
  const float b0 = bh[0], b1 = bh[1];
  // in_sc != NULL: `a` holds the PRE-BatchNorm tensor z and the activation relu(fmaf(z, sc, sh)) is formed here
  const bool bnin = in_sc != nullptr;
  const f32x4 isc = bnin ? ld4(in_sc + 4 * q) : f32x4{1.f, 1.f, 1.f, 1.f};
  const f32x4 ish = bnin ? ld4(in_sh + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  float acc[DC_HEAD_SUMS];
#pragma unroll
  for (int k = 0; k < DC_HEAD_SUMS; ++k) acc[k] = 0.f;
  // The C/4 lanes of a pixel share its two logits after the xor-reduce.  Pixels are taken C/4 at a time: lane q keeps
  // the logits of the q-th one, then EVERY lane runs the softmax / BCE / metric arithmetic (3 exp, 3 log per pixel) for
  // its own pixel -- C/4 times fewer issues of that transcendental-heavy block than one active lane per pixel.
  const long iters = (pixels + (long)gridDim.x * PPB - 1) / ((long)gridDim.x * PPB);
  for (long it0 = 0; it0 < iters; it0 += C4) {  // uniform trip counts: the shuffles below need every lane
    float kz0 = 0.f, kz1 = 0.f;
    long kpix = pixels;                           // "no pixel"
    auto one = [&](int j, long pix, bool ok, f32x4 v) {
      if (bnin) {
        v = fma4(v, isc, ish);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ok ? fmaxf(v[e], 0.f) : 0.f;
      }
      float z0 = head_dot4(v, k0);
      float z1 = head_dot4(v, k1);
      if constexpr (C4T != 0) {
#pragma unroll
        for (int s = 1; s < C4T; s <<= 1) { z0 += __shfl_xor(z0, s); z1 += __shfl_xor(z1, s); }
      } else {
        for (int s = 1; s < C4; s <<= 1) { z0 += __shfl_xor(z0, s); z1 += __shfl_xor(z1, s); }
      }
      if (j == q) { kz0 = z0; kz1 = z1; kpix = pix; }
    };
    if constexpr (C4T != 0) {
      f32x4 held[C4T];
#pragma unroll
      for (int j = 0; j < C4T; ++j) {
        const long pix = ((it0 + j) * gridDim.x + blockIdx.x) * PPB + pl;
        held[j] = pix < pixels ? ld4(a + pix * C + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < C4T; ++j) {
        const long pix = ((it0 + j) * gridDim.x + blockIdx.x) * PPB + pl;
        one(j, pix, pix < pixels, held[j]);
      }
    } else {
      for (int j = 0; j < C4; ++j) {
        const long pix = ((it0 + j) * gridDim.x + blockIdx.x) * PPB + pl;
        const bool ok = pix < pixels;
        one(j, pix, ok, ok ? ld4(a + pix * C + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f});
      }
    }
    if (kpix < pixels) {
      const long pix = kpix;
      const float z0 = kz0 + b0, z1 = kz1 + b1;
      // 2-way softmax, max-subtracted, class 1
      const float m = fmaxf(z0, z1);
      const float e0 = expf(z0 - m), e1 = expf(z1 - m);
      const float pr = e1 / (e0 + e1);
      p[pix] = pr;
      if (y) {
        const float yt = (float)y[pix];
        // Keras/TF binary_crossentropy (A.10): clip, logit, stable sigmoid-BCE
        const float pc = fminf(fmaxf(pr, 1e-7f), 1.f - 1e-7f);
        const float x = logf(pc / (1.f - pc));
        acc[0] += fmaxf(x, 0.f) - x * yt + log1pf(expf(-fabsf(x)));
        const float rp = round_half_even(pr);
        acc[1] += rp * yt;
        acc[2] += rp;
        acc[3] += fminf(fmaxf(yt - rp, 0.f), 1.f);
        acc[4] += yt;
        acc[5] += yt * pr;
        acc[6] += pr * pr;
        acc[7] += yt * yt;
        acc[8] += pr;
        // weighted_binary_crossentropy (utils/neurons.py:13-29): -(2*y*log(p+1e-7) + (1-y)*log(1-p+1e-7))
        acc[9] -= 2.f * yt * logf(pr + 1e-7f) + (1.f - yt) * logf(1.f - pr + 1e-7f);
      }
    }
  }
  if (partial) {
#pragma unroll
    for (int k = 0; k < DC_HEAD_SUMS; ++k) sm[tid][k] = acc[k];
    __syncthreads();
    if (tid < DC_HEAD_SUMS) {
      double s = 0.0;
      for (int t = 0; t < 256; ++t) s += (double)sm[t][tid];
      partial[(long)blockIdx.x * DC_HEAD_SUMS + tid] = (float)s;
    }
  }
}

// s = dL/dlogit1 (dlogit0 = -s).  loss_kind 0: Keras binary_crossentropy, s = (p - y)/M where the clip is inactive.
// Kinds 1..3 (utils/neurons.py:13-29,78-94) go through dL/dp * p(1-p); the global sums they need come from the
// forward's reduced sums (device memory, no host round trip):
//   1 weighted_binary_crossentropy: dL/dp = -(2y/(p+1e-7) - (1-y)/(1-p+1e-7)) / M
//   2 dice_loss   = 1 - 2I/D,  I = sum y*p, D = sum y + sum p + 1e-7:        dL/dp = -2 (y D - I) / D^2
//   3 dicesq_loss = -2I/D,     D = sum y^2 + sum p^2 + 1e-7:                 dL/dp = -2 (y D - 2 p I) / D^2
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ a, const float* __restrict__ p,
                                                      const uint8_t* __restrict__ y, const float* __restrict__ kh,
                                                      float* __restrict__ da, float* __restrict__ partial, long pixels,
                                                      int C, int loss_kind, const double* __restrict__ sums,
                                                      const float* __restrict__ in_sc, const float* __restrict__ in_sh,
                                                      const float* __restrict__ bn_mean,
                                                      const float* __restrict__ bn_invstd, float* __restrict__ bn_partial,
                                                      float* __restrict__ amax_partial) {
  __shared__ f32x4 sm[256];
  __shared__ float sms[256];
  const int C4 = C >> 2, PPB = 256 / C4;
  const int tid = threadIdx.x, q = tid % C4, pl = tid / C4;
  f32x4 kd;  // kh[c][1] - kh[c][0]
#pragma unroll
  for (int e = 0; e < 4; ++e) kd[e] = kh[(4 * q + e) * 2 + 1] - kh[(4 * q + e) * 2];
  const float invM = 1.f / (float)pixels;
  const bool bnin = in_sc != nullptr;
  const f32x4 isc = bnin ? ld4(in_sc + 4 * q) : f32x4{1.f, 1.f, 1.f, 1.f};
  const f32x4 ish = bnin ? ld4(in_sh + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  // bn_partial != NULL (needs bnin): also emit this block's (sum dy, sum dy*xhat) of the producing BatchNorm layer,
  // dy = da * [relu gate] -- the sums dc_bn_bwd_reduce would otherwise re-read da and z for
  const bool bnred = bn_partial != nullptr;
  const f32x4 bmu = bnred ? ld4(bn_mean + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 bis = bnred ? ld4(bn_invstd + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 r1 = {0.f, 0.f, 0.f, 0.f}, r2 = {0.f, 0.f, 0.f, 0.f};
  float I = 0.f, D = 1.f;
  if (loss_kind == 2) { I = (float)sums[5]; D = (float)(sums[4] + sums[8] + 1e-7); }
  if (loss_kind == 3) { I = (float)sums[5]; D = (float)(sums[7] + sums[6] + 1e-7); }
  const float invD2 = 1.f / (D * D);
  f32x4 sa = {0.f, 0.f, 0.f, 0.f};
  float ss = 0.f, smax = 0.f;     // smax: max |dL/dlogit| of this thread's pixels (da = kd * s: max |da_c| = |kd_c| * smax)
  for (long pix = (long)blockIdx.x * PPB + pl; pix < pixels; pix += (long)gridDim.x * PPB) {
    const float pr = p[pix], yt = (float)y[pix];
    float s;
    if (loss_kind == 0) {
      const bool inside = pr > 1e-7f && pr < 1.f - 1e-7f;
      s = inside ? (pr - yt) * invM : 0.f;
    } else {
      float dp;
      if (loss_kind == 1) dp = -(2.f * yt / (pr + 1e-7f) - (1.f - yt) / (1.f - pr + 1e-7f)) * invM;
      else if (loss_kind == 2) dp = -2.f * (yt * D - I) * invD2;
      else dp = -2.f * (yt * D - 2.f * pr * I) * invD2;
      s = dp * pr * (1.f - pr);
    }
    f32x4 v = ld4(a + pix * C + 4 * q);
    if (bnin) {
      const f32x4 zraw = v;
      v = fma4(v, isc, ish);
      if (bnred) {
        const f32x4 xh = (zraw - bmu) * bis;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dy = (v[e] > 0.f) ? kd[e] * s : 0.f;
          r1[e] += dy;
          r2[e] += dy * xh[e];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    st4(da + pix * C + 4 * q, kd * s);
    sa += v * s;
    smax = fmaxf(smax, fabsf(s));
    if (q == 0) ss += s;
  }
  sm[tid] = sa;
  sms[tid] = (q == 0) ? ss : 0.f;
  __syncthreads();
  if (pl == 0) {
    for (int k = 1; k < PPB; ++k) sa += sm[k * C4 + q];
    st4(partial + (long)blockIdx.x * (C + 4) + 4 * q, sa);
  }
  if (tid == 0) {
    float s = 0.f;
    for (int k = 0; k < PPB; ++k) s += sms[k * C4];
    partial[(long)blockIdx.x * (C + 4) + C] = s;
  }
  if (bnred) {
    __shared__ f32x4 sm2[256];
    __syncthreads();
    sm[tid] = r1; sm2[tid] = r2;
    __syncthreads();
    if (pl == 0) {
      for (int k = 1; k < PPB; ++k) { r1 += sm[k * C4 + q]; r2 += sm2[k * C4 + q]; }
      float* dst = bn_partial + ((long)blockIdx.x * C + 4 * q) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) { dst[2 * e] = r1[e]; dst[2 * e + 1] = r2[e]; }
    }
  }
  if (amax_partial) {          // every pixel's s is seen by all C/4 lanes of its group: a block-wide max of the per-thread maxima
    __shared__ float smx[4];
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) smax = fmaxf(smax, __shfl_xor(smax, o));
    if ((tid & 63) == 0) smx[tid >> 6] = smax;
    __syncthreads();
    if (pl == 0) {
      const float m = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
      f32x4 am;
#pragma unroll
      for (int e = 0; e < 4; ++e) am[e] = fabsf(kd[e]) * m;
      st4(amax_partial + (long)blockIdx.x * C + 4 * q, am);
    }
  }
}

// Training step with a per-pixel loss (loss_kind 0 / 1): the head's backward needs nothing but the pixel's own p and y, so
// the forward can do it while the C/4 values of the pixel are still in cache -- head_fwd_kernel's pixel-group scheme,
// then lane q hands the dL/dlogit of ITS pixel back to the group and every lane writes its 4 channels of da, the head's
// weight-gradient partials and (bn_partial) the producing BatchNorm layer's backward sums, as head_bwd_kernel does.
// One read of the 512^2 x nfb tensor instead of two, no read of p.  Per-block summation order differs from
// head_bwd_kernel's (same terms).
// C4T = 8 (C == 32, the network's nfb): the 8 loads of a lane's pixel group are issued TOGETHER and held in registers for the backward
// half.  The generic loop issues one load per iteration and waits for it (the shuffle reductions in between keep the compiler from
// hoisting the next one): 16 groups x 8 dependent loads per lane made the kernel latency-bound at 285-300 us for a 200-us sweep.
template <int C4T>
__global__ __launch_bounds__(256) void head_fwd_bwd_kernel(const float* __restrict__ a, const float* __restrict__ kh,
                                                          const float* __restrict__ bh, const uint8_t* __restrict__ y,
                                                          float* __restrict__ p, float* __restrict__ partial,
                                                          float* __restrict__ da, float* __restrict__ gpartial, long pixels,
                                                          int C, int loss_kind, const float* __restrict__ in_sc,
                                                          const float* __restrict__ in_sh, const float* __restrict__ bn_mean,
                                                          const float* __restrict__ bn_invstd, float* __restrict__ bn_partial,
                                                          float* __restrict__ amax_partial) {
  __shared__ float sm[256][DC_HEAD_SUMS];
  __shared__ f32x4 sm4[256], sm4b[256];
  __shared__ float sms[256];
  const int C4 = C4T ? C4T : (C >> 2), PPB = 256 / C4;
  const int tid = threadIdx.x, q = tid % C4, pl = tid / C4, lane = tid & 63;
  float k0[4], k1[4];
  f32x4 kd;
#pragma unroll
  for (int e = 0; e < 4; ++e) { k0[e] = kh[(4 * q + e) * 2]; k1[e] = kh[(4 * q + e) * 2 + 1]; kd[e] = k1[e] - k0[e]; }
  const float b0 = bh[0], b1 = bh[1];
  const float invM = 1.f / (float)pixels;
  const bool bnin = in_sc != nullptr, bnred = bn_partial != nullptr;
  const f32x4 isc = bnin ? ld4(in_sc + 4 * q) : f32x4{1.f, 1.f, 1.f, 1.f};
  const f32x4 ish = bnin ? ld4(in_sh + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 bmu = bnred ? ld4(bn_mean + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 bis = bnred ? ld4(bn_invstd + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  float acc[DC_HEAD_SUMS];
#pragma unroll
  for (int k = 0; k < DC_HEAD_SUMS; ++k) acc[k] = 0.f;
  f32x4 sa = {0.f, 0.f, 0.f, 0.f}, r1 = sa, r2 = sa;
  float ss = 0.f, smax = 0.f;
  const long iters = (pixels + (long)gridDim.x * PPB - 1) / ((long)gridDim.x * PPB);
  for (long it0 = 0; it0 < iters; it0 += C4) {
    float kz0 = 0.f, kz1 = 0.f;
    long kpix = pixels;
    f32x4 held[C4T ? C4T : 1];
    auto fwd_one = [&](int j, long pix, bool ok, f32x4 v) {
      if (bnin) {
        v = fma4(v, isc, ish);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ok ? fmaxf(v[e], 0.f) : 0.f;
      }
      float z0 = head_dot4(v, k0);
      float z1 = head_dot4(v, k1);
      if constexpr (C4T != 0) {
#pragma unroll
        for (int s = 1; s < C4T; s <<= 1) { z0 += __shfl_xor(z0, s); z1 += __shfl_xor(z1, s); }
      } else {
        for (int s = 1; s < C4; s <<= 1) { z0 += __shfl_xor(z0, s); z1 += __shfl_xor(z1, s); }
      }
      if (j == q) { kz0 = z0; kz1 = z1; kpix = pix; }
    };
    if constexpr (C4T != 0) {
#pragma unroll
      for (int j = 0; j < C4T; ++j) {
        const long pix = ((it0 + j) * gridDim.x + blockIdx.x) * PPB + pl;
        held[j] = pix < pixels ? ld4(a + pix * C + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < C4T; ++j) {
        const long pix = ((it0 + j) * gridDim.x + blockIdx.x) * PPB + pl;
        fwd_one(j, pix, pix < pixels, held[j]);
      }
    } else {
      for (int j = 0; j < C4; ++j) {
        const long pix = ((it0 + j) * gridDim.x + blockIdx.x) * PPB + pl;
        const bool ok = pix < pixels;
        fwd_one(j, pix, ok, ok ? ld4(a + pix * C + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f});
      }
    }
    float sown = 0.f;                              // dL/dlogit1 of this lane's pixel
    if (kpix < pixels) {
      const long pix = kpix;
      const float z0 = kz0 + b0, z1 = kz1 + b1;
      const float m = fmaxf(z0, z1);
      const float e0 = expf(z0 - m), e1 = expf(z1 - m);
      const float pr = e1 / (e0 + e1);
      p[pix] = pr;
      const float yt = (float)y[pix];
      const float pc = fminf(fmaxf(pr, 1e-7f), 1.f - 1e-7f);
      const float x = logf(pc / (1.f - pc));
      acc[0] += fmaxf(x, 0.f) - x * yt + log1pf(expf(-fabsf(x)));
      const float rp = round_half_even(pr);
      acc[1] += rp * yt;
      acc[2] += rp;
      acc[3] += fminf(fmaxf(yt - rp, 0.f), 1.f);
      acc[4] += yt;
      acc[5] += yt * pr;
      acc[6] += pr * pr;
      acc[7] += yt * yt;
      acc[8] += pr;
      acc[9] -= 2.f * yt * logf(pr + 1e-7f) + (1.f - yt) * logf(1.f - pr + 1e-7f);
      if (loss_kind == 0) {
        const bool inside = pr > 1e-7f && pr < 1.f - 1e-7f;
        sown = inside ? (pr - yt) * invM : 0.f;
      } else {
        const float dp = -(2.f * yt / (pr + 1e-7f) - (1.f - yt) / (1.f - pr + 1e-7f)) * invM;
        sown = dp * pr * (1.f - pr);
      }
      smax = fmaxf(smax, fabsf(sown));
    }
    // backward of the C4 pixels of this group
    auto bwd_one = [&](int j, f32x4 v, long pix) {
      const float s = __shfl(sown, lane - q + j);
      if (pix < pixels) {
        if (bnin) {
          const f32x4 zraw = v;
          v = fma4(v, isc, ish);
          if (bnred) {
            const f32x4 xh = (zraw - bmu) * bis;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float dy = (v[e] > 0.f) ? kd[e] * s : 0.f;
              r1[e] += dy;
              r2[e] += dy * xh[e];
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        st4(da + pix * C + 4 * q, kd * s);
        sa += v * s;
        if (q == 0) ss += s;
      }
    };
    if constexpr (C4T != 0) {
#pragma unroll
      for (int j = 0; j < C4T; ++j) bwd_one(j, held[j], ((it0 + j) * gridDim.x + blockIdx.x) * PPB + pl);
    } else {
      // (the values are re-read: they were loaded a moment ago)
      for (int j = 0; j < C4; ++j) {
        const long pix = ((it0 + j) * gridDim.x + blockIdx.x) * PPB + pl;
        bwd_one(j, pix < pixels ? ld4(a + pix * C + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f}, pix);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < DC_HEAD_SUMS; ++k) sm[tid][k] = acc[k];
  sm4[tid] = sa;
  sms[tid] = (q == 0) ? ss : 0.f;
  __syncthreads();
  if (tid < DC_HEAD_SUMS) {
    double s = 0.0;
    for (int t = 0; t < 256; ++t) s += (double)sm[t][tid];
    partial[(long)blockIdx.x * DC_HEAD_SUMS + tid] = (float)s;
  }
  if (pl == 0) {
    for (int k = 1; k < PPB; ++k) sa += sm4[k * C4 + q];
    st4(gpartial + (long)blockIdx.x * (C + 4) + 4 * q, sa);
  }
  if (tid == 0) {
    float s = 0.f;
    for (int k = 0; k < PPB; ++k) s += sms[k * C4];
    gpartial[(long)blockIdx.x * (C + 4) + C] = s;
  }
  if (bnred) {
    __syncthreads();
    sm4[tid] = r1; sm4b[tid] = r2;
    __syncthreads();
    if (pl == 0) {
      for (int k = 1; k < PPB; ++k) { r1 += sm4[k * C4 + q]; r2 += sm4b[k * C4 + q]; }
      float* dst = bn_partial + ((long)blockIdx.x * C + 4 * q) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) { dst[2 * e] = r1[e]; dst[2 * e + 1] = r2[e]; }
    }
  }
  if (amax_partial) {          // every pixel's s is seen by all C/4 lanes of its group: a block-wide max of the per-thread maxima
    __shared__ float smx[4];
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) smax = fmaxf(smax, __shfl_xor(smax, o));
    if ((tid & 63) == 0) smx[tid >> 6] = smax;
    __syncthreads();
    if (pl == 0) {
      const float m = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
      f32x4 am;
#pragma unroll
      for (int e = 0; e < 4; ++e) am[e] = fabsf(kd[e]) * m;
      st4(amax_partial + (long)blockIdx.x * C + 4 * q, am);
    }
  }
}

// partial rows are padded to C+4 floats to keep float4 alignment; one block per output element, fixed order
__global__ __launch_bounds__(256) void head_grad_finalize_kernel(const float* __restrict__ partial, int blocks, int C,
                                                                float* dkh, float* dbh) {
  __shared__ double sm[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  double s = 0.0;
  for (int b = tid; b < blocks; b += 256) s += (double)partial[(long)b * (C + 4) + c];
  sm[tid] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (tid < k) sm[tid] += sm[tid + k];
    __syncthreads();
  }
  if (tid == 0) {
    s = sm[0];
    if (c < C) { dkh[2 * c] = (float)(-s); dkh[2 * c + 1] = (float)s; }
    else { dbh[0] = (float)(-s); dbh[1] = (float)s; }
  }
}

static int head_blocks(long pixels) {
  long b = (pixels + 255) / 256;
  return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}
extern "C" int dc_head_blocks(long pixels) { return head_blocks(pixels); }

static int head_fwd_impl(const float* a, const float* in_sc, const float* in_sh, const float* kh, const float* bh,
                         const uint8_t* y, float* p, float* partial, long pixels, int C, dc_stream_t stream);
extern "C" int dc_head_fwd(const float* a, const float* kh, const float* bh, const uint8_t* y, float* p, float* partial,
                           long pixels, int C, dc_stream_t stream) {
  return head_fwd_impl(a, nullptr, nullptr, kh, bh, y, p, partial, pixels, C, stream);
}
extern "C" int dc_head_fwd_bnin(const float* z, const float* in_scale, const float* in_shift, const float* kh,
                                const float* bh, const uint8_t* y, float* p, float* partial, long pixels, int C,
                                dc_stream_t stream) {
  DC_REQUIRE(in_scale && in_shift, DC_EINVAL, "dc_head_fwd_bnin: null scale/shift");
  return head_fwd_impl(z, in_scale, in_shift, kh, bh, y, p, partial, pixels, C, stream);
}
static int head_fwd_impl(const float* a, const float* in_sc, const float* in_sh, const float* kh, const float* bh,
                         const uint8_t* y, float* p, float* partial, long pixels, int C, dc_stream_t stream) {
  DC_REQUIRE(a && kh && bh && p && pixels > 0, DC_EINVAL, "dc_head_fwd: bad arguments");
  DC_REQUIRE(!y || partial, DC_EINVAL, "dc_head_fwd: y given without a partial buffer");
  int rc = chan_check("dc_head_fwd", C);
  if (rc) return rc;
  DC_REQUIRE(C <= 256, DC_EUNSUP, "dc_head_fwd: C=%d > 256 (the C/4 lanes of a pixel must fit one wave)", C);
  hipLaunchKernelGGL(C == 32 ? head_fwd_kernel<8> : head_fwd_kernel<0>, dim3(head_blocks(pixels)), dim3(256), 0, (hipStream_t)stream, a, kh, bh, y, p,
                     y ? partial : nullptr, pixels, C, in_sc, in_sh);
  DC_CHECK_LAUNCH("dc_head_fwd");
  return DC_OK;
}
static int head_bwd_impl(const float* a, const float* in_sc, const float* in_sh, const float* p, const uint8_t* y,
                         const float* kh, float* da, float* partial, int loss_kind, const double* sums, long pixels,
                         int C, dc_stream_t stream, const float* bn_mean = nullptr, const float* bn_invstd = nullptr,
                         float* bn_partial = nullptr, float* amax_partial = nullptr);
extern "C" int dc_head_bwd(const float* a, const float* p, const uint8_t* y, const float* kh, float* da, float* partial,
                           int loss_kind, const double* sums, long pixels, int C, dc_stream_t stream) {
  return head_bwd_impl(a, nullptr, nullptr, p, y, kh, da, partial, loss_kind, sums, pixels, C, stream);
}
extern "C" int dc_head_bwd_bnin(const float* z, const float* in_scale, const float* in_shift, const float* p,
                                const uint8_t* y, const float* kh, float* da, float* partial, int loss_kind,
                                const double* sums, long pixels, int C, dc_stream_t stream) {
  DC_REQUIRE(in_scale && in_shift, DC_EINVAL, "dc_head_bwd_bnin: null scale/shift");
  return head_bwd_impl(z, in_scale, in_shift, p, y, kh, da, partial, loss_kind, sums, pixels, C, stream);
}
extern "C" int dc_head_bwd_bnin_bnred(const float* z, const float* in_scale, const float* in_shift, const float* p,
                                      const uint8_t* y, const float* kh, float* da, float* partial, int loss_kind,
                                      const double* sums, const float* bn_mean, const float* bn_invstd,
                                      float* bn_partial, float* amax_partial, long pixels, int C, dc_stream_t stream) {
  DC_REQUIRE(in_scale && in_shift && bn_mean && bn_invstd && bn_partial, DC_EINVAL, "dc_head_bwd_bnin_bnred: null pointer");
  return head_bwd_impl(z, in_scale, in_shift, p, y, kh, da, partial, loss_kind, sums, pixels, C, stream, bn_mean,
                       bn_invstd, bn_partial, amax_partial);
}
static int head_bwd_impl(const float* a, const float* in_sc, const float* in_sh, const float* p, const uint8_t* y,
                         const float* kh, float* da, float* partial, int loss_kind, const double* sums, long pixels,
                         int C, dc_stream_t stream, const float* bn_mean, const float* bn_invstd, float* bn_partial,
                         float* amax_partial) {
  DC_REQUIRE(a && p && y && kh && da && partial && pixels > 0, DC_EINVAL, "dc_head_bwd: bad arguments");
  DC_REQUIRE(loss_kind >= 0 && loss_kind <= 3 && (loss_kind < 2 || sums), DC_EINVAL, "dc_head_bwd: bad loss_kind / sums");
  int rc = chan_check("dc_head_bwd", C);
  if (rc) return rc;
  hipLaunchKernelGGL(head_bwd_kernel, dim3(head_blocks(pixels)), dim3(256), 0, (hipStream_t)stream, a, p, y, kh, da,
                     partial, pixels, C, loss_kind, sums, in_sc, in_sh, bn_mean, bn_invstd, bn_partial, amax_partial);
  DC_CHECK_LAUNCH("dc_head_bwd");
  return DC_OK;
}
extern "C" int dc_head_fwd_bwd(const float* a, const float* in_scale, const float* in_shift, const float* kh,
                               const float* bh, const uint8_t* y, float* p, float* partial, float* da, float* grad_partial,
                               int loss_kind, const float* bn_mean, const float* bn_invstd, float* bn_partial,
                               float* amax_partial, long pixels, int C, dc_stream_t stream) {
  DC_REQUIRE(a && kh && bh && y && p && partial && da && grad_partial && pixels > 0, DC_EINVAL, "dc_head_fwd_bwd: bad arguments");
  DC_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), DC_EINVAL, "dc_head_fwd_bwd: in_scale and in_shift go together");
  DC_REQUIRE(loss_kind == 0 || loss_kind == 1, DC_EUNSUP,
             "dc_head_fwd_bwd: loss_kind %d needs the global sums first (use dc_head_fwd + dc_head_bwd)", loss_kind);
  DC_REQUIRE(!bn_partial || (in_scale && bn_mean && bn_invstd), DC_EINVAL, "dc_head_fwd_bwd: bn_partial needs in_scale / bn_mean / bn_invstd");
  int rc = chan_check("dc_head_fwd_bwd", C);
  if (rc) return rc;
  DC_REQUIRE(C <= 64, DC_EUNSUP, "dc_head_fwd_bwd: C=%d > 64 (the C/4 lanes of a pixel group must divide a wave)", C);
  hipLaunchKernelGGL(C == 32 ? head_fwd_bwd_kernel<8> : head_fwd_bwd_kernel<0>, dim3(head_blocks(pixels)), dim3(256), 0, (hipStream_t)stream, a, kh, bh, y, p,
                     partial, da, grad_partial, pixels, C, loss_kind, in_scale, in_shift, bn_mean, bn_invstd, bn_partial,
                     amax_partial);
  DC_CHECK_LAUNCH("dc_head_fwd_bwd");
  return DC_OK;
}
extern "C" int dc_head_grad_finalize(const float* partial, int blocks, int C, float* dkh, float* dbh, dc_stream_t stream) {
  DC_REQUIRE(partial && dkh && dbh && blocks > 0 && C > 0, DC_EINVAL, "dc_head_grad_finalize: bad arguments");
  hipLaunchKernelGGL(head_grad_finalize_kernel, dim3(C + 1), dim3(256), 0, (hipStream_t)stream, partial, blocks, C,
                     dkh, dbh);
  DC_CHECK_LAUNCH("dc_head_grad_finalize");
  return DC_OK;
}

// ------------------------------------------------------------------------------------------------
// Deterministic partial-sum reductions.
// stage 1: grid (ceil(L/64), PY): block = 64 l x 4 p-lanes, each thread strides its p-chunk; stage 2 sums PY rows.
__global__ __launch_bounds__(256) void reduce_stage1_kernel(const float* __restrict__ in, int P, long L, int chunk,
                                                           float scale, float* __restrict__ out) {
  __shared__ double sm[4][64];
  const int lx = threadIdx.x & 63, pz = threadIdx.x >> 6;
  const long l = (long)blockIdx.x * 64 + lx;
  const int p0 = blockIdx.y * chunk, p1 = min(p0 + chunk, P);
  double s = 0.0;
  if (l < L)
    for (int p = p0 + pz; p < p1; p += 4) s += (double)in[(long)p * L + l];
  sm[pz][lx] = s;
  __syncthreads();
  if (pz == 0 && l < L) {
    s = sm[0][lx] + sm[1][lx] + sm[2][lx] + sm[3][lx];
    out[(long)blockIdx.y * L + l] = (float)(s * (gridDim.y == 1 ? (double)scale : 1.0));
  }
}
__global__ __launch_bounds__(256) void reduce_stage2_kernel(const float* __restrict__ tmp, int PY, long L, float scale,
                                                           float* __restrict__ out) {
  const long l = blockIdx.x * 256L + threadIdx.x;
  if (l >= L) return;
  double s = 0.0;
  for (int p = 0; p < PY; ++p) s += (double)tmp[(long)p * L + l];
  out[l] = (float)(s * (double)scale);
}

extern "C" int dc_reduce_partials(const float* in, int P, long L, float scale, float* out, float* tmp,
                                  dc_stream_t stream) {
  DC_REQUIRE(in && out && P > 0 && L > 0, DC_EINVAL, "dc_reduce_partials: bad arguments");
  int PY = P >= 64 ? 32 : 1;
  DC_REQUIRE(PY == 1 || tmp, DC_EINVAL, "dc_reduce_partials: tmp scratch required for P >= 64");
  const int chunk = dc_cdiv(P, PY);
  PY = dc_cdiv(P, chunk);
  dim3 grid((unsigned)dc_cdiv(L, 64), (unsigned)PY);
  hipLaunchKernelGGL(reduce_stage1_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, P, L, chunk, scale,
                     PY == 1 ? out : tmp);
  DC_CHECK_LAUNCH("dc_reduce_partials");
  if (PY > 1) {
    hipLaunchKernelGGL(reduce_stage2_kernel, dim3(dc_cdiv(L, 256)), dim3(256), 0, (hipStream_t)stream, tmp, PY, L, scale, out);
    DC_CHECK_LAUNCH("dc_reduce_partials(2)");
  }
  return DC_OK;
}

__global__ void reduce_f64_kernel(const float* __restrict__ in, int P, int L, double* __restrict__ out) {
  __shared__ double sm[256];
  const int l = blockIdx.x, tid = threadIdx.x;
  double s = 0.0;
  for (int p = tid; p < P; p += 256) s += (double)in[(long)p * L + l];
  sm[tid] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (tid < k) sm[tid] += sm[tid + k];
    __syncthreads();
  }
  if (tid == 0) out[l] = sm[0];
}
extern "C" int dc_reduce_partials_f64(const float* in, int P, int L, double* out, dc_stream_t stream) {
  DC_REQUIRE(in && out && P > 0 && L > 0, DC_EINVAL, "dc_reduce_partials_f64: bad arguments");
  hipLaunchKernelGGL(reduce_f64_kernel, dim3(L), dim3(256), 0, (hipStream_t)stream, in, P, L, out);
  DC_CHECK_LAUNCH("dc_reduce_partials_f64");
  return DC_OK;
}

// ------------------------------------------------------------------------------------------------
// Keras-2.0.6 Adam: eps outside the bias correction (lr_t carries the correction).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, long n, float lr_t, float b1, float b2, float eps,
                                                  float gscale) {
  const long n4 = n >> 2;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += gridDim.x * 256L) {
    const f32x4 gg = ld4(g + 4 * i) * gscale;
    f32x4 mm = ld4(m + 4 * i), vv = ld4(v + 4 * i), pp = ld4(p + 4 * i);
    mm = b1 * mm + (1.f - b1) * gg;
    vv = b2 * vv + (1.f - b2) * gg * gg;
#pragma unroll
    for (int e = 0; e < 4; ++e) pp[e] -= lr_t * mm[e] / (sqrtf(vv[e]) + eps);
    st4(m + 4 * i, mm); st4(v + 4 * i, vv); st4(p + 4 * i, pp);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    const float gg = g[i] * gscale;
    const float mm = b1 * m[i] + (1.f - b1) * gg, vv = b2 * v[i] + (1.f - b2) * gg * gg;
    m[i] = mm; v[i] = vv;
    p[i] -= lr_t * mm / (sqrtf(vv) + eps);
  }
}
extern "C" int dc_adam_step_flat(float* p, const float* g, float* m, float* v, long n, float lr_t, float b1, float b2,
                                 float eps, float gscale, dc_stream_t stream) {
  DC_REQUIRE(p && g && m && v && n > 0, DC_EINVAL, "dc_adam_step_flat: bad arguments");
  DC_REQUIRE(dc_aligned16(p) && dc_aligned16(g) && dc_aligned16(m) && dc_aligned16(v), DC_EINVAL,
             "dc_adam_step_flat: pointers must be 16-byte aligned");
  const long n4 = (n + 3) / 4;
  const int blocks = (int)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr_t, b1, b2, eps, gscale);
  DC_CHECK_LAUNCH("dc_adam_step_flat");
  return DC_OK;
}

__global__ void fill_kernel(float* p, long n, float value) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) p[i] = value;
}
extern "C" int dc_fill(float* p, long n, float value, dc_stream_t stream) {
  DC_REQUIRE(p && n > 0, DC_EINVAL, "dc_fill: bad arguments");
  const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, n, value);
  DC_CHECK_LAUNCH("dc_fill");
  return DC_OK;
}

__global__ void scale_kernel(float* p, long n, float s) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) p[i] *= s;
}
extern "C" int dc_scale_flat(float* p, long n, float s, dc_stream_t stream) {
  DC_REQUIRE(p && n > 0, DC_EINVAL, "dc_scale_flat: bad arguments");
  const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, n, s);
  DC_CHECK_LAUNCH("dc_scale_flat");
  return DC_OK;
}

// ------------------------------------------------------------------------------------------------
// Test-time augmentation on the device (UNet2DSummary.predict(augmentation=True),
// /root/reference/deepcalcium/models/neurons/unet_2d_summary.py:585-595 with the 8-entry table of
// utils/neurons.py:112-137).  The augmentations are pixel permutations: the host applies the table's own numpy
// functions to an index image ONCE and uploads the resulting maps, so the device cannot disagree with the table.
//   dc_gather_maps : out[k][p] = in[maps[k][p]]                         (the K augmented copies of one padded image)
//   dc_tta_merge   : m[p] = sum_k preds[k][invmaps[k][p]] / K  accumulated in DOUBLE in table order (= numpy's
//                    float64 `mp += inv(...) / 8`), cropped to hs x ws, mask = m > threshold (uint8), as :588-595.
__global__ __launch_bounds__(256) void gather_maps_kernel(const float* __restrict__ in, const int* __restrict__ maps,
                                                         float* __restrict__ out, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) out[i] = in[maps[i]];
}
extern "C" int dc_gather_maps(const float* in, const int* maps, float* out, int K, long n, dc_stream_t stream) {
  DC_REQUIRE(in && maps && out && K > 0 && n > 0, DC_EINVAL, "dc_gather_maps: bad arguments");
  const long total = (long)K * n;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(gather_maps_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, maps, out, total);
  DC_CHECK_LAUNCH("dc_gather_maps");
  return DC_OK;
}
__global__ __launch_bounds__(256) void tta_merge_kernel(const float* __restrict__ preds, const int* __restrict__ invmaps,
                                                       int K, long n, int W, int hs, int ws, double threshold,
                                                       uint8_t* __restrict__ mask, float* __restrict__ mean_out) {
  const long total = (long)hs * ws;
  const double invK = 1.0 / (double)K;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const int y = (int)(i / ws), x = (int)(i - (long)y * ws);
    const long p = (long)y * W + x;
    double m = 0.0;
    for (int k = 0; k < K; ++k) m += (double)preds[(long)k * n + invmaps[(long)k * n + p]] * invK;
    mask[i] = m > threshold ? 1 : 0;
    if (mean_out) mean_out[i] = (float)m;
  }
}
extern "C" int dc_tta_merge(const float* preds, const int* invmaps, int K, int H, int W, int hs, int ws, double threshold,
                            uint8_t* mask, float* mean_out, dc_stream_t stream) {
  DC_REQUIRE(preds && invmaps && mask && K > 0 && H > 0 && W > 0 && hs > 0 && ws > 0 && hs <= H && ws <= W, DC_EINVAL,
             "dc_tta_merge: bad arguments");
  const long total = (long)hs * ws;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(tta_merge_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, preds, invmaps, K, (long)H * W, W,
                     hs, ws, threshold, mask, mean_out);
  DC_CHECK_LAUNCH("dc_tta_merge");
  return DC_OK;
}

// ------------------------------------------------------------------------------------------------
// Training-batch assembly on the device: the array half of UNet2DSummary._batch_gen
// (/root/reference/deepcalcium/models/neurons/unet_2d_summary.py:434-530).  The host keeps drawing the reference's single
// numpy random stream (dataset, neuron pixel, jitter, augmentation indices :479-527) and reduces every item to four longs;
// the summaries / masks of all datasets are resident in HBM (a few tens of MB), so no image data crosses PCIe per step:
//   crop [y0 : y0+eh, x0 : x0+ew] of dataset k  ->  zero-filled (hw, hw) window (:505-521)  ->  the composition of the drawn
//   flips / rot90s (:524-527), ONE of the 8 dihedral pixel permutations:  out[i][j] = win[r][c],
//   (r, c) = (d4 & 1) ? (j, i) : (i, j);  r = (d4 & 2) ? hw-1-r : r;  c = (d4 & 4) ? hw-1-c : c.
// The items ride in the KERNEL ARGUMENT block (<= DC_CROP_MAX_ITEMS per launch): nothing is staged, copied or retained.
struct CropItems { long v[DC_CROP_MAX_ITEMS][4]; };       // { element offset of (y0, x0), row stride, eh << 32 | ew, d4 }
__global__ __launch_bounds__(256) void crop_augment_kernel(const float* __restrict__ S, const uint8_t* __restrict__ M,
                                                           CropItems items, int hw, float* __restrict__ x,
                                                           uint8_t* __restrict__ y) {
  const int b = blockIdx.y;
  const long org = items.v[b][0], ld = items.v[b][1];
  const int eh = (int)(items.v[b][2] >> 32), ew = (int)(items.v[b][2] & 0xffffffffL), d4 = (int)items.v[b][3];
  const int q4 = hw >> 2;                                   // 4 consecutive output pixels of a row per thread
  const long outb = (long)b * hw * hw;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < hw * q4; t += gridDim.x * 256) {
    const int i = t / q4, j0 = (t - i * q4) * 4;
    f32x4 v;
    uchar4 m;
    uint8_t* mm = reinterpret_cast<uint8_t*>(&m);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int r = (d4 & 1) ? j0 + e : i, c = (d4 & 1) ? i : j0 + e;
      if (d4 & 2) r = hw - 1 - r;
      if (d4 & 4) c = hw - 1 - c;
      const bool in = r < eh && c < ew;
      const long src = org + (long)r * ld + c;
      v[e] = in ? S[src] : 0.f;
      mm[e] = in ? M[src] : (uint8_t)0;
    }
    st4(x + outb + (long)i * hw + j0, v);
    *reinterpret_cast<uchar4*>(y + outb + (long)i * hw + j0) = m;
  }
}
extern "C" int dc_crop_augment(const float* S, const uint8_t* M, long src_elems, const long* items_host, int B, int hw,
                               float* x, uint8_t* y, dc_stream_t stream) {
  DC_REQUIRE(S && M && items_host && x && y && B > 0 && hw > 0 && src_elems > 0, DC_EINVAL, "dc_crop_augment: bad arguments");
  DC_REQUIRE(hw % 4 == 0 && dc_aligned16(x) && (reinterpret_cast<uintptr_t>(y) & 3) == 0, DC_EINVAL,
             "dc_crop_augment: the window side must be a multiple of 4, x 16-byte and y 4-byte aligned");
  for (int b = 0; b < B; ++b) {                               // the items come from the host: a bad one must not read out of bounds
    const long org = items_host[4 * b], ld = items_host[4 * b + 1];
    const long eh = items_host[4 * b + 2] >> 32, ew = items_host[4 * b + 2] & 0xffffffffL, d4 = items_host[4 * b + 3];
    DC_REQUIRE(org >= 0 && ld > 0 && eh >= 0 && ew >= 0 && eh <= hw && ew <= hw && ew <= ld && d4 >= 0 && d4 < 8 &&
               (eh == 0 || ew == 0 || org + (eh - 1) * ld + ew <= src_elems), DC_EINVAL,
               "dc_crop_augment: item %d {offset %ld, stride %ld, %ld x %ld, d4 %ld} leaves the %ld-element source", b, org, ld, eh, ew,
               d4, src_elems);
  }
  const int per = hw * (hw / 4);
  const int gx = (per + 255) / 256 > 64 ? 64 : (per + 255) / 256;
  for (int b0 = 0; b0 < B; b0 += DC_CROP_MAX_ITEMS) {
    const int nb = B - b0 < DC_CROP_MAX_ITEMS ? B - b0 : DC_CROP_MAX_ITEMS;
    CropItems it;
    for (int b = 0; b < nb; ++b)
      for (int e = 0; e < 4; ++e) it.v[b][e] = items_host[4 * (b0 + b) + e];
    hipLaunchKernelGGL(crop_augment_kernel, dim3(gx, nb), dim3(256), 0, (hipStream_t)stream, S, M, it, hw,
                       x + (long)b0 * hw * hw, y + (long)b0 * hw * hw);
  }
  DC_CHECK_LAUNCH("dc_crop_augment");
  return DC_OK;
}

// p > 0.5 as uint8 over the [y0:y1, x0:x1] window of each (H, W) probability map: numpy's `mp[y0:y1, x0:x1].round()` on
// probabilities (round half to even: 0.5 -> 0) of the validation callback (unet_2d_summary.py:90-91), so that only the scored
// stripe, one byte per pixel, leaves the device.
__global__ __launch_bounds__(256) void round_window_kernel(const float* __restrict__ p, int H, int W, int y0, int y1, int x0,
                                                           int x1, uint8_t* __restrict__ out) {
  const int hh = y1 - y0, ww = x1 - x0;
  const long per = (long)hh * ww;
  const long n = blockIdx.y;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < per; i += gridDim.x * 256L) {
    const int r = (int)(i / ww), c = (int)(i - (long)r * ww);
    out[n * per + i] = p[(n * H + y0 + r) * (long)W + x0 + c] > 0.5f ? 1 : 0;
  }
}
extern "C" int dc_round_window_u8(const float* p, int N, int H, int W, int y0, int y1, int x0, int x1, uint8_t* out,
                                  dc_stream_t stream) {
  DC_REQUIRE(p && out && N > 0 && 0 <= y0 && y0 < y1 && y1 <= H && 0 <= x0 && x0 < x1 && x1 <= W, DC_EINVAL,
             "dc_round_window_u8: bad window [%d:%d, %d:%d] of %d x %d", y0, y1, x0, x1, H, W);
  const long per = (long)(y1 - y0) * (x1 - x0);
  const int gx = (int)((per + 255) / 256 > 256 ? 256 : (per + 255) / 256);
  hipLaunchKernelGGL(round_window_kernel, dim3(gx, N), dim3(256), 0, (hipStream_t)stream, p, H, W, y0, y1, x0, x1, out);
  DC_CHECK_LAUNCH("dc_round_window_u8");
  return DC_OK;
}

// Shared helpers for libdcunet (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <mutex>
#include "../../include/dcunet.h"

#define DC_OK 0
#define DC_EINVAL (-1)
#define DC_EHIP (-2)
#define DC_EUNSUP (-3)

void dc_set_error(const char* fmt, ...);

// The library's only environment switch, read ONCE (common.cpp): DC_IGEMM_PP = 1 (default) the persistent role-split
// conv3x3 kernel serves every launch it can, 2 = forward launches only, 0 = never (A/B against the 256-thread kernels).
struct DcConfig { int igemm_pp; };
const DcConfig& dc_config();

#define DC_REQUIRE(cond, code, ...)        \
  do {                                     \
    if (!(cond)) {                         \
      dc_set_error(__VA_ARGS__);           \
      return (code);                       \
    }                                      \
  } while (0)

// Check the launch that just happened (asynchronous: catches configuration errors only).
#define DC_CHECK_LAUNCH(name)                                               \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      dc_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
      return DC_EHIP;                                                       \
    }                                                                       \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the kernel ON ONE DEVICE: set it once per (kernel,
// device), from whichever host thread launches there first (dcunet.h: concurrent callers on different devices).
struct DcLdsAttr {
  std::mutex mu;
  bool done[64] = {};
};
static inline int dc_func_max_lds(DcLdsAttr& st, const void* kern, int bytes, const char* name) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  DC_REQUIRE(e == hipSuccess, DC_EHIP, "%s: hipGetDevice: %s", name, hipGetErrorString(e));
  std::lock_guard<std::mutex> lock(st.mu);
  if (dev >= 0 && dev < 64 && st.done[dev]) return DC_OK;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  DC_REQUIRE(e == hipSuccess, DC_EHIP, "%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e));
  if (dev >= 0 && dev < 64) st.done[dev] = true;
  return DC_OK;
}

// dc_bracket_next_launch (dcunet.h): the pending event pair of this host thread, taken by the next entry point that honours it
bool dc_take_bracket(hipEvent_t* before, hipEvent_t* after);

static inline int dc_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline bool dc_is_pow2(long v) { return v > 0 && (v & (v - 1)) == 0; }
static inline bool dc_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Counter-based dropout RNG: keep-decision for element `idx` under `seed` (same bits in fwd and bwd).
__device__ __forceinline__ uint32_t dc_hash32(uint64_t seed, uint64_t idx) {
  uint64_t x = idx * 0x9E3779B97F4A7C15ull + seed;
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32;
  return (uint32_t)x;
}
__device__ __forceinline__ float dc_keep_factor(uint64_t seed, uint64_t idx, float keep, float inv_keep) {
  // uniform in [0,1): floor(keep + u) == 1  <=>  u >= 1 - keep   (Keras/TF dropout form)
  float u = (float)(dc_hash32(seed, idx) >> 8) * (1.0f / 16777216.0f);
  return (u >= 1.0f - keep) ? inv_keep : 0.0f;
}

// Training-mode BatchNorm as ONE fused multiply-add per element: y = fmaf(z, sc, sh) -- the form TensorFlow's
// tf.nn.batch_normalization (the op behind Keras-2.0.x BatchNormalization on the TF backend) evaluates:
// inv = gamma*rsqrt(var+eps); y = x*inv + (beta - mean*inv).  Every kernel that applies BN
// (bn_relu_drop_fwd, the BN-backward ReLU gate, and the consumers that apply BN + ReLU while staging a
// non-materialised activation) derives (sc, sh) with exactly these two operations, so the ReLU gate is the same
// bit pattern everywhere.
__host__ __device__ __forceinline__ void dc_bn_affine(float mu, float is, float ga, float be, float& sc, float& sh) {
  sc = ga * is;
  sh = __builtin_fmaf(-mu, sc, be);
}

// ---- per-channel moments that survive |mean| >> sigma ------------------------------------------------------------
// BatchNorm statistics leave the convolution epilogues as per-(tile, channel) partials.  Summing v and v*v in fp32
// loses the variance once |mean|/sigma reaches ~1e3 (catastrophic cancellation in E[v^2] - E[v]^2).  Instead every lane
// accumulates SHIFTED sums around its own first value K (d = v - K: s1 = sum d, s2 = sum d^2), turns them into
// (count, mean, M2 = sum (v - mean)^2), and the lanes / waves of a tile are merged with Chan's parallel update.  The
// tile's result is stored as DOUBLES (S = n*mean, Q = M2 + S*mean = sum v^2): the finalize kernels keep summing
// (S, Q) over tiles -- in double, where the cancellation is harmless -- so their interface is unchanged.
struct DcMoments { float n, mean, m2; };
__device__ __forceinline__ DcMoments dc_moments_from_shifted(float n, float K, float s1, float s2) {
  DcMoments r;
  r.n = n;
  const float ms = n > 0.f ? s1 / n : 0.f;
  r.mean = n > 0.f ? K + ms : 0.f;
  r.m2 = n > 0.f ? fmaxf(__builtin_fmaf(-s1, ms, s2), 0.f) : 0.f;
  return r;
}
__device__ __forceinline__ DcMoments dc_moments_merge(const DcMoments a, const DcMoments b) {
  DcMoments r;
  r.n = a.n + b.n;
  const float w = r.n > 0.f ? b.n / r.n : 0.f;
  const float d = b.mean - a.mean;
  r.mean = __builtin_fmaf(d, w, a.mean);
  r.m2 = a.m2 + b.m2 + d * d * a.n * w;
  return r;
}
__device__ __forceinline__ void dc_moments_store(double* dst, const DcMoments m) {
  const double S = (double)m.n * (double)m.mean;
  dst[0] = S;
  dst[1] = (double)m.m2 + S * (double)m.mean;
}

// ---- fp16 range guard of the split-fp16 (f16x3) contractions -----------------------------------------------------
// The fp16 hi/lo split keeps fp32-grade accuracy only while the operand sits inside fp16's exponent range: above
// 65504 hi becomes inf, far below 2^-3 the lo term is an fp16 subnormal.  Every f16x3 consumer therefore multiplies
// its activation operand by an exact POWER OF TWO derived from a per-channel magnitude bound of that tensor
// (`abound[c]`, written by the producing layer: |gamma|*sqrt(count) + |beta| for a training-mode BatchNorm output --
// |xhat| <= sqrt(count - 1) holds for ANY data -- or the measured max|a| in inference) and undoes it in the epilogue:
// max_c abound[c] lands in [2^14, 2^15).  Block-cooperative: every thread of the block calls it (two barriers);
// tmp = 16 floats of LDS that nothing else uses until the call returns.  abound == NULL: scale 1.
__device__ __forceinline__ float dc_pow2_guard(float amax) {
  const unsigned e = (__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu;     // biased exponent of the bound
  if (amax <= 0.f || e < 14u || e == 0xffu) return 1.f;                     // zero / denormal / inf / nan: leave alone
  return __builtin_bit_cast(float, (268u - e) << 23);                       // 2^(14 - floor(log2 amax))
}
// In inference the bound is MEASURED: the producing kernel folds max |a| per channel into DC_ABOUND_SLOTS replicas of
// the array (slot = workgroup id mod 8, `ld` floats apart) -- thousands of workgroups hammering ONE address per channel
// with atomics cost 10-50 % of the memory-bound kernels, eight replicas make it noise -- and the consumer takes the max
// over all slots (ld > 0); training writes slot 0 only (ld == 0).
__device__ __forceinline__ float dc_block_guard_scale(const float* __restrict__ abound, int n, float* tmp, long ld = 0) {
  if (abound == nullptr) return 1.f;
  float m = 0.f;
  // one flat sweep over (replica, channel): independent loads, all in flight together
  const int total = (ld > 0 ? DC_ABOUND_SLOTS : 1) * n;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int sl = i / n, c = i - sl * n;
    m = fmaxf(m, fabsf(abound[sl * ld + c]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) tmp[wave] = m;
  __syncthreads();
  m = tmp[0];
  for (int w = 1; w < nw; ++w) m = fmaxf(m, tmp[w]);
  __syncthreads();
  return dc_pow2_guard(m);
}
// Gradient operands: the power of two that brings max|dz| to [target/2, target] (dc_bn_bwd_apply emits one max per block).
// dc_pow2_from_absmax is THE formula (dc_pow2_scale_from_absmax / dc_bn_bwd_apply_finalize and the data-gradient kernels that
// derive the scale themselves evaluate it on the same maximum: same scale, and a power of two is exact either way).
__device__ __forceinline__ float dc_pow2_from_absmax(float m, float target) {
  float e = (m > 0.f && isfinite(m)) ? floorf(log2f(target / m)) : 0.f;
  e = fminf(fmaxf(e, -100.f), 100.f);
  return exp2f(e);
}
// Block-cooperative (every thread calls it; two barriers; tmp = 16 floats of LDS): max over part[0..n) -> the scale.
__device__ __forceinline__ float dc_block_absmax_scale(const float* __restrict__ part, int n, float target, float* tmp) {
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, part[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) tmp[wave] = m;
  __syncthreads();
  m = tmp[0];
  for (int w = 1; w < nw; ++w) m = fmaxf(m, tmp[w]);
  __syncthreads();
  return dc_pow2_from_absmax(m, target);
}
// running max |v| per channel (inference: the measured bound of a folded-BN activation); order-independent.
// Thousands of workgroups fold into the same few addresses: read first and skip the atomic unless it would raise the
// value (a stale cached read only costs a redundant atomic; after the first workgroups nearly every call is a read).
__device__ __forceinline__ void dc_atomic_absmax(float* dst, float v) {
  const float a = fabsf(v);
  if (a > __builtin_nontemporal_load(dst)) atomicMax(reinterpret_cast<unsigned*>(dst), __builtin_bit_cast(unsigned, a));
}
// Optimistic inference (out_absmax_ld < 0): activations of a BatchNorm network are O(1) and need no scale; the producing
// kernels only raise out_absmax[0] when an output exceeds what the consumers' unscaled fp16 split can take, and the host
// then repeats the forward pass with measured bounds.
#define DC_F16_SAFE_MAX 32768.f
__device__ __forceinline__ long dc_absmax_slot(long ld) { return (long)(blockIdx.x & (DC_ABOUND_SLOTS - 1)) * ld; }

// Buffer descriptor from provably wave-uniform inputs (readfirstlane), so hipcc does not wrap every buffer op in a
// waterfall loop.  Out-of-range offsets read zeros / drop stores (hardware bounds check on num_records = bytes).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dc_make_rsrc(const void* ptr, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

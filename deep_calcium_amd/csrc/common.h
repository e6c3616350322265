// Shared helpers for libdcunet (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/dcunet.h"

#define DC_OK 0
#define DC_EINVAL (-1)
#define DC_EHIP (-2)
#define DC_EUNSUP (-3)

void dc_set_error(const char* fmt, ...);

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

// Buffer descriptor from provably wave-uniform inputs (readfirstlane), so hipcc does not wrap every buffer op in a
// waterfall loop.  Out-of-range offsets read zeros / drop stores (hardware bounds check on num_records = bytes).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dc_make_rsrc(const void* ptr, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// v_mfma_f32_32x32x16_f16 vs v_mfma_f32_16x16x32_f16 with LDS-FED operands in the conv kernel's real fragment layout
// (csrc/igemm_pp.hip: [hi|lo][8-channel group][pixel] 16-byte slots for the activation patch, [tap][g8][hi|lo][col] slots
// for the weights), random data, split-fp16 product triples (lo*hi, hi*lo, hi*hi), one and two consumer waves per SIMD.
//   32x32x16: wave = 2 x 2 blocks of 32x32; per tap and 16-channel chunk 8 ds_read_b128 + 12 MFMAs (the shipped schedule)
//   16x16x32: wave = 4 x 4 blocks of 16x16 (same 64 px x 64 col macro tile, same fragment reuse); K = 32 = two taps of a
//             16-channel chunk (lanes 0-31 read tap t, lanes 32-63 tap t+1: per-lane LDS addresses); per TWO taps
//             16 ds_read_b128 + 48 MFMAs = the same reads and the same FLOPs as two taps of the other shape.
// Question (round-3 review, item 4): does the 16x16x32 shape hold a higher clock / deliver more FLOP/s under the real
// operand traffic?  Build + run:  hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_shape_lds.hip -o /tmp/msl && /tmp/msl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TWI = 34, NPIXH = 10 * 34, PS = ((NPIXH + 5) / 8) * 8 + 2, G8 = 2, BN = 64, TAPS = 9;
constexpr int A_SLOTS = 2 * G8 * PS, B_SLOTS = TAPS * G8 * 2 * BN, STAGE = A_SLOTS + B_SLOTS;    // 16-byte slots

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k32(const u32x4* __restrict__ src, float* __restrict__ out, int iters) {
  extern __shared__ u32x4 lds[];
  for (int i = threadIdx.x; i < STAGE; i += blockDim.x) lds[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, li = lane & 31, h = lane >> 5;
  const u32x4* ldsA = lds;
  const u32x4* ldsB = lds + A_SLOTS;
  int a_base[2];
  for (int mb = 0; mb < 2; ++mb) a_base[mb] = h * PS + (wave * 2 + mb) * TWI + li;
  const int b_base = (h * 2) * BN + li;
  f32x16 acc[2][2] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int toff = (tap / 3) * TWI + (tap % 3), boff = (tap * G8) * 2 * BN;
      f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        ah[mb] = __builtin_bit_cast(f16x8, ldsA[a_base[mb] + toff]);
        al[mb] = __builtin_bit_cast(f16x8, ldsA[a_base[mb] + toff + G8 * PS]);
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        bh[nb] = __builtin_bit_cast(f16x8, ldsB[b_base + boff + nb * 32]);
        bl[nb] = __builtin_bit_cast(f16x8, ldsB[b_base + boff + BN + nb * 32]);
      }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mb], bh[nb], acc[mb][nb], 0, 0, 0);
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], bl[nb], acc[mb][nb], 0, 0, 0);
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], bh[nb], acc[mb][nb], 0, 0, 0);
        }
    }
  }
  float s = 0.f;
  for (int mb = 0; mb < 2; ++mb) for (int nb = 0; nb < 2; ++nb) for (int r = 0; r < 16; ++r) s += acc[mb][nb][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 16x16x32: A fragment lane l: row (pixel) l % 16, k-group l / 16 (8 consecutive k each).  k-groups 0,1 = the two 8-channel
// groups of the chunk at tap t, k-groups 2,3 = the same at tap t+1 (a shifted window of the same patch).
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k16(const u32x4* __restrict__ src, float* __restrict__ out, int iters) {
  extern __shared__ u32x4 lds[];
  for (int i = threadIdx.x; i < STAGE; i += blockDim.x) lds[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, r16 = lane & 15, kg = lane >> 4;
  const u32x4* ldsA = lds;
  const u32x4* ldsB = lds + A_SLOTS;
  const int g8 = kg & 1, tsel = kg >> 1;          // channel group, which tap of the pair
  int a_base[4];
  for (int mb = 0; mb < 4; ++mb) a_base[mb] = g8 * PS + (wave * 2 + (mb >> 1)) * TWI + (mb & 1) * 16 + r16;
  const int b_base = (g8 * 2) * BN + r16;
  f32x4 acc[4][4] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {               // 4 tap pairs + (the ninth tap pairs with the next chunk's first: 9 per 2 chunks)
      const int t0 = 2 * tp, t1 = 2 * tp + 1;
      const int toff = tsel ? (t1 / 3) * TWI + (t1 % 3) : (t0 / 3) * TWI + (t0 % 3);
      const int boff = ((tsel ? t1 : t0) * G8) * 2 * BN;
      f16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        ah[mb] = __builtin_bit_cast(f16x8, ldsA[a_base[mb] + toff]);
        al[mb] = __builtin_bit_cast(f16x8, ldsA[a_base[mb] + toff + G8 * PS]);
      }
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        bh[nb] = __builtin_bit_cast(f16x8, ldsB[b_base + boff + nb * 16]);
        bl[nb] = __builtin_bit_cast(f16x8, ldsB[b_base + boff + BN + nb * 16]);
      }
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mb], bh[nb], acc[mb][nb], 0, 0, 0);
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mb], bl[nb], acc[mb][nb], 0, 0, 0);
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mb], bh[nb], acc[mb][nb], 0, 0, 0);
        }
    }
  }
  float s = 0.f;
  for (int mb = 0; mb < 4; ++mb) for (int nb = 0; nb < 4; ++nb) for (int r = 0; r < 4; ++r) s += acc[mb][nb][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static float run(K kern, int waves, const u32x4* d, float* o, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int lds = STAGE * 16;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), lds, 0, d, o, iters / 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), lds, 0, d, o, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  std::vector<_Float16> h((size_t)STAGE * 8);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
  u32x4* d; float* o;
  hipMalloc(&d, h.size() * 2); hipMalloc(&o, 4 << 20);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const int iters = 6000;
  for (int rep = 0; rep < 3; ++rep) {
    const float a1 = run(k32<4>, 4, d, o, iters), b1 = run(k16<4>, 4, d, o, iters);
    const float a2 = run(k32<8>, 8, d, o, iters), b2 = run(k16<8>, 8, d, o, iters);
    // FLOPs per wave and iteration: k32 9 taps x 12 MFMAs x 32768; k16 4 pairs x 48 x 16384 (= 8 taps' worth)
    const double f32_1 = 256.0 * 4 * iters * 9 * 12 * 32768.0, f16_1 = 256.0 * 4 * iters * 4 * 48 * 16384.0;
    printf("1 consumer wave/SIMD: 32x32x16 %.0f TF/s (%.2f ms) | 16x16x32 %.0f TF/s (%.2f ms) | ratio %.3f\n", f32_1 / a1 / 1e9, a1,
           f16_1 / b1 / 1e9, b1, (f16_1 / b1) / (f32_1 / a1));
    printf("2 consumer waves/SIMD: 32x32x16 %.0f TF/s (%.2f ms) | 16x16x32 %.0f TF/s (%.2f ms) | ratio %.3f\n", 2 * f32_1 / a2 / 1e9, a2,
           2 * f16_1 / b2 / 1e9, b2, (f16_1 / b2) / (f32_1 / a2));
  }
  return 0;
}

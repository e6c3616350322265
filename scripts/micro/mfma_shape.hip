// Bare fp16 MFMA loops on random operands held in registers: v_mfma_f32_32x32x16_f16 vs v_mfma_f32_16x16x32_f16 at the
// same FLOPs per wave (DVFS: the chip may hold a different clock per shape -- MI355X_MICROARCH.md 'DVFS give-back' (7)).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_shape.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k32(const f16x8* __restrict__ in, float* __restrict__ out, int iters) {
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x * 8 + i) & 4095]; b[i] = in[(threadIdx.x * 8 + 4 + i) & 4095]; }
  f32x16 acc[4] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[j], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[(j + 1) & 3], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(j + 1) & 3], b[j], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(j + 2) & 3], b[(j + 3) & 3], acc[3], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k16(const f16x8* __restrict__ in, float* __restrict__ out, int iters) {
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x * 8 + i) & 4095]; b[i] = in[(threadIdx.x * 8 + 4 + i) & 4095]; }
  f32x4 acc[16] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q)      // 4 x (16x16x32) = the FLOPs of 2 x (32x32x16); 16 per j = 4 x 32x32x16-equivalents x 2
        acc[4 * j + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(j + q) & 3], b[(q + 2 * j) & 3], acc[4 * j + q], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  std::vector<_Float16> h(4096 * 8);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
  f16x8* d; float* o;
  hipMalloc(&d, h.size() * 2); hipMalloc(&o, 4 << 20);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = 256 * wps, iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
      float ms32, ms16;
      hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, d, o, 2000);
      hipEventRecord(e0); hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, d, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms32, e0, e1);
      hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, d, o, 2000);
      hipEventRecord(e0); hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, d, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms16, e0, e1);
      const double fl32 = (double)blocks * 4 * iters * 16 * 32768.0;        // 16 MFMAs of 32x32x16 per iteration per wave
      const double fl16 = (double)blocks * 4 * iters * 16 * 16384.0;        // 16 MFMAs of 16x16x32
      printf("%d wave(s)/SIMD: 32x32x16 %.1f TF/s (%.2f ms) | 16x16x32 %.1f TF/s (%.2f ms)\n", wps, fl32 / ms32 / 1e9, ms32,
             fl16 / ms16 / 1e9, ms16);
    }
  }
  return 0;
}

#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// NCH independent accumulation chains, operands in registers: how fast does one wave per SIMD issue dependent MFMAs?
template <int NCH>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x16 acc[NCH] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 12; ++u) acc[u % NCH] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u % NCH], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < NCH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NCH> void run(float* o, int iters, int grid) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NCH>, dim3(grid), dim3(256), 0, 0, o, iters / 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NCH>, dim3(grid), dim3(256), 0, 0, o, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double n = (double)grid * 4 * iters * 12;
  printf("grid %d chains %d: %.2f ms, %.1f ns per MFMA per wave, %.0f TF/s\n", grid, NCH, ms, ms * 1e6 / ((double)iters * 12), n * 32768 / ms / 1e9);
}
int main() {
  float* o; hipMalloc(&o, 4 << 20);
  for (int grid : {256, 8}) {
    run<1>(o, 20000, grid); run<2>(o, 20000, grid); run<4>(o, 20000, grid);
  }
  return 0;
}

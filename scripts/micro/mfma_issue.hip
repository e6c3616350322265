// How many shader cycles does one v_mfma_f32_32x32x16_f16 cost a SIMD when ONE wave feeds the pipe (the role-split conv
// kernel's situation) vs two, with independent vs triple-dependent accumulator chains -- and at which clock?
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_issue.hip -o scripts/micro/mfma_issue && scripts/micro/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAIN>
__global__ __launch_bounds__(256) void k(const f16x8* __restrict__ in, float* __restrict__ out, unsigned long long* clk, int iters) {
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x * 8 + i) & 4095]; b[i] = in[(threadIdx.x * 8 + 4 + i) & 4095]; }
  f32x16 acc[4] = {};
  const unsigned long long r0 = wall_clock64(), c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (CHAIN) {        // three dependent MFMAs per accumulator back to back (the conv kernel's hi/lo triple), 12 per j
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[q], acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(j + 1) & 3], b[q], acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(j + 2) & 3], b[(q + 1) & 3], acc[q], 0, 0, 0);
        }
      } else {            // consecutive MFMAs on different accumulators, 12 per j
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(j + t) & 3], b[(q + t) & 3], acc[q], 0, 0, 0);
      }
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}
int main() {
  std::vector<_Float16> h(4096 * 8);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
  f16x8* d; float* o; unsigned long long* c;
  hipMalloc(&d, h.size() * 2); hipMalloc(&o, 4 << 20); hipMalloc(&c, 64);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 8000;
  for (int chain = 0; chain <= 1; ++chain)
    for (int wps = 1; wps <= 2; ++wps) {
      const int blocks = 256 * wps;
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (chain) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, o, c, iters);
        else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, o, c, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      unsigned long long hc[2]; hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
      const double nmfma = (double)iters * 48;                 // per wave
      const double fl = (double)blocks * 4 * nmfma * 32768.0;
      printf("%s, %d wave(s)/SIMD: %.0f TF/s | %.1f shader cycles per MFMA per wave (%.1f per SIMD) | clock %.0f MHz\n",
             chain ? "dependent triples  " : "independent chains ", wps, fl / ms / 1e9, hc[0] / nmfma, hc[0] / nmfma / wps,
             hc[0] * 100.0 / hc[1]);
    }
  return 0;
}

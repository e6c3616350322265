// Which waves of a 512-thread workgroup share a SIMD on gfx950?  (HW_ID: wave_id[3:0], simd_id[5:4], pipe[7:6], cu_id[11:8], sh[12], se[15:13])
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/wave_simd_probe.hip -o scripts/micro/wave_simd_probe && ./scripts/micro/wave_simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 1) void probe(unsigned* out) {
  extern __shared__ char smem[];
  smem[threadIdx.x] = 1;
  unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));      // hwreg(HW_REG_HW_ID, 0, 32)
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
}
int main() {
  unsigned* d; hipMalloc(&d, 1024 * 8 * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  for (int vg = 0; vg < 1; ++vg) {
    hipLaunchKernelGGL(probe, dim3(1024), dim3(512), 150 * 1024, 0, d);
    hipDeviceSynchronize();
    unsigned h[1024 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b : {0, 1, 255, 600}) {
      printf("block %4d:", b);
      for (int w = 0; w < 8; ++w) printf("  w%d: simd %u wave %u cu %u", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
      printf("\n");
    }
    int hist[8][4] = {};
    for (int b = 0; b < 1024; ++b) for (int w = 0; w < 8; ++w) hist[w][(h[b * 8 + w] >> 4) & 3]++;
    for (int w = 0; w < 8; ++w) printf("wave %d -> simd histogram: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  }
  return 0;
}

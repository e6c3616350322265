// Shader-clock probe: one wave spins for `spin_ticks` of the constant 100 MHz counter and reports how many shader
// cycles went by -> the clock the chip actually ran at while other kernels (launched on other streams) were busy.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/micro/clock_probe.hip -o scripts/micro/libclock_probe.so
#include <hip/hip_runtime.h>
__global__ void clock_probe_kernel(unsigned long long* out, unsigned long long spin_ticks) {
  const unsigned long long r0 = wall_clock64(), c0 = __builtin_readcyclecounter();
  unsigned long long r1 = r0;
  while (r1 - r0 < spin_ticks) { __builtin_amdgcn_s_sleep(8); r1 = wall_clock64(); }
  const unsigned long long c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[0] = r1 - r0; out[1] = c1 - c0; }
}
extern "C" int clock_probe(unsigned long long* out, unsigned long long spin_ticks, void* stream) {
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, spin_ticks);
  return (int)hipGetLastError();
}

// Synthetic co-runners for scripts/corun_probe.py: which shared resource slows the weight-gradient kernel when a
// BatchNorm pass runs beside it?  Each kernel isolates one: VALU issue slots, HBM reads, HBM reads + writes.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/micro/corunners.hip -o scripts/micro/libcorunners.so
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void valu_spin_kernel(float* out, int iters) {
  float a = threadIdx.x, b = 1.0001f, c = 0.5f, d = 0.25f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) { a = __builtin_fmaf(a, b, c); c = __builtin_fmaf(c, b, d); d = __builtin_fmaf(d, b, a); }
  }
  if (a + c + d == 12345.f) out[0] = a;
}
__global__ __launch_bounds__(256) void stream_read_kernel(const f32x4* __restrict__ in, long n, float* out) {
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const long stride = (long)gridDim.x * 256;
  long i = blockIdx.x * 256L + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const f32x4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
    s += a + b + c + d;
  }
  if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[0] = s[0];
}
__global__ __launch_bounds__(256) void stream_copy_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ o, long n) {
  const long stride = (long)gridDim.x * 256;
  long i = blockIdx.x * 256L + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const f32x4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
    o[i] = a; o[i + stride] = b; o[i + 2 * stride] = c; o[i + 3 * stride] = d;
  }
}
extern "C" int valu_spin(float* out, int blocks, int iters, void* st) {
  hipLaunchKernelGGL(valu_spin_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)st, out, iters); return (int)hipGetLastError(); }
extern "C" int stream_read(const void* in, long n16, float* out, int blocks, void* st) {
  hipLaunchKernelGGL(stream_read_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)st, (const f32x4*)in, n16, out); return (int)hipGetLastError(); }
extern "C" int stream_copy(const void* in, void* o, long n16, int blocks, void* st) {
  hipLaunchKernelGGL(stream_copy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)st, (const f32x4*)in, (f32x4*)o, n16); return (int)hipGetLastError(); }

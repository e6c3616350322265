// Stream hand-off events by creation flag (round 5, follow-up of handoff_cost.hip): what the RECORDING queue loses per hand-off with
//   0 hipEventDisableTiming                       (what the engine used)
//   1 ... | hipEventReleaseToDevice               (agent-scope release)
//   2 ... | hipEventDisableSystemFence            (no system-scope fence at the marker)
// and whether the hand-off still publishes the producer's stores: a producer kernel on the main queue fills 256 MB with the repetition's
// number, the consumer on the side queue (behind hipStreamWaitEvent) counts every element that is not that number, 200 repetitions with
// the buffer rewritten each time (stale lines of the previous repetition in another XCD's L2 would be counted).
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/handoff_flags.hip -o scripts/micro/handoff_flags && ./scripts/micro/handoff_flags
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void work(float* p, int iters) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  float a = p[i & 65535];
  for (int k = 0; k < iters; ++k) a = __builtin_fmaf(a, 1.0001f, 0.5f);
  p[i & 65535] = a;
}
__global__ void fill(unsigned* p, long n, unsigned v) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void check(const unsigned* p, long n, unsigned v, unsigned long long* bad) {
  unsigned long long b = 0;
  // walk the buffer from the far end so that a consumer block reads lines another XCD's producer block wrote
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b += p[n - 1 - i] != v;
  if (b) atomicAdd(bad, b);
}
int main() {
  float* buf; CK(hipMalloc(&buf, 1 << 20));
  hipStream_t m, s; CK(hipStreamCreate(&m)); CK(hipStreamCreate(&s));
  const int NK = 44, REP = 30;
  const unsigned flags[3] = {hipEventDisableTiming, hipEventDisableTiming | hipEventReleaseToDevice, hipEventDisableTiming | hipEventDisableSystemFence};
  const char* names[3] = {"DisableTiming                   ", "DisableTiming|ReleaseToDevice   ", "DisableTiming|DisableSystemFence"};
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  for (int iters : {200, 4000}) {
    for (int variant = -1; variant < 3; ++variant) {
      std::vector<hipEvent_t> ev(NK);
      for (auto& e : ev) CK(hipEventCreateWithFlags(&e, flags[variant < 0 ? 0 : variant]));
      auto step = [&]() {
        for (int k = 0; k < NK; ++k) {
          hipLaunchKernelGGL(work, dim3(1024), dim3(256), 0, m, buf, iters);
          if (variant >= 0 && (k & 1)) {
            hipEventRecord(ev[k], m);
            hipStreamWaitEvent(s, ev[k], 0);
            hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, buf + 131072, iters / 2);
          }
        }
      };
      for (int i = 0; i < 3; ++i) step();
      CK(hipDeviceSynchronize());
      float best = 1e9f;
      for (int r = 0; r < REP; ++r) {
        CK(hipEventRecord(t0, m)); step(); CK(hipEventRecord(t1, m)); CK(hipEventSynchronize(t1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, t0, t1)); if (ms < best) best = ms;
      }
      printf("kernel iters %4d, %s: main chain of %d kernels %.1f us (%.2f us per kernel)\n", iters, variant < 0 ? "no hand-offs                    " : names[variant], NK,
             best * 1e3, best * 1e3 / NK);
      for (auto& e : ev) CK(hipEventDestroy(e));
    }
  }
  // visibility
  const long n = 64L << 20;
  unsigned* big; CK(hipMalloc(&big, n * 4));
  unsigned long long* bad; CK(hipMalloc(&bad, 8));
  for (int variant = 0; variant < 3; ++variant) {
    hipEvent_t e, back; CK(hipEventCreateWithFlags(&e, flags[variant])); CK(hipEventCreateWithFlags(&back, flags[variant]));
    CK(hipMemset(bad, 0, 8)); CK(hipDeviceSynchronize());
    for (unsigned r = 1; r <= 200; ++r) {
      hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, m, big, n, r);
      CK(hipEventRecord(e, m));
      CK(hipStreamWaitEvent(s, e, 0));
      hipLaunchKernelGGL(check, dim3(2048), dim3(256), 0, s, big, n, r, bad);
      CK(hipEventRecord(back, s));                 // the producer may not overwrite the buffer before the consumer has read it
      CK(hipStreamWaitEvent(m, back, 0));
    }
    CK(hipDeviceSynchronize());
    unsigned long long h = 0; CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    printf("visibility, %s: %llu stale elements of %ld x 200\n", names[variant], h, n);
    CK(hipEventDestroy(e)); CK(hipEventDestroy(back));
  }
  return 0;
}

// What a stream hand-off costs the queue that records it (round 5): main queue = chain of dependent kernels, every 2nd one followed by a
// hand-off to a side queue (the side queue launches a kernel behind it).  Variants:  A no hand-offs;  B hipEventRecord(main) + hipStreamWaitEvent(side)
// (what the engine does);  C the producing kernel launched with hipExtLaunchKernelGGL(..., stopEvent) -- its own completion signal IS the event,
// no separate marker packet on the main queue --, then hipStreamWaitEvent(side).   GPU time of the main chain per variant (HIP events around it).
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/handoff_cost.hip -o scripts/micro/handoff_cost && ./scripts/micro/handoff_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void work(float* p, int iters) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  float a = p[i & 65535];
  for (int k = 0; k < iters; ++k) a = __builtin_fmaf(a, 1.0001f, 0.5f);
  p[i & 65535] = a;
}
int main() {
  float* buf; CK(hipMalloc(&buf, 1 << 20));
  hipStream_t m, s; CK(hipStreamCreate(&m)); CK(hipStreamCreate(&s));
  const int NK = 44, REP = 30;
  std::vector<hipEvent_t> ev(NK);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  for (int iters : {200, 4000}) {                          // ~2 us and ~30 us kernels
    for (int variant = 0; variant < 3; ++variant) {
      auto step = [&]() {
        for (int k = 0; k < NK; ++k) {
          const bool hand = variant > 0 && (k & 1);
          if (hand && variant == 2) hipExtLaunchKernelGGL(work, dim3(1024), dim3(256), 0, m, nullptr, ev[k], 0, buf, iters);
          else hipLaunchKernelGGL(work, dim3(1024), dim3(256), 0, m, buf, iters);
          if (hand) {
            if (variant == 1) hipEventRecord(ev[k], m);
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
      printf("kernel iters %4d, %s: main chain of %d kernels %.1f us (%.2f us per kernel)\n", iters,
             variant == 0 ? "no hand-offs                     " : variant == 1 ? "hipEventRecord + StreamWaitEvent " : "ext launch stopEvent + WaitEvent  ", NK, best * 1e3, best * 1e3 / NK);
    }
  }
  return 0;
}

// Host cost of enqueuing a 230-launch step: hipLaunchKernel loop vs hipGraphLaunch (round 5, review item 2).
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/launch_cost.hip -o /tmp/launch_cost && /tmp/launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void tiny(const float* a, const float* b, float* c, const float* d, const float* e, float* f, long n, int p0, int p1,
                     int p2, int p3, float s0, float s1, double cnt, unsigned long long seed, int us) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  float acc = 0.f;
  // ~us microseconds of dependent work per thread
  for (int k = 0; k < us * 60; ++k) acc = __builtin_fmaf(acc, 1.0001f, s0);
  if (i < n) c[i] = acc + (a ? a[i] : 0.f);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const int NL = 230, REP = 50;
  float* buf; CK(hipMalloc(&buf, 1 << 22));
  hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
  std::vector<hipEvent_t> evs(64);
  for (auto& e : evs) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (int us : {0, 10}) {
    auto step = [&](bool two) {
      int ei = 0;
      for (int l = 0; l < NL; ++l) {
        hipStream_t st = s;
        if (two && l % 8 == 7) {                      // every 8th launch goes to the side stream behind an event
          hipEventRecord(evs[ei], s); hipStreamWaitEvent(s2, evs[ei], 0); ei = (ei + 1) % 64; st = s2;
        }
        hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, st, buf, buf, buf, buf, buf, buf, 65536L, 1, 2, 3, 4, 1.f, 2.f, 3.0, 7ULL, us);
      }
      if (two) { hipEventRecord(evs[ei], s2); hipStreamWaitEvent(s, evs[ei], 0); }
    };
    for (int two = 0; two < 2; ++two) {
      for (int i = 0; i < 3; ++i) step(two);
      CK(hipStreamSynchronize(s));
      double t0 = now();
      for (int r = 0; r < REP; ++r) step(two);
      double t1 = now();
      CK(hipStreamSynchronize(s));
      double t2 = now();
      printf("kernel ~%2d us, %d stream(s): launch loop  host %.3f ms/step (%.2f us/launch), wall %.3f ms/step\n", us, two + 1,
             (t1 - t0) / REP * 1e3, (t1 - t0) / REP / NL * 1e6, (t2 - t0) / REP * 1e3);
      // the same step as a graph
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
      step(two);
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
      CK(hipStreamSynchronize(s));
      t0 = now();
      for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, s));
      t1 = now();
      CK(hipStreamSynchronize(s));
      t2 = now();
      printf("kernel ~%2d us, %d stream(s): hipGraphLaunch host %.3f ms/step, wall %.3f ms/step\n", us, two + 1,
             (t1 - t0) / REP * 1e3, (t2 - t0) / REP * 1e3);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
  }
  return 0;
}

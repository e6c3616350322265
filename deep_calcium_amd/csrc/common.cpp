// Error reporting, version and HIP-event timing helpers of libdcunet.
#include "common.h"
#include <stdlib.h>
#include <map>
#include <utility>

static thread_local char g_err[512] = "";

void dc_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const DcConfig& dc_config() {
  static const DcConfig cfg = [] {
    DcConfig c;
    const char* v = getenv("DC_IGEMM_PP");
    c.igemm_pp = v ? atoi(v) : 1;
    return c;
  }();
  return cfg;
}

int dc_stream_ws(hipStream_t stream, size_t bytes, void** out) {
  struct Buf { void* ptr; size_t size; };
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, Buf> table;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  DC_REQUIRE(e == hipSuccess, DC_EHIP, "dc_stream_ws: hipGetDevice: %s", hipGetErrorString(e));
  std::lock_guard<std::mutex> lock(mu);
  Buf& b = table[{dev, stream}];
  if (b.size < bytes) {
    if (b.ptr) {
      e = hipStreamSynchronize(stream);
      DC_REQUIRE(e == hipSuccess, DC_EHIP, "dc_stream_ws: hipStreamSynchronize: %s", hipGetErrorString(e));
      (void)hipFree(b.ptr);
      b.ptr = nullptr; b.size = 0;
    }
    const size_t want = (bytes + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
    e = hipMalloc(&b.ptr, want);
    DC_REQUIRE(e == hipSuccess, DC_EHIP, "dc_stream_ws: hipMalloc(%zu): %s", want, hipGetErrorString(e));
    b.size = want;
  }
  *out = b.ptr;
  return DC_OK;
}

extern "C" int dc_version(void) { return 101; }
extern "C" const char* dc_last_error(void) { return g_err; }

extern "C" int dc_event_create(void** ev) {
  DC_REQUIRE(ev, DC_EINVAL, "dc_event_create: null");
  hipEvent_t e;
  hipError_t rc = hipEventCreate(&e);
  DC_REQUIRE(rc == hipSuccess, DC_EHIP, "hipEventCreate: %s", hipGetErrorString(rc));
  *ev = (void*)e;
  return DC_OK;
}
extern "C" int dc_event_record(void* ev, dc_stream_t stream) {
  hipError_t rc = hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
  DC_REQUIRE(rc == hipSuccess, DC_EHIP, "hipEventRecord: %s", hipGetErrorString(rc));
  return DC_OK;
}
extern "C" int dc_event_elapsed_ms(void* start, void* stop, float* ms) {
  hipError_t rc = hipEventSynchronize((hipEvent_t)stop);
  DC_REQUIRE(rc == hipSuccess, DC_EHIP, "hipEventSynchronize: %s", hipGetErrorString(rc));
  rc = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
  DC_REQUIRE(rc == hipSuccess, DC_EHIP, "hipEventElapsedTime: %s", hipGetErrorString(rc));
  return DC_OK;
}
extern "C" int dc_event_destroy(void* ev) {
  (void)hipEventDestroy((hipEvent_t)ev);
  return DC_OK;
}

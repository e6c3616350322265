// Error reporting, version and HIP-event timing helpers of libdcunet.
#include "common.h"
#include <stdlib.h>

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

extern "C" int dc_version(void) { return DC_ABI_VERSION; }
extern "C" const char* dc_last_error(void) { return g_err; }

extern "C" int dc_event_create(void** ev) {
  DC_REQUIRE(ev, DC_EINVAL, "dc_event_create: null");
  hipEvent_t e;
  hipError_t rc = hipEventCreate(&e);
  DC_REQUIRE(rc == hipSuccess, DC_EHIP, "hipEventCreate: %s", hipGetErrorString(rc));
  *ev = (void*)e;
  return DC_OK;
}
static thread_local hipEvent_t g_bracket[2] = {nullptr, nullptr};
extern "C" int dc_bracket_next_launch(void* ev_before, void* ev_after) {
  g_bracket[0] = (hipEvent_t)ev_before;
  g_bracket[1] = (hipEvent_t)ev_after;
  return DC_OK;
}
bool dc_take_bracket(hipEvent_t* before, hipEvent_t* after) {
  if (!g_bracket[0] && !g_bracket[1]) return false;
  *before = g_bracket[0]; *after = g_bracket[1];
  g_bracket[0] = g_bracket[1] = nullptr;
  return true;
}
extern "C" int dc_event_create_sync(void** ev) {
  DC_REQUIRE(ev, DC_EINVAL, "dc_event_create_sync: null");
  hipEvent_t e;
  // Same-device stream ordering only: the producing kernel's own agent-scope release publishes its stores to the consumer's queue, the
  // marker's system-scope fence (for HOST visibility: nobody synchronises the host on these events) is skipped.  Measured
  // (scripts/micro/handoff_flags.hip, profiles/r05_ab.txt item 11): 5.7 -> 3.4 us lost by the recording queue per hand-off, no stale
  // element in 200 x 256 MB cross-queue hand-overs; hipEventReleaseToDevice alone changes nothing.  DC_EVENT_SYSTEM_FENCE=1 keeps the fence.
  static const bool fence = [] { const char* v = getenv("DC_EVENT_SYSTEM_FENCE"); return v && v[0] == '1'; }();
  hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming | (fence ? 0u : (unsigned)hipEventDisableSystemFence));
  DC_REQUIRE(rc == hipSuccess, DC_EHIP, "hipEventCreateWithFlags: %s", hipGetErrorString(rc));
  *ev = (void*)e;
  return DC_OK;
}
extern "C" int dc_event_create_fenced(void** ev) {
  DC_REQUIRE(ev, DC_EINVAL, "dc_event_create_fenced: null");
  hipEvent_t e;
  // Ordering in front of / behind a COLLECTIVE: the buffer's next reader (or last writer) is a peer GPU over xGMI, so the marker keeps its
  // system-scope release; ~2.3 us per record, three or four of them per data-parallel step.
  hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
  DC_REQUIRE(rc == hipSuccess, DC_EHIP, "hipEventCreateWithFlags: %s", hipGetErrorString(rc));
  *ev = (void*)e;
  return DC_OK;
}
extern "C" int dc_stream_wait_event(dc_stream_t stream, void* ev) {
  DC_REQUIRE(ev, DC_EINVAL, "dc_stream_wait_event: null event");
  hipError_t rc = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0);
  DC_REQUIRE(rc == hipSuccess, DC_EHIP, "hipStreamWaitEvent: %s", hipGetErrorString(rc));
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

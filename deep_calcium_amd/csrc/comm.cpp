// Gradient exchange through the C ABI: RCCL (librccl.so.1) behind dc_comm_* (include/dcunet.h; SURVEY 8b / 8e: one
// ncclAllReduce(sum, fp32) over the flat gradient buffer per step, optionally in buckets).  RCCL is resolved at FIRST USE with
// dlopen -- the library itself must load (and every other entry point work) on a box where nothing links RCCL -- and the copy
// the process already holds is taken when there is one: PyTorch-ROCm ships its own librccl.so.1, and two RCCL instances in
// one process would each bring their own bootstrap threads and IPC state.  Every call is an asynchronous enqueue on the
// caller's stream (tape-able like a kernel launch); nothing here synchronises the host or allocates device memory.
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>
#include <mutex>

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  char err[256] = "";
};
Rccl g_rccl;
std::once_flag g_once;

template <class F>
bool sym(void* h, const char* name, F& out) {
  out = reinterpret_cast<F>(dlsym(h, name));
  return out != nullptr;
}
void load_rccl() {
  Rccl& r = g_rccl;
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : names)                                  // the instance already in the process (torch's), if any
    if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
  for (const char* n : names)
    if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  if (!r.h) { snprintf(r.err, sizeof(r.err), "dlopen(librccl.so.1): %s", dlerror()); return; }
  const bool ok = sym(r.h, "ncclGetUniqueId", r.GetUniqueId) && sym(r.h, "ncclCommInitRank", r.CommInitRank) &&
                  sym(r.h, "ncclAllReduce", r.AllReduce) && sym(r.h, "ncclGroupStart", r.GroupStart) &&
                  sym(r.h, "ncclGroupEnd", r.GroupEnd) && sym(r.h, "ncclCommDestroy", r.CommDestroy) &&
                  sym(r.h, "ncclGetErrorString", r.GetErrorString);
  if (!ok) { snprintf(r.err, sizeof(r.err), "librccl.so.1 lacks an expected nccl* symbol"); r.h = nullptr; }
}
const Rccl* rccl() {
  std::call_once(g_once, load_rccl);
  return g_rccl.h ? &g_rccl : nullptr;
}
}  // namespace

#define DC_RCCL(r, fn)                                                            \
  const Rccl* r = rccl();                                                         \
  DC_REQUIRE(r, DC_EUNSUP, "%s: RCCL is not available: %s", fn, g_rccl.err)
#define DC_NCCL_OK(r, call, fn)                                                   \
  do {                                                                            \
    ncclResult_t rc_ = (call);                                                    \
    DC_REQUIRE(rc_ == ncclSuccess, DC_EHIP, "%s: %s", fn, r->GetErrorString(rc_)); \
  } while (0)

extern "C" int dc_comm_unique_id(void* id_host) {
  DC_REQUIRE(id_host, DC_EINVAL, "dc_comm_unique_id: null");
  static_assert(sizeof(ncclUniqueId) == DC_COMM_ID_BYTES, "DC_COMM_ID_BYTES is RCCL's NCCL_UNIQUE_ID_BYTES");
  DC_RCCL(r, "dc_comm_unique_id");
  ncclUniqueId id;
  DC_NCCL_OK(r, r->GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(id_host, &id, sizeof(id));
  return DC_OK;
}
extern "C" int dc_comm_init_rank(void** comm, const void* id_host, int nranks, int rank) {
  DC_REQUIRE(comm && id_host && nranks > 0 && rank >= 0 && rank < nranks, DC_EINVAL, "dc_comm_init_rank: bad arguments (rank %d of %d)", rank, nranks);
  DC_RCCL(r, "dc_comm_init_rank");
  ncclUniqueId id;
  memcpy(&id, id_host, sizeof(id));
  ncclComm_t c = nullptr;
  DC_NCCL_OK(r, r->CommInitRank(&c, nranks, id, rank), "ncclCommInitRank");      // binds the CURRENT device (the caller set it)
  *comm = (void*)c;
  return DC_OK;
}
// buf[0..n) <- sum over the communicator's ranks, in place, fp32; asynchronous on `stream`
extern "C" int dc_comm_all_reduce_sum(void* comm, float* buf, long n, dc_stream_t stream) {
  DC_REQUIRE(comm && buf && n > 0, DC_EINVAL, "dc_comm_all_reduce_sum: bad arguments");
  DC_RCCL(r, "dc_comm_all_reduce_sum");
  DC_NCCL_OK(r, r->AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream), "ncclAllReduce");
  return DC_OK;
}
// the same for doubles (the step's 12 loss / metric sums)
extern "C" int dc_comm_all_reduce_sum_f64(void* comm, double* buf, long n, dc_stream_t stream) {
  DC_REQUIRE(comm && buf && n > 0, DC_EINVAL, "dc_comm_all_reduce_sum_f64: bad arguments");
  DC_RCCL(r, "dc_comm_all_reduce_sum_f64");
  DC_NCCL_OK(r, r->AllReduce(buf, buf, (size_t)n, ncclFloat64, ncclSum, (ncclComm_t)comm, (hipStream_t)stream), "ncclAllReduce");
  return DC_OK;
}
extern "C" int dc_comm_group_start(void) {
  DC_RCCL(r, "dc_comm_group_start");
  DC_NCCL_OK(r, r->GroupStart(), "ncclGroupStart");
  return DC_OK;
}
extern "C" int dc_comm_group_end(void) {
  DC_RCCL(r, "dc_comm_group_end");
  DC_NCCL_OK(r, r->GroupEnd(), "ncclGroupEnd");
  return DC_OK;
}
extern "C" int dc_comm_destroy(void* comm) {
  const Rccl* r = rccl();
  if (r && comm) (void)r->CommDestroy((ncclComm_t)comm);
  return DC_OK;
}

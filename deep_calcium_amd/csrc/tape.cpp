// Launch tape: a training step's enqueue sequence replayed from C.
// The step of the reference's own training configuration (128^2 windows x batch 20, unet_2d_summary.py:333-335 via
// examples/neurons/unet2ds_nf.py:36-40) is ~230 launches of 5-50 us: issued one ctypes call at a time, the Python host spends
// 1.7-2.4 ms per 3-ms step on them, with the GIL held (fit()'s generator thread and 8 ranks on 16 cores wait for it).  The
// sequence of a steady-state step is FIXED -- same entry points, same pointers, same shapes --; only a handful of scalars
// (dropout seeds, Adam's lr_t, the batch pointers) change.  The engine records it once (deep_calcium_amd/_lib.py Tape), this
// file replays it: one C call per phase of the step, the recorded arguments handed to the very same extern "C" entry points
// through generated trampolines (tape_tramp.inc), the per-step scalars patched in from a small value table.
// Stream choreography (event record / stream wait) is recorded like any other entry point.  Host memory only.
#include "common.h"
#include <string.h>
#include <vector>

union DcArg {
  void* p;
  long l;
  double d;
  uint64_t u;
};
struct DcTapeFn {
  const char* name;
  int (*fn)(const DcArg*);
  int nargs;
  const char* kinds;      // per argument: p pointer / stream, l int / long, d float / double, u uint64
};
#define DC_TAPE_MAX_ARGS 32
#include "tape_tramp.inc"

namespace {
struct Op {
  const DcTapeFn* f;
  DcArg a[DC_TAPE_MAX_ARGS];
};
struct Patch { int op, arg, slot; };
struct Tape {
  std::vector<Op> ops;
  std::vector<Patch> patches;      // sorted by op (appended in op order by the binding)
};
const DcTapeFn* find_fn(const char* name) {
  for (const DcTapeFn& f : kTapeFns)
    if (strcmp(f.name, name) == 0) return &f;
  return nullptr;
}
}  // namespace

extern "C" int dc_tape_create(void** tape) {
  DC_REQUIRE(tape, DC_EINVAL, "dc_tape_create: null");
  *tape = new Tape();
  return DC_OK;
}
extern "C" int dc_tape_destroy(void* tape) {
  delete reinterpret_cast<Tape*>(tape);
  return DC_OK;
}
// args8: nargs raw 8-byte slots (integers and pointers as longs, float / double arguments as the bits of a double)
extern "C" int dc_tape_append(void* tape, const char* fn, const long* args8, int nargs) {
  DC_REQUIRE(tape && fn && (args8 || nargs == 0), DC_EINVAL, "dc_tape_append: null");
  const DcTapeFn* f = find_fn(fn);
  DC_REQUIRE(f, DC_EUNSUP, "dc_tape_append: %s is not a tape-able entry point", fn);
  DC_REQUIRE(nargs == f->nargs && nargs <= DC_TAPE_MAX_ARGS, DC_EINVAL, "dc_tape_append: %s takes %d arguments, got %d", fn, f->nargs, nargs);
  Op op;
  op.f = f;
  for (int i = 0; i < nargs; ++i) op.a[i].l = args8[i];
  reinterpret_cast<Tape*>(tape)->ops.push_back(op);
  return DC_OK;
}
// at replay, argument `arg` of operation `op` is values[slot] instead of the recorded one
extern "C" int dc_tape_patch(void* tape, int op, int arg, int slot) {
  Tape* t = reinterpret_cast<Tape*>(tape);
  DC_REQUIRE(t && op >= 0 && op < (int)t->ops.size() && arg >= 0 && arg < t->ops[op].f->nargs && slot >= 0, DC_EINVAL,
             "dc_tape_patch: bad (op %d, arg %d, slot %d)", op, arg, slot);
  DC_REQUIRE(t->patches.empty() || t->patches.back().op <= op, DC_EINVAL, "dc_tape_patch: patches must come in operation order");
  t->patches.push_back({op, arg, slot});
  return DC_OK;
}
// Replays operations [first, first + count): every one goes through its entry point's own argument checks and returns its
// status; the first failure stops the replay (dc_last_error() names the operation).
extern "C" int dc_tape_replay(void* tape, int first, int count, const long* values, int nvalues) {
  Tape* t = reinterpret_cast<Tape*>(tape);
  DC_REQUIRE(t && first >= 0 && count >= 0 && first + count <= (int)t->ops.size(), DC_EINVAL, "dc_tape_replay: bad range [%d, %d + %d) of %d",
             first, first, count, t ? (int)t->ops.size() : 0);
  size_t pi = 0;
  while (pi < t->patches.size() && t->patches[pi].op < first) ++pi;
  for (int k = first; k < first + count; ++k) {
    Op& op = t->ops[k];
    for (; pi < t->patches.size() && t->patches[pi].op == k; ++pi) {
      const Patch& pt = t->patches[pi];
      DC_REQUIRE(pt.slot < nvalues && values, DC_EINVAL, "dc_tape_replay: value slot %d of %d", pt.slot, nvalues);
      op.a[pt.arg].l = values[pt.slot];
    }
    const int rc = op.f->fn(op.a);
    if (rc != DC_OK) {
      char msg[400];
      snprintf(msg, sizeof(msg), "%s", dc_last_error());
      dc_set_error("dc_tape_replay: operation %d (%s) failed: %s", k, op.f->name, msg);
      return rc;
    }
  }
  return DC_OK;
}
extern "C" int dc_tape_len(void* tape) { return tape ? (int)reinterpret_cast<Tape*>(tape)->ops.size() : 0; }

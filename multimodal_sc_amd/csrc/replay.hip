// replay.hip — C-side replay of a recorded training step (host code only; no kernel lives here).
//
// One eager step of the path is a FIXED sequence of ~450 C-ABI calls: the same entry points with the same argument values
// (device addresses from the per-step bump arena, sizes, stream handles, event waits) every step; step and noise-draw numbers
// live in device memory (msc_counter_add), so nothing on the host changes between steps.  Python needs ~12 us per call to get
// there (autograd node, argument marshalling, ctypes); below ~40 frames per GPU that makes the step host-bound.  The host side
// (multimodal_sc_amd/replay.py) therefore records the calls of one eager step and this file replays them: one ctypes call per
// SEGMENT, each recorded call re-issued through the same extern "C" entry point on the same stream, in the recorded order.
// Segments end where Python must act between launches (the data-parallel all-reduces).
// Not a hipGraph: a graph replay of this step measured slower than eager because it serialises the captured branches
// (DESIGN.md §4); this replay keeps the three streams and their event waits exactly as the eager loop issues them.
#include <vector>
#include "msc_common.h"
#include "replay_dispatch.inc"

namespace {
struct Rec { int id, nargs; int64_t a[MSC_REPLAY_MAX_ARGS]; double f[MSC_REPLAY_MAX_ARGS]; };
struct Program {
  std::vector<Rec> recs;
  std::vector<size_t> seg_begin{0};     // seg_begin[s] .. seg_begin[s+1] (or recs.size()) = segment s
  int64_t last_failed = -1;             // index of the record whose call failed in the last run (-1: none)
};
}  // namespace

extern "C" int msc_replay_fn_id(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < MSC_REPLAY_NFN; ++i)
    if (!strcmp(name, msc_replay_names[i])) return i;
  return -1;
}
extern "C" int msc_replay_fn_nargs(int fn_id) { return (fn_id >= 0 && fn_id < MSC_REPLAY_NFN) ? msc_replay_nargs[fn_id] : -1; }

extern "C" int64_t msc_replay_create(void) { return (int64_t)(intptr_t) new (std::nothrow) Program(); }
extern "C" int msc_replay_destroy(int64_t prog) {
  delete (Program*)(intptr_t)prog;
  return MSC_OK;
}
// append one recorded call: iargs[j] = integer / pointer arguments (as int64), fargs[j] = float arguments (as double); the slot
// that is not the argument's kind is ignored
extern "C" int msc_replay_add(int64_t prog, int fn_id, const int64_t* iargs, const double* fargs, int nargs) {
  Program* p = (Program*)(intptr_t)prog;
  MSC_CHECK_ARG(p && iargs && fargs && fn_id >= 0 && fn_id < MSC_REPLAY_NFN && nargs == msc_replay_nargs[fn_id] && nargs <= MSC_REPLAY_MAX_ARGS);
  Rec r;
  r.id = fn_id;
  r.nargs = nargs;
  for (int j = 0; j < nargs; ++j) { r.a[j] = iargs[j]; r.f[j] = fargs[j]; }
  p->recs.push_back(r);
  return MSC_OK;
}
// close the current segment; returns the index of the segment that begins here
extern "C" int msc_replay_cut(int64_t prog) {
  Program* p = (Program*)(intptr_t)prog;
  if (!p) return MSC_ERR_BAD_ARG;
  p->seg_begin.push_back(p->recs.size());
  return (int)p->seg_begin.size() - 1;
}
extern "C" int64_t msc_replay_count(int64_t prog, int segment) {
  Program* p = (Program*)(intptr_t)prog;
  if (!p) return -1;
  if (segment < 0) return (int64_t)p->recs.size();
  if ((size_t)segment >= p->seg_begin.size()) return -1;
  const size_t e = (size_t)segment + 1 < p->seg_begin.size() ? p->seg_begin[segment + 1] : p->recs.size();
  return (int64_t)(e - p->seg_begin[segment]);
}
// re-issue segment `segment` (all segments when < 0); stops at the first failing call and returns its code
extern "C" int msc_replay_run(int64_t prog, int segment) {
  Program* p = (Program*)(intptr_t)prog;
  if (!p || (segment >= 0 && (size_t)segment >= p->seg_begin.size())) return MSC_ERR_BAD_ARG;
  const size_t b = segment < 0 ? 0 : p->seg_begin[segment];
  const size_t e = (segment < 0 || (size_t)segment + 1 >= p->seg_begin.size()) ? p->recs.size() : p->seg_begin[segment + 1];
  p->last_failed = -1;
  for (size_t i = b; i < e; ++i) {
    const Rec& r = p->recs[i];
    const int rc = msc_replay_dispatch(r.id, r.a, r.f);
    if (rc != MSC_OK) { p->last_failed = (int64_t)i; return rc; }
  }
  return MSC_OK;
}
extern "C" int64_t msc_replay_last_failed(int64_t prog) {
  Program* p = (Program*)(intptr_t)prog;
  return p ? p->last_failed : -1;
}

// abi.hip — host-only entry points of the C-ABI: error strings, version / build id, stream and event helpers.
// The Makefile rebuilds this file whenever any kernel source or include/msc.h changes and embeds the hash of all of them
// (MSC_BUILD_ID), so msc_build_id() identifies the sources a loaded libmsc_hip.so was built from.
#include "msc_common.h"

extern "C" const char* msc_strerror(int code) {
  switch (code) {
    case MSC_OK: return "ok";
    case MSC_ERR_BAD_ARG: return "bad argument (null pointer, size or alignment the kernel does not support)";
    case MSC_ERR_LAUNCH: return "HIP launch failed";
    case MSC_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown error";
  }
}
extern "C" int msc_version(void) { return 300; }
extern "C" int msc_abi_version(void) { return MSC_ABI_VERSION; }
#ifndef MSC_BUILD_ID
#define MSC_BUILD_ID "unknown"
#endif
extern "C" const char* msc_build_id(void) { return MSC_BUILD_ID; }

// A non-blocking HIP stream at a priority level relative to the device's range: -1 lowest, 0 default, +1 highest.
// Returns the hipStream_t as an integer (0 on failure).  The side stream of the training step (weight gradients, bias
// column sums) is created at the lowest level so the dgrad / normalisation chain wins the CUs when both have work.
extern "C" int64_t msc_stream_create(int level) {
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return 0;
  const int prio = level < 0 ? least : (level > 0 ? greatest : 0);
  hipStream_t st = nullptr;
  if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio) != hipSuccess) return 0;
  return (int64_t)(intptr_t)st;
}
// One reusable event + "waiter waits for everything enqueued so far on signaler" (event record + stream wait).
extern "C" int64_t msc_event_create(void) {
  hipEvent_t ev = nullptr;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return 0;
  return (int64_t)(intptr_t)ev;
}
extern "C" int msc_stream_wait_stream(void* waiter, void* signaler, int64_t event) {
  hipEvent_t ev = (hipEvent_t)(intptr_t)event;
  if (!ev) return MSC_ERR_BAD_ARG;
  if (hipEventRecord(ev, (hipStream_t)signaler) != hipSuccess) return MSC_ERR_LAUNCH;
  if (hipStreamWaitEvent((hipStream_t)waiter, ev, 0) != hipSuccess) return MSC_ERR_LAUNCH;
  return MSC_OK;
}
// host -> device staging of the next batch of frames: hipMemcpyAsync from PINNED host memory on `stream` (the copy engine moves it
// while the compute streams run; ordered on `stream` like a kernel).  src_host must be page-locked for the copy to be asynchronous.
extern "C" int msc_memcpy_h2d_async(void* dst, const void* src_host, int64_t bytes, void* stream) {
  MSC_CHECK_ARG(dst && src_host && bytes > 0);
  return hipMemcpyAsync(dst, src_host, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream) == hipSuccess ? MSC_OK : MSC_ERR_LAUNCH;
}
// the two halves of msc_stream_wait_stream for a dependency that is signalled now and waited for later (the staged batch: the copy
// is enqueued a step ahead; the consumer must not wait for what the signalling stream was given AFTER the copy)
extern "C" int msc_event_record(int64_t event, void* stream) {
  hipEvent_t ev = (hipEvent_t)(intptr_t)event;
  if (!ev) return MSC_ERR_BAD_ARG;
  return hipEventRecord(ev, (hipStream_t)stream) == hipSuccess ? MSC_OK : MSC_ERR_LAUNCH;
}
// HOST-side wait (the one blocking call of the ABI): returns once the work recorded with `event` has completed — the frame stager's
// filler thread calls it before it overwrites a pinned buffer whose copy may still be in flight.  Never called from the step.
extern "C" int msc_event_synchronize(int64_t event) {
  hipEvent_t ev = (hipEvent_t)(intptr_t)event;
  if (!ev) return MSC_ERR_BAD_ARG;
  return hipEventSynchronize(ev) == hipSuccess ? MSC_OK : MSC_ERR_LAUNCH;
}
extern "C" int msc_stream_wait_event(void* waiter, int64_t event) {
  hipEvent_t ev = (hipEvent_t)(intptr_t)event;
  if (!ev) return MSC_ERR_BAD_ARG;
  return hipStreamWaitEvent((hipStream_t)waiter, ev, 0) == hipSuccess ? MSC_OK : MSC_ERR_LAUNCH;
}
extern "C" int msc_stream_destroy(int64_t stream) {
  return hipStreamDestroy((hipStream_t)(intptr_t)stream) == hipSuccess ? MSC_OK : MSC_ERR_BAD_ARG;
}

// fgmm_device.h — the device layer under the host orchestration (fgmm_encode.cpp, fgmm_decode*.cpp, fgmm_capi.cpp).
//
// Everything the orchestration asks of the GPU runtime goes through these few functions - memory, streams, events, async
// copies - plus the kernel launchers declared in fgmm_internal.h.  Two implementations exist, chosen at LINK time:
//   fgmm_device_hip.cpp          the product: thin forwards to the HIP runtime (libflashgmm_amd.so)
//   tests/fake/fake_device.cpp   test infrastructure: "device" memory is host memory, every stream is an in-order queue run by
//                                its own thread with random delays, events are flags, the kernels are restated on the CPU from the
//                                oracle - so that the whole concurrent pipeline (planner, staging, task queue, event waits,
//                                re-runs) runs WITHOUT a GPU under ThreadSanitizer / AddressSanitizer (scripts/tsan_host.sh)
// Semantics are HIP's: work on one stream runs in order, streams are independent unless joined by stream_wait_event, an event
// completes when everything enqueued on its stream before event_record has, event_sync on a never-recorded event returns at once.
// Every function returns 0 or a backend error code (error_string gives the text).
#pragma once
#include <stddef.h>

namespace fgmm {
namespace dev {

using Stream = void *; // hipStream_t; nullptr = the default stream
using Event = void *;  // hipEvent_t

enum CopyKind { kH2D = 1, kD2H = 2, kD2D = 3 };
enum EventFlags { kEventTiming = 1, kEventBlocking = 2 }; // blocking: a waiter sleeps (interrupt) instead of polling

int device_count(int *n);
int get_device(int *d);
int set_device(int d);

int malloc_device(void **p, size_t bytes);
int free_device(void *p);
int malloc_pinned(void **p, size_t bytes);
int free_pinned(void *p);
int mem_info(size_t *free_bytes, size_t *total_bytes);

int stream_create(Stream *s, bool high_priority); // non-blocking w.r.t. the default stream
int stream_destroy(Stream s);
int stream_sync(Stream s);
int stream_wait_event(Stream s, Event e);

int event_create(Event *e, int flags);
int event_destroy(Event e);
int event_record(Event e, Stream s);
int event_sync(Event e);
int event_elapsed_ms(float *ms, Event begin, Event end);

int copy_async(void *dst, const void *src, size_t bytes, CopyKind kind, Stream s);
int copy_sync(void *dst, const void *src, size_t bytes, CopyKind kind); // returns with the copy complete
int memset_async(void *p, int value, size_t bytes, Stream s);

const char *error_string(int e);

} // namespace dev
} // namespace fgmm

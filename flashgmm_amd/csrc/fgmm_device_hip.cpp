// fgmm_device_hip.cpp — the device layer of the product (fgmm_device.h): thin forwards to the HIP runtime.
#include <hip/hip_runtime.h>

#include "fgmm_device.h"

namespace fgmm {
namespace dev {

int device_count(int *n) { return (int)hipGetDeviceCount(n); }
int get_device(int *d) { return (int)hipGetDevice(d); }
int set_device(int d) { return (int)hipSetDevice(d); }

int malloc_device(void **p, size_t bytes) { return (int)hipMalloc(p, bytes); }
int free_device(void *p) { return (int)hipFree(p); }
int malloc_pinned(void **p, size_t bytes) { return (int)hipHostMalloc(p, bytes, hipHostMallocDefault); }
int free_pinned(void *p) { return (int)hipHostFree(p); }
int mem_info(size_t *free_bytes, size_t *total_bytes) { return (int)hipMemGetInfo(free_bytes, total_bytes); }

int stream_create(Stream *s, bool high_priority) {
  hipStream_t st = nullptr;
  hipError_t e;
  if (high_priority) {
    // the table copies are shader copies on this runtime: they share the CUs with the table kernels of the later launches, and
    // PCIe - the longest leg of a decode call - must not wait for a CU (10.05 against 10.17 ms per step at the default priority)
    int lo = 0, hi = 0;
    if ((e = hipDeviceGetStreamPriorityRange(&lo, &hi)) != hipSuccess) return (int)e;
    e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi);
  } else {
    e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  }
  *s = st;
  return (int)e;
}
int stream_destroy(Stream s) { return (int)hipStreamDestroy((hipStream_t)s); }
int stream_sync(Stream s) { return (int)hipStreamSynchronize((hipStream_t)s); }
int stream_wait_event(Stream s, Event e) { return (int)hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0); }

int event_create(Event *e, int flags) {
  hipEvent_t ev = nullptr;
  unsigned f = (flags & kEventTiming) ? hipEventDefault : hipEventDisableTiming;
  if (flags & kEventBlocking) f |= hipEventBlockingSync;
  const hipError_t rc = hipEventCreateWithFlags(&ev, f);
  *e = ev;
  return (int)rc;
}
int event_destroy(Event e) { return (int)hipEventDestroy((hipEvent_t)e); }
int event_record(Event e, Stream s) { return (int)hipEventRecord((hipEvent_t)e, (hipStream_t)s); }
int event_sync(Event e) { return (int)hipEventSynchronize((hipEvent_t)e); }
int event_elapsed_ms(float *ms, Event begin, Event end) { return (int)hipEventElapsedTime(ms, (hipEvent_t)begin, (hipEvent_t)end); }

static hipMemcpyKind kind_of(CopyKind k) { return k == kH2D ? hipMemcpyHostToDevice : k == kD2H ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice; }
int copy_async(void *dst, const void *src, size_t bytes, CopyKind kind, Stream s) {
  return (int)hipMemcpyAsync(dst, src, bytes, kind_of(kind), (hipStream_t)s);
}
int copy_sync(void *dst, const void *src, size_t bytes, CopyKind kind) { return (int)hipMemcpy(dst, src, bytes, kind_of(kind)); }
int memset_async(void *p, int value, size_t bytes, Stream s) { return (int)hipMemsetAsync(p, value, bytes, (hipStream_t)s); }

const char *error_string(int e) { return hipGetErrorString((hipError_t)e); }

} // namespace dev
} // namespace fgmm

// fgmm_capi.cpp — the C ABI of libflashgmm_amd.so (include/flashgmm_amd.h): context, staging, orchestration.
//
// One fgmm_ctx per process per GPU owns
//   * a device workspace and a pinned host staging area (both grow on demand and are then reused),
//   * HIP events used to hand finished tables to the host coder item by item,
//   * a pool of host worker threads, one rANS state machine per bitstream.
// The float work is enqueued for ALL items of a call first (batched kernels, blockIdx.z = item), the tables
// come back by pinned hipMemcpyAsync, and the workers start on item i as soon as its copy has landed, so the
// PCIe transfer of item i+1 overlaps the host coding of item i.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <queue>
#include <thread>
#include <vector>

#include "../../include/flashgmm_amd.h"
#include "fgmm_internal.h"

using namespace fgmm;

namespace {

thread_local char t_err[512] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(t_err, sizeof t_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                                       \
  do {                                                                                                      \
    hipError_t e_ = (expr);                                                                                 \
    if (e_ != hipSuccess) return fail(FGMM_ERR_HIP, "%s -> %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

#define LAUNCH_TRY(expr)                                                                                    \
  do {                                                                                                      \
    int e_ = (expr);                                                                                        \
    if (e_ != 0) return fail(FGMM_ERR_HIP, "%s -> %s", #expr, hipGetErrorString((hipError_t)e_));          \
  } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// FGMM_TRACE=1: phase timestamps of every batched call on stderr (development aid)
struct Trace {
  bool on;
  std::chrono::steady_clock::time_point t0, last;
  const char *what;
  explicit Trace(const char *w) : what(w) {
    static const bool enabled = getenv("FGMM_TRACE") && atoi(getenv("FGMM_TRACE")) > 0;
    on = enabled;
    if (on) t0 = last = std::chrono::steady_clock::now();
  }
  void mark(const char *phase) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[fgmm %s] %-28s +%8.3f ms  (t=%8.3f)\n", what, phase,
            std::chrono::duration<double, std::milli>(now - last).count(),
            std::chrono::duration<double, std::milli>(now - t0).count());
    last = now;
  }
  double ms() const { return on ? std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() : 0.0; }
};

// ---- host worker pool ----------------------------------------------------------------------------------
class Pool {
public:
  explicit Pool(int n) {
    for (int i = 0; i < n; ++i) th_.emplace_back([this] { run(); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> l(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  int size() const { return (int)th_.size(); }
  void submit(std::function<void()> f) {
    {
      std::lock_guard<std::mutex> l(m_);
      q_.push(std::move(f));
      ++pending_;
    }
    cv_.notify_one();
  }
  void wait_all() {
    std::unique_lock<std::mutex> l(m_);
    done_cv_.wait(l, [this] { return pending_ == 0; });
  }

private:
  void run() {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [this] { return stop_ || !q_.empty(); });
        if (stop_ && q_.empty()) return;
        f = std::move(q_.front());
        q_.pop();
      }
      f();
      {
        std::lock_guard<std::mutex> l(m_);
        if (--pending_ == 0) done_cv_.notify_all();
      }
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_cv_;
  std::queue<std::function<void()>> q_;
  int pending_ = 0;
  bool stop_ = false;
};

// waits for every submitted job before the enclosing scope is left (jobs reference locals of that scope)
struct PoolDrain {
  Pool *p;
  ~PoolDrain() { p->wait_all(); }
};

// bump allocator over one device buffer + one pinned host buffer with identical offsets
struct Arena {
  size_t off = 0;
  size_t take(size_t bytes, size_t align = 256) {
    off = align_up(off, align);
    const size_t o = off;
    off += bytes;
    return o;
  }
};

} // namespace

struct fgmm_ctx {
  int device = 0;
  std::mutex mu; // one call at a time per context
  Pool *pool = nullptr;
  char *d_ws = nullptr;
  size_t d_cap = 0;
  char *h_ws = nullptr; // pinned
  size_t h_cap = 0;
  std::vector<hipEvent_t> events;
  hipStream_t fill_stream = nullptr; // decode: fill passes (the count passes of later groups run beside them)
  hipStream_t copy_stream = nullptr; // bulk D2H of the decode tables (overlaps the table kernels of later groups)
  hipStream_t aux_stream = nullptr;  // the few bytes of per-group pool counters
  // pinned receive area of the decode tables: a list of chunks, bump-allocated per call, never moved while
  // copies are in flight (sizes are only known group by group)
  struct Chunk {
    char *p;
    size_t cap, used;
  };
  std::vector<Chunk> chunks;
  // device staging area of the decode rows (same scheme): the fill pass writes there, one copy per group fetches it
  std::vector<Chunk> dchunks;
  void chunks_reset() {
    for (auto &c : chunks) c.used = 0;
    for (auto &c : dchunks) c.used = 0;
  }
  char *d_tmp = nullptr; // decode: the edges the count passes evaluated, read back by the fill passes
  size_t d_tmp_cap = 0;
  int ensure_tmp(size_t bytes) {
    if (bytes <= d_tmp_cap) return FGMM_OK;
    if (d_tmp) HIP_TRY(hipFree(d_tmp));
    d_tmp = nullptr;
    d_tmp_cap = 0;
    HIP_TRY(hipMalloc((void **)&d_tmp, bytes));
    d_tmp_cap = bytes;
    return FGMM_OK;
  }
  int dchunk_alloc(size_t bytes, char **out) {
    bytes = align_up(bytes, 256);
    for (auto &c : dchunks)
      if (c.cap - c.used >= bytes) {
        *out = c.p + c.used;
        c.used += bytes;
        return FGMM_OK;
      }
    Chunk c{nullptr, std::max(bytes, (size_t)256 << 20), 0};
    HIP_TRY(hipMalloc((void **)&c.p, c.cap));
    c.used = bytes;
    dchunks.push_back(c);
    *out = c.p;
    return FGMM_OK;
  }
  int chunk_alloc(size_t bytes, char **out) {
    bytes = align_up(bytes, 256);
    for (auto &c : chunks)
      if (c.cap - c.used >= bytes) {
        *out = c.p + c.used;
        c.used += bytes;
        return FGMM_OK;
      }
    Chunk c{nullptr, std::max(bytes, (size_t)256 << 20), 0};
    HIP_TRY(hipHostMalloc((void **)&c.p, c.cap, hipHostMallocDefault));
    c.used = bytes;
    chunks.push_back(c);
    *out = c.p;
    return FGMM_OK;
  }
  int ensure_streams() {
    if (!fill_stream) HIP_TRY(hipStreamCreateWithFlags(&fill_stream, hipStreamNonBlocking));
    if (!copy_stream) HIP_TRY(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    if (!aux_stream) HIP_TRY(hipStreamCreateWithFlags(&aux_stream, hipStreamNonBlocking));
    return FGMM_OK;
  }
  bool profiling = false;
  unsigned long long stat[4] = {0, 0, 0, 0}; // last batched call: [0] encode table bytes D2H, [1] decode hdr+row bytes D2H,
                                             // [2] decode latents, [3] decode rows that are Elias-Fano coded (unused: 0)
  hipEvent_t prof[4][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
  bool prof_valid[4] = {false, false, false, false};

  int prof_begin(int which, hipStream_t s) {
    if (!profiling) return FGMM_OK;
    HIP_TRY(hipEventRecord(prof[which][0], s));
    return FGMM_OK;
  }
  int prof_end(int which, hipStream_t s) {
    if (!profiling) return FGMM_OK;
    HIP_TRY(hipEventRecord(prof[which][1], s));
    prof_valid[which] = true;
    return FGMM_OK;
  }

  int ensure_device(size_t bytes) {
    if (bytes <= d_cap) return FGMM_OK;
    if (d_ws) HIP_TRY(hipFree(d_ws));
    d_ws = nullptr;
    d_cap = 0;
    const size_t want = align_up(bytes + bytes / 4, 1 << 20);
    HIP_TRY(hipMalloc((void **)&d_ws, want));
    d_cap = want;
    return FGMM_OK;
  }
  int ensure_host(size_t bytes) {
    if (bytes <= h_cap) return FGMM_OK;
    if (h_ws) HIP_TRY(hipHostFree(h_ws));
    h_ws = nullptr;
    h_cap = 0;
    const size_t want = align_up(bytes + bytes / 4, 1 << 20);
    HIP_TRY(hipHostMalloc((void **)&h_ws, want, hipHostMallocDefault));
    h_cap = want;
    return FGMM_OK;
  }
  int ensure_events(size_t n) {
    while (events.size() < n) {
      hipEvent_t e;
      HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      events.push_back(e);
    }
    return FGMM_OK;
  }
};

namespace {

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && (prev == dev || hipSetDevice(dev) == hipSuccess)) ok = true;
  }
  ~DeviceGuard() {
    int cur;
    if (ok && prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};

bool mode_ok(int mode) { return mode >= 0 && mode <= 2; }

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// can item use the 16-B-per-lane symtab kernel?
bool enc_vec4_ok(const EncDesc &d, bool f16) {
  const uintptr_t pm = f16 ? 7 : 15; // 4 parameters per load: 8 B (fp16) or 16 B (fp32)
  auto al = [pm](const void *p) { return (reinterpret_cast<uintptr_t>(p) & pm) == 0; };
  return d.stride_p == 1 && (d.hw & 3) == 0 && (d.stride_c & 3) == 0 && (d.stride_k & 3) == 0 && al(d.scales) &&
         al(d.means) && al(d.weights) && (d.y ? aligned16(d.y) : aligned16(d.sym)) && aligned16(d.packed);
}

// ---------------------------------------------------------------------------------------------------------
// encode, batched.  Items are described in the latent-codec layout (fgmm_item) or as raw (n,K) rows.
// ---------------------------------------------------------------------------------------------------------
struct EncItem {
  // inputs
  const float *y = nullptr;      // device
  const int32_t *sym_dev = nullptr; // device (raw boundary)
  const int32_t *sym_host = nullptr; // host copy of the raw symbols when the caller has one
  fgmm_params prm{};
  int64_t stride_p = 1;
  int32_t M = 0;
  int64_t hw = 0;
  int clamp = 0;
  float *yq = nullptr; // device out
  fgmm_symbuf *symbuf = nullptr; // raw boundary, buffered form: append the symbols instead of flushing a stream
  // outputs
  int64_t *zero_bitmap = nullptr; // host [M] or null
  int32_t abs_max = 0;
  uint8_t *bytes = nullptr;
  size_t bytes_len = 0;
  int status = FGMM_OK;
  // workspace offsets
  size_t o_min = 0, o_max = 0, o_nz = 0, o_list = 0, o_meta = 0, o_packed = 0, meta_count = 0;
  // what the item's host job needs (set when its side information has been read)
  const int32_t *job_syms = nullptr;
  int64_t job_n = 0, job_bypass = 0;
  double t_sub = 0, t_start = 0, t_end = 0; // FGMM_TRACE=2: job timeline
};

int encode_batch(fgmm_ctx *ctx, hipStream_t stream, std::vector<EncItem> &items, int mode) {
  const int count = (int)items.size();
  if (count == 0) return FGMM_OK;
  Trace tr("encode");
  // ---- plan the workspace: [descs][small: per item min|max|nz|meta][tables: per item packed] -------
  Arena ar;
  const size_t o_descs = ar.take(sizeof(EncDesc) * count);
  const size_t o_small = ar.take(0);
  int M_max = 0;
  int64_t hw_max = 0;
  for (auto &it : items) {
    it.o_min = ar.take(sizeof(float) * it.M, 16);
    it.o_max = ar.take(sizeof(float) * it.M, 16);
    it.o_nz = ar.take(sizeof(int32_t) * it.M, 16);
    it.o_list = ar.take(sizeof(int32_t) * ((size_t)it.M + 1), 16);
    it.meta_count = (size_t)it.M * (size_t)((it.hw + 255) / 256) * 4; // one slot per wave, sized for the 1-symbol-per-lane form
    it.o_meta = ar.take(sizeof(uint32_t) * it.meta_count, 16);
    M_max = std::max(M_max, it.M);
    hw_max = std::max(hw_max, it.hw);
  }
  const size_t small_bytes = ar.off - o_small;
  for (auto &it : items) it.o_packed = ar.take(sizeof(uint32_t) * (size_t)it.M * (size_t)it.hw + 64);
  const size_t total = ar.off;
  int rc;
  if ((rc = ctx->ensure_device(total)) || (rc = ctx->ensure_host(total)) || (rc = ctx->ensure_events(count + 1))) return rc;

  // ---- descriptors ------------------------------------------------------------------------------
  EncDesc *hd = reinterpret_cast<EncDesc *>(ctx->h_ws + o_descs);
  bool vec4 = true, any_y = false;
  for (int i = 0; i < count; ++i) {
    const EncItem &it = items[i];
    EncDesc &d = hd[i];
    memset(&d, 0, sizeof d);
    d.y = it.y;
    d.sym = it.sym_dev;
    d.scales = it.prm.scales;
    d.means = it.prm.means;
    d.weights = it.prm.weights;
    d.stride_k = it.prm.stride_k;
    d.stride_c = it.prm.stride_c;
    d.stride_p = it.stride_p;
    d.hw = it.hw;
    d.M = it.M;
    d.clamp = it.clamp;
    d.yq = it.yq;
    d.chan_min = reinterpret_cast<float *>(ctx->d_ws + it.o_min);
    d.chan_max = reinterpret_cast<float *>(ctx->d_ws + it.o_max);
    d.chan_nz = it.y ? reinterpret_cast<int32_t *>(ctx->d_ws + it.o_nz) : nullptr;
    d.chan_list = it.y ? reinterpret_cast<int32_t *>(ctx->d_ws + it.o_list) : nullptr;
    d.packed = reinterpret_cast<uint32_t *>(ctx->d_ws + it.o_packed);
    d.meta = reinterpret_cast<uint32_t *>(ctx->d_ws + it.o_meta);
    vec4 = vec4 && enc_vec4_ok(d, it.prm.dtype == FGMM_F16);
    any_y = any_y || it.y;
  }
  HIP_TRY(hipMemcpyAsync(ctx->d_ws + o_descs, hd, sizeof(EncDesc) * count, hipMemcpyHostToDevice, stream));
  HIP_TRY(hipMemsetAsync(ctx->d_ws + o_small, 0, small_bytes, stream));
  const EncDesc *dd = reinterpret_cast<const EncDesc *>(ctx->d_ws + o_descs);
  // a batch is homogeneous by construction: all latent-layout items (y given) or one raw (n,K) item
  if (any_y) {
    if ((rc = ctx->prof_begin(2, stream))) return rc;
    LAUNCH_TRY(launch_quant_stats(dd, count, M_max, stream));
    if ((rc = ctx->prof_end(2, stream))) return rc;
  }
  if ((rc = ctx->prof_begin(0, stream))) return rc;
  static const int force_vec = getenv("FGMM_VEC") ? atoi(getenv("FGMM_VEC")) : 0; // dev: A/B the load width
  const int vec = vec4 ? (force_vec ? force_vec : 4) : 1;
  static const int no_linear = getenv("FGMM_NO_LINEAR") ? atoi(getenv("FGMM_NO_LINEAR")) : 0; // dev: A/B the two grid forms
  int64_t n_max = 0;
  bool linear = !no_linear;
  for (auto &it : items) {
    n_max = std::max(n_max, (int64_t)it.M * it.hw);
    linear = linear && it.hw % (64 * vec) == 0;
  }
  LAUNCH_TRY(launch_symtab(dd, count, M_max, hw_max, n_max, linear, mode, vec, items[0].clamp != 0,
                           items[0].prm.dtype == FGMM_F16, stream));
  if ((rc = ctx->prof_end(0, stream))) return rc;
  // ---- tables back to the host: small region first, then one copy + event per item ----------------
  HIP_TRY(hipMemcpyAsync(ctx->h_ws + o_small, ctx->d_ws + o_small, small_bytes, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipEventRecord(ctx->events[count], stream));
  // the per-item tables are contiguous in the workspace: a handful of large copies instead of one per item
  const int group_size = count >= 16 ? std::max(2, count / 6) : 1;
  std::vector<int> group_of(count);
  int n_groups = 0;
  for (int i0 = 0; i0 < count; i0 += group_size, ++n_groups) {
    const int i1 = std::min(count, i0 + group_size);
    const size_t beg = items[i0].o_packed;
    const size_t end = items[i1 - 1].o_packed + sizeof(uint32_t) * (size_t)items[i1 - 1].M * (size_t)items[i1 - 1].hw;
    if (end > beg) HIP_TRY(hipMemcpyAsync(ctx->h_ws + beg, ctx->d_ws + beg, end - beg, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipEventRecord(ctx->events[n_groups], stream));
    for (int i = i0; i < i1; ++i) group_of[i] = n_groups;
  }
  tr.mark("enqueued");
  HIP_TRY(hipEventSynchronize(ctx->events[count]));
  tr.mark("kernels + meta landed");
  ctx->stat[0] = 0;
  for (auto &it : items) ctx->stat[0] += sizeof(uint32_t) * (unsigned long long)it.M * (unsigned long long)it.hw;

  // ---- host side: per item side information, then one rANS job per item -----------------------------
  std::vector<std::vector<int32_t>> wide_syms(count); // only for bypass symbols beyond int16 (rare)
  PoolDrain drain{ctx->pool};
  for (int i = 0; i < count; ++i) {
    EncItem &it = items[i];
    int64_t n = (int64_t)it.M * it.hw;
    unsigned long long n_bypass = 0;
    for (size_t k = 0; k < it.meta_count; ++k) n_bypass += reinterpret_cast<const uint32_t *>(ctx->h_ws + it.o_meta)[k];
    const int32_t *syms_for_bypass = it.sym_host;
    if (it.y) {
      const float *mn = reinterpret_cast<const float *>(ctx->h_ws + it.o_min);
      const float *mx = reinterpret_cast<const float *>(ctx->h_ws + it.o_max);
      const int32_t *nz = reinterpret_cast<const int32_t *>(ctx->h_ws + it.o_nz);
      float gmin = INFINITY, gmax = -INFINITY;
      int n_nz = 0;
      for (int c = 0; c < it.M; ++c) {
        gmin = fminf(gmin, mn[c]);
        gmax = fmaxf(gmax, mx[c]);
        n_nz += nz[c] != 0;
        if (it.zero_bitmap) it.zero_bitmap[c] = nz[c] != 0;
      }
      // max(torch.abs(y.max()).int(), torch.abs(y.min()).int()) + 1, floored at 1   (entropy_models.py:834-837)
      auto trunc_abs = [](float v) -> int64_t {
        const float a = fabsf(v);
        if (!(a < 2147483648.0f)) return INT32_MIN; // torch .int() of an out-of-range float: x86 cvttss2si
        return (int64_t)(int32_t)a;
      };
      int64_t am = (it.M * it.hw) ? std::max(trunc_abs(gmax), trunc_abs(gmin)) + 1 : 1;
      if (am < 1) am = 1;
      it.abs_max = (int32_t)am;
      n = (int64_t)n_nz * it.hw;
      if (n_bypass && am > 32767) {
        // a bypassed symbol may not fit the 16 bits the table carries: fetch the GPU-rounded latents (y_q, written
        // by quant_stats_kernel) and convert them to the int32 symbols — an integer conversion, no arithmetic.
        // Without a y_q buffer the raw latents are fetched and rounded to nearest-even here (rintf semantics).
        std::vector<float> yv((size_t)it.M * it.hw);
        HIP_TRY(hipMemcpy(yv.data(), it.yq ? it.yq : it.y, sizeof(float) * yv.size(), hipMemcpyDeviceToHost));
        wide_syms[i].reserve((size_t)n);
        for (int c = 0; c < it.M; ++c)
          if (nz[c])
            for (int64_t p = 0; p < it.hw; ++p) {
              const float v = yv[(size_t)c * it.hw + p];
              wide_syms[i].push_back((int32_t)(it.yq ? v : nearbyintf(v)));
            }
        syms_for_bypass = wide_syms[i].data();
      }
    } else if (n_bypass && !it.sym_host) {
      wide_syms[i].resize((size_t)n);
      HIP_TRY(hipMemcpy(wide_syms[i].data(), it.sym_dev, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
      syms_for_bypass = wide_syms[i].data();
    }
    it.job_syms = syms_for_bypass;
    it.job_n = n;
    it.job_bypass = (int64_t)n_bypass;
    // More bitstreams than workers: consecutive items go to one worker two at a time, coded in turn symbol by symbol
    // (rans_encode_symtab2) — 48 streams on 16 threads are then two rounds of a pair (2 x 1.5 ns/symbol) instead of
    // three rounds of a single stream (2.4 ns/symbol).
    const bool pairing = count > ctx->pool->size() && !it.symbuf;
    if (pairing && (i & 1) == 0 && i + 1 < count && !items[i + 1].symbuf) continue; // submitted together with item i + 1
    const bool pair = pairing && (i & 1) == 1 && !items[i - 1].symbuf;
    HIP_TRY(hipEventSynchronize(ctx->events[group_of[i]])); // item i - 1 lies in the same or an earlier copy
    const char *h_ws = ctx->h_ws;
    EncItem *pa = pair ? &items[i - 1] : &it, *pb = pair ? &it : nullptr;
    pa->t_sub = tr.ms();
    if (pb) pb->t_sub = pa->t_sub;
    auto job = [pa, pb, h_ws, &tr] {
      pa->t_start = tr.ms();
      const uint32_t *ta = reinterpret_cast<const uint32_t *>(h_ws + pa->o_packed);
      if (pb) {
        pb->t_start = pa->t_start;
        const uint32_t *const packed[2] = {ta, reinterpret_cast<const uint32_t *>(h_ws + pb->o_packed)};
        const int32_t *const syms[2] = {pa->job_syms, pb->job_syms};
        const int64_t n2[2] = {pa->job_n, pb->job_n}, nb2[2] = {pa->job_bypass, pb->job_bypass};
        uint8_t **out[2] = {&pa->bytes, &pb->bytes};
        size_t *len[2] = {&pa->bytes_len, &pb->bytes_len};
        pa->status = pb->status = rans_encode_symtab2(packed, syms, n2, nb2, out, len);
        pb->t_end = tr.ms();
      } else if (pa->symbuf) {
        pa->status = fgmm_symbuf_append_symtab(pa->symbuf, ta, pa->job_syms, pa->job_n);
      } else {
        pa->status = rans_encode_symtab(ta, pa->job_syms, pa->job_n, pa->job_bypass, &pa->bytes, &pa->bytes_len);
      }
      pa->t_end = tr.ms();
    };
    if (count == 1) job(); else ctx->pool->submit(job);
  }
  tr.mark("all tables landed, jobs out");
  if (count > 1) ctx->pool->wait_all();
  tr.mark("host rANS done");
  if (tr.on && getenv("FGMM_TRACE") && atoi(getenv("FGMM_TRACE")) > 1)
    for (int i = 0; i < count; ++i)
      fprintf(stderr, "[fgmm encode]   item %2d  submitted %7.3f  job %7.3f .. %7.3f  (%.3f ms)\n", i, items[i].t_sub,
              items[i].t_start, items[i].t_end, items[i].t_end - items[i].t_start);
  for (auto &it : items)
    if (it.status) return fail(it.status, "host rANS encode failed (%d)", it.status);
  return FGMM_OK;
}

// ---------------------------------------------------------------------------------------------------------
// decode, batched
// ---------------------------------------------------------------------------------------------------------
struct DecItem {
  const uint8_t *enc = nullptr;
  size_t enc_len = 0;
  fgmm_params prm{};
  int64_t stride_p = 1;
  int32_t M = 0;
  int64_t hw = 0;
  int clamp = 0;
  int32_t max_bs = 1;
  const int64_t *zero_bitmap = nullptr; // host [M] or null (= all channels coded)
  float *y_hat = nullptr;               // device [M*hw] or null
  int32_t *sym_host_out = nullptr;      // host [n] or null
  int status = FGMM_OK;
  // derived
  int32_t n_ch = 0;
  int64_t n = 0;
  size_t o_list = 0, o_rank = 0, o_hdr = 0, o_used = 0, o_bsum = 0, o_boff = 0;
  int32_t tiles = 0;
  size_t hdr_bytes = 0;                   // header array, rounded up to 256 B
  const char *h_hdr = nullptr;            // pinned: headers (uint32, or uint16 when hdr16)
  bool hdr16 = false;                     // the item's headers travel in the 2-byte form
  char *h_out = nullptr;                  // pinned: decoded symbols (host-written, read by the scatter kernel)
  uint64_t pool_used = 0;
  int wide = 0; // h_out holds int32 symbols (some symbol outside int16), else int16
  // how the rows reach the host (see decode_batch): whole, or in n_piece pieces for the items of the tail window
  int n_piece = 1;
  uint64_t piece_end[kMaxPieces] = {};       // latents on the host once piece k has landed
  const uint8_t *piece_base[kMaxPieces] = {}; // pinned: rows of piece k
  hipEvent_t piece_ev[kMaxPieces] = {};      // [0] is the event the dispatcher waits for
  std::atomic<int> copy_queued{0};           // piece_ev[0] recorded in this call (events are reused: never wait on a stale one)
  std::atomic<int> done{0};
  double t_taken = 0, t_start = 0, t_end = 0; // FGMM_TRACE=2: job timeline
  DecItem() = default;
  DecItem(const DecItem &) = delete;
};

constexpr size_t kCounterBytes = sizeof(unsigned long long) * (3 + kMaxPieces); // per item, see DecDesc::pool_used

// Decode, batched and pipelined.  Items are cut into groups; the sizes of the variable-length tables are only known
// on the device, so every group takes one host round trip between its two passes:
//   caller's stream : [H2D descs][count+scan g0][count+scan g1] ...              ... [y_hat scatter, item by item]
//   aux stream      : after count+scan g -> D2H of group g's row-pool sizes
//   this thread     : sizes of group g known -> pack the group's rows into one staging range (device) and one
//                     pinned range (host) of exactly that size, patch the descriptors
//   fill stream     : [H2D descs g][hdr_pack g][fill g]   (beside the count passes of later groups; the headers go
//                     into the same range: 2 bytes per latent where the item's half-width fits and no row is
//                     non-monotone, else 4; the fill pass formats rows from the edges the count pass kept)
//   copy stream     : after fill g -> ONE copy for the group's range, headers + rows (few large copies: measured
//                     55.7 GB/s, against 51 GB/s for a copy per item)
//   dispatcher job  : waits for the groups' copies in order and hands each item to the workers as it lands
//   host workers    : one bitstream each; symbols go to pinned memory as int16 (int32 if one does not fit)
//   caller's stream : yhat_scatter_kernel reads them from there and writes the full float latent, zero channels too
// so the PCIe transfer of the tables, the longest leg, overlaps the table kernels of later groups and the host
// coding of earlier items.
// Tail window: a bitstream decodes sequentially (~9 ns/symbol), so whatever lands last leaves one whole item of
// host work behind it.  The last few items are therefore transferred in pieces (channel ranges), piece k of all of
// them in one copy: their decoders start on piece 0 and follow the pieces as they land (rows are in latent order),
// and what remains after the final copy is one piece of work instead of one item.
int decode_batch(fgmm_ctx *ctx, hipStream_t stream, std::vector<DecItem> &items, int mode) {
  const int count = (int)items.size();
  if (count == 0) return FGMM_OK;
  Trace tr("decode");
  int rc;
  if ((rc = ctx->ensure_streams())) return rc;
  // tail window: how many trailing items land in pieces, and in how many pieces each
  const int tail_cfg = getenv("FGMM_TAIL_ITEMS") ? atoi(getenv("FGMM_TAIL_ITEMS")) : 8;
  const int piece_cfg = getenv("FGMM_TAIL_PIECES") ? atoi(getenv("FGMM_TAIL_PIECES")) : 4;
  const int n_piece = std::min(std::max(piece_cfg, 1), (int)kMaxPieces);
  const int tail_items = (n_piece > 1) ? std::min({std::max(tail_cfg, 0), count, std::max(ctx->pool->size() / 2, 1)}) : 0;
  const int tail_begin = count - tail_items;
  // groups of items: small first (the first tables land as early as possible), then larger; the tail window is
  // one group of its own
  std::vector<int> gbeg;
  {
    const int steady_cfg = getenv("FGMM_DEC_GROUP") ? atoi(getenv("FGMM_DEC_GROUP")) : 0;
    const int steady = steady_cfg > 0 ? steady_cfg : (count >= 16 ? std::max(2, count / 8) : count);
    const int first_cfg = getenv("FGMM_DEC_FIRST") ? atoi(getenv("FGMM_DEC_FIRST")) : 2;
    int i = 0, sz = count >= 16 ? std::min(std::max(first_cfg, 1), steady) : count;
    while (i < tail_begin) {
      gbeg.push_back(i);
      i = std::min(i + sz, tail_begin);
      sz = std::min(steady, sz * 2);
    }
    if (tail_items) gbeg.push_back(tail_begin);
    gbeg.push_back(count);
  }
  const int n_groups = (int)gbeg.size() - 1;

  Arena ar; // device workspace; the part before the counters is mirrored in h_ws
  const size_t o_descs = ar.take(sizeof(DecDesc) * (size_t)count);
  const size_t o_pdescs = ar.take(sizeof(DecDesc) * (size_t)std::max(tail_items * n_piece, 1)); // piece k of tail item t: [k * tail_items + t]
  for (auto &it : items) {
    if (it.max_bs < 0 || it.max_bs > FGMM_MAX_BS)
      return fail(FGMM_ERR_UNSUPPORTED, "max_bs_value %d outside [0, %d]", it.max_bs, FGMM_MAX_BS);
    it.n_ch = 0;
    for (int c = 0; c < it.M; ++c) it.n_ch += it.zero_bitmap ? (it.zero_bitmap[c] != 0) : 1;
    it.n = (int64_t)it.n_ch * it.hw;
    it.o_list = ar.take(sizeof(int32_t) * std::max(it.n_ch, 1), 16);
    it.o_rank = ar.take(sizeof(int32_t) * std::max(it.M, 1), 16);
  }
  const size_t o_used = ar.take(kCounterBytes * (size_t)count, 256);
  const size_t upload_bytes = o_used;
  for (int i = 0; i < count; ++i) items[i].o_used = o_used + kCounterBytes * (size_t)i;
  const size_t host_fixed = ar.off;
  for (auto &it : items) { // the header arrays of consecutive items are adjacent: one copy per group fetches them
    it.hdr_bytes = align_up(sizeof(uint32_t) * (size_t)it.n, 256);
    it.o_hdr = ar.take(it.hdr_bytes);
  }
  for (auto &it : items) {
    it.tiles = (int32_t)((it.hw + 255) / 256);
    const size_t nblk = (size_t)it.n_ch * (size_t)it.tiles;
    it.o_bsum = ar.take(sizeof(uint32_t) * nblk + 64);
    it.o_boff = ar.take(sizeof(uint64_t) * nblk + 64);
  }
  // temp buffer of evaluated edges, [block][W][256] uint16 per item: 300 B/latent at max_bs = 74 (1.9 GB for the Kodak
  // batch, of 288 GB); batches that would need more than FGMM_TMP_MAX_MB (default 16 GiB) evaluate the rows twice instead
  std::vector<size_t> tmp_off((size_t)count, 0);
  size_t tmp_total = 0;
  {
    const size_t cap = (getenv("FGMM_TMP_MAX_MB") ? strtoull(getenv("FGMM_TMP_MAX_MB"), nullptr, 10) : 16384ull) << 20;
    for (int i = 0; i < count; ++i) {
      const DecItem &it = items[i];
      tmp_off[(size_t)i] = tmp_total;
      tmp_total += sizeof(uint16_t) * (size_t)it.n_ch * (size_t)it.tiles * 256 * (size_t)(2 * (int64_t)it.max_bs + 2 + kTmpHdrRows);
      tmp_total = align_up(tmp_total, 256);
    }
    if (tmp_total > cap) tmp_total = 0;
  }
  if (tmp_total && (rc = ctx->ensure_tmp(tmp_total))) return rc;
  // events: per group [scan done][counters landed][fill done][tables landed], plus per piece of the tail window
  // [fill done][landed]
  if ((rc = ctx->ensure_device(ar.off)) || (rc = ctx->ensure_host(host_fixed)) ||
      (rc = ctx->ensure_events(4 * (size_t)n_groups + 2 * (size_t)n_piece)))
    return rc;
  ctx->chunks_reset();
  hipEvent_t *ev_scan = ctx->events.data(), *ev_counters = ev_scan + n_groups, *ev_fill = ev_counters + n_groups,
             *ev_landed = ev_fill + n_groups, *ev_pfill = ev_landed + n_groups, *ev_pland = ev_pfill + n_piece;

  DecDesc *hd = reinterpret_cast<DecDesc *>(ctx->h_ws + o_descs);
  DecDesc *hpd = reinterpret_cast<DecDesc *>(ctx->h_ws + o_pdescs);
  const DecDesc *dd = reinterpret_cast<const DecDesc *>(ctx->d_ws + o_descs);
  const DecDesc *dpd = reinterpret_cast<const DecDesc *>(ctx->d_ws + o_pdescs);
  for (int i = 0; i < count; ++i) {
    DecItem &it = items[i];
    int32_t *list = reinterpret_cast<int32_t *>(ctx->h_ws + it.o_list);
    int32_t *rank = reinterpret_cast<int32_t *>(ctx->h_ws + it.o_rank);
    int r = 0;
    for (int c = 0; c < it.M; ++c) {
      const bool coded = !it.zero_bitmap || it.zero_bitmap[c] != 0;
      rank[c] = coded ? r : -1;
      if (coded) list[r++] = c;
    }
    DecDesc &d = hd[i];
    memset(&d, 0, sizeof d);
    d.scales = it.prm.scales;
    d.means = it.prm.means;
    d.weights = it.prm.weights;
    d.stride_k = it.prm.stride_k;
    d.stride_c = it.prm.stride_c;
    d.stride_p = it.stride_p;
    d.hw = it.hw;
    d.chan_list = reinterpret_cast<const int32_t *>(ctx->d_ws + it.o_list);
    d.n_ch = it.n_ch;
    d.ch_begin = 0;
    d.ch_end = it.n_ch;
    d.max_bs = it.max_bs;
    d.clamp = it.clamp;
    d.prune = 1;
    d.tiles = it.tiles;
    d.hdr = reinterpret_cast<uint32_t *>(ctx->d_ws + it.o_hdr);
    d.pool = nullptr;   // set once the size is known
    d.pool_cap = ~0ull; // the pool is carved to the exact size: the overflow flag of the scan pass stays clear
    d.pool_used = reinterpret_cast<unsigned long long *>(ctx->d_ws + it.o_used);
    d.n_piece = i >= tail_begin ? n_piece : 1;
    d.tmp = tmp_total ? reinterpret_cast<uint16_t *>(ctx->d_tmp + tmp_off[(size_t)i]) : nullptr;
    d.blk_sums = reinterpret_cast<uint32_t *>(ctx->d_ws + it.o_bsum);
    d.blk_off = reinterpret_cast<unsigned long long *>(ctx->d_ws + it.o_boff);
  }
  HIP_TRY(hipMemcpyAsync(ctx->d_ws, ctx->h_ws, upload_bytes, hipMemcpyHostToDevice, stream));
  HIP_TRY(hipMemsetAsync(ctx->d_ws + o_used, 0, kCounterBytes * (size_t)count, stream));
  const bool clamped = items[0].clamp != 0, f16 = items[0].prm.dtype == FGMM_F16;
  auto extent = [&](int i0, int i1, int *n_ch_max, int64_t *hw_max) {
    *n_ch_max = 0;
    *hw_max = 0;
    for (int i = i0; i < i1; ++i) {
      *n_ch_max = std::max(*n_ch_max, items[i].n_ch);
      *hw_max = std::max(*hw_max, items[i].hw);
    }
  };
  if ((rc = ctx->prof_begin(1, stream))) return rc;
  for (int g = 0; g < n_groups; ++g) {
    const int i0 = gbeg[g], i1 = gbeg[g + 1];
    int n_ch_max;
    int64_t hw_max;
    extent(i0, i1, &n_ch_max, &hw_max);
    LAUNCH_TRY(launch_cdftab_count(dd + i0, i1 - i0, n_ch_max, hw_max, mode, clamped, f16, stream));
    HIP_TRY(hipEventRecord(ev_scan[g], stream));
    HIP_TRY(hipStreamWaitEvent(ctx->aux_stream, ev_scan[g], 0));
    HIP_TRY(hipMemcpyAsync(ctx->h_ws + items[i0].o_used, ctx->d_ws + items[i0].o_used, kCounterBytes * (size_t)(i1 - i0),
                           hipMemcpyDeviceToHost, ctx->aux_stream));
    HIP_TRY(hipEventRecord(ev_counters[g], ctx->aux_stream));
  }
  if ((rc = ctx->prof_end(1, stream))) return rc; // brackets the count + scan passes of all groups
  tr.mark("enqueued");

  std::mutex done_mu;
  std::condition_variable done_cv;

  auto submit_job = [&](int i, bool here) { // here: the item's tables (or their first piece) are on the host
    DecItem *pit = &items[i];
    pit->t_taken = tr.ms();
    auto job = [pit, here, &done_mu, &done_cv, &tr] {
      pit->t_start = tr.ms();
      // decoded symbols: the caller's buffer, or a per-thread scratch (a fresh 600 KB malloc per stream is an mmap)
      static thread_local std::vector<int32_t> scratch;
      int32_t *sym = pit->sym_host_out;
      if (!sym) {
        try {
          if (scratch.size() < (size_t)std::max<int64_t>(pit->n, 1)) scratch.resize((size_t)std::max<int64_t>(pit->n, 1));
          sym = scratch.data();
        } catch (const std::bad_alloc &) {
          sym = nullptr;
        }
      }
      if (!here) { // a kernel or copy of the item's group failed
        pit->status = FGMM_ERR_HIP;
      } else if (!sym) {
        pit->status = FGMM_ERR_NOMEM;
      } else {
        Landing land{pit->n_piece, pit->piece_end, pit->piece_base, pit, [](void *arg, int k) -> int {
                       return hipEventSynchronize(static_cast<DecItem *>(arg)->piece_ev[k]) == hipSuccess ? (int)FGMM_OK : (int)FGMM_ERR_HIP;
                     }};
        pit->status = rans_decode_cdftab(pit->enc, pit->enc_len, reinterpret_cast<const uint32_t *>(pit->h_hdr), pit->piece_base[0],
                                         pit->n, pit->max_bs, sym, pit->n_piece > 1 ? &land : nullptr,
                                         pit->hdr16 ? reinterpret_cast<const uint16_t *>(pit->h_hdr) : nullptr);
        if (pit->status == FGMM_OK && pit->y_hat) {
          // symbols -> pinned memory for the scatter kernel: int16 unless some (bypass-coded) symbol does not fit
          int16_t *s16 = reinterpret_cast<int16_t *>(pit->h_out);
          int32_t acc = 0;
          for (int64_t k = 0; k < pit->n; ++k) {
            const int32_t v = sym[k];
            s16[k] = (int16_t)v;
            acc |= v ^ (int32_t)(int16_t)v;
          }
          pit->wide = acc != 0;
          if (pit->wide) memcpy(pit->h_out, sym, sizeof(int32_t) * (size_t)pit->n);
        }
      }
      pit->t_end = tr.ms();
      {
        std::lock_guard<std::mutex> l(done_mu);
        pit->done.store(1);
      }
      done_cv.notify_all();
    };
    if (count == 1) job(); else ctx->pool->submit(job);
  };

  // One dispatcher (itself a pool job) waits for the copies in order and hands each item to the workers as it
  // lands, while this thread keeps feeding the fill and copy streams group by group.
  std::atomic<int> abandon{0};
  auto dispatcher = [&] {
    for (int i = 0; i < count; ++i) {
      while (!items[i].copy_queued.load(std::memory_order_acquire)) {
        if (abandon.load()) return;
        __builtin_ia32_pause();
      }
      submit_job(i, hipEventSynchronize(items[i].piece_ev[0]) == hipSuccess);
    }
  };
  // Order of destruction on any return: release the dispatcher, wait for every job, then the objects they use.
  PoolDrain drain{ctx->pool};
  struct Abandon {
    std::atomic<int> &f;
    ~Abandon() { f.store(1); }
  } abandon_on_exit{abandon};
  if (count > 1) ctx->pool->submit(dispatcher);

  if ((rc = ctx->prof_begin(3, ctx->fill_stream))) return rc;
  for (int g = 0; g < n_groups; ++g) {
    const int i0 = gbeg[g], i1 = gbeg[g + 1];
    bool tail = tail_items && i0 == tail_begin;
    if (tail) { // pieces only pay for rows that take a while to cross: small tail groups travel whole
      int64_t lat = 0;
      for (int i = i0; i < i1; ++i) lat += items[i].n;
      tail = lat >= 65536;
    }
    const int np = tail ? n_piece : 1;
    HIP_TRY(hipEventSynchronize(ev_counters[g]));
    // ---- layout of the group's range, the same in the device staging area and in pinned memory:
    //   [headers of every item: 2 bytes per latent where the item allows it, else 4][rows, piece-major: piece k of
    //   every item, then piece k+1 ...], every part 256-byte aligned
    size_t hdr_total = 0, n_lat_max = 0;
    std::vector<size_t> hdr_at((size_t)(i1 - i0));
    for (int i = i0; i < i1; ++i) {
      DecItem &it = items[i];
      const unsigned long long *u = reinterpret_cast<const unsigned long long *>(ctx->h_ws + it.o_used);
      it.hdr16 = tab_hdr_fits16(it.max_bs) && u[2 + kMaxPieces] == 0;
      hdr_at[(size_t)(i - i0)] = hdr_total;
      hdr_total += align_up((it.hdr16 ? sizeof(uint16_t) : sizeof(uint32_t)) * (size_t)it.n, 256);
      n_lat_max = std::max(n_lat_max, (size_t)it.n);
    }
    size_t piece_off[kMaxPieces + 1] = {0}; // byte range of piece k within the group's rows
    std::vector<size_t> at((size_t)(i1 - i0) * (size_t)np); // [k * (i1-i0) + t]: where piece k of item t starts
    size_t off = 0, out_total = 0;
    for (int k = 0; k < np; ++k) {
      piece_off[k] = off;
      for (int i = i0; i < i1; ++i) {
        DecItem &it = items[i];
        const unsigned long long *u = reinterpret_cast<const unsigned long long *>(ctx->h_ws + it.o_used);
        it.pool_used = u[0];
        const uint64_t b0 = k ? u[2 + k - 1] : 0, b1 = k + 1 < np ? u[2 + k] : u[0];
        if (b1 < b0 || b1 > u[0]) return fail(FGMM_ERR_HIP, "inconsistent piece offsets for item %d", i);
        at[(size_t)k * (size_t)(i1 - i0) + (size_t)(i - i0)] = off;
        off = align_up(off + (size_t)(b1 - b0), 256);
      }
    }
    piece_off[np] = off;
    const size_t range_bytes = hdr_total + off + 256; // + slack: the host's SIMD search reads a little past a row
    for (int i = i0; i < i1; ++i) out_total += align_up(sizeof(int32_t) * (size_t)std::max<int64_t>(items[i].n, 1), 256);
    char *d_range = nullptr, *h_range = nullptr, *h_outs = nullptr;
    if ((rc = ctx->dchunk_alloc(range_bytes, &d_range)) || (rc = ctx->chunk_alloc(range_bytes, &h_range)) ||
        (rc = ctx->chunk_alloc(out_total + 256, &h_outs)))
      return rc;
    char *const d_rows = d_range + hdr_total, *const h_rows = h_range + hdr_total;
    memset(h_rows + off, 0, 256);
    size_t out_off = 0;
    for (int i = i0; i < i1; ++i) {
      DecItem &it = items[i];
      const unsigned long long *u = reinterpret_cast<const unsigned long long *>(ctx->h_ws + it.o_used);
      it.h_hdr = h_range + hdr_at[(size_t)(i - i0)];
      hd[i].hdr_out = d_range + hdr_at[(size_t)(i - i0)];
      hd[i].hdr_compact = it.hdr16 ? 1 : 0;
      it.h_out = h_outs + out_off;
      out_off += align_up(sizeof(int32_t) * (size_t)std::max<int64_t>(it.n, 1), 256);
      it.n_piece = np;
      for (int k = 0; k < np; ++k) {
        const size_t a = at[(size_t)k * (size_t)(i1 - i0) + (size_t)(i - i0)];
        const uint64_t b0 = k ? u[2 + k - 1] : 0;
        it.piece_base[k] = reinterpret_cast<const uint8_t *>(h_rows + a);
        it.piece_end[k] = (uint64_t)((int64_t)it.n_ch * (k + 1) / np) * (uint64_t)it.hw;
        it.piece_ev[k] = tail ? ev_pland[k] : ev_landed[g];
        // the fill pass addresses rows as pool + (offset within the item): bias the base by the piece's start
        DecDesc &d = tail ? hpd[k * tail_items + (i - i0)] : hd[i];
        if (tail) d = hd[i];
        d.pool = reinterpret_cast<uint8_t *>(d_rows + a) - b0;
        d.ch_begin = (int32_t)((int64_t)it.n_ch * k / np);
        d.ch_end = (int32_t)((int64_t)it.n_ch * (k + 1) / np);
      }
    }
    int n_ch_max;
    int64_t hw_max;
    extent(i0, i1, &n_ch_max, &hw_max);
    // ---- headers into the range, fill, then fetch: ONE copy for an ordinary group (headers + rows), one per piece
    // for the tail window (the first carries the headers)
    HIP_TRY(hipMemcpyAsync(ctx->d_ws + o_descs + sizeof(DecDesc) * (size_t)i0, &hd[i0], sizeof(DecDesc) * (size_t)(i1 - i0),
                           hipMemcpyHostToDevice, ctx->fill_stream));
    LAUNCH_TRY(launch_hdr_pack(dd + i0, i1 - i0, (int64_t)n_lat_max, ctx->fill_stream));
    if (!tail) {
      LAUNCH_TRY(launch_cdftab_fill(dd + i0, i1 - i0, n_ch_max, hw_max, mode, clamped, f16, ctx->fill_stream));
      HIP_TRY(hipEventRecord(ev_fill[g], ctx->fill_stream));
      HIP_TRY(hipStreamWaitEvent(ctx->copy_stream, ev_fill[g], 0));
      if (hdr_total + off) HIP_TRY(hipMemcpyAsync(h_range, d_range, hdr_total + off, hipMemcpyDeviceToHost, ctx->copy_stream));
      HIP_TRY(hipEventRecord(ev_landed[g], ctx->copy_stream));
    } else {
      HIP_TRY(hipMemcpyAsync(ctx->d_ws + o_pdescs, hpd, sizeof(DecDesc) * (size_t)(tail_items * n_piece), hipMemcpyHostToDevice,
                             ctx->fill_stream));
      for (int k = 0; k < np; ++k) {
        LAUNCH_TRY(launch_cdftab_fill(dpd + k * tail_items, tail_items, n_ch_max, hw_max, mode, clamped, f16, ctx->fill_stream));
        HIP_TRY(hipEventRecord(ev_pfill[k], ctx->fill_stream));
        HIP_TRY(hipStreamWaitEvent(ctx->copy_stream, ev_pfill[k], 0));
        const size_t c0 = k ? hdr_total + piece_off[k] : 0, c1 = hdr_total + piece_off[k + 1];
        if (c1 > c0) HIP_TRY(hipMemcpyAsync(h_range + c0, d_range + c0, c1 - c0, hipMemcpyDeviceToHost, ctx->copy_stream));
        HIP_TRY(hipEventRecord(ev_pland[k], ctx->copy_stream));
      }
    }
    for (int i = i0; i < i1; ++i) items[i].copy_queued.store(1, std::memory_order_release);
  }
  if ((rc = ctx->prof_end(3, ctx->fill_stream))) return rc;
  if (count == 1) dispatcher();
  ctx->stat[1] = ctx->stat[2] = 0;
  for (auto &it : items) {
    ctx->stat[1] += it.pool_used + (it.hdr16 ? sizeof(uint16_t) : sizeof(uint32_t)) * (unsigned long long)it.n;
    ctx->stat[2] += (unsigned long long)it.n;
  }
  tr.mark("sizes known, fill + copies queued");

  // ---- symbols back to the GPU item by item: scatter kernel on the caller's stream ---------------------------
  int first_err = FGMM_OK;
  for (int i = 0; i < count; ++i) {
    DecItem &it = items[i];
    {
      std::unique_lock<std::mutex> l(done_mu);
      done_cv.wait(l, [&it] { return it.done.load() != 0; });
    }
    if (it.status && !first_err) first_err = it.status;
    if (it.status == FGMM_OK && it.y_hat && it.M * it.hw)
      LAUNCH_TRY(launch_yhat_scatter(it.h_out, it.wide, reinterpret_cast<const int32_t *>(ctx->d_ws + it.o_rank), it.y_hat, it.M,
                                     it.hw, stream));
  }
  tr.mark("host rANS done");
  HIP_TRY(hipStreamSynchronize(stream));
  tr.mark("y_hat written");
  if (tr.on && getenv("FGMM_TRACE") && atoi(getenv("FGMM_TRACE")) > 1)
    for (int i = 0; i < count; ++i)
      fprintf(stderr, "[fgmm decode]   item %2d  pieces %d  taken %7.3f  job %7.3f .. %7.3f  (%.3f ms)\n", i, items[i].n_piece,
              items[i].t_taken, items[i].t_start, items[i].t_end, items[i].t_end - items[i].t_start);
  if (first_err)
    return fail(first_err, "host rANS decode failed (%d)%s", first_err, first_err == FGMM_ERR_STREAM ? ": bitstream too short" : "");
  return FGMM_OK;
}

// stage an (n,K) host parameter triple on the device; returns device pointers + strides to use
struct StagedRows {
  const float *s = nullptr, *m = nullptr, *w = nullptr;
  int64_t stride_n = 0, stride_k = 0;
};

} // namespace

// ===========================================================================================================
// extern "C"
// ===========================================================================================================
extern "C" {

int fgmm_abi_version(void) { return FGMM_ABI_VERSION; }
const char *fgmm_last_error(void) { return t_err; }

int fgmm_ctx_create(int device, int n_threads, fgmm_ctx **out) {
  if (!out) return fail(FGMM_ERR_INVALID, "out == NULL");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(FGMM_ERR_NO_DEVICE, "no HIP device: libflashgmm_amd has no CPU path for the float work");
  if (device < 0 && hipGetDevice(&device) != hipSuccess) return fail(FGMM_ERR_NO_DEVICE, "hipGetDevice failed");
  if (device >= ndev) return fail(FGMM_ERR_INVALID, "device %d of %d", device, ndev);
  DeviceGuard g(device);
  if (!g.ok) return fail(FGMM_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
  if (n_threads <= 0) {
    n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads <= 0) n_threads = 4;
    n_threads = std::min(n_threads, 16);
  }
  fgmm_ctx *c = new (std::nothrow) fgmm_ctx;
  if (!c) return fail(FGMM_ERR_NOMEM, "ctx");
  c->device = device;
  c->pool = new Pool(n_threads);
  *out = c;
  return FGMM_OK;
}

void fgmm_ctx_destroy(fgmm_ctx *ctx) {
  if (!ctx) return;
  {
    DeviceGuard g(ctx->device);
    delete ctx->pool;
    for (auto e : ctx->events) (void)hipEventDestroy(e);
    for (auto &pr : ctx->prof)
      for (auto e : pr)
        if (e) (void)hipEventDestroy(e);
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    if (ctx->h_ws) (void)hipHostFree(ctx->h_ws);
    for (auto &c : ctx->chunks) (void)hipHostFree(c.p);
    for (auto &c : ctx->dchunks) (void)hipFree(c.p);
    if (ctx->d_tmp) (void)hipFree(ctx->d_tmp);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->fill_stream) (void)hipStreamDestroy(ctx->fill_stream);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
  }
  delete ctx;
}

int fgmm_ctx_set_profiling(fgmm_ctx *ctx, int enable) {
  if (!ctx) return fail(FGMM_ERR_INVALID, "ctx == NULL");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  if (enable)
    for (auto &pr : ctx->prof)
      for (auto &e : pr)
        if (!e) HIP_TRY(hipEventCreate(&e));
  ctx->profiling = enable != 0;
  return FGMM_OK;
}

int fgmm_ctx_kernel_ms(fgmm_ctx *ctx, int which, float *ms_out) {
  if (!ctx || which < 0 || which > 3 || !ms_out) return fail(FGMM_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> lock(ctx->mu);
  if (!ctx->profiling || !ctx->prof_valid[which]) return fail(FGMM_ERR_INVALID, "no profiled launch of kernel %d yet", which);
  DeviceGuard g(ctx->device);
  HIP_TRY(hipEventSynchronize(ctx->prof[which][1]));
  HIP_TRY(hipEventElapsedTime(ms_out, ctx->prof[which][0], ctx->prof[which][1]));
  return FGMM_OK;
}

int fgmm_ctx_stat(fgmm_ctx *ctx, int which, uint64_t *out) {
  if (!ctx || which < 0 || which > 3 || !out) return fail(FGMM_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> lock(ctx->mu);
  *out = ctx->stat[which];
  return FGMM_OK;
}

int fgmm_ctx_device(const fgmm_ctx *ctx) { return ctx ? ctx->device : -1; }
int fgmm_ctx_threads(const fgmm_ctx *ctx) { return ctx && ctx->pool ? ctx->pool->size() : 0; }

// ---- section 2: entropy-model level ---------------------------------------------------------------------

int fgmm_gmc_compress_batch(fgmm_ctx *ctx, void *stream, fgmm_item *items, int count, int mode, int clamp_scales) {
  if (!ctx || count < 0 || (count && !items) || !mode_ok(mode)) return fail(FGMM_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  if (!g.ok) return fail(FGMM_ERR_NO_DEVICE, "hipSetDevice(%d) failed", ctx->device);
  std::vector<EncItem> v((size_t)count);
  for (int i = 0; i < count; ++i) {
    const fgmm_item &s = items[i];
    if (s.K != FGMM_K) return fail(FGMM_ERR_INVALID, "K = %d: the reference binds K = 4 only", s.K);
    if (s.M < 0 || s.hw < 0 || (s.M * s.hw && (!s.y || !s.params.scales || !s.params.means || !s.params.weights)))
      return fail(FGMM_ERR_INVALID, "item %d: null tensor / negative size", i);
    if (s.params.dtype != items[0].params.dtype || (s.params.dtype != FGMM_F32 && s.params.dtype != FGMM_F16))
      return fail(FGMM_ERR_INVALID, "item %d: parameter dtype must be FGMM_F32 or FGMM_F16 and the same for a whole batch", i);
    EncItem &e = v[i];
    e.y = s.y;
    e.prm = s.params;
    e.M = s.M;
    e.hw = s.hw;
    e.clamp = clamp_scales;
    e.yq = s.yq_out;
    e.zero_bitmap = s.zero_bitmap;
  }
  const int rc = encode_batch(ctx, (hipStream_t)stream, v, mode);
  for (int i = 0; i < count; ++i) {
    items[i].abs_max = v[i].abs_max;
    items[i].bytes = v[i].bytes;
    items[i].bytes_len = v[i].bytes_len;
    items[i].status = v[i].status;
  }
  return rc;
}

int fgmm_gmc_compress(fgmm_ctx *ctx, void *stream, const float *y, const fgmm_params *params, int M, int K,
                      int64_t hw, int mode, int clamp_scales, float *yq_out, int32_t *abs_max_out,
                      int64_t *zero_bitmap_out, uint8_t **out, size_t *out_len) {
  if (!params || !out || !out_len) return fail(FGMM_ERR_INVALID, "null argument");
  fgmm_item it;
  memset(&it, 0, sizeof it);
  it.y = y;
  it.params = *params;
  it.M = M;
  it.K = K;
  it.hw = hw;
  it.yq_out = yq_out;
  it.zero_bitmap = zero_bitmap_out;
  const int rc = fgmm_gmc_compress_batch(ctx, stream, &it, 1, mode, clamp_scales);
  if (rc) return rc;
  if (abs_max_out) *abs_max_out = it.abs_max;
  *out = it.bytes;
  *out_len = it.bytes_len;
  return FGMM_OK;
}

int fgmm_gmc_decompress_batch(fgmm_ctx *ctx, void *stream, fgmm_item *items, int count, int mode, int clamp_scales) {
  if (!ctx || count < 0 || (count && !items) || !mode_ok(mode)) return fail(FGMM_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  if (!g.ok) return fail(FGMM_ERR_NO_DEVICE, "hipSetDevice(%d) failed", ctx->device);
  std::vector<DecItem> v((size_t)count);
  for (int i = 0; i < count; ++i) {
    const fgmm_item &s = items[i];
    if (s.K != FGMM_K) return fail(FGMM_ERR_INVALID, "K = %d: the reference binds K = 4 only", s.K);
    if (s.M < 0 || s.hw < 0 || !s.bytes || !s.zero_bitmap || !s.yq_out ||
        (s.M * s.hw && (!s.params.scales || !s.params.means || !s.params.weights)))
      return fail(FGMM_ERR_INVALID, "item %d: null tensor / negative size", i);
    if (s.params.dtype != items[0].params.dtype || (s.params.dtype != FGMM_F32 && s.params.dtype != FGMM_F16))
      return fail(FGMM_ERR_INVALID, "item %d: parameter dtype must be FGMM_F32 or FGMM_F16 and the same for a whole batch", i);
    DecItem &d = v[i];
    d.enc = s.bytes;
    d.enc_len = s.bytes_len;
    d.prm = s.params;
    d.M = s.M;
    d.hw = s.hw;
    d.clamp = clamp_scales;
    d.max_bs = s.abs_max + 1; // entropy_models.py:888
    d.zero_bitmap = s.zero_bitmap;
    d.y_hat = s.yq_out;
  }
  const int rc = decode_batch(ctx, (hipStream_t)stream, v, mode);
  for (int i = 0; i < count; ++i) items[i].status = v[i].status;
  return rc;
}

int fgmm_gmc_decompress(fgmm_ctx *ctx, void *stream, const uint8_t *encoded, size_t encoded_len, int32_t abs_max,
                        const int64_t *zero_bitmap, const fgmm_params *params, int M, int K, int64_t hw, int mode,
                        int clamp_scales, float *y_hat_out) {
  if (!params) return fail(FGMM_ERR_INVALID, "null argument");
  fgmm_item it;
  memset(&it, 0, sizeof it);
  it.params = *params;
  it.M = M;
  it.K = K;
  it.hw = hw;
  it.yq_out = y_hat_out;
  it.zero_bitmap = const_cast<int64_t *>(zero_bitmap);
  it.abs_max = abs_max;
  it.bytes = const_cast<uint8_t *>(encoded);
  it.bytes_len = encoded_len;
  return fgmm_gmc_decompress_batch(ctx, stream, &it, 1, mode, clamp_scales);
}

// ---- section 1: the reference's native boundary ------------------------------------------------------------

namespace {

// Host (n,K) rows -> device.  The three arrays are copied as the smallest span covering every addressed element
// when that span is dense enough; otherwise they are gathered into (n,4) row-major staging first (a copy, no
// arithmetic).  Device rows are used in place.
int stage_rows(fgmm_ctx *ctx, hipStream_t stream, const float *scales, const float *means, const float *weights,
               int64_t n, int64_t stride_n, int64_t stride_k, int memspace, std::vector<void *> &to_free, StagedRows *out) {
  if (memspace == FGMM_DEVICE || n == 0) {
    *out = {scales, means, weights, stride_n, stride_k};
    return FGMM_OK;
  }
  (void)ctx;
  const float *src[3] = {scales, means, weights};
  const float *dst[3];
  const bool dense = stride_n >= 0 && stride_k >= 0 && ((n - 1) * stride_n + 3 * stride_k + 1) <= 8 * n;
  const size_t span = dense ? (size_t)((n - 1) * stride_n + 3 * stride_k + 1) : (size_t)n * 4;
  for (int a = 0; a < 3; ++a) {
    float *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, span * sizeof(float) + 64));
    to_free.push_back(d);
    if (dense) {
      HIP_TRY(hipMemcpyAsync(d, src[a], span * sizeof(float), hipMemcpyHostToDevice, stream));
    } else {
      std::vector<float> tmp((size_t)n * 4);
      for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < 4; ++k) tmp[(size_t)i * 4 + k] = src[a][i * stride_n + k * stride_k];
      HIP_TRY(hipMemcpy(d, tmp.data(), tmp.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    dst[a] = d;
  }
  *out = {dst[0], dst[1], dst[2], dense ? stride_n : 4, dense ? stride_k : 1};
  return FGMM_OK;
}

struct FreeList {
  std::vector<void *> v;
  ~FreeList() {
    for (void *p : v) (void)hipFree(p);
  }
};

} // namespace

namespace {
int encode_rows(fgmm_ctx *ctx, const int32_t *symbols, const float *scales, const float *means, const float *weights,
                int64_t n, int64_t stride_n, int64_t stride_k, int K, int mode, int memspace, fgmm_symbuf *symbuf,
                uint8_t **out, size_t *out_len) {
  if (!ctx || n < 0 || !mode_ok(mode)) return fail(FGMM_ERR_INVALID, "bad argument");
  if (K != FGMM_K) return fail(FGMM_ERR_INVALID, "K = %d: the reference binds K = 4 only", K);
  if (n && (!symbols || !scales || !means || !weights)) return fail(FGMM_ERR_INVALID, "null tensor");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  if (!g.ok) return fail(FGMM_ERR_NO_DEVICE, "hipSetDevice(%d) failed", ctx->device);
  hipStream_t stream = nullptr;
  FreeList fl;
  StagedRows r;
  int rc = stage_rows(ctx, stream, scales, means, weights, n, stride_n, stride_k, memspace, fl.v, &r);
  if (rc) return rc;
  const int32_t *sym_dev = symbols;
  if (memspace == FGMM_HOST && n) {
    int32_t *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, sizeof(int32_t) * (size_t)n + 64));
    fl.v.push_back(d);
    HIP_TRY(hipMemcpyAsync(d, symbols, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, stream));
    sym_dev = d;
  }
  std::vector<EncItem> v(1);
  EncItem &e = v[0];
  e.sym_dev = sym_dev;
  e.sym_host = memspace == FGMM_HOST ? symbols : nullptr;
  e.prm = {r.s, r.m, r.w, r.stride_k, 0, FGMM_F32, 0};
  e.stride_p = r.stride_n;
  e.M = 1;
  e.hw = n;
  e.clamp = 0;
  e.symbuf = symbuf;
  rc = encode_batch(ctx, stream, v, mode);
  if (rc) return rc;
  if (out) {
    *out = e.bytes;
    *out_len = e.bytes_len;
  }
  return FGMM_OK;
}
} // namespace

int fgmm_encode_with_indexes_gmm(fgmm_ctx *ctx, const int32_t *symbols, const float *scales, const float *means,
                                 const float *weights, int64_t n, int64_t stride_n, int64_t stride_k, int K,
                                 int mode, int memspace, int32_t max_value, uint8_t **out, size_t *out_len) {
  (void)max_value; // ignored by the reference too (rans_interface.cpp:462)
  if (!out || !out_len) return fail(FGMM_ERR_INVALID, "bad argument");
  return encode_rows(ctx, symbols, scales, means, weights, n, stride_n, stride_k, K, mode, memspace, nullptr, out, out_len);
}

int fgmm_symbuf_append_gmm(fgmm_ctx *ctx, fgmm_symbuf *b, const int32_t *symbols, const float *scales,
                           const float *means, const float *weights, int64_t n, int64_t stride_n, int64_t stride_k,
                           int K, int mode, int memspace) {
  if (!b) return fail(FGMM_ERR_INVALID, "symbuf == NULL");
  return encode_rows(ctx, symbols, scales, means, weights, n, stride_n, stride_k, K, mode, memspace, b, nullptr, nullptr);
}

int fgmm_decode_with_indexes_gmm(fgmm_ctx *ctx, const uint8_t *encoded, size_t encoded_len, const float *scales,
                                 const float *means, const float *weights, int64_t n, int64_t stride_n,
                                 int64_t stride_k, int K, int mode, int memspace, int32_t max_bs_value,
                                 int32_t *out_symbols) {
  if (!ctx || !encoded || n < 0 || !mode_ok(mode) || (n && !out_symbols)) return fail(FGMM_ERR_INVALID, "bad argument");
  if (K != FGMM_K) return fail(FGMM_ERR_INVALID, "K = %d: the reference binds K = 4 only", K);
  if (n && (!scales || !means || !weights)) return fail(FGMM_ERR_INVALID, "null tensor");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  if (!g.ok) return fail(FGMM_ERR_NO_DEVICE, "hipSetDevice(%d) failed", ctx->device);
  hipStream_t stream = nullptr;
  FreeList fl;
  StagedRows r;
  int rc = stage_rows(ctx, stream, scales, means, weights, n, stride_n, stride_k, memspace, fl.v, &r);
  if (rc) return rc;
  std::vector<DecItem> v(1);
  DecItem &d = v[0];
  d.enc = encoded;
  d.enc_len = encoded_len;
  d.prm = {r.s, r.m, r.w, r.stride_k, 0, FGMM_F32, 0};
  d.stride_p = r.stride_n;
  d.M = 1;
  d.hw = n;
  d.max_bs = max_bs_value;
  d.sym_host_out = out_symbols;
  return decode_batch(ctx, stream, v, mode);
}

// ---- section 3: building blocks ---------------------------------------------------------------------------

int fgmm_gmm_cdf_hip(fgmm_ctx *ctx, void *stream, const int32_t *v, const float *scales, const float *means,
                     const float *weights, int64_t n, int64_t stride_n, int64_t stride_k, int mode, float *c1,
                     float *c2) {
  if (!ctx || n < 0 || !mode_ok(mode)) return fail(FGMM_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  LAUNCH_TRY(launch_cdf_pair(v, scales, means, weights, n, stride_n, stride_k, mode, c1, c2, stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return FGMM_OK;
}

int fgmm_build_symtab_hip(fgmm_ctx *ctx, void *stream, const int32_t *symbols, const float *scales,
                          const float *means, const float *weights, int64_t n, int64_t stride_n, int64_t stride_k,
                          int mode, uint32_t *packed) {
  if (!ctx || n < 0 || !mode_ok(mode)) return fail(FGMM_ERR_INVALID, "bad argument");
  if (n == 0) return FGMM_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  int rc;
  const size_t meta_bytes = sizeof(uint32_t) * (size_t)((n + 255) / 256) * 4; // per-wave bypass counts (unused here)
  if ((rc = ctx->ensure_device(1024 + meta_bytes + 256)) || (rc = ctx->ensure_host(4096))) return rc;
  EncDesc *hd = reinterpret_cast<EncDesc *>(ctx->h_ws);
  memset(hd, 0, sizeof *hd);
  hd->sym = symbols;
  hd->scales = scales;
  hd->means = means;
  hd->weights = weights;
  hd->stride_k = stride_k;
  hd->stride_p = stride_n;
  hd->hw = n;
  hd->M = 1;
  hd->packed = packed;
  hd->meta = reinterpret_cast<uint32_t *>(ctx->d_ws + 1024);
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipMemcpyAsync(ctx->d_ws, hd, sizeof *hd, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(ctx->d_ws + 1024, 0, meta_bytes, s));
  LAUNCH_TRY(launch_symtab(reinterpret_cast<const EncDesc *>(ctx->d_ws), 1, 1, n, n, false, mode, enc_vec4_ok(*hd, false) ? 4 : 1, false, false, s));
  HIP_TRY(hipStreamSynchronize(s));
  return FGMM_OK;
}

int fgmm_build_cdftab_hip(fgmm_ctx *ctx, void *stream, const float *scales, const float *means,
                          const float *weights, int64_t n, int64_t stride_n, int64_t stride_k, int mode,
                          int32_t max_bs, int flags, uint32_t *hdr, uint8_t *pool, uint64_t pool_cap,
                          uint64_t *pool_used) {
  if (!ctx || n < 0 || !mode_ok(mode) || !pool_used) return fail(FGMM_ERR_INVALID, "bad argument");
  if (max_bs < 0 || max_bs > FGMM_MAX_BS) return fail(FGMM_ERR_UNSUPPORTED, "max_bs %d outside [0, %d]", max_bs, FGMM_MAX_BS);
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  const int32_t tiles = (int32_t)((n + 255) / 256);
  const size_t o_bsum = 2048, o_boff = o_bsum + align_up(sizeof(uint32_t) * (size_t)tiles + 64, 256);
  int rc;
  if ((rc = ctx->ensure_device(o_boff + sizeof(uint64_t) * (size_t)tiles + 64)) || (rc = ctx->ensure_host(4096))) return rc;
  hipStream_t s = (hipStream_t)stream;
  DecDesc *hd = reinterpret_cast<DecDesc *>(ctx->h_ws);
  memset(hd, 0, sizeof *hd);
  hd->scales = scales;
  hd->means = means;
  hd->weights = weights;
  hd->stride_k = stride_k;
  hd->stride_p = stride_n;
  hd->hw = n;
  hd->n_ch = 1;
  hd->ch_begin = 0;
  hd->ch_end = 1;
  hd->max_bs = max_bs;
  hd->prune = (flags & FGMM_TAB_NO_PRUNE) ? 0 : 1;
  hd->clamp = (flags & FGMM_TAB_CLAMP) ? 1 : 0;
  hd->tiles = tiles;
  hd->hdr = hdr;
  hd->pool = pool;
  hd->pool_cap = pool_cap;
  hd->pool_used = reinterpret_cast<unsigned long long *>(ctx->d_ws + 1024);
  hd->blk_sums = reinterpret_cast<uint32_t *>(ctx->d_ws + o_bsum);
  hd->blk_off = reinterpret_cast<unsigned long long *>(ctx->d_ws + o_boff);
  HIP_TRY(hipMemcpyAsync(ctx->d_ws, hd, sizeof *hd, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(ctx->d_ws + 1024, 0, 16, s));
  if (n) LAUNCH_TRY(launch_cdftab(reinterpret_cast<const DecDesc *>(ctx->d_ws), 1, 1, n, mode, (flags & FGMM_TAB_CLAMP) != 0, false, s));
  unsigned long long used[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(used, ctx->d_ws + 1024, 16, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipMemcpy(pool_used, used, sizeof(uint64_t), hipMemcpyHostToDevice));
  if (used[1]) return fail(FGMM_ERR_NOMEM, "pool_cap %llu bytes too small (need %llu)", (unsigned long long)pool_cap, used[0]);
  return FGMM_OK;
}

static int ckbd(fgmm_ctx *ctx, void *stream, const void *src, void *dst, int64_t planes, int64_t h, int64_t w, int elem_bytes,
                int anchor_odd, bool embed) {
  if (!ctx || planes < 0 || h < 0 || w < 0 || (w & 1) || (elem_bytes != 2 && elem_bytes != 4) || (anchor_odd & ~1))
    return fail(FGMM_ERR_INVALID, "checkerboard split/merge: bad argument (w must be even, elem_bytes 2 or 4)");
  if (planes == 0 || h == 0 || w == 0) return FGMM_OK;
  if (!src || !dst) return fail(FGMM_ERR_INVALID, "checkerboard split/merge: null tensor");
  // the full tensor is accessed pair-wise (2 * elem_bytes), the halves element-wise
  const void *full = embed ? dst : src, *halves = embed ? src : dst;
  if (reinterpret_cast<uintptr_t>(full) % (2 * (size_t)elem_bytes) || reinterpret_cast<uintptr_t>(halves) % (size_t)elem_bytes)
    return fail(FGMM_ERR_INVALID, "checkerboard split/merge: misaligned tensor");
  DeviceGuard g(ctx->device);
  if (!g.ok) return fail(FGMM_ERR_NO_DEVICE, "cannot select HIP device %d", ctx->device);
  LAUNCH_TRY(launch_ckbd(src, dst, planes, h, w, elem_bytes, anchor_odd, embed, stream));
  return FGMM_OK;
}
int fgmm_ckbd_unembed(fgmm_ctx *ctx, void *stream, const void *src, void *dst, int64_t planes, int64_t h, int64_t w,
                      int elem_bytes, int anchor_odd) {
  return ckbd(ctx, stream, src, dst, planes, h, w, elem_bytes, anchor_odd, false);
}
int fgmm_ckbd_embed(fgmm_ctx *ctx, void *stream, const void *src, void *dst, int64_t planes, int64_t h, int64_t w,
                    int elem_bytes, int anchor_odd) {
  return ckbd(ctx, stream, src, dst, planes, h, w, elem_bytes, anchor_odd, true);
}

int fgmm_selftest_fastmath(fgmm_ctx *ctx, int which, uint64_t n, uint64_t seed, uint64_t *n_bad_out) {
  if (!ctx || which < 0 || which > 6 || !n_bad_out) return fail(FGMM_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  int rc;
  if ((rc = ctx->ensure_device(4096))) return rc;
  unsigned long long *d = reinterpret_cast<unsigned long long *>(ctx->d_ws);
  HIP_TRY(hipMemsetAsync(d, 0, 24, nullptr));
  LAUNCH_TRY(launch_fastmath_selftest(which, n, seed, d, nullptr));
  unsigned long long bad[3] = {0, 0, 0};
  HIP_TRY(hipMemcpy(bad, d, 24, hipMemcpyDeviceToHost));
  *n_bad_out = bad[0];
  if (bad[0]) snprintf(t_err, sizeof t_err, "fastmath selftest %d: %llu mismatches, witness a=0x%08llx s=0x%08llx", which, bad[0], bad[1], bad[2]);
  return FGMM_OK;
}

int fgmm_selftest_saturation(fgmm_ctx *ctx, int mode, uint64_t *n_bad_out) {
  if (!ctx || !mode_ok(mode) || !n_bad_out) return fail(FGMM_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  int rc;
  if ((rc = ctx->ensure_device(4096))) return rc;
  unsigned long long *d = reinterpret_cast<unsigned long long *>(ctx->d_ws);
  HIP_TRY(hipMemsetAsync(d, 0, 8, nullptr));
  LAUNCH_TRY(launch_saturation_selftest(mode, d, nullptr));
  unsigned long long bad = 0;
  HIP_TRY(hipMemcpy(&bad, d, 8, hipMemcpyDeviceToHost));
  *n_bad_out = bad;
  return FGMM_OK;
}

} // extern "C"

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
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <queue>
#include <string>
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

// ctx option "trace" >= 1: phase timestamps of every batched call on stderr (development aid)
struct Trace {
  bool on;
  int level;
  std::chrono::steady_clock::time_point t0, last;
  const char *what;
  Trace(const char *w, int lvl) : on(lvl > 0), level(lvl), what(w) { t0 = last = std::chrono::steady_clock::now(); }
  void mark(const char *phase) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[fgmm %s] %-28s +%8.3f ms  (t=%8.3f)\n", what, phase,
            std::chrono::duration<double, std::milli>(now - last).count(),
            std::chrono::duration<double, std::milli>(now - t0).count());
    last = now;
  }
  double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } // (always: the call log)
};

// ---- host worker pool ----------------------------------------------------------------------------------
class Pool {
public:
  explicit Pool(int n) {
    for (int i = 0; i < n; ++i)
      th_.emplace_back([this, i] {
        char name[16];
        snprintf(name, sizeof name, "fgmm-w%d", i); // (visible in /proc/<pid>/task/<tid>/comm: bench.py's step_diag names the threads that waited for a CPU)
        pthread_setname_np(pthread_self(), name);
        run();
      });
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

// Device -> pinned-host copies issued straight to the SDMA engines (hsa_amd_memory_async_copy_on_engine): what this
// runtime's hipMemcpyAsync does with shader ("blit") kernels on the CUs - next to the table kernels of the later launches
// (profiles/r03_bench_kernel_stats.csv: __amd_rocclr_copyBuffer, 43 % of the GPU time of a step).  The caller has seen the
// producing kernel complete (it needs the launch's byte count anyway), so a copy has no dependency; completion is an HSA
// signal the host workers sleep on.  HIP sits on the same HSA runtime: hsa_init() only takes another reference.
struct HsaCopier {
  bool tried = false, ok = false;
  hsa_agent_t gpu{}, cpu{};
  uint32_t engine[2] = {0, 0}; // two engines in turn: 56.9 GB/s against 55.4 on one (scripts/proto/sdma_copy.cpp)
  std::vector<hsa_signal_t> sigs;
  unsigned next = 0;
  // agents from the buffers themselves: `dptr` device memory of the context's GPU, `hptr` pinned host memory
  bool init(const void *dptr, const void *hptr) {
    if (tried) return ok;
    tried = true;
    if (hsa_init() != HSA_STATUS_SUCCESS) return false;
    hsa_amd_pointer_info_t pd, ph;
    memset(&pd, 0, sizeof pd);
    memset(&ph, 0, sizeof ph);
    pd.size = sizeof pd;
    ph.size = sizeof ph;
    if (hsa_amd_pointer_info(dptr, &pd, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS ||
        hsa_amd_pointer_info(hptr, &ph, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS || pd.type == HSA_EXT_POINTER_TYPE_UNKNOWN ||
        ph.type == HSA_EXT_POINTER_TYPE_UNKNOWN)
      return false;
    gpu = pd.agentOwner;
    cpu = ph.agentOwner;
    uint32_t avail = 0, pref = 0;
    if (hsa_amd_memory_copy_engine_status(cpu, gpu, &avail) != HSA_STATUS_SUCCESS || !avail) return false;
    (void)hsa_amd_memory_get_preferred_copy_engine(cpu, gpu, &pref);
    uint32_t pick = pref & avail ? pref & avail : avail & 0xFu ? avail & 0xFu : avail; // (engines beyond the first four serve the xGMI links: 6-13 GB/s to the host)
    engine[0] = pick & (0u - pick);
    pick &= ~engine[0];
    engine[1] = pick ? pick & (0u - pick) : engine[0];
    ok = true;
    return true;
  }
  // -> a signal that reaches 0 when the copy has landed; handle 0: not issued (the caller copies the HIP way)
  hsa_signal_t copy(void *dst, const void *src, size_t bytes, bool two_engines) {
    hsa_signal_t none{0};
    if (!ok) return none;
    if (next >= sigs.size()) {
      hsa_signal_t sg;
      if (hsa_signal_create(1, 0, nullptr, &sg) != HSA_STATUS_SUCCESS) return none;
      sigs.push_back(sg);
    }
    const hsa_signal_t sg = sigs[next];
    hsa_signal_store_relaxed(sg, 1);
    if (hsa_amd_memory_async_copy_on_engine(dst, cpu, src, gpu, bytes, 0, nullptr, sg, (hsa_amd_sdma_engine_id_t)engine[two_engines ? next & 1 : 0], false) != HSA_STATUS_SUCCESS)
      return none;
    ++next;
    return sg;
  }
  static bool wait(hsa_signal_t sg, bool spin) {
    while (hsa_signal_wait_scacquire(sg, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, spin ? HSA_WAIT_STATE_ACTIVE : HSA_WAIT_STATE_BLOCKED) >= 1) {
    }
    return true;
  }
  void quiesce() { // every copy issued so far has landed (before its buffers are reused or freed)
    for (unsigned k = 0; k < next && k < sigs.size(); ++k) wait(sigs[k], false);
    next = 0;
  }
  void destroy() {
    quiesce();
    for (auto &sg : sigs) hsa_signal_destroy(sg);
    sigs.clear();
    if (tried && ok) hsa_shut_down();
    ok = false;
  }
};

} // namespace

struct fgmm_ctx {
  int device = 0;
  HsaCopier hsa;
  std::mutex mu; // one call at a time per context
  Pool *pool = nullptr;
  char *d_ws = nullptr;
  size_t d_cap = 0;
  char *h_ws = nullptr; // pinned
  size_t h_cap = 0;
  std::vector<hipEvent_t> events;
  std::vector<hipEvent_t> sleep_events; // hipEventBlockingSync: waited for by the host workers (see ensure_events)
  hipStream_t copy_stream = nullptr; // bulk D2H of the decode tables (overlaps the table kernels of later launches)
  hipStream_t aux_stream = nullptr;  // the few bytes of per-launch counters
  // tuning knobs (fgmm_ctx_set_option); the FGMM_* environment variables of the same meaning are read once, at creation
  struct Opts {
    int64_t pieces = 0, dec_group = 0, dec_first = 2, tab_cap_e = kTabCapE, stage_max_mb = 0, trace = 0, enc_vec = 0, enc_linear = 1, ef_rows = 0, ef_min = kTabEfDefault, dec_pair = 0, enc_ways = 0, ckpt_decode = 0, spin_lat = 400000, gpu_decode = 0, tab_place = 0, tab_spin = kTabSpinLimit, copy_engine = 0, enc_segs = 1, scatter_rounds = 1;
  } opt;
  // pinned receive area of the decode tables: a list of chunks, bump-allocated per call, never moved while copies
  // are in flight (sizes are only known launch by launch)
  struct Chunk {
    char *p;
    size_t cap, used;
  };
  std::vector<Chunk> chunks;
  void chunks_reset() {
    for (auto &c : chunks) c.used = 0;
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
  // device staging area of the decode tables (headers, block offsets, rows): the table kernels write there, one copy
  // per launch fetches what was used.  Provisioned for the worst case of a call where memory allows (rows are placed by a
  // cursor, nothing is touched beyond it), else capped: a launch that overflows is re-run with the exact size.
  char *d_stage = nullptr;
  size_t d_stage_cap = 0;
  int ensure_stage(size_t bytes) {
    if (bytes <= d_stage_cap) return FGMM_OK;
    if (d_stage) HIP_TRY(hipFree(d_stage));
    d_stage = nullptr;
    d_stage_cap = 0;
    HIP_TRY(hipMalloc((void **)&d_stage, bytes));
    d_stage_cap = bytes;
    return FGMM_OK;
  }
  size_t stage_budget() const { // bytes the staging area may take
    if (opt.stage_max_mb > 0) return (size_t)opt.stage_max_mb << 20;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return (size_t)4 << 30;
    return std::max((free_b + d_stage_cap) / 4, (size_t)64 << 20);
  }
  int ensure_streams() {
    if (!copy_stream) {
      // the table copies are shader copies on this runtime: they share the CUs with the table kernels of the later launches,
      // and PCIe - the longest leg of a decode call - must not wait for a CU: highest priority (10.05 against 10.17 ms per
      // step at the default priority and 10.6 at the lowest, four runs each)
      int lo = 0, hi = 0;
      HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
      HIP_TRY(hipStreamCreateWithPriority(&copy_stream, hipStreamNonBlocking, hi));
    }
    if (!aux_stream) HIP_TRY(hipStreamCreateWithFlags(&aux_stream, hipStreamNonBlocking));
    return FGMM_OK;
  }
  std::vector<int32_t> h_sym; // decode: int32 symbols of every bitstream of a call (grown, kept)
  void trim() {
    hsa.quiesce();
    std::vector<int32_t>().swap(h_sym);
    if (d_ws) (void)hipFree(d_ws);
    if (h_ws) (void)hipHostFree(h_ws);
    if (d_stage) (void)hipFree(d_stage);
    for (auto &c : chunks) (void)hipHostFree(c.p);
    chunks.clear();
    d_ws = h_ws = d_stage = nullptr;
    d_cap = h_cap = d_stage_cap = 0;
  }
  // the call log: phase marks of the most recent batched calls (fgmm_ctx_call_log), always kept - a few clock reads per call
  const std::chrono::steady_clock::time_point born = std::chrono::steady_clock::now();
  static constexpr int kLogCap = 64;
  fgmm_call_marks log[kLogCap];
  unsigned long long log_n = 0;
  void log_call(int kind, int count, const Trace &tr, const double ms[5], double busy, double wait) {
    fgmm_call_marks &m = log[log_n++ % kLogCap];
    m.kind = kind;
    m.count = count;
    m.t_begin_ms = std::chrono::duration<double, std::milli>(tr.t0 - born).count();
    for (int k = 0; k < 5; ++k) m.ms[k] = ms[k];
    m.ms[5] = tr.ms();
    m.worker_busy_ms = busy;
    m.worker_wait_ms = wait;
  }
  bool profiling = false;
  unsigned long long stat[7] = {0, 0, 0, 0, 0, 0, 0}; // [6] table launches re-run with the cursor after a look-back gave up (since the context exists); [4] bitstreams the GPU's segment decoder decoded, [5] ... handed back to the table path;
                                                   // last batched call: [0] encode table bytes D2H, [1] decode table bytes D2H,
                                             // [2] decode latents, [3] edges the decode-side kernels evaluated
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
  // `events` are waited for by the calling thread for microseconds (spinning is right); `sleep_events` mark the landing
  // of table copies and are waited for by up to 16 host workers for up to milliseconds: those must SLEEP — a GPU box gives
  // the process a CPU quota of 16 cores, and sixteen spinning waiters plus the calling thread exceed it, which the
  // scheduler answers by throttling the whole process for the rest of its period (measured: +-10 % from run to run).
  int ensure_events(size_t n, size_t n_sleep = 0) {
    while (events.size() < n) {
      hipEvent_t e;
      HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      events.push_back(e);
    }
    while (sleep_events.size() < n_sleep) {
      hipEvent_t e;
      HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync));
      sleep_events.push_back(e);
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
  int64_t ckpt_stride = 0;       // note a checkpoint every this many symbols (fgmm_ckpt; 0: none)
  // outputs
  fgmm_ckpt *ckpt = nullptr;     // malloc'ed, n_ckpt entries
  int64_t n_ckpt = 0;
  int64_t *zero_bitmap = nullptr; // host [M] or null
  int32_t abs_max = 0;
  uint8_t *bytes = nullptr;
  size_t bytes_len = 0;
  int status = FGMM_OK;
  // workspace offsets
  size_t o_min = 0, o_max = 0, o_nz = 0, o_list = 0, o_meta = 0, o_packed = 0, meta_count = 0;
  // the table in segments of compact channels (EncDesc::packed_seg): offsets, channels per segment, segments, copy group of each
  size_t o_seg[kEncSegs] = {0, 0, 0, 0};
  int32_t cps = 0, n_seg = 0, seg_group[kEncSegs] = {0, 0, 0, 0};
  // what the item's host job needs (set when its side information has been read)
  const int32_t *job_syms = nullptr;
  int64_t job_n = 0, job_bypass = 0;
  double t_sub = 0, t_start = 0, t_end = 0; // FGMM_TRACE=2: job timeline
};

int encode_batch(fgmm_ctx *ctx, hipStream_t stream, std::vector<EncItem> &items, int mode) {
  const int count = (int)items.size();
  if (count == 0) return FGMM_OK;
  Trace tr("encode", (int)ctx->opt.trace);
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
  // The bitstreams of a call in the order of their size, LARGEST FIRST (equal sizes: as given): their tables cross PCIe in that
  // order and their host jobs are handed out in that order - the long jobs start first and the short ones fill the workers'
  // tails (ELIC's groups differ 12x in size: in the order given the largest tables landed last and their jobs ended the call)
  std::vector<int> order((size_t)count);
  for (int i = 0; i < count; ++i) order[(size_t)i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return (int64_t)items[a].M * items[a].hw > (int64_t)items[b].M * items[b].hw; });
  // A bitstream is encoded BACKWARDS (rANS), so its encoder needs the END of its table first - and with whole tables crossing
  // PCIe one after another the call ends a whole job (0.3-0.45 ms for a Kodak half) after the last table has landed.  When every
  // bitstream has a worker of its own, the tables are therefore laid out in up to four SEGMENTS of compact channels each, LAST
  // SEGMENT FIRST across all bitstreams: the encoders start on the tails after an eighth of the transfer and follow the landing;
  // what is left after the last byte is a quarter of a job (48 Kodak halves: 1.18 -> 0.9 ms per call).
  const int enc_T = std::max(ctx->pool->size(), 1);
  // automatic: pairs as soon as there are more bitstreams than workers (measured on the box, 48 bitstreams on 16 workers:
  // pairs 1.78 ms per call, threes 2.28, workers pulling one or two as the tables land 1.95-2.04)
  const int enc_ways = ctx->opt.enc_ways > 0 ? (int)ctx->opt.enc_ways : (count > enc_T ? 2 : 1);
  bool segmented = ctx->opt.enc_segs != 0 && count >= 2 && enc_ways == 1;
  {
    size_t table_bytes = 0;
    for (auto &it : items) {
      table_bytes += sizeof(uint32_t) * (size_t)it.M * (size_t)it.hw;
      segmented = segmented && it.y && !it.symbuf && it.M >= 2 * kEncSegs;
    }
    segmented = segmented && table_bytes >= ((size_t)4 << 20); // (smaller calls: the transfer is not what they wait for)
  }
  if (!segmented) {
    for (int i : order) items[i].o_packed = ar.take(sizeof(uint32_t) * (size_t)items[i].M * (size_t)items[i].hw + 64);
  } else {
    for (auto &it : items) {
      it.cps = (it.M + kEncSegs - 1) / kEncSegs;
      it.n_seg = (it.M + it.cps - 1) / it.cps;
    }
    for (int sg = kEncSegs - 1; sg >= 0; --sg)
      for (int i : order) {
        EncItem &it = items[i];
        if (sg >= it.n_seg) continue;
        const int32_t ch = std::min(it.M, (sg + 1) * it.cps) - sg * it.cps;
        it.o_seg[sg] = ar.take(sizeof(uint32_t) * (size_t)ch * (size_t)it.hw + 64);
      }
    for (auto &it : items) it.o_packed = it.o_seg[0];
  }
  const size_t o_tables_end = ar.off;
  const size_t total = ar.off;
  int rc;
  if ((rc = ctx->ensure_device(total)) || (rc = ctx->ensure_host(total)) || (rc = ctx->ensure_events((size_t)count + 17, 16))) return rc;
  const size_t ev_meta = (size_t)count + 16; // (the copy groups of the tables use the events before it: at most count, or nine)

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
    d.logits = (it.prm.flags & FGMM_PARAMS_LOGITS) ? 1 : 0;
    d.yq = it.yq;
    d.chan_min = reinterpret_cast<float *>(ctx->d_ws + it.o_min);
    d.chan_max = reinterpret_cast<float *>(ctx->d_ws + it.o_max);
    d.chan_nz = it.y ? reinterpret_cast<int32_t *>(ctx->d_ws + it.o_nz) : nullptr;
    d.chan_list = it.y ? reinterpret_cast<int32_t *>(ctx->d_ws + it.o_list) : nullptr;
    d.packed = reinterpret_cast<uint32_t *>(ctx->d_ws + it.o_packed);
    d.seg_b[0] = d.seg_b[1] = d.seg_b[2] = INT32_MAX;
    d.packed_seg[0] = d.packed;
    if (segmented) {
      d.cps = it.cps;
      for (int sg = 0; sg < it.n_seg; ++sg) {
        d.packed_seg[sg] = reinterpret_cast<uint32_t *>(ctx->d_ws + it.o_seg[sg]);
        if (sg + 1 < it.n_seg) d.seg_b[sg] = (sg + 1) * it.cps;
      }
    }
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
  const int vec = vec4 ? (ctx->opt.enc_vec == 1 ? 1 : ctx->opt.enc_vec == 2 ? 2 : 4) : 1; // option "enc_vec" = 1, 2: A/B narrower loads
  int64_t n_max = 0;
  bool linear = ctx->opt.enc_linear != 0; // option "enc_linear" = 0: A/B the per-channel grid
  for (auto &it : items) {
    n_max = std::max(n_max, (int64_t)it.M * it.hw);
    linear = linear && it.hw % (64 * vec) == 0;
  }
  LAUNCH_TRY(launch_symtab(dd, count, M_max, hw_max, n_max, linear, mode, vec, items[0].clamp != 0,
                           items[0].prm.dtype == FGMM_F16, stream));
  if ((rc = ctx->prof_end(0, stream))) return rc;
  // ---- tables back to the host: small region first, then one copy + event per item ----------------
  HIP_TRY(hipMemcpyAsync(ctx->h_ws + o_small, ctx->d_ws + o_small, small_bytes, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipEventRecord(ctx->events[ev_meta], stream));
  // the per-item tables are contiguous in the workspace (in `order`): a handful of large copies instead of one per item
  std::vector<int> group_of(count);
  int n_groups = 0;
  if (segmented) {
    // the segments in the order they were laid out (tails of all bitstreams first), in about eight copies
    struct Chunk {
      int item, sg;
      size_t beg, end;
    };
    std::vector<Chunk> chunks;
    size_t bytes = 0;
    for (int sg = kEncSegs - 1; sg >= 0; --sg)
      for (int i : order) {
        const EncItem &it = items[i];
        if (sg >= it.n_seg) continue;
        const int32_t ch = std::min(it.M, (sg + 1) * it.cps) - sg * it.cps;
        chunks.push_back(Chunk{i, sg, it.o_seg[sg], it.o_seg[sg] + sizeof(uint32_t) * (size_t)ch * (size_t)it.hw});
        bytes += chunks.back().end - chunks.back().beg;
      }
    const size_t per_group = bytes / 8 + 1;
    for (size_t c0 = 0; c0 < chunks.size(); ++n_groups) {
      size_t c1 = c0, got = 0;
      do {
        got += chunks[c1].end - chunks[c1].beg;
        ++c1;
      } while (c1 < chunks.size() && got < per_group);
      const size_t beg = chunks[c0].beg, end = chunks[c1 - 1].end; // (laid out in this order: one contiguous range)
      if (end > beg) HIP_TRY(hipMemcpyAsync(ctx->h_ws + beg, ctx->d_ws + beg, end - beg, hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipEventRecord(ctx->sleep_events[n_groups], stream));
      for (size_t c = c0; c < c1; ++c) items[chunks[c].item].seg_group[chunks[c].sg] = n_groups;
      c0 = c1;
    }
    (void)o_tables_end;
  } else {
    size_t table_bytes = 0;
    for (auto &it : items) table_bytes += sizeof(uint32_t) * (size_t)it.M * (size_t)it.hw;
    const size_t per_group = count >= 16 ? table_bytes / 6 + 1 : 0; // (fewer than 16 bitstreams: a copy each)
    for (int p0 = 0; p0 < count; ++n_groups) {
      int p1 = p0;
      size_t got = 0;
      do {
        got += sizeof(uint32_t) * (size_t)items[order[(size_t)p1]].M * (size_t)items[order[(size_t)p1]].hw;
        ++p1;
      } while (p1 < count && got < per_group);
      const EncItem &a = items[order[(size_t)p0]], &b = items[order[(size_t)p1 - 1]];
      const size_t beg = a.o_packed, end = b.o_packed + sizeof(uint32_t) * (size_t)b.M * (size_t)b.hw;
      if (end > beg) HIP_TRY(hipMemcpyAsync(ctx->h_ws + beg, ctx->d_ws + beg, end - beg, hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipEventRecord(ctx->events[n_groups], stream));
      for (int p = p0; p < p1; ++p) group_of[order[(size_t)p]] = n_groups;
      p0 = p1;
    }
  }
  tr.mark("enqueued");
  double marks[5] = {tr.ms(), 0, 0, 0, 0}; // the call log: enqueued | kernels + side information here | jobs out | last table (segment) seen landed | last job done
  HIP_TRY(hipEventSynchronize(ctx->events[ev_meta]));
  tr.mark("kernels + meta landed");
  marks[1] = tr.ms();
  ctx->stat[0] = 0;
  for (auto &it : items) ctx->stat[0] += sizeof(uint32_t) * (unsigned long long)it.M * (unsigned long long)it.hw;

  // ---- host side: per item side information, then one rANS job per item -----------------------------
  std::vector<std::vector<int32_t>> wide_syms(count); // only for bypass symbols beyond int16 (rare)
  // segmented tables: an encoder asks for a segment before it enters it and SLEEPS on the event of the copy that carries it
  // (hipEventBlockingSync, like the decode workers on their pieces: no thread of this process polls or hands events on)
  struct SegWaitArg {
    hipEvent_t *ev;       // the copy groups' events
    const int32_t *group; // EncItem::seg_group
    Trace *tr;
    double waited, last; // the time spent waiting, per job; when the last wait returned
  };
  std::vector<SegWaitArg> seg_args((size_t)count);
  PoolDrain drain{ctx->pool};
  auto seg_wait = +[](void *arg, int sg) -> int {
    SegWaitArg *a = static_cast<SegWaitArg *>(arg);
    const double t0 = a->tr->ms();
    const bool ok = hipEventSynchronize(a->ev[a->group[sg]]) == hipSuccess;
    a->last = a->tr->ms();
    a->waited += a->last - t0;
    return ok ? FGMM_OK : FGMM_ERR_HIP;
  };
  // jobs: runs of up to `enc_ways` bitstreams adjacent in `order` (similar sizes), coded in turn by one worker; a bitstream that
  // is a worker's fair share by itself (>= 1 / (2 * workers) of the call) is a job of its own - sixteen large pairs on eight
  // workers would leave the other eight idle
  std::vector<int> job_last((size_t)count, 0), job_first((size_t)count, 0); // by position in `order`
  {
    int64_t n_total = 0;
    for (auto &it : items) n_total += (int64_t)it.M * it.hw;
    const int64_t big = ctx->opt.enc_ways > 0 ? INT64_MAX : n_total / (2 * (int64_t)enc_T) + 1;
    // ... and the streams that would form a last, half-empty round of pairs (48 on 16 workers: 16 pairs, then 8 pairs on 8
    // workers while 8 idle) are singles instead: every worker gets a pair and a single
    const int tail = ctx->opt.enc_ways > 0 || enc_ways != 2 ? 0 : count % (2 * enc_T);
    const int first_single = tail <= enc_T ? count - tail : count;
    for (int p = 0; p < count;) {
      int q = p + 1;
      const EncItem &a = items[order[(size_t)p]];
      if (!a.symbuf && (int64_t)a.M * a.hw < big && p < first_single)
        while (q < first_single && q - p < enc_ways && !items[order[(size_t)q]].symbuf && (int64_t)items[order[(size_t)q]].M * items[order[(size_t)q]].hw < big) ++q;
      for (int r = p; r < q; ++r) job_first[(size_t)r] = p, job_last[(size_t)r] = q - 1;
      p = q;
    }
  }
  std::vector<EncItem *> job_items((size_t)count);
  for (int p = 0; p < count; ++p) job_items[(size_t)p] = &items[order[(size_t)p]];
  for (int pos = 0; pos < count; ++pos) {
    const int i = order[(size_t)pos];
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
    // More bitstreams than workers: the members of a job go to one worker, coded in turn symbol by symbol
    // (rans_encode_symtab_ways) - two dependency chains share a core: 1.5 instead of 2.4 ns/symbol.  The job is submitted with
    // its last member (the later copies hold the smaller tables)
    if (pos < job_last[(size_t)pos]) continue;
    const int g_begin = job_first[(size_t)pos], n_in = pos - g_begin + 1;
    int last_group = 0;
    for (int r = g_begin; r <= pos; ++r) last_group = std::max(last_group, group_of[order[(size_t)r]]);
    if (!segmented) {
      HIP_TRY(hipEventSynchronize(ctx->events[last_group])); // copies complete in the order they were queued
      marks[3] = tr.ms();
    }
    const char *h_ws = ctx->h_ws;
    if (segmented) seg_args[(size_t)i] = SegWaitArg{ctx->sleep_events.data(), it.seg_group, &tr, 0.0, 0.0};
    SegWaitArg *const seg_arg = segmented ? &seg_args[(size_t)i] : nullptr;
    EncItem *const *first = &job_items[(size_t)g_begin];
    const double t_sub = tr.ms();
    for (int q = 0; q < n_in; ++q) first[q]->t_sub = t_sub;
    auto job = [first, n_in, h_ws, &tr, seg_arg, seg_wait] {
      const double t_start = tr.ms();
      if (seg_arg) { // (n_in == 1) the table lies in segments that land tail first: the encoder asks for each before it enters it
        EncItem &e = *first[0];
        e.t_start = t_start;
        SegTable t;
        t.n_seg = e.n_seg;
        t.seg_len = (int64_t)e.cps * e.hw;
        for (int sg = 0; sg < kEncSegs; ++sg) t.seg[sg] = sg < e.n_seg ? reinterpret_cast<const uint32_t *>(h_ws + e.o_seg[sg]) : nullptr;
        t.wait = seg_wait;
        t.arg = seg_arg;
        const int64_t stride = e.ckpt_stride;
        const int64_t n_ck = stride > 0 && e.job_n > 0 ? (e.job_n - 1) / stride : 0;
        int rc = FGMM_OK;
        if (n_ck > 0) {
          e.ckpt = static_cast<fgmm_ckpt *>(malloc(sizeof(fgmm_ckpt) * (size_t)n_ck));
          if (!e.ckpt) rc = FGMM_ERR_NOMEM;
          e.n_ckpt = e.ckpt ? n_ck : 0;
        }
        if (rc == FGMM_OK) rc = rans_encode_symtab_segs(t, e.job_syms, e.job_n, e.job_bypass, &e.bytes, &e.bytes_len, n_ck > 0 ? stride : 0, e.ckpt);
        if (rc != FGMM_OK) {
          free(e.ckpt);
          e.ckpt = nullptr;
          e.n_ckpt = 0;
        }
        e.status = rc;
        e.t_end = tr.ms();
        return;
      }
      if (n_in == 1 && first[0]->symbuf) {
        first[0]->t_start = t_start;
        first[0]->status = fgmm_symbuf_append_symtab(first[0]->symbuf, reinterpret_cast<const uint32_t *>(h_ws + first[0]->o_packed), first[0]->job_syms, first[0]->job_n);
        first[0]->t_end = tr.ms();
        return;
      }
      const uint32_t *packed[kMaxEncWays];
      const int32_t *syms[kMaxEncWays];
      int64_t n[kMaxEncWays], nb[kMaxEncWays];
      uint8_t **out[kMaxEncWays];
      size_t *len[kMaxEncWays];
      fgmm_ckpt *ck[kMaxEncWays];
      const int64_t stride = first[0]->ckpt_stride; // one stride per call (checked at the boundary)
      int rc = FGMM_OK;
      for (int q = 0; q < n_in; ++q) {
        first[q]->t_start = t_start;
        packed[q] = reinterpret_cast<const uint32_t *>(h_ws + first[q]->o_packed);
        syms[q] = first[q]->job_syms;
        n[q] = first[q]->job_n;
        nb[q] = first[q]->job_bypass;
        out[q] = &first[q]->bytes;
        len[q] = &first[q]->bytes_len;
        ck[q] = nullptr;
        const int64_t n_ck = stride > 0 && n[q] > 0 ? (n[q] - 1) / stride : 0;
        if (n_ck > 0) {
          ck[q] = first[q]->ckpt = static_cast<fgmm_ckpt *>(malloc(sizeof(fgmm_ckpt) * (size_t)n_ck));
          if (!ck[q]) rc = FGMM_ERR_NOMEM;
          first[q]->n_ckpt = ck[q] ? n_ck : 0;
        }
      }
      if (rc == FGMM_OK) rc = rans_encode_symtab_ways(n_in, packed, syms, n, nb, out, len, stride, ck);
      if (rc != FGMM_OK)
        for (int q = 0; q < n_in; ++q) {
          free(first[q]->ckpt);
          first[q]->ckpt = nullptr;
          first[q]->n_ckpt = 0;
        }
      const double t_end = tr.ms();
      for (int q = 0; q < n_in; ++q) {
        first[q]->status = rc;
        first[q]->t_end = t_end;
      }
    };
    if (count == 1) job(); else ctx->pool->submit(job);
  }
  tr.mark(segmented ? "jobs out" : "all tables landed, jobs out");
  marks[2] = tr.ms();
  if (count > 1) ctx->pool->wait_all();
  tr.mark("host rANS done");
  {
    double busy = 0, wait = 0;
    for (int i = 0; i < count; ++i) {
      const double w = segmented ? seg_args[(size_t)i].waited : 0.0;
      if (segmented) marks[3] = std::max(marks[3], seg_args[(size_t)i].last);
      marks[4] = std::max(marks[4], items[i].t_end);
      busy += items[i].t_end - items[i].t_start - w;
      wait += w;
    }
    ctx->log_call(0, count, tr, marks, busy, wait);
  }
  if (tr.level > 1)
    for (int i = 0; i < count; ++i)
      fprintf(stderr, "[fgmm encode]   item %2d  submitted %7.3f  job %7.3f .. %7.3f  (%.3f ms, %.3f of it waiting for its table's segments)\n", i,
              items[i].t_sub, items[i].t_start, items[i].t_end, items[i].t_end - items[i].t_start, segmented ? seg_args[(size_t)i].waited : 0.0);
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
  const fgmm_ckpt *ckpt = nullptr;      // checkpoints of the bitstream (out-of-band notes of its encoder) or null
  int64_t n_ckpt = 0, ckpt_stride = 0;
  int status = FGMM_OK;
  // derived
  int32_t n_ch = 0;
  int64_t n = 0;
  size_t o_list = 0, o_rank = 0;
  int hdr_form = 4;
  uint32_t ef_min = kTabEfMin;
  int32_t tl = 0;     // latents per block of the single-pass kernel; 0: generic two-pass path
  int64_t nblk = 0;   // blocks of tl latents
  uint64_t table_bytes = 0; // headers + block offsets + rows that crossed PCIe
  // how the tables reach the host (see decode_batch): in n_piece pieces (block ranges)
  int n_piece = 1;
  TabPiece piece[kMaxPieces] = {};
  hipEvent_t piece_ev[kMaxPieces] = {}; // recorded (this call) before the item's job is submitted
  hsa_signal_t piece_sig[kMaxPieces] = {}; // ... or, for a piece copied by an SDMA engine (HsaCopier), the signal of its copy (handle 0: the event)
  char *h_out = nullptr;                // pinned: decoded symbols (host-written, read by the scatter kernel)
  int wide = 0; // h_out holds int32 symbols (some symbol outside int16), else int16
  int64_t narrowed = 0; // symbols already converted to int16 in h_out (piece by piece)
  // schedule state, guarded by the call's mutex: pieces whose copy is queued | next piece to decode | a worker holds the
  // item | it is in the ready heap
  int queued = 0, next_piece = 0;
  bool busy = false, in_ready = false;
  TabDecoder dec;
  TabView view;
  int32_t *sym = nullptr;               // int32 symbols (sym_host_out or a slice of the context's scratch)
  std::atomic<int> done{0};
  bool rounds = false; // its symbols go back to the GPU round by round (ScatDesc), not in one launch when it has finished
  // checkpointed streams decode as independent SEGMENTS (n_seg = n_ckpt + 1; 0: sequentially, piece by piece)
  int n_seg = 0, next_seg_push = 0;       // next_seg_push: guarded by the call's mutex
  int64_t piece_end[kMaxPieces] = {};     // one past the last latent of every piece (known when the call is planned)
  std::atomic<int> segs_left{0}, ckpt_bad{0}, wide_any{0};
  std::vector<uint32_t> enc_aligned;      // a misaligned bitstream of a checkpointed item, copied ONCE (every segment starts a decoder on it)
  double t_taken = 0, t_start = 0, t_end = 0, t_waited = 0, t_lastland = 0, t_work = 0; // trace level 2: job timeline
  DecItem() = default;
  DecItem(const DecItem &) = delete;
};

constexpr size_t kCounterBytes = kTabCounters * sizeof(unsigned long long); // per launch unit, see DecDesc::counters

// frees what a call allocated outside the context's reusable buffers (rare paths: overflow re-runs, generic items)
struct TempDevice {
  std::vector<void *> v;
  ~TempDevice() {
    for (void *p : v) (void)hipFree(p);
  }
  int alloc(size_t bytes, char **out) {
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, std::max<size_t>(bytes, 256)));
    v.push_back(p);
    *out = static_cast<char *>(p);
    return FGMM_OK;
  }
};

// Decode, batched and pipelined.
//   caller's stream : [H2D descriptors][tab_kernel unit 0][tab_kernel unit 1] ...          [y_hat scatter, item by item]
//   aux stream      : after unit u's kernel -> D2H of its four counters (cursor = bytes of rows placed)
//   this thread     : unit u's size known -> a pinned range of exactly that size; copy stream: ONE copy per unit
//                     (headers + block offsets + rows; few large copies reach 55.7 GB/s, a copy per item 51);
//                     the unit's pieces are marked queued: the workers' (bitstream, piece) tasks become ready
//   host workers    : take the earliest-landing ready task, sleep on its copy's event, decode the piece, hand the
//                     bitstream's coder state back; symbols go to pinned memory as int16 (int32 if one does not fit)
//   caller's stream : scatter kernels read them from there and write the float latent - round by round (piece r of every
//                     bitstream in one launch, as soon as all have decoded it; zero channels up front), so that the last
//                     decoder is followed by its last piece only
// A launch unit is one ROUND of pieces: block range p of every item of the call (the first round is cut into small
// launches: the first tables reach the host as early as possible).  The single-pass kernel needs no host decision
// before its rows exist - they go to a provisioned staging area, placed by a cursor - so every kernel of the call is
// enqueued up front and the PCIe transfer, the longest leg, starts as soon as the first small unit is done.
// Why pieces: a bitstream decodes sequentially (~9 ns/symbol), so whatever lands last leaves that much host work behind
// it.  Every item therefore crosses in shrinking pieces, piece-major; its decoder follows the pieces as they land,
// on whichever worker is free, and what remains after the final copy is 1/36 of each bitstream.
// Items whose half-width does not fit the single-pass kernel (tab_tl() == 0) take the generic two-pass kernels, one
// item at a time, synchronously (8-byte headers past max_bs 16382).
// can this item be decoded by the GPU's segment decoder?  (checkpoints exactly as an encoder notes them for this many symbols;
// a float latent to write; a half-width whose window fits the kernel's 16-bit fields)
static bool gpu_decodable(const DecItem &it, int64_t n) {
  return it.y_hat && !it.sym_host_out && it.ckpt && it.n_ckpt > 0 && it.ckpt_stride >= 256 && !(it.ckpt_stride & (it.ckpt_stride - 1)) &&
         n > 0 && it.n_ckpt == (n - 1) / it.ckpt_stride && it.n_ckpt < (1 << 24) && it.max_bs >= 0 && 2 * (int64_t)it.max_bs + 2 <= 2048 /* kSegCapE: a latent's window in the wave's LDS */ &&
         it.enc_len >= 8 && !(it.enc_len & 3) && it.stride_p == 1;
}

// Checkpointed bitstreams decoded ON THE GPU (segdec_kernel: one workgroup of two or three waves per segment, no tables, nothing but the bitstreams and
// their notes crosses PCIe).  `which`: the items to decode; on return `redo` holds those whose segments did not all verify
// (a row the kernel leaves to the reference's bisection, wrong notes): the caller sends them through the table path.
int decode_batch_gpu(fgmm_ctx *ctx, hipStream_t stream, std::vector<DecItem> &items, const std::vector<int> &which, int mode,
                     std::vector<int> &redo) {
  Trace tr("decode-gpu", (int)ctx->opt.trace);
  const int count = (int)which.size();
  const bool clamped = items[which[0]].clamp != 0, f16 = items[which[0]].prm.dtype == FGMM_F16;
  // ---- device workspace: [descs][segment list][per item: channel list | status | checkpoints | bitstream]
  Arena ar;
  int64_t n_segs = 0;
  for (int k = 0; k < count; ++k) n_segs += items[which[k]].n_ckpt + 1;
  const size_t o_descs = ar.take(sizeof(SegDesc) * (size_t)count);
  const size_t o_segs = ar.take(sizeof(SegRef) * (size_t)n_segs);
  struct Off {
    size_t list, ckpt, words, status;
  };
  std::vector<Off> off((size_t)count);
  for (int k = 0; k < count; ++k) {
    DecItem &it = items[which[k]];
    off[(size_t)k].list = ar.take(sizeof(int32_t) * (size_t)std::max(it.M, 1), 16); // live channels, then dead ones
    off[(size_t)k].ckpt = ar.take(sizeof(fgmm_ckpt) * (size_t)it.n_ckpt, 16);
    off[(size_t)k].words = ar.take(it.enc_len, 16);
  }
  const size_t upload_bytes = ar.off;
  const size_t o_status = ar.take(sizeof(uint32_t) * (size_t)n_segs, 256);
  {
    size_t at = o_status;
    for (int k = 0; k < count; ++k) {
      off[(size_t)k].status = at;
      at += sizeof(uint32_t) * (size_t)(items[which[k]].n_ckpt + 1);
    }
  }
  int rc;
  if ((rc = ctx->ensure_device(ar.off)) || (rc = ctx->ensure_host(ar.off))) return rc;
  SegDesc *hd = reinterpret_cast<SegDesc *>(ctx->h_ws + o_descs);
  SegRef *hs = reinterpret_cast<SegRef *>(ctx->h_ws + o_segs);
  int64_t max_dead = 0;
  for (int k = 0; k < count; ++k) {
    DecItem &it = items[which[k]];
    int32_t *list = reinterpret_cast<int32_t *>(ctx->h_ws + off[(size_t)k].list);
    int r = 0, dead = it.n_ch;
    for (int c = 0; c < it.M; ++c) {
      if (!it.zero_bitmap || it.zero_bitmap[c] != 0) list[r++] = c;
      else list[dead++] = c;
    }
    max_dead = std::max<int64_t>(max_dead, it.M - it.n_ch);
    memcpy(ctx->h_ws + off[(size_t)k].ckpt, it.ckpt, sizeof(fgmm_ckpt) * (size_t)it.n_ckpt);
    memcpy(ctx->h_ws + off[(size_t)k].words, it.enc, it.enc_len);
    SegDesc &d = hd[k];
    memset(&d, 0, sizeof d);
    d.scales = it.prm.scales;
    d.means = it.prm.means;
    d.weights = it.prm.weights;
    d.stride_k = it.prm.stride_k;
    d.stride_c = it.prm.stride_c;
    d.stride_p = it.stride_p;
    d.hw = it.hw;
    d.n = it.n;
    d.chan_list = reinterpret_cast<const int32_t *>(ctx->d_ws + off[(size_t)k].list);
    d.max_bs = it.max_bs;
    d.clamp = it.clamp;
    d.logits = (it.prm.flags & FGMM_PARAMS_LOGITS) ? 1 : 0;
    d.words = reinterpret_cast<const uint32_t *>(ctx->d_ws + off[(size_t)k].words);
    d.n_words = (int64_t)(it.enc_len / 4);
    d.ckpt = reinterpret_cast<const fgmm_ckpt *>(ctx->d_ws + off[(size_t)k].ckpt);
    d.n_ckpt = it.n_ckpt;
    d.stride = it.ckpt_stride;
    d.y_hat = it.y_hat;
    d.status = reinterpret_cast<uint32_t *>(ctx->d_ws + off[(size_t)k].status);
    d.dead_list = d.chan_list + it.n_ch; // channels without a coded symbol are zero in y_hat (entropy_models.py:903-908)
    d.n_dead = it.M - it.n_ch;
  }
  // The segments in the order of the launch's workgroups: HEAVIEST FIRST.  A segment costs its symbols plus its edges, and a latent's
  // window is wide where its symbol is expensive - so the words a segment takes of the bitstream (the distance between its notes) rank
  // the segments by weight; with the heavy ones (3x the median on Kodak-like latents) in front, the launch does not end on one that
  // started last.  (The notes are not trusted: a wrong one spoils an order, nothing else.)
  {
    constexpr int kBuckets = 256;
    std::vector<uint32_t> wgt((size_t)n_segs);
    uint32_t w_max = 1;
    int64_t at = 0;
    for (int k = 0; k < count; ++k) {
      const DecItem &it = items[which[k]];
      const uint64_t end_all = it.enc_len / 4 - 2;
      uint64_t prev = 0;
      for (int64_t sgm = 0; sgm <= it.n_ckpt; ++sgm) {
        const uint64_t pos = sgm < it.n_ckpt ? it.ckpt[sgm].pos : end_all;
        const uint64_t wds = pos >= prev ? pos - prev : 0;
        wgt[(size_t)at] = (uint32_t)std::min<uint64_t>(wds, 0x7FFFFFFFu);
        w_max = std::max(w_max, wgt[(size_t)at]);
        prev = pos;
        ++at;
      }
    }
    int64_t first[kBuckets + 1] = {};
    auto bucket = [&](uint32_t wv) { return kBuckets - 1 - (int)((uint64_t)wv * (kBuckets - 1) / w_max); }; // heavy -> bucket 0
    for (int64_t q = 0; q < n_segs; ++q) ++first[bucket(wgt[(size_t)q]) + 1];
    for (int b = 0; b < kBuckets; ++b) first[b + 1] += first[b];
    at = 0;
    for (int k = 0; k < count; ++k)
      for (int64_t sgm = 0; sgm <= items[which[k]].n_ckpt; ++sgm, ++at) hs[first[bucket(wgt[(size_t)at])]++] = SegRef{k, (int32_t)sgm};
  }
  HIP_TRY(hipMemcpyAsync(ctx->d_ws, ctx->h_ws, upload_bytes, hipMemcpyHostToDevice, stream));
  LAUNCH_TRY(launch_segzero(reinterpret_cast<const SegDesc *>(ctx->d_ws + o_descs), count, max_dead, stream));
  if ((rc = ctx->prof_begin(3, stream))) return rc;
  LAUNCH_TRY(launch_segdec(reinterpret_cast<const SegDesc *>(ctx->d_ws + o_descs), reinterpret_cast<const SegRef *>(ctx->d_ws + o_segs), n_segs, mode,
                           clamped, f16, stream));
  if ((rc = ctx->prof_end(3, stream))) return rc;
  HIP_TRY(hipMemcpyAsync(ctx->h_ws + o_status, ctx->d_ws + o_status, sizeof(uint32_t) * (size_t)n_segs, hipMemcpyDeviceToHost, stream));
  tr.mark("enqueued");
  const double t_enq = tr.ms();
  HIP_TRY(hipStreamSynchronize(stream));
  tr.mark("segments decoded");
  {
    const double t_done = tr.ms(), mk[5] = {t_enq, t_enq, t_enq, t_done, t_done};
    ctx->log_call(2, count, tr, mk, 0.0, 0.0);
  }
  for (int k = 0; k < count; ++k) {
    DecItem &it = items[which[k]];
    const uint32_t *st = reinterpret_cast<const uint32_t *>(ctx->h_ws + off[(size_t)k].status);
    uint32_t worst = 0;
    for (int64_t sgm = 0; sgm <= it.n_ckpt; ++sgm) worst = std::max(worst, st[sgm]);
    if (worst != kSegOk) redo.push_back(which[k]);
    else it.status = FGMM_OK, it.done.store(1);
    if (tr.on && worst != kSegOk) {
      int64_t first = 0, n_bad = 0;
      for (int64_t sgm = it.n_ckpt; sgm >= 0; --sgm)
        if (st[sgm] != kSegOk) first = sgm, ++n_bad;
      fprintf(stderr, "[fgmm decode-gpu]   item %d: %lld of %lld segments not ok, the first: segment %lld status %u\n", which[k], (long long)n_bad,
              (long long)it.n_ckpt + 1, (long long)first, st[first]);
    }
  }
  ctx->stat[1] = 0; // no decode-side tables at all
  ctx->stat[2] = ctx->stat[3] = 0;
  ctx->stat[4] = (unsigned long long)(count - (int)redo.size());
  ctx->stat[5] = (unsigned long long)redo.size();
  for (int k = 0; k < count; ++k) ctx->stat[2] += (unsigned long long)items[which[k]].n;
  return FGMM_OK;
}

int decode_batch(fgmm_ctx *ctx, hipStream_t stream, std::vector<DecItem> &items, int mode) {
  const int count = (int)items.size();
  if (count == 0) return FGMM_OK;
  // ---- checkpointed bitstreams go to the GPU's segment decoder; whatever it does not take or cannot finish, and everything
  // else, takes the table path below (option "gpu_decode": 0 / 1 = when possible, 2 = never)
  if (ctx->opt.gpu_decode != 2) {
    std::vector<int> gpu, rest;
    for (int i = 0; i < count; ++i) {
      DecItem &it = items[i];
      int n_ch = 0;
      for (int c = 0; c < it.M; ++c) n_ch += it.zero_bitmap ? (it.zero_bitmap[c] != 0) : 1;
      it.n_ch = n_ch;
      it.n = (int64_t)n_ch * it.hw;
      (gpu_decodable(it, it.n) && it.clamp == items[0].clamp && it.prm.dtype == items[0].prm.dtype ? gpu : rest).push_back(i);
    }
    // Is the GPU the faster decoder for this call?  A segment's waves decode it at ~0.65 us per symbol however empty the chip is
    // (its widest rows included), and the chip as a whole at ~0.35 ns per symbol (Kodak-like latents); the host decodes at ~12 ns per symbol and worker and
    // is fed at 58 B per latent over PCIe.  Many segments (a batch, a 4K image's group): the GPU, by 2-4x; one Kodak half
    // in a few hundred long segments: the host workers.  ("gpu_decode" = 1: always)  The rates below are those of the boxes this
    // was measured on (MI355X + EPYC 9575F, 16 workers, PCIe 5 x16): another host overrides the choice with the option.
    if (!gpu.empty() && ctx->opt.gpu_decode == 0) {
      double syms = 0, stride_max = 0, work = 0;
      for (int i : gpu) {
        syms += (double)items[i].n;
        stride_max = std::max(stride_max, (double)std::min<int64_t>(items[i].ckpt_stride, items[i].n));
        work += (double)(items[i].n_ckpt + 1);
      }
      const double t_gpu = std::max(stride_max * 0.65, syms * 0.00035) + 100.0;
      const double workers = std::min<double>(std::max(ctx->pool->size(), 1), work);
      const double t_host = std::max(syms * 0.012 / workers, syms * 58.0 / 55700.0) + 450.0 + 3.0 * work / workers; // + a segment's set-up
      if (t_gpu >= t_host) {
        for (int i : gpu) rest.push_back(i);
        gpu.clear();
        std::sort(rest.begin(), rest.end());
      }
    }
    if (!gpu.empty()) {
      std::vector<int> redo;
      int rc = decode_batch_gpu(ctx, stream, items, gpu, mode, redo);
      if (rc) return rc;
      for (int i : redo) rest.push_back(i);
      if (rest.empty()) return FGMM_OK;
      // the rest through the table path, as a batch of its own (the notes of a bitstream that failed them are dropped)
      std::vector<DecItem> sub(rest.size());
      for (size_t k = 0; k < rest.size(); ++k) {
        const DecItem &s0 = items[rest[k]];
        DecItem &t = sub[k];
        t.enc = s0.enc, t.enc_len = s0.enc_len, t.prm = s0.prm, t.stride_p = s0.stride_p, t.M = s0.M, t.hw = s0.hw, t.clamp = s0.clamp;
        t.max_bs = s0.max_bs, t.zero_bitmap = s0.zero_bitmap, t.y_hat = s0.y_hat, t.sym_host_out = s0.sym_host_out;
        const bool failed = std::find(redo.begin(), redo.end(), rest[k]) != redo.end();
        if (!failed) t.ckpt = s0.ckpt, t.n_ckpt = s0.n_ckpt, t.ckpt_stride = s0.ckpt_stride;
      }
      const int64_t saved = ctx->opt.gpu_decode;
      ctx->opt.gpu_decode = 2;
      rc = decode_batch(ctx, stream, sub, mode);
      ctx->opt.gpu_decode = saved;
      for (size_t k = 0; k < rest.size(); ++k) items[rest[k]].status = sub[k].status;
      return rc;
    }
  }
  Trace tr("decode", (int)ctx->opt.trace);
  int rc;
  if ((rc = ctx->ensure_streams())) return rc;
  int cap_e = (int)std::min<int64_t>(std::max<int64_t>(ctx->opt.tab_cap_e, 256), 32768) & ~31;
  if (ctx->opt.tab_cap_e == kTabCapE) // (the default: a half-width that only fits the wider budget gets it)
    for (const DecItem &it : items)
      if (!tab_tl(it.max_bs, cap_e) && tab_tl(it.max_bs, kTabCapEWide)) {
        cap_e = kTabCapEWide;
        break;
      }
  const bool clamped = items[0].clamp != 0, f16 = items[0].prm.dtype == FGMM_F16;
  // Elias-Fano rows (long rows: ef_min) are 18 % fewer bytes than uint16 rows and 40 % more nanoseconds to search (57.6 B and
  // 12 ns per latent against 70 B and 8.7 ns on the Kodak workload): with P = min(workers, bitstreams) decoders at work a
  // latent costs max(bytes / 55.7 GB/s, ns / P) - Elias-Fano rows pay when 12 / P < 70 B / 55.7 GB/s = 1.26 ns, P >= 10.
  // (a checkpointed bitstream keeps as many decoders busy as it has segments)
  // Segments pay when a call has fewer bitstreams than workers (one image, ELIC's stages of a few images); a call with a
  // bitstream per worker or more keeps them all busy piece by piece and would only pay the segments' bookkeeping
  // (24 Kodak halves on 16 workers: 8.57 ms of decode per step sequentially, 8.73 in segments): the notes are ignored there.
  const bool use_ckpt = ctx->opt.ckpt_decode == 1 || (ctx->opt.ckpt_decode == 0 && count < std::max(ctx->pool->size(), 1));
  int64_t streams_of_work = 0;
  for (auto &it : items) {
    if (!use_ckpt) it.ckpt = nullptr, it.n_ckpt = 0;
    streams_of_work += it.ckpt && it.n_ckpt > 0 ? it.n_ckpt + 1 : 1;
  }
  const int decoders = (int)std::min<int64_t>(std::max(ctx->pool->size(), 1), streams_of_work);
  const uint32_t ef_min = ctx->opt.ef_rows == 1 || (ctx->opt.ef_rows == 0 && decoders >= 10) ? (uint32_t)ctx->opt.ef_min : kTabNoEf;

  // ---- items: coded channels, header form, path --------------------------------------------------------------------
  Arena ar; // device workspace, mirrored in h_ws up to the counters
  std::vector<int> fast, generic;
  for (int i = 0; i < count; ++i) {
    DecItem &it = items[i];
    if (it.max_bs < 0 || it.max_bs > FGMM_MAX_BS)
      return fail(FGMM_ERR_UNSUPPORTED, "max_bs_value %d outside [0, %d]", it.max_bs, FGMM_MAX_BS);
    it.n_ch = 0;
    for (int c = 0; c < it.M; ++c) it.n_ch += it.zero_bitmap ? (it.zero_bitmap[c] != 0) : 1;
    it.n = (int64_t)it.n_ch * it.hw;
    it.o_list = ar.take(sizeof(int32_t) * std::max(it.n_ch, 1), 16);
    it.o_rank = ar.take(sizeof(int32_t) * std::max(it.M, 1), 16);
    it.hdr_form = tab_hdr_form(it.max_bs);
    it.ef_min = ef_min;
    it.tl = tab_tl(it.max_bs, cap_e);
    it.nblk = it.tl ? (it.n + it.tl - 1) / it.tl : 0;
    if (it.nblk > 0x7FFFFFFFll) it.tl = 0, it.nblk = 0;
    (it.tl ? fast : generic).push_back(i);
  }
  const int n_fast = (int)fast.size();

  // ---- launch units over the fast items (in item order) -----------------------------------------------------------
  struct Part { // one item's share of a unit
    int item;
    int64_t blk_begin, blk_end;
    size_t o_hdr, o_blkoff; // within the unit's range
    int piece;              // which piece of the item this is
  };
  struct Unit {
    std::vector<Part> parts;
    size_t fixed = 0, rows_cap = 0; // bytes: headers + block offsets | provisioned rows
    size_t o_stage = 0;             // where the unit's range starts in the staging area
    char *d_range = nullptr;        // device: [fixed | rows]
    size_t o_scan = 0;              // the unit's look-back states in the workspace (one word per block, launch order)
    int64_t scan_total = 0;         // blocks of the launch
    int placement = 1;              // DecDesc::placement
  };
  std::vector<Unit> units;
  // Every item crosses in `np` pieces (block ranges), PIECE-MAJOR: piece 0 of every item, then piece 1 ...  The host
  // workers take (item, piece) tasks as they land - a bitstream decodes sequentially, but its coder state moves from worker
  // to worker between pieces - so every worker starts on the first round, the load balances whatever count / workers is,
  // and what is left after the last copy is the last piece of each item.  Pieces therefore shrink linearly (4 pieces: 40,
  // 30, 20, 10 % of the blocks).  The first round is cut into small launches + copies (the first tables reach the host as
  // early as possible), later rounds are one launch + one copy each.
  int np = (int)std::min<int64_t>(std::max<int64_t>(ctx->opt.pieces, 1), kMaxPieces);
  {
    int64_t lat = 0, lat_max = 0;
    for (int k = 0; k < n_fast; ++k) lat += items[fast[k]].n, lat_max = std::max(lat_max, items[fast[k]].n);
    if (ctx->opt.pieces <= 0) {
      // automatic: eight pieces leave a Kodak half's decoder at most 4 096 latents (1 / 36 of its bitstream, 40 us) behind the bus's last
      // byte; a bitstream of an ELIC-4K stage is up to 24 times as long, and so were its first piece (the head of the call) and its last
      // (the tail) - as many pieces as keep the last one at that size, at most 24 (ELIC-4K, 16 images: 231 -> 210 ms per step with 24,
      // 32 no better; profiles/r04_elic_pieces_ab.txt)
      np = 8;
      while (np < kAutoPiecesMax && (int64_t)np * (np + 1) / 2 * 4096 < lat_max) ++np;
    }
    if (lat < 65536) np = 1; // pieces only pay for rows that take a while to cross
    if (decoders == 1) np = std::min(np, 3); // one decoder: pieces only let it start early, and each costs a hand-over (20 us)
    // every round costs this thread ~60 us of launch / counter / copy round trips: no more rounds than the tables' time on the bus
    // is worth (a lone Kodak half: 7.7 MB = 0.14 ms -> 2 pieces; measured 0.45 ms per call against 0.72 with 8)
    np = (int)std::min<int64_t>(np, std::max<int64_t>(1, lat * 58 / 55700 / 60)); // lat * 58 B / 55.7 GB/s in units of 60 us
  }
  auto piece_bound = [np](int64_t nblk, int p) { // first block of piece p: weights np, np-1 ... 1
    const int64_t tot = (int64_t)np * (np + 1) / 2, cum = (int64_t)p * (2 * np - p + 1) / 2;
    return (int64_t)((__int128)nblk * cum / tot);
  };
  {
    const int steady = ctx->opt.dec_group > 0 ? (int)ctx->opt.dec_group : std::max(n_fast, 1);
    for (int p = 0; p < np && n_fast; ++p) {
      int k = 0, sz = p == 0 && n_fast >= 8 ? (int)std::min<int64_t>(std::max<int64_t>(ctx->opt.dec_first, 1), steady) : steady;
      while (k < n_fast) {
        Unit u;
        const int k1 = std::min(k + sz, n_fast);
        for (; k < k1; ++k) {
          const DecItem &it = items[fast[(size_t)k]];
          u.parts.push_back(Part{fast[(size_t)k], piece_bound(it.nblk, p), piece_bound(it.nblk, p + 1), 0, 0, p});
        }
        units.push_back(std::move(u));
        sz = std::min(steady, sz * 2);
      }
    }
    for (int k = 0; k < n_fast; ++k) {
      DecItem &it = items[fast[k]];
      it.n_piece = np;
      for (int p = 0; p < np; ++p) it.piece_end[p] = std::min<int64_t>(piece_bound(it.nblk, p + 1) * it.tl, it.n);
      // segments: the notes must be exactly the ones an encoder writes for this many symbols (anything else: sequential)
      const bool seekable = it.ckpt && it.n_ckpt > 0 && it.ckpt_stride >= 256 && !(it.ckpt_stride & (it.ckpt_stride - 1)) &&
                            it.n_ckpt == (it.n - 1) / it.ckpt_stride && it.n_ckpt < (1 << 24);
      it.n_seg = seekable ? (int)it.n_ckpt + 1 : 0;
      it.segs_left.store(it.n_seg);
      // TabDecoder::begin copies a bitstream that is not 4-byte aligned (a C caller's; Python's bytes are aligned): once per
      // item here, not once per segment there
      if (it.n_seg && (reinterpret_cast<uintptr_t>(it.enc) & 3) && it.enc_len >= 8 && !(it.enc_len & 3)) {
        try {
          it.enc_aligned.resize(it.enc_len / 4);
        } catch (const std::bad_alloc &) {
          return fail(FGMM_ERR_NOMEM, "out of memory (%zu bytes of bitstream)", it.enc_len);
        }
        memcpy(it.enc_aligned.data(), it.enc, it.enc_len);
        it.enc = reinterpret_cast<const uint8_t *>(it.enc_aligned.data());
      }
    }
  }
  const int n_units = (int)units.size();
  size_t n_parts = 0, stage_total = 0, rows_worst_total = 0;
  for (auto &u : units) {
    size_t off = 0;
    for (auto &p : u.parts) {
      const DecItem &it = items[p.item];
      const int64_t lat = std::min<int64_t>(p.blk_end * it.tl, it.n) - std::min<int64_t>(p.blk_begin * it.tl, it.n);
      p.o_hdr = off;
      off += align_up((size_t)it.hdr_form * (size_t)lat, 256);
      p.o_blkoff = off;
      off += align_up(sizeof(uint32_t) * (size_t)(p.blk_end - p.blk_begin), 256);
      // worst case of a row: every edge of the window kept as a uint16, plus the 2-byte form's escape header
      u.rows_cap += (size_t)lat * (2 * (size_t)(2 * (int64_t)it.max_bs + 2) + 4) + 2 * (size_t)(p.blk_end - p.blk_begin);
    }
    u.fixed = off;
    u.rows_cap = align_up(u.rows_cap, 256);
    rows_worst_total += u.rows_cap;
    n_parts += u.parts.size();
  }
  // staging: the worst case when it fits the budget, else every unit's row area shrinks by the same factor (a unit
  // that then overflows is re-run with the exact size its cursor reports)
  {
    size_t fixed_total = 0;
    for (auto &u : units) fixed_total += u.fixed + 512;
    const size_t budget = fixed_total + rows_worst_total + 256 * (size_t)n_units <= ctx->d_stage_cap && ctx->opt.stage_max_mb <= 0
                              ? ctx->d_stage_cap : ctx->stage_budget(); // the device is asked only when the area has to grow
    if (fixed_total + rows_worst_total + 256 * (size_t)n_units > budget && rows_worst_total) {
      const double f = budget > fixed_total ? (double)(budget - fixed_total) / (double)rows_worst_total : 0.0;
      for (auto &u : units) u.rows_cap = align_up(std::max<size_t>((size_t)((double)u.rows_cap * f), 4096), 256);
    }
    for (auto &u : units) {
      u.o_stage = stage_total;
      stage_total += align_up(u.fixed + u.rows_cap + 256, 256);
    }
  }
  const size_t o_descs = ar.take(sizeof(DecDesc) * std::max<size_t>(n_parts, 1));
  // Symbols back to the GPU ROUND BY ROUND (ScatDesc): the sequentially decoded items of the single-pass path with a latent to write.
  // (a checkpointed item's segments finish in any order, a generic item has one piece: those are scattered whole, as they finish)
  int n_round = 0, M_round = 0;
  int64_t hw_round = 0;
  bool dead_round = false;
  if (count <= 65535 && ctx->opt.scatter_rounds != 0)
    for (int i : fast) {
      DecItem &it = items[i];
      it.rounds = it.y_hat && it.n_seg == 0 && (int64_t)it.M * it.hw > 0 && it.M <= 65535;
      if (!it.rounds) continue;
      n_round = std::max(n_round, it.n_piece);
      M_round = std::max(M_round, it.M);
      hw_round = std::max(hw_round, it.hw);
      dead_round = dead_round || it.n_ch < it.M;
    }
  const size_t o_scat = ar.take(sizeof(ScatDesc) * (size_t)count, 16);
  const size_t o_counters = ar.take(kCounterBytes * (size_t)std::max(n_units, 1), 256);
  const size_t upload_bytes = o_counters;
  // look-back states of every launch, right behind the counters: zeroed with them in one memset
  for (auto &u : units) {
    for (auto &p : u.parts) u.scan_total += p.blk_end - p.blk_begin;
    u.placement = ctx->opt.tab_place == 1 ? 1 : 0;
    u.o_scan = ar.take(sizeof(unsigned long long) * (size_t)std::max<int64_t>(u.scan_total, 1), 8);
  }
  const size_t zero_bytes = ar.off - o_counters;
  // events: per unit [kernel done][counters landed] (this thread waits, briefly) and [tables landed] (the workers wait)
  // Workers SLEEP on the copies' events (sixteen spinning waiters exceed the box's CPU quota: ensure_events) - except in a small
  // call (one image: a few hundred microseconds in all), where being woken by an interrupt costs as much as the work itself:
  // there they wait on plain events, which the runtime polls
  int64_t lat_total = 0;
  for (auto &it : items) lat_total += it.n;
  const bool spin = ctx->opt.spin_lat < 0 ? false : lat_total <= ctx->opt.spin_lat;
  ctx->hsa.quiesce(); // (an earlier call that returned early may have left engine copies in flight: before any buffer moves)
  if ((rc = ctx->ensure_device(ar.off)) || (rc = ctx->ensure_host(upload_bytes + kCounterBytes * (size_t)std::max(n_units, 1))) ||
      (rc = ctx->ensure_events((spin ? 3 : 2) * (size_t)std::max(n_units, 1) + 2, (size_t)n_units + 2)) || (rc = ctx->ensure_stage(stage_total)))
    return rc;
  ctx->chunks_reset();
  tr.mark("planned, buffers ensured");
  double marks[5] = {tr.ms(), 0, 0, 0, 0}; // the call log: planned | first copy queued | last copy queued | last piece seen landed | last decoder done
  // pinned output areas (decoded symbols) of all items
  {
    size_t out_total = 256;
    for (auto &it : items) out_total += align_up(sizeof(int32_t) * (size_t)std::max<int64_t>(it.n, 1), 256);
    char *h_outs = nullptr;
    if ((rc = ctx->chunk_alloc(out_total, &h_outs))) return rc;
    size_t o = 0;
    for (auto &it : items) {
      it.h_out = h_outs + o;
      o += align_up(sizeof(int32_t) * (size_t)std::max<int64_t>(it.n, 1), 256);
    }
  }
  hipEvent_t *ev_kernel = ctx->events.data(), *ev_counters = ev_kernel + n_units,
             *ev_landed = spin ? ev_counters + n_units : ctx->sleep_events.data();

  // ---- channel lists, descriptors ------------------------------------------------------------------------------------
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
    ScatDesc &sd = reinterpret_cast<ScatDesc *>(ctx->h_ws + o_scat)[i];
    memset(&sd, 0, sizeof sd);
    if (it.rounds) {
      sd.sym = reinterpret_cast<const int16_t *>(it.h_out);
      sd.chan_list = reinterpret_cast<const int32_t *>(ctx->d_ws + it.o_list);
      sd.rank = reinterpret_cast<const int32_t *>(ctx->d_ws + it.o_rank);
      sd.y_hat = it.y_hat;
      sd.hw = it.hw;
      sd.M = it.M;
      for (int p = 0; p < kMaxPieces; ++p) sd.bound[p + 1] = p < it.n_piece ? it.piece_end[p] : it.n;
    }
  }
  const ScatDesc *d_scat = reinterpret_cast<const ScatDesc *>(ctx->d_ws + o_scat);
  auto base_desc = [&](const DecItem &it) {
    DecDesc d;
    memset(&d, 0, sizeof d);
    d.scales = it.prm.scales;
    d.means = it.prm.means;
    d.weights = it.prm.weights;
    d.stride_k = it.prm.stride_k;
    d.stride_c = it.prm.stride_c;
    d.stride_p = it.stride_p;
    d.hw = it.hw;
    d.n = it.n;
    d.chan_list = reinterpret_cast<const int32_t *>(ctx->d_ws + it.o_list);
    d.n_ch = it.n_ch;
    d.max_bs = it.max_bs;
    d.clamp = it.clamp;
    d.logits = (it.prm.flags & FGMM_PARAMS_LOGITS) ? 1 : 0;
    d.prune = 1;
    d.hdr_form = it.hdr_form;
    d.ef_min = ef_min;
    d.tl = it.tl;
    d.count_edges = ctx->profiling ? 1 : 0; // measurement aid only (bench.py's roofline_decode)
    return d;
  };
  DecDesc *hd = reinterpret_cast<DecDesc *>(ctx->h_ws + o_descs);
  const DecDesc *dd = reinterpret_cast<const DecDesc *>(ctx->d_ws + o_descs);
  std::vector<size_t> unit_desc0((size_t)n_units + 1, 0);
  auto fill_unit_descs = [&](int u) {
    Unit &un = units[(size_t)u];
    int64_t scan_base = 0;
    for (size_t k = 0; k < un.parts.size(); ++k) {
      const Part &p = un.parts[k];
      DecDesc &d = hd[unit_desc0[(size_t)u] + k];
      d = base_desc(items[p.item]);
      d.blk_begin = (int32_t)p.blk_begin;
      d.blk_end = (int32_t)p.blk_end;
      d.placement = un.placement;
      d.scan = reinterpret_cast<unsigned long long *>(ctx->d_ws + un.o_scan);
      d.scan_base = scan_base;
      d.scan_total = un.scan_total;
      d.spin_limit = (int32_t)ctx->opt.tab_spin;
      scan_base += p.blk_end - p.blk_begin;
      d.hdr_out = un.d_range + p.o_hdr;
      d.blkoff_out = reinterpret_cast<uint32_t *>(un.d_range + p.o_blkoff);
      d.rows = reinterpret_cast<uint8_t *>(un.d_range + un.fixed);
      d.rows_cap = un.rows_cap;
      d.counters = reinterpret_cast<unsigned long long *>(ctx->d_ws + o_counters + kCounterBytes * (size_t)u);
    }
  };
  for (int u = 0; u < n_units; ++u) {
    unit_desc0[(size_t)u + 1] = unit_desc0[(size_t)u] + units[(size_t)u].parts.size();
    units[(size_t)u].d_range = ctx->d_stage + units[(size_t)u].o_stage;
    fill_unit_descs(u);
  }
  tr.mark("descriptors built");
  HIP_TRY(hipMemcpyAsync(ctx->d_ws, ctx->h_ws, upload_bytes, hipMemcpyHostToDevice, stream));
  HIP_TRY(hipMemsetAsync(ctx->d_ws + o_counters, 0, zero_bytes, stream));
  if (n_round && dead_round) LAUNCH_TRY(launch_yhat_zero_dead(d_scat, count, M_round, hw_round, stream)); // (channels without a coded symbol)
  auto launch_unit = [&](int u) -> int {
    const Unit &un = units[(size_t)u];
    int64_t blocks_max = 0;
    int tl_max = 16;
    for (auto &p : un.parts) {
      blocks_max = std::max(blocks_max, p.blk_end - p.blk_begin);
      tl_max = std::max(tl_max, (int)items[p.item].tl);
    }
    LAUNCH_TRY(launch_tab(dd + unit_desc0[(size_t)u], (int)un.parts.size(), (int)blocks_max, tl_max, cap_e, mode, clamped, f16, stream));
    return FGMM_OK;
  };
  unsigned long long *h_counters = reinterpret_cast<unsigned long long *>(ctx->h_ws + o_counters);
  // The launches are enqueued a few units AHEAD of the unit whose size this thread waits for, not all up front: a launch costs this
  // thread ~11 us of API calls (kernel, two events, a stream wait, the counters' copy), eleven of them 0.12 ms - by which time the
  // first unit's kernel had long finished and its copy, the first bytes on the bus, was 0.08 ms late.
  constexpr int kLaunchAhead = 3;
  int launched = 0;
  auto launch_next = [&]() -> int {
    const int u = launched;
    int rc_ = launch_unit(u);
    if (rc_) return rc_;
    HIP_TRY(hipEventRecord(ev_kernel[u], stream));
    HIP_TRY(hipStreamWaitEvent(ctx->aux_stream, ev_kernel[u], 0));
    HIP_TRY(hipMemcpyAsync(h_counters + kTabCounters * (size_t)u, ctx->d_ws + o_counters + kCounterBytes * (size_t)u, kCounterBytes, hipMemcpyDeviceToHost,
                           ctx->aux_stream));
    HIP_TRY(hipEventRecord(ev_counters[u], ctx->aux_stream));
    ++launched;
    if (launched == n_units) return ctx->prof_end(1, stream); // brackets every table kernel of the call
    return FGMM_OK;
  };
  if ((rc = ctx->prof_begin(1, stream))) return rc;
  if (n_units == 0 && (rc = ctx->prof_end(1, stream))) return rc;
  while (launched < std::min(n_units, kLaunchAhead))
    if ((rc = launch_next())) return rc;
  tr.mark("first launches enqueued");

  // ---- the host side: (item, piece) tasks --------------------------------------------------------------------------------
  std::mutex mu; // guards the schedule state of the items, the ready heap, `abandon`, `unfinished`
  std::condition_variable work_cv, done_cv;
  bool abandon = false; // this call is returning early: workers must not wait for copies that will never be queued
  int unfinished = count;
  // a task: piece `piece` of a sequentially decoded item (seg < 0), or segment `seg` of a checkpointed one whose last table
  // piece is `piece`; the earliest-landing task first
  struct Key {
    int64_t piece;
    int item, seg;
    bool operator>(const Key &o) const { return piece != o.piece ? piece > o.piece : (item != o.item ? item > o.item : seg > o.seg); }
  };
  std::priority_queue<Key, std::vector<Key>, std::greater<Key>> ready;
  {
    size_t need = 0;
    for (auto &it : items) need += it.sym_host_out ? 0 : (size_t)std::max<int64_t>(it.n, 1);
    try {
      if (ctx->h_sym.size() < need) ctx->h_sym.resize(need);
    } catch (const std::bad_alloc &) {
      return fail(FGMM_ERR_NOMEM, "out of memory (%zu decoded symbols)", need);
    }
    size_t at = 0;
    for (auto &it : items) {
      it.sym = it.sym_host_out ? it.sym_host_out : ctx->h_sym.data() + at;
      if (!it.sym_host_out) at += (size_t)std::max<int64_t>(it.n, 1);
    }
  }
  auto seg_last_piece = [&](const DecItem &it, int sg) { // the piece that holds the last latent of segment sg
    const int64_t hi = sg + 1 == it.n_seg ? it.n : (int64_t)(sg + 1) * it.ckpt_stride;
    int p = 0;
    while (p + 1 < it.n_piece && it.piece_end[p] < hi) ++p;
    return p;
  };
  auto push_if_ready = [&](int i) { // under mu
    DecItem &it = items[i];
    if (it.n_seg) { // every segment whose tables are queued by now; segments are independent of one another
      int pushed = 0;
      while (it.next_seg_push < it.n_seg) {
        const int lp = seg_last_piece(it, it.next_seg_push);
        if (lp >= it.queued) break;
        ready.push(Key{lp, i, it.next_seg_push++});
        ++pushed;
      }
      return pushed;
    }
    if (!it.busy && !it.in_ready && !it.done.load() && it.next_piece < it.queued) {
      it.in_ready = true;
      ready.push(Key{it.next_piece, i, -1});
      return 1;
    }
    return 0;
  };
  // waits until piece p of the item is in host memory: the event of its copy, or the signal of the SDMA engine that copies it
  auto landed = [spin](const DecItem &it, int p) {
    if (it.piece_sig[p].handle) return HsaCopier::wait(it.piece_sig[p], spin);
    return hipEventSynchronize(it.piece_ev[p]) == hipSuccess;
  };
  // before / after a piece is decoded (no lock held)
  auto prepare = [&](DecItem &it, int p) {
    if (p == 0) {
      it.t_taken = tr.ms();
      it.view = TabView{it.ef_min, it.hdr_form, it.tl, it.n_piece, it.piece, nullptr, nullptr};
      if (it.status == FGMM_OK) it.status = it.dec.begin(it.enc, it.enc_len, &it.view, it.n, it.max_bs, it.sym);
    }
    const double tw0 = tr.ms();
    if (it.status == FGMM_OK && !landed(it, p)) it.status = FGMM_ERR_HIP;
    const double tw1 = tr.ms();
    it.t_waited += tw1 - tw0;
    it.t_lastland = tw1;
    if (p == 0) it.t_start = tw1;
  };
  auto complete = [&](DecItem &it, int p) { // true: the item is finished
    if (it.status == FGMM_OK && it.y_hat) {
      // this piece's symbols -> pinned memory for the scatter kernel: int16 unless some (bypass-coded) symbol does not fit
      int16_t *s16 = reinterpret_cast<int16_t *>(it.h_out);
      const int64_t k1 = std::min<int64_t>(it.dec.i, it.n);
      int32_t acc = 0;
      for (int64_t k = it.narrowed; k < k1; ++k) {
        const int32_t v = it.sym[k];
        s16[k] = (int16_t)v;
        acc |= v ^ (int32_t)(int16_t)v;
      }
      it.narrowed = k1;
      it.wide |= acc != 0;
    }
    const bool last = it.status != FGMM_OK || p + 1 == it.n_piece;
    if (!last) return false;
    const int rf = it.dec.finish();
    if (it.status == FGMM_OK) it.status = rf;
    // (an item scattered round by round: its int16 symbols may still be being read - this thread's final loop redoes a wide one)
    if (it.status == FGMM_OK && it.y_hat && it.wide && !it.rounds) memcpy(it.h_out, it.sym, sizeof(int32_t) * (size_t)it.n);
    it.t_end = tr.ms();
    return true;
  };
  // One segment of a checkpointed bitstream, start to end on this thread (no lock held): from its checkpoint - the stream's
  // own head for segment 0 - to the next one, which it must hit exactly.  The last segment to finish closes the item; if any
  // segment missed its checkpoint the whole bitstream is decoded sequentially then (the notes were wrong: nothing of what the
  // segments wrote is kept).  -> true: the item is finished
  auto run_segment = [&](DecItem &it, int sg) {
    const int64_t lo = (int64_t)sg * it.ckpt_stride, hi = sg + 1 == it.n_seg ? it.n : (int64_t)(sg + 1) * it.ckpt_stride;
    int p0 = 0;
    while (p0 + 1 < it.n_piece && it.piece_end[p0] <= lo) ++p0;
    const int p1 = seg_last_piece(it, sg);
    const double tw0 = tr.ms();
    bool ok = !it.ckpt_bad.load(std::memory_order_relaxed);
    for (int p = p0; p <= p1 && ok; ++p)
      if (!landed(it, p)) ok = false;
    const double tw1 = tr.ms();
    if (ok) {
      TabDecoder td;
      int rc2 = td.begin(it.enc, it.enc_len, &it.view, it.n, it.max_bs, it.sym);
      uint64_t x1 = 0, pos1 = 0;
      if (rc2 == FGMM_OK) rc2 = td.segment(lo, hi, sg ? it.ckpt[sg - 1].x : td.x, sg ? it.ckpt[sg - 1].pos : 0, &x1, &pos1);
      td.rc = FGMM_OK;
      td.i = it.n;
      (void)td.finish();
      ok = rc2 == FGMM_OK && (sg + 1 == it.n_seg || (x1 == it.ckpt[sg].x && pos1 == it.ckpt[sg].pos));
    }
    if (!ok) {
      it.ckpt_bad.store(1);
    } else if (it.y_hat) { // this segment's symbols -> pinned memory for the scatter kernel, int16 unless one does not fit
      int16_t *s16 = reinterpret_cast<int16_t *>(it.h_out);
      int32_t acc = 0;
      for (int64_t k = lo; k < hi; ++k) {
        const int32_t v = it.sym[k];
        s16[k] = (int16_t)v;
        acc |= v ^ (int32_t)(int16_t)v;
      }
      if (acc) it.wide_any.store(1);
    }
    {
      std::lock_guard<std::mutex> l(mu);
      it.t_waited += tw1 - tw0;
      it.t_work += tr.ms() - tw1;
      it.t_lastland = std::max(it.t_lastland, tw1);
      if (sg == 0) it.t_taken = it.t_start = tw0;
    }
    if (it.segs_left.fetch_sub(1) != 1) return false;
    // the last segment: close the item
    if (it.ckpt_bad.load()) { // sequential decode of the whole bitstream (every piece is queued: the last segment needed the last one)
      it.status = it.dec.begin(it.enc, it.enc_len, &it.view, it.n, it.max_bs, it.sym);
      for (int p = 0; p < it.n_piece && it.status == FGMM_OK; ++p) {
        if (!landed(it, p)) it.status = FGMM_ERR_HIP;
        else it.status = it.dec.piece(p);
      }
      const int rf = it.dec.finish();
      if (it.status == FGMM_OK) it.status = rf;
      if (it.status == FGMM_OK && it.y_hat) {
        int16_t *s16 = reinterpret_cast<int16_t *>(it.h_out);
        int32_t acc = 0;
        for (int64_t k = 0; k < it.n; ++k) {
          const int32_t v = it.sym[k];
          s16[k] = (int16_t)v;
          acc |= v ^ (int32_t)(int16_t)v;
        }
        it.wide_any.store(acc != 0);
      }
    }
    it.wide = it.wide_any.load();
    if (it.status == FGMM_OK && it.y_hat && it.wide) memcpy(it.h_out, it.sym, sizeof(int32_t) * (size_t)it.n);
    it.t_end = tr.ms();
    return true;
  };
  int waiting = 0; // workers asleep on work_cv (under mu)
  // two bitstreams in turn per worker: 8.9 -> 5.9 ns/symbol per thread with uint16 rows, nothing with Elias-Fano rows, and a
  // loss whenever it leaves workers idle - automatic only with at least two bitstreams per worker
  const bool pairing = ctx->opt.dec_pair == 1 || (ctx->opt.dec_pair == 0 && count >= 2 * std::max(ctx->pool->size(), 1) && ef_min == kTabNoEf);
  auto worker = [&] {
    std::unique_lock<std::mutex> l(mu);
    auto take = [&](int *p_out) { // under mu: the earliest-landing ready task of a sequentially decoded item
      const int i = ready.top().item;
      ready.pop();
      items[i].in_ready = false;
      items[i].busy = true;
      *p_out = items[i].next_piece;
      return i;
    };
    auto give_back = [&](int i, int p, bool finished) { // under mu
      DecItem &it = items[i];
      it.busy = false;
      it.next_piece = p + 1;
      if (finished) {
        it.done.store(1);
        if (--unfinished == 0) work_cv.notify_all();
        done_cv.notify_all();
      } else {
        push_if_ready(i);
        if (it.rounds) done_cv.notify_all(); // (a piece's symbols are in pinned memory: its round may be complete)
      }
    };
    for (;;) {
      if (unfinished == 0) return;
      if (ready.empty()) {
        if (abandon) return;
        ++waiting;
        work_cv.wait(l);
        --waiting;
        continue;
      }
      if (ready.top().seg >= 0) { // a segment of a checkpointed bitstream: independent of every other task
        const Key k = ready.top();
        ready.pop();
        l.unlock();
        const bool fin = run_segment(items[k.item], k.seg);
        l.lock();
        if (fin) {
          items[k.item].done.store(1);
          if (--unfinished == 0) work_cv.notify_all();
          done_cv.notify_all();
        }
        continue;
      }
      int p0 = 0, p1 = 0;
      const int i0 = take(&p0);
      // a second bitstream for this thread (decoded latent by latent in turn with the first: two dependency chains share
      // the core) - unless that would leave a sleeping worker without a task
      const int i1 = pairing && !ready.empty() && ready.top().seg < 0 && (int)ready.size() > waiting ? take(&p1) : -1;
      l.unlock();
      DecItem &a = items[i0];
      prepare(a, p0);
      const double t0 = tr.ms();
      if (i1 < 0) {
        if (a.status == FGMM_OK) a.status = a.dec.piece(p0);
        a.t_work += tr.ms() - t0;
        const bool fa = complete(a, p0);
        l.lock();
        give_back(i0, p0, fa);
      } else {
        DecItem &b = items[i1];
        prepare(b, p1);
        const double t1 = tr.ms();
        if (a.status == FGMM_OK && b.status == FGMM_OK) {
          rans_decode_pieces2(a.dec, p0, b.dec, p1, &a.status, &b.status);
        } else {
          if (a.status == FGMM_OK) a.status = a.dec.piece(p0);
          if (b.status == FGMM_OK) b.status = b.dec.piece(p1);
        }
        {
          const double dt = tr.ms() - t1;
          a.t_work += dt / 2;
          b.t_work += dt / 2;
        }
        const bool fa = complete(a, p0), fb = complete(b, p1);
        l.lock();
        give_back(i0, p0, fa);
        give_back(i1, p1, fb);
      }
    }
  };
  auto mark_queued = [&](int i, int pieces) {
    int pushed;
    {
      std::lock_guard<std::mutex> l(mu);
      items[i].queued = pieces;
      pushed = push_if_ready(i);
    }
    if (pushed > 1) work_cv.notify_all(); else work_cv.notify_one();
  };
  struct Abandon { // any return: release workers that wait for copies (before PoolDrain waits for the workers)
    std::mutex &mu;
    std::condition_variable &cv;
    bool &flag;
    ~Abandon() {
      {
        std::lock_guard<std::mutex> l(mu);
        flag = true;
      }
      cv.notify_all();
    }
  };
  PoolDrain drain{ctx->pool}; // on any return: wait for every job before the objects they use go away
  Abandon abandon_on_exit{mu, work_cv, abandon};
  TempDevice temp;

  for (auto &it : items)
    if (it.n_seg) it.view = TabView{it.ef_min, it.hdr_form, it.tl, it.n_piece, it.piece, nullptr, nullptr};
  // (a single bitstream too: its decoder starts on piece 0 while this thread is still queuing the later pieces' copies)
  const int n_workers = (int)std::min<int64_t>(std::max(ctx->pool->size(), 1), streams_of_work);
  for (int j = 0; j < n_workers; ++j) ctx->pool->submit(worker);

  // ---- unit by unit: size known -> pinned range, ONE copy; the workers are told ----------------------------------------
  unsigned long long edges = 0;
  std::vector<std::array<double, 3>> unit_trace; // trace level 2: [queued at, bytes, landed at] per unit
  for (int u = 0; u < n_units; ++u) {
    Unit &un = units[(size_t)u];
    while (launched < std::min(n_units, u + 1 + kLaunchAhead))
      if ((rc = launch_next())) return rc;
    HIP_TRY(hipEventSynchronize(ev_counters[u]));
    unsigned long long *cn = h_counters + kTabCounters * (size_t)u;
    // Not placed as launched: a look-back gave up (bit 1; never seen outside the test that forces it) - once more with the
    // cursor -, or the provisioned row area was too small (bit 0) - once more into an area of exactly the size asked for
    for (int attempt = 0; cn[1] && attempt < 3; ++attempt) {
      if (cn[1] & 2) {
        un.placement = 0;
        ++ctx->stat[6];
      } else {
        const size_t need = align_up((size_t)cn[0], 256);
        char *d_new = nullptr;
        if ((rc = temp.alloc(un.fixed + need + 256, &d_new))) return rc;
        un.d_range = d_new;
        un.rows_cap = need;
      }
      fill_unit_descs(u);
      HIP_TRY(hipMemcpyAsync(ctx->d_ws + o_descs + sizeof(DecDesc) * unit_desc0[(size_t)u], hd + unit_desc0[(size_t)u],
                             sizeof(DecDesc) * un.parts.size(), hipMemcpyHostToDevice, stream));
      HIP_TRY(hipMemsetAsync(ctx->d_ws + o_counters + kCounterBytes * (size_t)u, 0, kCounterBytes, stream));
      HIP_TRY(hipMemsetAsync(ctx->d_ws + un.o_scan, 0, sizeof(unsigned long long) * (size_t)std::max<int64_t>(un.scan_total, 1), stream));
      if ((rc = launch_unit(u))) return rc;
      HIP_TRY(hipEventRecord(ev_kernel[u], stream));
      HIP_TRY(hipMemcpyAsync(cn, ctx->d_ws + o_counters + kCounterBytes * (size_t)u, kCounterBytes, hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
    }
    if (cn[1]) return fail(FGMM_ERR_HIP, "decode tables could not be placed (unit %d: flags %llu, %llu of %zu bytes)", u, cn[1], cn[0], un.rows_cap);
    const size_t used = (size_t)cn[0];
    for (int q = 0; q < kTabEdgeSlots; ++q) edges += cn[4 + q];
    char *h_range = nullptr;
    if ((rc = ctx->chunk_alloc(un.fixed + used + 256, &h_range))) return rc;
    memset(h_range + un.fixed + used, 0, 256); // slack: the host's SIMD search reads a little past a row
    // the unit's kernel is complete (its counters are here): the copy depends on nothing.  Straight to an SDMA engine
    // (option copy_engine), or the runtime's way
    hsa_signal_t sig{0};
    if (ctx->opt.copy_engine >= 1 && un.fixed + used && ctx->hsa.init(un.d_range, h_range))
      sig = ctx->hsa.copy(h_range, un.d_range, un.fixed + used, ctx->opt.copy_engine == 2);
    if (!sig.handle) {
      HIP_TRY(hipStreamWaitEvent(ctx->copy_stream, ev_kernel[u], 0));
      if (un.fixed + used) HIP_TRY(hipMemcpyAsync(h_range, un.d_range, un.fixed + used, hipMemcpyDeviceToHost, ctx->copy_stream));
      HIP_TRY(hipEventRecord(ev_landed[u], ctx->copy_stream));
    }
    for (auto &p : un.parts) {
      DecItem &it = items[p.item];
      TabPiece &pc = it.piece[p.piece];
      pc.hdr = h_range + p.o_hdr;
      pc.blk_off = reinterpret_cast<const uint32_t *>(h_range + p.o_blkoff);
      pc.rows = reinterpret_cast<const uint8_t *>(h_range + un.fixed);
      pc.rows_len = used + 256;
      pc.end = std::min<int64_t>(p.blk_end * it.tl, it.n);
      it.piece_ev[p.piece] = ev_landed[u];
      it.piece_sig[p.piece] = sig;
      const int64_t lat = pc.end - std::min<int64_t>(p.blk_begin * it.tl, it.n);
      it.table_bytes += (uint64_t)it.hdr_form * (uint64_t)lat + sizeof(uint32_t) * (uint64_t)(p.blk_end - p.blk_begin);
      mark_queued(p.item, p.piece + 1); // pieces reach an item in order: rounds are piece-major
    }
    // rows are shared by the unit's items: account them once
    if (!un.parts.empty()) items[un.parts[0].item].table_bytes += used;
    marks[2] = tr.ms();
    if (u == 0) marks[1] = marks[2];
    if (tr.level > 1) unit_trace.push_back({tr.ms(), (double)(un.fixed + used), 0.0});
  }
  tr.mark("sizes known, copies queued");
  if (tr.level > 1 && !ctx->opt.copy_engine) { // the copies' own timeline: this thread watches every unit land (delays the scatter rounds a little)
    for (int u = 0; u < n_units; ++u) {
      (void)hipEventSynchronize(ev_landed[u]);
      unit_trace[(size_t)u][2] = tr.ms();
    }
    for (int u = 0; u < n_units; ++u)
      fprintf(stderr, "[fgmm decode]   unit %2d  %2zu parts  %9.0f bytes  queued %7.3f  landed %7.3f  (%.1f GB/s since the unit before landed or this one was queued)\n", u,
              units[(size_t)u].parts.size(), unit_trace[(size_t)u][1], unit_trace[(size_t)u][0], unit_trace[(size_t)u][2],
              unit_trace[(size_t)u][1] / 1e6 / std::max(1e-6, unit_trace[(size_t)u][2] - std::max(unit_trace[(size_t)u][0], u ? unit_trace[(size_t)u - 1][2] : 0.0)));
  }

  // ---- generic path: items too wide for the single-pass kernel, one at a time ------------------------------------------
  for (int i : generic) {
    DecItem &it = items[i];
    it.n_piece = 1;
    if (it.n == 0) {
      it.piece[0] = TabPiece{ctx->h_ws, nullptr, reinterpret_cast<const uint8_t *>(ctx->h_ws), 0, 0};
      it.piece_ev[0] = ev_landed[n_units];
      HIP_TRY(hipEventRecord(ev_landed[n_units], stream));
      mark_queued(i, 1);
      continue;
    }
    const int32_t tiles = (int32_t)((it.hw + 255) / 256);
    const size_t nblk = (size_t)it.n_ch * (size_t)tiles;
    const size_t hdr_bytes = align_up((size_t)(it.hdr_form == 8 ? 8 : 4) * (size_t)it.n, 256);
    Arena ga;
    const size_t g_desc = ga.take(sizeof(DecDesc)), g_used = ga.take(64), g_hdr = ga.take(hdr_bytes), g_bsum = ga.take(4 * nblk + 64),
                 g_boff = ga.take(8 * nblk + 64);
    char *d_g = nullptr;
    if ((rc = temp.alloc(ga.off, &d_g))) return rc;
    DecDesc d = base_desc(it);
    d.hdr_form = it.hdr_form == 8 ? 8 : 4; // the generic kernels write 4- or 8-byte headers
    it.hdr_form = d.hdr_form;
    d.hdr = d_g + g_hdr;
    d.tiles = tiles;
    d.pool = nullptr;
    d.pool_cap = ~0ull;
    d.pool_used = reinterpret_cast<unsigned long long *>(d_g + g_used);
    d.blk_sums = reinterpret_cast<uint32_t *>(d_g + g_bsum);
    d.blk_off = reinterpret_cast<unsigned long long *>(d_g + g_boff);
    HIP_TRY(hipMemsetAsync(d_g + g_used, 0, 64, stream));
    HIP_TRY(hipMemcpyAsync(d_g + g_desc, &d, sizeof d, hipMemcpyHostToDevice, stream));
    LAUNCH_TRY(launch_cdftab_count(reinterpret_cast<const DecDesc *>(d_g + g_desc), 1, it.n_ch, it.hw, mode, clamped, f16, stream));
    unsigned long long used4[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(used4, d_g + g_used, sizeof used4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (used4[3]) {
      it.status = FGMM_ERR_UNSUPPORTED;
      it.piece_ev[0] = ev_landed[n_units];
      HIP_TRY(hipEventRecord(ev_landed[n_units], stream));
      mark_queued(i, 1);
      continue;
    }
    const size_t pool_bytes = (size_t)used4[0];
    char *d_pool = nullptr, *h_range = nullptr;
    if ((rc = temp.alloc(pool_bytes + 256, &d_pool)) || (rc = ctx->chunk_alloc(hdr_bytes + pool_bytes + 256, &h_range))) return rc;
    d.pool = reinterpret_cast<uint8_t *>(d_pool);
    HIP_TRY(hipMemcpyAsync(d_g + g_desc, &d, sizeof d, hipMemcpyHostToDevice, stream));
    LAUNCH_TRY(launch_cdftab_fill(reinterpret_cast<const DecDesc *>(d_g + g_desc), 1, it.n_ch, it.hw, mode, clamped, f16, stream));
    HIP_TRY(hipMemcpyAsync(h_range, d_g + g_hdr, hdr_bytes, hipMemcpyDeviceToHost, stream));
    if (pool_bytes) HIP_TRY(hipMemcpyAsync(h_range + hdr_bytes, d_pool, pool_bytes, hipMemcpyDeviceToHost, stream));
    memset(h_range + hdr_bytes + pool_bytes, 0, 256);
    HIP_TRY(hipStreamSynchronize(stream));
    it.piece[0] = TabPiece{h_range, nullptr, reinterpret_cast<const uint8_t *>(h_range + hdr_bytes), pool_bytes + 256, it.n};
    it.piece_ev[0] = ev_landed[n_units]; // nothing left to wait for: the stream has just been synchronised
    HIP_TRY(hipEventRecord(ev_landed[n_units], stream));
    it.table_bytes = hdr_bytes + pool_bytes;
    mark_queued(i, 1);
  }

  ctx->stat[1] = ctx->stat[2] = 0;
  ctx->stat[3] = edges;
  for (auto &it : items) {
    ctx->stat[1] += it.table_bytes;
    ctx->stat[2] += (unsigned long long)it.n;
  }

  // ---- symbols back to the GPU: scatter kernels on the caller's stream -------------------------------------------------
  // Round r (piece r of every item that is decoded piece by piece) goes as soon as every such item has decoded it: one launch
  // for all of them, while the later pieces are still on the bus - what is left after the last decoder is the last, smallest
  // piece (one launch + 0.1 MB over PCIe instead of a launch per item + the symbols of the items that finish together:
  // 0.06 -> 0.02-0.03 ms between the last decoder and the call's end, profiles/r04_scatter_rounds_ab.txt).
  for (int r = 0; r < n_round; ++r) {
    int64_t max_range = 0;
    {
      std::unique_lock<std::mutex> l(mu);
      done_cv.wait(l, [&] {
        for (int i : fast)
          if (items[i].rounds && items[i].next_piece <= r && !items[i].done.load()) return false;
        return true;
      });
    }
    for (int i : fast)
      if (items[i].rounds && r < items[i].n_piece) max_range = std::max(max_range, items[i].piece_end[r] - (r ? items[i].piece_end[r - 1] : 0));
    LAUNCH_TRY(launch_yhat_scatter_round(d_scat, count, r, max_range, stream));
  }
  int first_err = FGMM_OK;
  for (int i = 0; i < count; ++i) {
    DecItem &it = items[i];
    {
      std::unique_lock<std::mutex> l(mu);
      done_cv.wait(l, [&it] { return it.done.load() != 0; });
    }
    if (it.status && !first_err) first_err = it.status;
    if (it.status != FGMM_OK || !it.y_hat || !(it.M * it.hw) || (it.rounds && !it.wide)) continue;
    if (it.rounds) { // a symbol that does not fit int16 (bypass-coded, rare): once more, whole and wide - after the rounds have read
      HIP_TRY(hipStreamSynchronize(stream));
      memcpy(it.h_out, it.sym, sizeof(int32_t) * (size_t)it.n);
    }
    LAUNCH_TRY(launch_yhat_scatter(it.h_out, it.wide, reinterpret_cast<const int32_t *>(ctx->d_ws + it.o_rank), it.y_hat, it.M, it.hw, stream));
  }
  tr.mark("host rANS done");
  HIP_TRY(hipStreamSynchronize(stream));
  tr.mark("y_hat written");
  {
    double busy = 0, wait = 0;
    for (auto &it : items) {
      marks[3] = std::max(marks[3], it.t_lastland);
      marks[4] = std::max(marks[4], it.t_end);
      busy += it.t_work;
      wait += it.t_waited;
    }
    ctx->log_call(1, count, tr, marks, busy, wait);
  }
  if (tr.level > 1)
    for (int i = 0; i < count; ++i)
      fprintf(stderr, "[fgmm decode]   item %2d  pieces %d  taken %7.3f  job %7.3f .. %7.3f  (decoding %.3f ms, waiting for copies %.3f, last piece at %.3f)\n",
              i, items[i].n_piece, items[i].t_taken, items[i].t_start, items[i].t_end, items[i].t_work, items[i].t_waited, items[i].t_lastland);
  if (first_err)
    return fail(first_err, "host rANS decode failed (%d)%s", first_err,
                first_err == FGMM_ERR_STREAM ? ": bitstream too short"
                : first_err == FGMM_ERR_UNSUPPORTED ? ": an evaluation window beyond 2^20 edges (see FGMM_MAX_BS)" : "");
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

namespace {
struct OptName {
  const char *name;
  int64_t fgmm_ctx::Opts::*field;
  int64_t lo, hi;
  const char *env; // read once at context creation (compatibility with round-1 scripts)
};
const OptName kOpts[] = {
    {"pieces", &fgmm_ctx::Opts::pieces, 0, kMaxPieces, "FGMM_PIECES"},
    {"dec_group", &fgmm_ctx::Opts::dec_group, 0, 1 << 20, "FGMM_DEC_GROUP"},
    {"dec_first", &fgmm_ctx::Opts::dec_first, 1, 1 << 20, "FGMM_DEC_FIRST"},
    {"tab_cap_e", &fgmm_ctx::Opts::tab_cap_e, 256, 32768, "FGMM_TAB_CAP_E"},
    {"stage_max_mb", &fgmm_ctx::Opts::stage_max_mb, 0, 1 << 30, "FGMM_STAGE_MAX_MB"},
    {"trace", &fgmm_ctx::Opts::trace, 0, 2, "FGMM_TRACE"},
    {"enc_vec", &fgmm_ctx::Opts::enc_vec, 0, 4, "FGMM_VEC"},
    {"enc_linear", &fgmm_ctx::Opts::enc_linear, 0, 1, nullptr},
    {"ef_rows", &fgmm_ctx::Opts::ef_rows, 0, 2, "FGMM_EF_ROWS"},
    {"ef_min", &fgmm_ctx::Opts::ef_min, kTabEfMin, 1 << 20, "FGMM_EF_MIN_ROWS"},
    {"dec_pair", &fgmm_ctx::Opts::dec_pair, 0, 2, "FGMM_DEC_PAIR"},
    {"enc_ways", &fgmm_ctx::Opts::enc_ways, 0, kMaxEncWays, "FGMM_ENC_WAYS"},
    {"ckpt_decode", &fgmm_ctx::Opts::ckpt_decode, 0, 2, "FGMM_CKPT_DECODE"},
    {"spin_lat", &fgmm_ctx::Opts::spin_lat, -1, 1ll << 40, "FGMM_SPIN_LAT"},
    {"gpu_decode", &fgmm_ctx::Opts::gpu_decode, 0, 2, "FGMM_GPU_DECODE"},
    // tab_kernel's placement of a block's rows: 0 = one atomic add per block on a cursor (arrival order), 1 = decoupled look-back
    // (launch order: deterministic tables, no same-address atomics - and a third slower: every block waits for all of its
    // predecessors to arrive, profiles/r04_tab_place_sweep.txt; a launch in which a look-back gives up is re-run with the cursor)
    {"tab_place", &fgmm_ctx::Opts::tab_place, 0, 1, "FGMM_TAB_PLACE"},
    {"tab_spin", &fgmm_ctx::Opts::tab_spin, 0, 1 << 30, nullptr}, // look-back polls before giving up (tests set 0: every wait gives up)
    // decode tables device -> pinned host: 0 = hipMemcpyAsync (shader copies on this runtime), 1 = straight to ONE SDMA engine,
    // 2 = two engines in turn (measured slower in situ than the shader copies: profiles/r04_copy_engine.md)
    {"copy_engine", &fgmm_ctx::Opts::copy_engine, 0, 2, "FGMM_COPY_ENGINE"},
    // encode: 1 = the tables of a call with a worker per bitstream cross PCIe TAIL FIRST in four segments per bitstream and the
    // encoders (which walk a table backwards) follow the landing; 0 = whole tables, bitstream after bitstream
    {"enc_segs", &fgmm_ctx::Opts::enc_segs, 0, 1, "FGMM_ENC_SEGS"},
    {"scatter_rounds", &fgmm_ctx::Opts::scatter_rounds, 0, 1, nullptr},
};
} // namespace

namespace {
// CPUs the cgroup lets this process use per period (v2: cpu.max of the process's group and its ancestors as far as they are
// visible; v1: cpu.cfs_quota_us / cpu.cfs_period_us), or a negative number when there is no quota
double cgroup_cpu_quota() {
  double best = -1.0;
  auto take = [&best](double q) {
    if (q > 0 && (best < 0 || q < best)) best = q;
  };
  auto read_line = [](const std::string &path, char *buf, size_t cap) {
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return false;
    const bool ok = fgets(buf, (int)cap, f) != nullptr;
    fclose(f);
    return ok;
  };
  char buf[256];
  std::string v2_rel, v1_rel;
  // FGMM_SYSROOT: a directory that stands in for "/" (tests/test_host_cpu.py builds cgroup trees there)
  const std::string sysroot = getenv("FGMM_SYSROOT") ? getenv("FGMM_SYSROOT") : "";
  if (FILE *f = fopen((sysroot + "/proc/self/cgroup").c_str(), "r")) {
    while (fgets(buf, sizeof buf, f)) {
      std::string ln(buf);
      while (!ln.empty() && (ln.back() == '\n' || ln.back() == '\r')) ln.pop_back();
      const size_t a = ln.find(':'), b = a == std::string::npos ? a : ln.find(':', a + 1);
      if (b == std::string::npos) continue;
      const std::string ctrl = ln.substr(a + 1, b - a - 1), rel = ln.substr(b + 1);
      if (ctrl.empty()) v2_rel = rel;
      else if (("," + ctrl + ",").find(",cpu,") != std::string::npos) v1_rel = rel;
    }
    fclose(f);
  }
  // v2: the group itself, then every ancestor up to the mount point
  for (const char *root : {"/sys/fs/cgroup", "/sys/fs/cgroup/unified"}) {
    std::string rel = v2_rel;
    for (;;) {
      if (read_line(sysroot + root + rel + (rel.empty() || rel.back() != '/' ? "/" : "") + "cpu.max", buf, sizeof buf)) {
        long long q = 0, p = 0;
        if (sscanf(buf, "%lld %lld", &q, &p) == 2 && q > 0 && p > 0) take((double)q / (double)p);
      }
      if (rel.empty() || rel == "/") break;
      const size_t cut = rel.find_last_of('/');
      rel = cut == std::string::npos ? std::string() : rel.substr(0, cut);
    }
  }
  // v1
  for (const char *root : {"/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"}) {
    std::string rel = v1_rel;
    for (;;) {
      const std::string dir = sysroot + root + rel + (rel.empty() || rel.back() != '/' ? "/" : "");
      long long q = 0, p = 0;
      if (read_line(dir + "cpu.cfs_quota_us", buf, sizeof buf) && sscanf(buf, "%lld", &q) == 1 && q > 0 &&
          read_line(dir + "cpu.cfs_period_us", buf, sizeof buf) && sscanf(buf, "%lld", &p) == 1 && p > 0)
        take((double)q / (double)p);
      if (rel.empty() || rel == "/") break;
      const size_t cut = rel.find_last_of('/');
      rel = cut == std::string::npos ? std::string() : rel.substr(0, cut);
    }
  }
  return best;
}
} // namespace

int fgmm_host_cpu_budget(double *cpus_out, int *affinity_out, double *quota_out) {
  cpu_set_t set;
  CPU_ZERO(&set);
  int aff = 0;
  if (sched_getaffinity(0, sizeof set, &set) == 0) aff = CPU_COUNT(&set);
  if (aff <= 0) aff = (int)std::thread::hardware_concurrency();
  if (aff <= 0) aff = 4;
  const double quota = cgroup_cpu_quota();
  if (affinity_out) *affinity_out = aff;
  if (quota_out) *quota_out = quota;
  if (cpus_out) *cpus_out = quota > 0 ? std::min((double)aff, quota) : (double)aff;
  return FGMM_OK;
}

int fgmm_host_thread_budget(int ranks_sharing) {
  double cpus = 4, quota = -1;
  int aff = 4;
  (void)fgmm_host_cpu_budget(&cpus, &aff, &quota);
  double aff_share = aff;
  if (ranks_sharing > 1) cpus /= ranks_sharing, aff_share /= ranks_sharing;
  // The workers SLEEP on the copies' events and are busy three quarters of a decode call (less, the more of them there are), and a
  // cgroup quota limits CPU TIME per period, not how many threads may run at once: where the affinity mask is wider than the quota
  // (a GPU box: 128 CPUs of the GPU's node, a quota of 16) a pool of up to three workers per CPU of the share runs the bursts of a
  // step - 48 bitstreams to encode, 24 to decode - on as many cores, stays inside the quota (a Kodak step uses 9-10 CPUs' worth of
  // time; nr_throttled does not move, the bench line carries the counters) and is 3 % faster than 16 workers (48: step median
  // 9.41-9.60 ms, 16: 9.73-9.80 on quiet boxes; checkpointed streams +6 %: profiles/r04_host_threads.txt).  Never more workers than
  // the rank's share of the affinity mask; 16 / 14 / 12 workers: 962 / 938 / 891 Mpixels/s (profiles/r03_host_threads.md).
  // FGMM_WORKERS_PER_CPU (1..4, default 3): the multiplier, for hosts where other processes of the same cgroup need part of the quota
  int per_cpu = 3;
  if (const char *e = getenv("FGMM_WORKERS_PER_CPU")) per_cpu = std::min(std::max(atoi(e), 1), 4);
  const int by_time = (int)floor(cpus + 1e-9), by_mask = (int)floor(aff_share + 1e-9);
  const int t = quota > 0 && by_mask > by_time ? std::min(by_mask, per_cpu * by_time) : by_time;
  return std::max(1, std::min(t, 48));
}
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
  if (n_threads <= 0) n_threads = fgmm_host_thread_budget(1);
  fgmm_ctx *c = new (std::nothrow) fgmm_ctx;
  if (!c) return fail(FGMM_ERR_NOMEM, "ctx");
  c->device = device;
  c->pool = new Pool(n_threads);
  for (const OptName &o : kOpts)
    if (o.env && getenv(o.env)) c->opt.*(o.field) = std::min(std::max<int64_t>(atoll(getenv(o.env)), o.lo), o.hi);
  *out = c;
  return FGMM_OK;
}

void fgmm_ctx_destroy(fgmm_ctx *ctx) {
  if (!ctx) return;
  {
    DeviceGuard g(ctx->device);
    delete ctx->pool;
    for (auto e : ctx->events) (void)hipEventDestroy(e);
    for (auto e : ctx->sleep_events) (void)hipEventDestroy(e);
    for (auto &pr : ctx->prof)
      for (auto e : pr)
        if (e) (void)hipEventDestroy(e);
    ctx->trim();
    ctx->hsa.destroy();
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
  }
  delete ctx;
}

int fgmm_ctx_set_option(fgmm_ctx *ctx, const char *name, int64_t value) {
  if (!ctx || !name) return fail(FGMM_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lock(ctx->mu);
  for (const OptName &o : kOpts)
    if (!strcmp(o.name, name)) {
      if (value < o.lo || value > o.hi) return fail(FGMM_ERR_INVALID, "option %s: %lld outside [%lld, %lld]", name, (long long)value, (long long)o.lo, (long long)o.hi);
      ctx->opt.*(o.field) = value;
      return FGMM_OK;
    }
  return fail(FGMM_ERR_INVALID, "unknown option '%s'", name);
}

int fgmm_ctx_get_option(fgmm_ctx *ctx, const char *name, int64_t *value_out) {
  if (!ctx || !name || !value_out) return fail(FGMM_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lock(ctx->mu);
  for (const OptName &o : kOpts)
    if (!strcmp(o.name, name)) {
      *value_out = ctx->opt.*(o.field);
      return FGMM_OK;
    }
  return fail(FGMM_ERR_INVALID, "unknown option '%s'", name);
}

int fgmm_ctx_trim(fgmm_ctx *ctx) {
  if (!ctx) return fail(FGMM_ERR_INVALID, "ctx == NULL");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  if (!g.ok) return fail(FGMM_ERR_NO_DEVICE, "hipSetDevice(%d) failed", ctx->device);
  ctx->trim();
  return FGMM_OK;
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
  if (!ctx || which < 0 || which > 6 || !out) return fail(FGMM_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> lock(ctx->mu);
  *out = ctx->stat[which];
  return FGMM_OK;
}

int fgmm_ctx_call_log(fgmm_ctx *ctx, fgmm_call_marks *out, int cap, int *n_out) {
  if (!ctx || cap < 0 || (cap && !out) || !n_out) return fail(FGMM_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> lock(ctx->mu);
  const unsigned long long have = std::min<unsigned long long>(ctx->log_n, (unsigned long long)std::min(cap, (int)fgmm_ctx::kLogCap));
  for (unsigned long long k = 0; k < have; ++k) out[k] = ctx->log[(ctx->log_n - have + k) % fgmm_ctx::kLogCap];
  *n_out = (int)have;
  return FGMM_OK;
}

int fgmm_ctx_set_threads(fgmm_ctx *ctx, int n_threads) {
  if (!ctx) return fail(FGMM_ERR_INVALID, "ctx == NULL");
  std::lock_guard<std::mutex> lock(ctx->mu); // no call is in flight: the pool is idle
  if (n_threads <= 0) n_threads = fgmm_host_thread_budget(1);
  if (n_threads > 256) return fail(FGMM_ERR_INVALID, "n_threads %d", n_threads);
  if (ctx->pool && ctx->pool->size() == n_threads) return FGMM_OK;
  Pool *fresh = new (std::nothrow) Pool(n_threads);
  if (!fresh) return fail(FGMM_ERR_NOMEM, "worker pool");
  delete ctx->pool;
  ctx->pool = fresh;
  return FGMM_OK;
}

int fgmm_ctx_take_buffers(fgmm_ctx *ctx, void *const *dst, void *const *src, const size_t *len, int count) {
  if (!ctx || count < 0 || (count && (!dst || !src || !len))) return fail(FGMM_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> lock(ctx->mu);
  constexpr size_t kChunk = 1u << 20; // (a 3 MB bitstream is three workers' copies)
  size_t total = 0;
  for (int i = 0; i < count; ++i) {
    if (len[i] && (!dst[i] || !src[i])) return fail(FGMM_ERR_INVALID, "buffer %d is NULL", i);
    total += len[i];
  }
  if (total < (8u << 20)) { // (a Kodak batch's 2.5 MB: waking the workers costs more than the copy)
    for (int i = 0; i < count; ++i)
      if (len[i]) memcpy(dst[i], src[i], len[i]);
  } else {
    PoolDrain drain{ctx->pool};
    for (int i = 0; i < count; ++i)
      for (size_t at = 0; at < len[i]; at += kChunk) {
        char *d = static_cast<char *>(dst[i]) + at;
        const char *s_ = static_cast<const char *>(src[i]) + at;
        const size_t nb = std::min(kChunk, len[i] - at);
        ctx->pool->submit([d, s_, nb] { memcpy(d, s_, nb); });
      }
    ctx->pool->wait_all();
  }
  for (int i = 0; i < count; ++i) free(src[i]);
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
    if (s.params.flags & ~FGMM_PARAMS_LOGITS) return fail(FGMM_ERR_INVALID, "item %d: unknown fgmm_params.flags %d", i, s.params.flags);
    EncItem &e = v[i];
    e.y = s.y;
    e.prm = s.params;
    e.M = s.M;
    e.hw = s.hw;
    e.clamp = clamp_scales;
    e.yq = s.yq_out;
    e.zero_bitmap = s.zero_bitmap;
    if (s.ckpt_stride != items[0].ckpt_stride || s.ckpt_stride < 0 || (s.ckpt_stride && (s.ckpt_stride < 256 || (s.ckpt_stride & (s.ckpt_stride - 1)))))
      return fail(FGMM_ERR_INVALID, "item %d: ckpt_stride must be 0 or a power of two >= 256, the same for a whole batch", i);
    e.ckpt_stride = s.ckpt_stride;
  }
  const int rc = encode_batch(ctx, (hipStream_t)stream, v, mode);
  for (int i = 0; i < count; ++i) {
    items[i].abs_max = v[i].abs_max;
    items[i].bytes = v[i].bytes;
    items[i].bytes_len = v[i].bytes_len;
    items[i].status = v[i].status;
    items[i].ckpt = v[i].ckpt;
    items[i].n_ckpt = v[i].n_ckpt;
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
    if (s.params.flags & ~FGMM_PARAMS_LOGITS) return fail(FGMM_ERR_INVALID, "item %d: unknown fgmm_params.flags %d", i, s.params.flags);
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
    if (s.ckpt && s.n_ckpt > 0 && s.ckpt_stride > 0) { // (notes that do not fit the stream are ignored: sequential decode)
      d.ckpt = s.ckpt;
      d.n_ckpt = s.n_ckpt;
      d.ckpt_stride = s.ckpt_stride;
    }
  }
  ctx->stat[4] = ctx->stat[5] = 0;
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

int fgmm_softmax4_hip(fgmm_ctx *ctx, void *stream, const float *logits, float *weights, int64_t n) {
  if (!ctx || n < 0 || (n && (!logits || !weights))) return fail(FGMM_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  LAUNCH_TRY(launch_softmax_probe(logits, weights, n, stream));
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
  hd->packed_seg[0] = packed;
  hd->seg_b[0] = hd->seg_b[1] = hd->seg_b[2] = INT32_MAX; // the table in one piece
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
  if (max_bs < 0 || max_bs > FGMM_MAX_BS_H4) return fail(FGMM_ERR_UNSUPPORTED, "max_bs %d outside [0, %d] (4-byte headers)", max_bs, FGMM_MAX_BS_H4);
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
  hd->n = n;
  hd->n_ch = 1;
  hd->max_bs = max_bs;
  hd->prune = (flags & FGMM_TAB_NO_PRUNE) ? 0 : 1;
  hd->clamp = (flags & FGMM_TAB_CLAMP) ? 1 : 0;
  hd->hdr_form = 4;
  hd->ef_min = (flags & FGMM_TAB_RAW_ROWS) ? kTabNoEf : kTabEfMin;
  hd->tiles = tiles;
  hd->hdr = hdr;
  hd->pool = pool;
  hd->pool_cap = pool_cap;
  hd->pool_used = reinterpret_cast<unsigned long long *>(ctx->d_ws + 1024);
  hd->blk_sums = reinterpret_cast<uint32_t *>(ctx->d_ws + o_bsum);
  hd->blk_off = reinterpret_cast<unsigned long long *>(ctx->d_ws + o_boff);
  HIP_TRY(hipMemcpyAsync(ctx->d_ws, hd, sizeof *hd, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(ctx->d_ws + 1024, 0, 32, s));
  if (n) LAUNCH_TRY(launch_cdftab(reinterpret_cast<const DecDesc *>(ctx->d_ws), 1, 1, n, mode, (flags & FGMM_TAB_CLAMP) != 0, false, s));
  unsigned long long used[4] = {0, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(used, ctx->d_ws + 1024, 32, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipMemcpy(pool_used, used, sizeof(uint64_t), hipMemcpyHostToDevice));
  if (used[3]) return fail(FGMM_ERR_UNSUPPORTED, "an evaluation window beyond 2^20 edges");
  if (used[1]) return fail(FGMM_ERR_NOMEM, "pool_cap %llu bytes too small (need %llu)", (unsigned long long)pool_cap, used[0]);
  return FGMM_OK;
}

int fgmm_build_tab_hip(fgmm_ctx *ctx, void *stream, const float *scales, const float *means, const float *weights,
                       int64_t n, int64_t stride_n, int64_t stride_k, int mode, int32_t max_bs, int flags, void *hdr,
                       uint32_t *blk_off, uint8_t *rows, uint64_t rows_cap, uint64_t *rows_used, int32_t *tl_out) {
  if (!ctx || n < 0 || !mode_ok(mode) || !rows_used || !tl_out) return fail(FGMM_ERR_INVALID, "bad argument");
  if (max_bs < 0 || max_bs > FGMM_MAX_BS) return fail(FGMM_ERR_UNSUPPORTED, "max_bs %d outside [0, %d]", max_bs, FGMM_MAX_BS);
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  int cap_e = (int)std::min<int64_t>(std::max<int64_t>(ctx->opt.tab_cap_e, 256), 32768) & ~31;
  if (ctx->opt.tab_cap_e == kTabCapE && !tab_tl(max_bs, cap_e) && tab_tl(max_bs, kTabCapEWide)) cap_e = kTabCapEWide;
  const int tl = tab_tl(max_bs, cap_e);
  *tl_out = tl;
  if (!tl) return fail(FGMM_ERR_UNSUPPORTED, "2*max_bs+2 = %lld edges per latent do not fit the single-pass kernel (tab_cap_e = %d)", 2ll * max_bs + 2, cap_e);
  const int64_t nblk = (n + tl - 1) / tl;
  if (nblk > 0x7FFFFFFFll) return fail(FGMM_ERR_UNSUPPORTED, "too many blocks");
  int rc;
  const size_t o_scan = 1024 + align_up(kCounterBytes, 256);
  if ((rc = ctx->ensure_device(o_scan + sizeof(unsigned long long) * (size_t)std::max<int64_t>(nblk, 1))) || (rc = ctx->ensure_host(4096))) return rc;
  hipStream_t s = (hipStream_t)stream;
  DecDesc *hd = reinterpret_cast<DecDesc *>(ctx->h_ws);
  memset(hd, 0, sizeof *hd);
  hd->scales = scales;
  hd->means = means;
  hd->weights = weights;
  hd->stride_k = stride_k;
  hd->stride_p = stride_n;
  hd->hw = n;
  hd->n = n;
  hd->n_ch = 1;
  hd->max_bs = max_bs;
  hd->prune = (flags & FGMM_TAB_NO_PRUNE) ? 0 : 1;
  hd->clamp = (flags & FGMM_TAB_CLAMP) ? 1 : 0;
  hd->hdr_form = tab_hdr_form(max_bs);
  hd->ef_min = (flags & FGMM_TAB_RAW_ROWS) ? kTabNoEf : kTabEfMin;
  hd->tl = tl;
  hd->blk_begin = 0;
  hd->blk_end = (int32_t)nblk;
  hd->hdr_out = hdr;
  hd->blkoff_out = blk_off;
  hd->rows = rows;
  hd->rows_cap = rows_cap;
  hd->counters = reinterpret_cast<unsigned long long *>(ctx->d_ws + 1024);
  hd->count_edges = 1;
  hd->scan = reinterpret_cast<unsigned long long *>(ctx->d_ws + o_scan);
  hd->scan_base = 0;
  hd->scan_total = nblk;
  hd->spin_limit = (int32_t)ctx->opt.tab_spin;
  unsigned long long cn[kTabCounters] = {};
  for (int placement = ctx->opt.tab_place == 1 ? 1 : 0; placement >= 0; --placement) { // look-back; the cursor should a look-back give up
    hd->placement = placement;
    HIP_TRY(hipMemcpyAsync(ctx->d_ws, hd, sizeof *hd, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(ctx->d_ws + 1024, 0, o_scan - 1024 + sizeof(unsigned long long) * (size_t)std::max<int64_t>(nblk, 1), s));
    if (n) LAUNCH_TRY(launch_tab(reinterpret_cast<const DecDesc *>(ctx->d_ws), 1, (int)nblk, tl, cap_e, mode, (flags & FGMM_TAB_CLAMP) != 0, false, s));
    HIP_TRY(hipMemcpyAsync(cn, ctx->d_ws + 1024, sizeof cn, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (!(cn[1] & 2)) break;
    ++ctx->stat[6];
  }
  HIP_TRY(hipMemcpy(rows_used, cn, sizeof(uint64_t), hipMemcpyHostToDevice));
  ctx->stat[3] = 0;
  for (int q = 0; q < kTabEdgeSlots; ++q) ctx->stat[3] += cn[4 + q];
  if (cn[1] & 2) return fail(FGMM_ERR_HIP, "tab_kernel: placement failed");
  if (cn[1]) return fail(FGMM_ERR_NOMEM, "rows_cap %llu bytes too small (need %llu)", (unsigned long long)rows_cap, cn[0]);
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

// fgmm_ctx.h — what the host orchestration shares (fgmm_capi.cpp, fgmm_encode.cpp, fgmm_decode.cpp, fgmm_decode_gpu.cpp):
// the context, its worker pool, the per-call item structures.  Not part of the public ABI.  No GPU-runtime header is included
// here: the device is reached through fgmm_device.h only (the CPU test build links a fake behind it).
#pragma once
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
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
#include "fgmm_device.h"
#include "fgmm_internal.h"

namespace fgmm {

int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3))); // sets fgmm_last_error() of this thread, returns code
char *last_error_buffer(size_t *cap);

#define DEV_TRY(expr)                                                                                                       \
  do {                                                                                                                      \
    const int e_ = (expr);                                                                                                  \
    if (e_ != 0) return ::fgmm::fail(FGMM_ERR_HIP, "%s -> %s (%s:%d)", #expr, ::fgmm::dev::error_string(e_), __FILE__, __LINE__); \
  } while (0)
#define LAUNCH_TRY(expr) DEV_TRY(expr)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Phase clock of a batched call: always kept (the call log, fgmm_ctx_call_log); printed when option "trace" >= 1
struct Trace {
  bool on;
  int level;
  std::chrono::steady_clock::time_point t0, last;
  const char *what;
  Trace(const char *w, int lvl) : on(lvl > 0), level(lvl), what(w) { t0 = last = std::chrono::steady_clock::now(); }
  void mark(const char *phase) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[fgmm %s] %-28s +%8.3f ms  (t=%8.3f)\n", what, phase, std::chrono::duration<double, std::milli>(now - last).count(),
            std::chrono::duration<double, std::milli>(now - t0).count());
    last = now;
  }
  double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// Where a context's host workers may run.  They stream the decode-side tables (358 MB per Kodak step) through the L3 of whatever core
// complex they run on, and a calling thread that shares that L3 - a Python interpreter above all - runs the code between the calls
// from DRAM: 0.9 - 1.5 ms of glue per Kodak step instead of 0.5, by the luck of the scheduler's placement (profiles/r05_l3_ab.txt).
//   FGMM_WORKER_CPUS unset     the creating thread's mask MINUS the CPUs that share an L3 with the CPU it is on, when that leaves the
//                              workers enough room (at least 32 CPUs and two per worker); else the creating thread's mask
//   FGMM_WORKER_CPUS=inherit   the creating thread's mask, untouched
//   FGMM_WORKER_CPUS=<cpulist> exactly these ("0-7,16-23"), e.g. from a deployment that pins its calling thread somewhere else
// The library never touches the calling thread's own affinity.
struct WorkerCpus {
  cpu_set_t set;
  bool restricted = false; // false: the workers inherit the creating thread's mask
  bool automatic = false;  // `set` = the creating thread's mask minus its L3 (room() decides per pool size whether it is used)
  int cpus = 0;            // CPUs in `set`
  std::string error;       // FGMM_WORKER_CPUS could not be honoured: fgmm_ctx_create fails with this text
  // the automatic rule: the workers leave the creating thread's L3 alone when that leaves them at least 32 CPUs and two per worker
  bool room(int n_threads) const { return cpus >= std::max(32, 2 * n_threads); }
  void resize(int n_threads) { // fgmm_ctx_set_threads: the room rule is re-evaluated for the new pool (an explicit list stays)
    if (automatic) restricted = room(n_threads);
  }
  static bool parse(const char *text, cpu_set_t *out) {
    CPU_ZERO(out);
    int n = 0;
    for (const char *p = text; *p && *p != '\n';) {
      char *end = nullptr;
      const long lo = strtol(p, &end, 10);
      if (end == p || lo < 0) return false;
      long hi = lo;
      if (*end == '-') {
        const char *q = end + 1;
        hi = strtol(q, &end, 10);
        if (end == q || hi < lo) return false;
      }
      for (long c = lo; c <= hi && c < CPU_SETSIZE; ++c) CPU_SET((int)c, out), ++n;
      if (*end == ',') ++end;
      else if (*end && *end != '\n') return false;
      p = end;
    }
    return n > 0;
  }
  static WorkerCpus choose(int n_threads) {
    WorkerCpus w;
    CPU_ZERO(&w.set);
    const char *e = getenv("FGMM_WORKER_CPUS");
    if (e && !strcmp(e, "inherit")) return w;
    cpu_set_t have, l3;
    const bool have_mask = sched_getaffinity(0, sizeof have, &have) == 0;
    if (e && *e) {
      // an explicit list is honoured exactly or not at all: a typo must not silently put 48 workers on the caller's L3
      cpu_set_t asked;
      if (!parse(e, &asked)) {
        w.error = std::string("FGMM_WORKER_CPUS='") + e + "' is neither \"inherit\" nor a cpulist like \"16-63,80-127\"";
        return w;
      }
      // the EFFECTIVE set - what the kernel grants a thread of this process that asks for the list (the process's cpuset decides, not
      // the creating thread's own mask, which a caller may have narrowed to the CPUs it keeps for itself): asked for by a short-lived
      // thread and read back; fgmm_ctx_worker_cpus reports it
      bool granted = false;
      std::thread probe([&] {
        granted = sched_setaffinity(0, sizeof asked, &asked) == 0 && sched_getaffinity(0, sizeof w.set, &w.set) == 0;
      });
      probe.join();
      w.cpus = granted ? CPU_COUNT(&w.set) : 0;
      if (w.cpus == 0) {
        CPU_ZERO(&w.set);
        w.error = std::string("FGMM_WORKER_CPUS='") + e + "' names no CPU this process may run on";
        return w;
      }
      w.restricted = true;
      return w;
    }
    const int cpu = sched_getcpu();
    if (cpu < 0 || !have_mask) return w;
    char path[96], text[512];
    snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
    FILE *f = fopen(path, "r");
    if (!f) return w;
    const bool got = fgets(text, sizeof text, f) != nullptr;
    fclose(f);
    if (!got || !parse(text, &l3)) return w;
    for (int c = 0; c < CPU_SETSIZE; ++c)
      if (CPU_ISSET(c, &have) && !CPU_ISSET(c, &l3)) CPU_SET(c, &w.set), ++w.cpus;
    w.automatic = true;
    w.restricted = w.room(n_threads);
    return w;
  }
  // The same set with ONE hardware thread per core (the lowest-numbered sibling that is in the set): where the DECODE jobs run.  A host
  // decoder is a dependent chain that misses the cache on rows the DMA engine has just written; two of them on one core's two
  // hardware threads take 14 % more CPU time between them than on two cores (decode call 2 of a Kodak step: 48.6 -> 41.7 ms busy,
  // profiles/r05_smt_ab.txt) - while the encode call, 48 bitstreams at once, wants every hardware thread (it lost 10 % when the
  // whole pool was confined this way: hence two pools).  Unrestricted / unreadable topology / fewer than n_threads cores: *this.
  WorkerCpus one_per_core(int n_threads) const {
    cpu_set_t base;
    if (restricted) base = set;
    else if (sched_getaffinity(0, sizeof base, &base) != 0) return *this;
    WorkerCpus w;
    CPU_ZERO(&w.set);
    for (int c = 0; c < CPU_SETSIZE; ++c) {
      if (!CPU_ISSET(c, &base)) continue;
      char path[96], text[256];
      snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
      FILE *f = fopen(path, "r");
      if (!f) return *this;
      const bool got = fgets(text, sizeof text, f) != nullptr;
      fclose(f);
      cpu_set_t sib;
      if (!got || !parse(text, &sib)) return *this;
      bool first = true;
      for (int q = 0; q < c; ++q)
        if (CPU_ISSET(q, &sib) && CPU_ISSET(q, &base)) first = false;
      if (first) CPU_SET(c, &w.set), ++w.cpus;
    }
    if (w.cpus < n_threads || w.cpus == (restricted ? cpus : CPU_COUNT(&base))) return *this; // too few cores / no SMT here
    w.restricted = true;
    return w;
  }
  std::string cpulist() const { // "" when the workers inherit
    std::string out;
    if (!restricted) return out;
    for (int c = 0; c < CPU_SETSIZE; ++c) {
      if (!CPU_ISSET(c, &set)) continue;
      int hi = c;
      while (hi + 1 < CPU_SETSIZE && CPU_ISSET(hi + 1, &set)) ++hi;
      if (!out.empty()) out += ',';
      out += std::to_string(c);
      if (hi > c) out += '-' + std::to_string(hi);
      c = hi;
    }
    return out;
  }
};

// ---- host worker pool ----------------------------------------------------------------------------------
class Pool {
public:
  Pool(int n, const WorkerCpus &where, char tag = 'w') : slot_((size_t)n), lifo_(getenv("FGMM_POOL_FIFO") == nullptr) {
    for (int i = 0; i < n; ++i)
      th_.emplace_back([this, i, where, tag] {
        // (the affinity first: whoever finds a thread of this name in /proc/<pid>/task - bench.py's step_diag, the tests - sees its final mask)
        if (where.restricted) (void)sched_setaffinity(0, sizeof where.set, &where.set); // (refused by the kernel: the inherited mask)
        char name[16];
        snprintf(name, sizeof name, "fgmm-%c%d", tag, i); // fgmm-w*: the context's pool (encode, copies); fgmm-d*: the decode calls' pool // (/proc/<pid>/task/<tid>/comm: bench.py's step_diag names the threads that waited for a CPU)
        pthread_setname_np(pthread_self(), name);
        run(i);
      });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> l(m_);
      stop_ = true;
    }
    for (auto &s : slot_) s.cv.notify_one();
    for (auto &t : th_) t.join();
  }
  int size() const { return (int)th_.size(); }
  // A job wakes the worker that went idle LAST (a stack of idle workers, each with a condition variable of its own): a decode call
  // of 24 bitstreams after one of 24 runs on the same 24 workers - their stacks and the decoder's code warm, on the cores the
  // scheduler had settled them on - where one shared condition variable wakes the workers that have slept longest.
  void submit(std::function<void()> f) {
    Slot *wake = nullptr;
    {
      std::lock_guard<std::mutex> l(m_);
      q_.push(std::move(f));
      ++pending_;
      if (!idle_.empty()) {
        const size_t at = lifo_ ? idle_.size() - 1 : 0;
        wake = &slot_[(size_t)idle_[at]];
        idle_.erase(idle_.begin() + (std::ptrdiff_t)at);
        wake->signalled = true;
      }
    }
    if (wake) wake->cv.notify_one();
  }
  void wait_all() {
    std::unique_lock<std::mutex> l(m_);
    done_cv_.wait(l, [this] { return pending_ == 0; });
  }

private:
  struct Slot {
    std::condition_variable cv;
    bool signalled = false;
  };
  void run(int id) {
    Slot &me = slot_[(size_t)id];
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> l(m_);
        while (!stop_ && q_.empty()) { // nothing to do: onto the stack of idle workers until a job names this one (or the pool ends)
          idle_.push_back(id);
          me.signalled = false;
          me.cv.wait(l, [&] { return me.signalled || stop_; });
          if (!me.signalled) idle_.erase(std::find(idle_.begin(), idle_.end(), id)); // (woken by the pool's end: still on the stack)
        }
        if (q_.empty()) return; // (stop_, and every job taken)
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
  std::vector<Slot> slot_;
  std::vector<int> idle_; // ids of the workers that wait for a job, the one that went idle last at the back
  const bool lifo_;       // (FGMM_POOL_FIFO in the environment: the longest-idle worker first, as a shared condition variable does - A/B)
  std::mutex m_;
  std::condition_variable done_cv_;
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

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int d) {
    if (dev::get_device(&prev) == 0 && (prev == d || dev::set_device(d) == 0)) ok = true;
  }
  ~DeviceGuard() {
    int cur;
    if (ok && prev >= 0 && dev::get_device(&cur) == 0 && cur != prev) (void)dev::set_device(prev);
  }
};

// frees what a call allocated outside the context's reusable buffers (rare paths: overflow re-runs, generic items, raw rows)
struct TempDevice {
  std::vector<void *> v;
  ~TempDevice() {
    for (void *p : v) (void)dev::free_device(p);
  }
  int alloc(size_t bytes, char **out) {
    void *p = nullptr;
    DEV_TRY(dev::malloc_device(&p, std::max<size_t>(bytes, 256)));
    v.push_back(p);
    *out = static_cast<char *>(p);
    return FGMM_OK;
  }
};

} // namespace fgmm

struct fgmm_ctx {
  int device = 0;
  std::mutex mu; // one call at a time per context
  fgmm::Pool *pool = nullptr;
  fgmm::WorkerCpus worker_cpus; // decided when the context is created; fgmm_ctx_set_threads re-applies the automatic rule's room test
  fgmm::Pool *dec_pool = nullptr; // the decode calls' workers: as many, on one hardware thread per core (WorkerCpus::one_per_core);
                                  // == pool when the host has no second hardware threads, too few cores, or FGMM_DECODE_SMT=1 says so
  fgmm::Pool *decoders() const { return dec_pool ? dec_pool : pool; }
  char *d_ws = nullptr; // device workspace (descriptors, counters, encode tables): grown on demand, reused
  size_t d_cap = 0;
  char *h_ws = nullptr; // its pinned mirror
  size_t h_cap = 0;
  std::vector<fgmm::dev::Event> events;       // waited for by the calling thread for microseconds (the runtime polls)
  std::vector<fgmm::dev::Event> sleep_events; // mark the landing of table copies; waited for by the host workers for up to
                                              // milliseconds: those must SLEEP (a box's CPU quota is 16 cores: as many spinning
                                              // waiters plus the calling thread exceed it and the whole process is throttled)
  fgmm::dev::Stream copy_stream = nullptr; // bulk D2H of the decode tables (overlaps the table kernels of later launches)
  fgmm::dev::Stream aux_stream = nullptr;  // the few bytes of per-launch counters
  // tuning knobs (fgmm_ctx_set_option); the FGMM_* environment variables of the same meaning are read once, at creation
  struct Opts {
    int64_t pieces = 0, dec_first = 2, tab_cap_e = fgmm::kTabCapE, stage_max_mb = 0, trace = 0, enc_vec = 0, enc_linear = 1, ef_rows = 0,
            ef_min = fgmm::kTabEfDefault, enc_ways = 0, ckpt_decode = 0, spin_lat = 400000, gpu_decode = 0, enc_segs = 1, scatter_rounds = 1;
  } opt;
  // pinned receive area of the decode tables: a list of chunks, bump-allocated per call, never moved while copies are in
  // flight (sizes are only known launch by launch)
  struct Chunk {
    char *p;
    size_t cap, used;
  };
  std::vector<Chunk> chunks;
  void chunks_reset() {
    for (auto &c : chunks) c.used = 0;
  }
  int chunk_alloc(size_t bytes, char **out);
  // device staging area of the decode tables (headers, block offsets, rows): the table kernels write there, one copy per launch
  // fetches what was used.  Provisioned for the worst case of a call where memory allows (rows are placed by a cursor, nothing is
  // touched beyond it), else capped: a launch that overflows is re-run with the exact size.
  char *d_stage = nullptr;
  size_t d_stage_cap = 0;
  int ensure_stage(size_t bytes);
  size_t stage_budget() const; // bytes the staging area may take
  int ensure_streams();
  int ensure_device(size_t bytes);
  int ensure_host(size_t bytes);
  int ensure_events(size_t n, size_t n_sleep = 0);
  std::vector<int32_t> h_sym; // decode: int32 symbols of every bitstream of a call (grown, kept)
  void trim();                // gives every grown buffer back
  // the call log: phase marks of the most recent batched calls (fgmm_ctx_call_log), always kept - a few clock reads per call
  const std::chrono::steady_clock::time_point born = std::chrono::steady_clock::now();
  static constexpr int kLogCap = 64;
  fgmm_call_marks log[kLogCap];
  unsigned long long log_n = 0;
  void log_call(int kind, int count, const fgmm::Trace &tr, const double ms[5], double busy, double wait, const double *head = nullptr);
  // measurement aid (fgmm_ctx_set_profiling): timing events around the kernels
  bool profiling = false;
  fgmm::dev::Event prof[4][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
  bool prof_valid[4] = {false, false, false, false};
  int prof_begin(int which, fgmm::dev::Stream s);
  int prof_end(int which, fgmm::dev::Stream s);
  // counters (fgmm_ctx_stat): last batched call: [0] encode table bytes D2H, [1] decode table bytes D2H, [2] decode latents,
  // [3] edges the decode-side kernels evaluated, [4] bitstreams the GPU's segment decoder decoded, [5] ... handed back to the table path
  unsigned long long stat[6] = {0, 0, 0, 0, 0, 0};
};

namespace fgmm {

// ---- one bitstream of a batched encode --------------------------------------------------------------------------------------
struct EncItem {
  // inputs
  bool latent = false;               // latent-codec layout (y [M, hw], per-channel statistics, channel compaction) - also when M * hw == 0 and y is null
  const float *y = nullptr;          // device
  const float *x = nullptr;          // device: the parameter head's input features [c_in, hw] (fused head: encode_batch's `head`)
  const int32_t *sym_dev = nullptr;  // device (raw boundary)
  const int32_t *sym_host = nullptr; // host copy of the raw symbols when the caller has one
  fgmm_params prm{};
  int64_t stride_p = 1;
  int32_t M = 0;
  int64_t hw = 0;
  int clamp = 0;
  float *yq = nullptr;           // device out
  fgmm_symbuf *symbuf = nullptr; // raw boundary, buffered form: append the symbols instead of flushing a stream
  int64_t ckpt_stride = 0;       // note a checkpoint every this many symbols (fgmm_ckpt; 0: none)
  // outputs
  fgmm_ckpt *ckpt = nullptr; // malloc'ed, n_ckpt entries
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
  double t_sub = 0, t_start = 0, t_end = 0, t_waited = 0, t_lastland = 0; // job timeline (the call log; trace level 2)
};
int encode_batch(fgmm_ctx *ctx, dev::Stream stream, std::vector<EncItem> &items, int mode, const HeadW *head = nullptr, const fgmm_sink *sink = nullptr);

// ---- one bitstream of a batched decode --------------------------------------------------------------------------------------
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
  int32_t tl = 0;           // latents per block of the single-pass kernel; 0: generic two-pass path
  int64_t nblk = 0;         // blocks of tl latents
  uint64_t table_bytes = 0; // headers + block offsets + rows that crossed PCIe
  // how the tables reach the host: in n_piece pieces (block ranges); piece p arrives with launch unit piece_unit[p]
  int n_piece = 1;
  TabPiece piece[kMaxPieces] = {};
  int piece_unit[kMaxPieces] = {};
  char *h_out = nullptr; // pinned: decoded symbols (host-written, read by the scatter kernel)
  int wide = 0;          // h_out holds int32 symbols (some symbol outside int16), else int16
  int64_t narrowed = 0;  // symbols already converted to int16 in h_out (piece by piece)
  // schedule state, guarded by the call's mutex: pieces whose copy is queued | next piece to decode | a worker holds the item |
  // it is in the ready heap
  int queued = 0, next_piece = 0;
  bool busy = false, in_ready = false;
  TabDecoder dec;
  TabView view;
  int32_t *sym = nullptr; // int32 symbols (sym_host_out or a slice of the context's scratch)
  std::atomic<int> done{0};
  bool rounds = false; // its symbols go back to the GPU round by round (ScatDesc), not in one launch when it has finished
  // checkpointed streams decode as independent SEGMENTS (n_seg = n_ckpt + 1; 0: sequentially, piece by piece)
  int n_seg = 0, next_seg_push = 0;   // next_seg_push: guarded by the call's mutex
  int64_t piece_end[kMaxPieces] = {}; // one past the last latent of every piece (known when the call is planned)
  std::atomic<int> segs_left{0}, ckpt_bad{0}, wide_any{0};
  std::vector<uint32_t> enc_aligned; // a misaligned bitstream of a checkpointed item, copied ONCE (every segment starts a decoder on it)
  double t_taken = 0, t_start = 0, t_end = 0, t_waited = 0, t_lastland = 0, t_work = 0; // job timeline (the call log; trace level 2)
  DecItem() = default;
  DecItem(const DecItem &) = delete;
};
int decode_batch(fgmm_ctx *ctx, dev::Stream stream, std::vector<DecItem> &items, int mode);
// checkpointed bitstreams decoded ON THE GPU (segdec_kernel); `redo`: items whose segments did not all verify
int decode_batch_gpu(fgmm_ctx *ctx, dev::Stream stream, std::vector<DecItem> &items, const std::vector<int> &which, int mode, std::vector<int> &redo);
bool gpu_decodable(const DecItem &it, int64_t n);

constexpr size_t kCounterBytes = kTabCounters * sizeof(unsigned long long); // per launch unit, see DecDesc::counters

} // namespace fgmm

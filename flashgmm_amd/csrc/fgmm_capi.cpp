// fgmm_capi.cpp — the C ABI of libflashgmm_amd.so (include/flashgmm_amd.h): the context and its buffers, options, the host's CPU
// budget, and the entry points, which validate their arguments and hand batches to fgmm_encode.cpp / fgmm_decode.cpp.
//
// One fgmm_ctx per process per GPU owns a device workspace and a pinned host staging area (both grow on demand and are then
// reused), the events used to hand finished tables to the host coders, and a pool of host worker threads - one rANS state machine
// per bitstream.  The device is reached through fgmm_device.h only (fgmm_device_hip.cpp in the product).
#include <sched.h>

#include <string>

#include "fgmm_ctx.h"

using namespace fgmm;

namespace fgmm {

static thread_local char t_err[512] = "";
char *last_error_buffer(size_t *cap) {
  if (cap) *cap = sizeof t_err;
  return t_err;
}
int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(t_err, sizeof t_err, fmt, ap);
  va_end(ap);
  return code;
}

} // namespace fgmm

// ---- the context's buffers --------------------------------------------------------------------------------------------------
int fgmm_ctx::chunk_alloc(size_t bytes, char **out) {
  bytes = align_up(bytes, 256);
  for (auto &c : chunks)
    if (c.cap - c.used >= bytes) {
      *out = c.p + c.used;
      c.used += bytes;
      return FGMM_OK;
    }
  Chunk c{nullptr, std::max(bytes, (size_t)256 << 20), 0};
  DEV_TRY(dev::malloc_pinned((void **)&c.p, c.cap));
  c.used = bytes;
  chunks.push_back(c);
  *out = c.p;
  return FGMM_OK;
}
int fgmm_ctx::ensure_stage(size_t bytes) {
  if (bytes <= d_stage_cap) return FGMM_OK;
  if (d_stage) DEV_TRY(dev::free_device(d_stage));
  d_stage = nullptr;
  d_stage_cap = 0;
  DEV_TRY(dev::malloc_device((void **)&d_stage, bytes));
  d_stage_cap = bytes;
  return FGMM_OK;
}
size_t fgmm_ctx::stage_budget() const {
  if (opt.stage_max_mb > 0) return (size_t)opt.stage_max_mb << 20;
  size_t free_b = 0, total_b = 0;
  if (dev::mem_info(&free_b, &total_b) != 0) return (size_t)4 << 30;
  return std::max((free_b + d_stage_cap) / 4, (size_t)64 << 20);
}
int fgmm_ctx::ensure_streams() {
  if (!copy_stream) DEV_TRY(dev::stream_create(&copy_stream, true)); // (PCIe, the longest leg of a decode call, must not wait for a CU)
  if (!aux_stream) DEV_TRY(dev::stream_create(&aux_stream, false));
  return FGMM_OK;
}
int fgmm_ctx::ensure_device(size_t bytes) {
  if (bytes <= d_cap) return FGMM_OK;
  if (d_ws) DEV_TRY(dev::free_device(d_ws));
  d_ws = nullptr;
  d_cap = 0;
  const size_t want = align_up(bytes + bytes / 4, 1 << 20);
  DEV_TRY(dev::malloc_device((void **)&d_ws, want));
  d_cap = want;
  return FGMM_OK;
}
int fgmm_ctx::ensure_host(size_t bytes) {
  if (bytes <= h_cap) return FGMM_OK;
  if (h_ws) DEV_TRY(dev::free_pinned(h_ws));
  h_ws = nullptr;
  h_cap = 0;
  const size_t want = align_up(bytes + bytes / 4, 1 << 20);
  DEV_TRY(dev::malloc_pinned((void **)&h_ws, want));
  h_cap = want;
  return FGMM_OK;
}
int fgmm_ctx::ensure_events(size_t n, size_t n_sleep) {
  while (events.size() < n) {
    dev::Event e;
    DEV_TRY(dev::event_create(&e, 0));
    events.push_back(e);
  }
  while (sleep_events.size() < n_sleep) {
    dev::Event e;
    DEV_TRY(dev::event_create(&e, dev::kEventBlocking));
    sleep_events.push_back(e);
  }
  return FGMM_OK;
}
void fgmm_ctx::trim() {
  std::vector<int32_t>().swap(h_sym);
  if (d_ws) (void)dev::free_device(d_ws);
  if (h_ws) (void)dev::free_pinned(h_ws);
  if (d_stage) (void)dev::free_device(d_stage);
  for (auto &c : chunks) (void)dev::free_pinned(c.p);
  chunks.clear();
  d_ws = h_ws = d_stage = nullptr;
  d_cap = h_cap = d_stage_cap = 0;
}
void fgmm_ctx::log_call(int kind, int count, const Trace &tr, const double ms[5], double busy, double wait, const double *head) {
  fgmm_call_marks &m = log[log_n++ % kLogCap];
  m.kind = kind;
  m.count = count;
  m.t_begin_ms = std::chrono::duration<double, std::milli>(tr.t0 - born).count();
  for (int k = 0; k < 5; ++k) m.ms[k] = ms[k];
  m.ms[5] = tr.ms();
  m.worker_busy_ms = busy;
  m.worker_wait_ms = wait;
  for (int k = 0; k < 3; ++k) m.head_ms[k] = head ? head[k] : 0.0;
}
int fgmm_ctx::prof_begin(int which, dev::Stream s) {
  if (!profiling) return FGMM_OK;
  DEV_TRY(dev::event_record(prof[which][0], s));
  return FGMM_OK;
}
int fgmm_ctx::prof_end(int which, dev::Stream s) {
  if (!profiling) return FGMM_OK;
  DEV_TRY(dev::event_record(prof[which][1], s));
  prof_valid[which] = true;
  return FGMM_OK;
}

namespace {
bool mode_ok(int mode) { return mode >= 0 && mode <= 2; }
bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// can a raw (n,K) table use the 16-B-per-lane symtab kernel?
bool enc_vec4_ok(const EncDesc &d, bool f16) {
  const uintptr_t pm = f16 ? 7 : 15;
  auto al = [pm](const void *p) { return (reinterpret_cast<uintptr_t>(p) & pm) == 0; };
  return d.stride_p == 1 && (d.hw & 3) == 0 && (d.stride_c & 3) == 0 && (d.stride_k & 3) == 0 && al(d.scales) && al(d.means) &&
         al(d.weights) && (d.y ? aligned16(d.y) : aligned16(d.sym)) && aligned16(d.packed);
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
const char *fgmm_last_error(void) { return fgmm::last_error_buffer(nullptr); }

namespace {
struct OptName {
  const char *name;
  int64_t fgmm_ctx::Opts::*field;
  int64_t lo, hi;
  const char *env; // read once at context creation (compatibility with round-1 scripts)
};
const OptName kOpts[] = {
    {"pieces", &fgmm_ctx::Opts::pieces, 0, kMaxPieces, "FGMM_PIECES"},
    {"dec_first", &fgmm_ctx::Opts::dec_first, 1, 1 << 20, "FGMM_DEC_FIRST"},
    {"tab_cap_e", &fgmm_ctx::Opts::tab_cap_e, 256, 32768, "FGMM_TAB_CAP_E"},
    {"stage_max_mb", &fgmm_ctx::Opts::stage_max_mb, 0, 1 << 30, "FGMM_STAGE_MAX_MB"},
    {"trace", &fgmm_ctx::Opts::trace, 0, 2, "FGMM_TRACE"},
    {"enc_vec", &fgmm_ctx::Opts::enc_vec, 0, 8, "FGMM_VEC"},
    {"enc_linear", &fgmm_ctx::Opts::enc_linear, 0, 1, nullptr},
    {"ef_rows", &fgmm_ctx::Opts::ef_rows, 0, 2, "FGMM_EF_ROWS"},
    {"ef_min", &fgmm_ctx::Opts::ef_min, kTabEfMin, 1 << 20, "FGMM_EF_MIN_ROWS"},
    {"enc_ways", &fgmm_ctx::Opts::enc_ways, 0, kMaxEncWays, "FGMM_ENC_WAYS"},
    {"ckpt_decode", &fgmm_ctx::Opts::ckpt_decode, 0, 2, "FGMM_CKPT_DECODE"},
    {"spin_lat", &fgmm_ctx::Opts::spin_lat, -1, 1ll << 40, "FGMM_SPIN_LAT"},
    {"gpu_decode", &fgmm_ctx::Opts::gpu_decode, 0, 2, "FGMM_GPU_DECODE"},
    {"enc_segs", &fgmm_ctx::Opts::enc_segs, 0, 2, "FGMM_ENC_SEGS"}, // (2: segments whatever the tables' size - tests)
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
// the decode calls' pool (fgmm_ctx.h: one hardware thread per core), or none when that is the encode pool's own set
static void make_decode_pool(fgmm_ctx *c, int n_threads) {
  delete c->dec_pool;
  c->dec_pool = nullptr;
  const char *e = getenv("FGMM_DECODE_SMT");
  if (e && *e == '1') return; // (A/B: both hardware threads of a core, as the encode pool)
  const WorkerCpus one = c->worker_cpus.one_per_core(n_threads);
  if (one.restricted && one.cpus != (c->worker_cpus.restricted ? c->worker_cpus.cpus : -1)) c->dec_pool = new (std::nothrow) Pool(n_threads, one, 'd');
}

int fgmm_ctx_create(int device, int n_threads, fgmm_ctx **out) {
  if (!out) return fail(FGMM_ERR_INVALID, "out == NULL");
  *out = nullptr;
  int ndev = 0;
  if (dev::device_count(&ndev) != 0 || ndev <= 0)
    return fail(FGMM_ERR_NO_DEVICE, "no HIP device: libflashgmm_amd has no CPU path for the float work");
  if (device < 0 && dev::get_device(&device) != 0) return fail(FGMM_ERR_NO_DEVICE, "no current device");
  if (device >= ndev) return fail(FGMM_ERR_INVALID, "device %d of %d", device, ndev);
  DeviceGuard g(device);
  if (!g.ok) return fail(FGMM_ERR_NO_DEVICE, "cannot select device %d", device);
  if (n_threads <= 0) n_threads = fgmm_host_thread_budget(1);
  fgmm_ctx *c = new (std::nothrow) fgmm_ctx;
  if (!c) return fail(FGMM_ERR_NOMEM, "ctx");
  c->device = device;
  c->worker_cpus = WorkerCpus::choose(n_threads);
  if (!c->worker_cpus.error.empty()) {
    const std::string why = c->worker_cpus.error;
    delete c;
    return fail(FGMM_ERR_INVALID, "%s", why.c_str());
  }
  c->pool = new Pool(n_threads, c->worker_cpus);
  make_decode_pool(c, n_threads);
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
    delete ctx->dec_pool;
    for (auto e : ctx->events) (void)dev::event_destroy(e);
    for (auto e : ctx->sleep_events) (void)dev::event_destroy(e);
    for (auto &pr : ctx->prof)
      for (auto e : pr)
        if (e) (void)dev::event_destroy(e);
    ctx->trim();
    if (ctx->copy_stream) (void)dev::stream_destroy(ctx->copy_stream);
    if (ctx->aux_stream) (void)dev::stream_destroy(ctx->aux_stream);
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
  if (!g.ok) return fail(FGMM_ERR_NO_DEVICE, "cannot select device %d", ctx->device);
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
        if (!e) DEV_TRY(dev::event_create(&e, dev::kEventTiming));
  ctx->profiling = enable != 0;
  return FGMM_OK;
}

int fgmm_ctx_kernel_ms(fgmm_ctx *ctx, int which, float *ms_out) {
  if (!ctx || which < 0 || which > 3 || !ms_out) return fail(FGMM_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> lock(ctx->mu);
  if (!ctx->profiling || !ctx->prof_valid[which]) return fail(FGMM_ERR_INVALID, "no profiled launch of kernel %d yet", which);
  DeviceGuard g(ctx->device);
  DEV_TRY(dev::event_sync(ctx->prof[which][1]));
  DEV_TRY(dev::event_elapsed_ms(ms_out, ctx->prof[which][0], ctx->prof[which][1]));
  return FGMM_OK;
}

int fgmm_ctx_stat(fgmm_ctx *ctx, int which, uint64_t *out) {
  if (!ctx || which < 0 || which > 5 || !out) return fail(FGMM_ERR_INVALID, "bad argument");
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
  ctx->worker_cpus.resize(n_threads);
  Pool *fresh = new (std::nothrow) Pool(n_threads, ctx->worker_cpus);
  if (!fresh) return fail(FGMM_ERR_NOMEM, "worker pool");
  delete ctx->pool;
  ctx->pool = fresh;
  make_decode_pool(ctx, n_threads);
  return FGMM_OK;
}

int fgmm_ctx_worker_cpus(fgmm_ctx *ctx, char *out, size_t cap) {
  if (!ctx || !out || cap == 0) return fail(FGMM_ERR_INVALID, "bad argument");
  const std::string s = ctx->worker_cpus.cpulist();
  if (s.size() + 1 > cap) return fail(FGMM_ERR_INVALID, "cpulist of %zu characters, room for %zu", s.size(), cap - 1);
  memcpy(out, s.c_str(), s.size() + 1);
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

// the packed weights of a parameter head (fgmm_head.hip), owned by the caller through fgmm_head_create / _destroy
struct fgmm_head {
  int device = 0;
  float *packed = nullptr; // [wp | bp]
  fgmm::HeadW w{};
  // bf16x6: the features' split copy of the call in progress (grown on demand; calls are serialised by the context's lock)
  mutable void *xs = nullptr;
  mutable size_t xs_cap = 0;
};

// FGMM_HEAD_BF16X6: split the features of a call's items into their three bf16 parts (fgmm_head16.hip) -> the pointers the kernels'
// descriptors carry as `x`.  One launch when the items are evenly spaced and of one size (stacked tensors), one per item otherwise.
static int head16_split(const fgmm_head *head, void *stream, const float *const *x, const int64_t *hw, int count, std::vector<const float *> &out) {
  out.assign((size_t)count, nullptr);
  std::vector<size_t> at((size_t)count + 1, 0);
  for (int i = 0; i < count; ++i) at[(size_t)i + 1] = at[(size_t)i] + head16_split_elems(head->w.c_in, hw[i]);
  const size_t bytes = at[(size_t)count] * sizeof(uint16_t);
  if (bytes > head->xs_cap) {
    if (head->xs) (void)dev::free_device(head->xs);
    head->xs = nullptr, head->xs_cap = 0;
    void *p = nullptr;
    if (dev::malloc_device(&p, bytes + 256) != 0) return fail(FGMM_ERR_NOMEM, "%zu bytes of device memory for the split features", bytes);
    head->xs = p, head->xs_cap = bytes;
  }
  uint16_t *base = static_cast<uint16_t *>(head->xs);
  for (int i = 0; i < count; ++i) out[(size_t)i] = reinterpret_cast<const float *>(base + at[(size_t)i]);
  bool even = count > 0;
  for (int i = 1; i < count && even; ++i) even = hw[i] == hw[0] && x[i] - x[i - 1] == x[1] - x[0];
  if (even && hw[0] > 0) {
    LAUNCH_TRY(launch_head16_split(x[0], base, hw[0], head->w.c_in, count, count > 1 ? (int64_t)(x[1] - x[0]) : 0, (int64_t)at[1], stream));
  } else {
    for (int i = 0; i < count; ++i)
      if (hw[i] > 0) LAUNCH_TRY(launch_head16_split(x[i], base + at[(size_t)i], hw[i], head->w.c_in, 1, 0, 0, stream));
  }
  return FGMM_OK;
}

static int compress_batch_impl(fgmm_ctx *ctx, void *stream, fgmm_item *items, int count, int mode, int clamp_scales, const fgmm_head *head,
                               const float *const *x, const fgmm_sink *sink = nullptr) {
  if (!ctx || count < 0 || (count && !items) || !mode_ok(mode) || (sink && !sink->alloc)) return fail(FGMM_ERR_INVALID, "bad argument");
  if (head && (head->device != ctx->device || (count && !x))) return fail(FGMM_ERR_INVALID, "head: made for device %d, context on %d (or x == NULL)", head->device, ctx->device);
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  if (!g.ok) return fail(FGMM_ERR_NO_DEVICE, "cannot select device %d", ctx->device);
  std::vector<EncItem> v((size_t)count);
  std::vector<const float *> xsplit;
  if (head && head->w.arith == FGMM_HEAD_BF16X6 && count) { // the features' three bf16 parts, once for all of the call's blocks
    std::vector<int64_t> hws((size_t)count);
    for (int i = 0; i < count; ++i) {
      if (items[i].hw < 0 || (items[i].M * items[i].hw && !x[i])) return fail(FGMM_ERR_INVALID, "item %d: null tensor / negative size", i);
      hws[(size_t)i] = (int64_t)items[i].M * items[i].hw ? items[i].hw : 0;
    }
    if (int rc = head16_split(head, stream, x, hws.data(), count, xsplit)) return rc;
    x = xsplit.data();
  }
  for (int i = 0; i < count; ++i) {
    const fgmm_item &s = items[i];
    if (s.K != FGMM_K) return fail(FGMM_ERR_INVALID, "K = %d: the reference binds K = 4 only", s.K);
    EncItem &e = v[i];
    if (head) {
      // the parameters come out of the head's matrix product: the item names its features instead of parameter planes
      if (s.M != head->w.M) return fail(FGMM_ERR_INVALID, "item %d: M = %d, the head was made for M = %d", i, s.M, head->w.M);
      if (s.hw < 0 || (s.M * s.hw && (!s.y || !x[i]))) return fail(FGMM_ERR_INVALID, "item %d: null tensor / negative size", i);
      e.x = x[i];
      e.prm = fgmm_params{};
      e.prm.dtype = FGMM_F32;
      e.prm.flags = FGMM_PARAMS_LOGITS;
    } else {
      if (s.M < 0 || s.hw < 0 || (s.M * s.hw && (!s.y || !s.params.scales || !s.params.means || !s.params.weights)))
        return fail(FGMM_ERR_INVALID, "item %d: null tensor / negative size", i);
      if (s.params.dtype != items[0].params.dtype || (s.params.dtype != FGMM_F32 && s.params.dtype != FGMM_F16))
        return fail(FGMM_ERR_INVALID, "item %d: parameter dtype must be FGMM_F32 or FGMM_F16 and the same for a whole batch", i);
      if (s.params.flags & ~FGMM_PARAMS_LOGITS) return fail(FGMM_ERR_INVALID, "item %d: unknown fgmm_params.flags %d", i, s.params.flags);
      e.prm = s.params;
    }
    e.latent = true;
    e.y = s.y;
    e.M = s.M;
    e.hw = s.hw;
    e.clamp = clamp_scales;
    e.yq = s.yq_out;
    e.zero_bitmap = s.zero_bitmap;
    if (s.ckpt_stride != items[0].ckpt_stride || s.ckpt_stride < 0 || (s.ckpt_stride && (s.ckpt_stride < 256 || (s.ckpt_stride & (s.ckpt_stride - 1)))))
      return fail(FGMM_ERR_INVALID, "item %d: ckpt_stride must be 0 or a power of two >= 256, the same for a whole batch", i);
    e.ckpt_stride = s.ckpt_stride;
  }
  const int rc = encode_batch(ctx, (dev::Stream)stream, v, mode, head ? &head->w : nullptr, sink);
  if (rc != FGMM_OK) // a failed call returns no buffer: what the bitstreams that had finished hold is released here, not leaked by a
    for (auto &e : v) { // binding that raises on the status
      if (!sink) free(e.bytes); // (a sink's storage is the caller's)
      free(e.ckpt);
      e.bytes = nullptr, e.ckpt = nullptr, e.bytes_len = 0, e.n_ckpt = 0;
    }
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

int fgmm_gmc_compress_batch(fgmm_ctx *ctx, void *stream, fgmm_item *items, int count, int mode, int clamp_scales) {
  return compress_batch_impl(ctx, stream, items, count, mode, clamp_scales, nullptr, nullptr);
}
int fgmm_gmc_compress_batch_to(fgmm_ctx *ctx, void *stream, fgmm_item *items, int count, int mode, int clamp_scales, const fgmm_sink *sink) {
  return compress_batch_impl(ctx, stream, items, count, mode, clamp_scales, nullptr, nullptr, sink);
}

// ---- the parameter head ---------------------------------------------------------------------------------------
int fgmm_head_create(fgmm_ctx *ctx, void *stream, const float *weight, const float *bias, int M, int K, int c_in, fgmm_head **out) {
  return fgmm_head_create_ex(ctx, stream, weight, bias, M, K, c_in, 0, out);
}

int fgmm_head_create_ex(fgmm_ctx *ctx, void *stream, const float *weight, const float *bias, int M, int K, int c_in, int flags, fgmm_head **out) {
  if (!out) return fail(FGMM_ERR_INVALID, "out == NULL");
  *out = nullptr;
  if (!ctx || !weight || M <= 0 || c_in <= 0 || M > (1 << 20) || c_in > (1 << 20) || (flags & ~FGMM_HEAD_BF16X6)) return fail(FGMM_ERR_INVALID, "bad argument");
  if (K != FGMM_K) return fail(FGMM_ERR_INVALID, "K = %d: the reference binds K = 4 only", K);
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  if (!g.ok) return fail(FGMM_ERR_NO_DEVICE, "cannot select device %d", ctx->device);
  fgmm_head *h = new (std::nothrow) fgmm_head;
  if (!h) return fail(FGMM_ERR_NOMEM, "head");
  h->device = ctx->device;
  h->w.M = M, h->w.c_in = c_in;
  h->w.n_cg = (M + kHeadCG - 1) / kHeadCG, h->w.n_kt = (c_in + kHeadBK - 1) / kHeadBK;
  const size_t floats = head_packed_floats(M, c_in), bias_floats = (size_t)h->w.n_cg * 12 * kHeadCG;
  const bool b16 = (flags & FGMM_HEAD_BF16X6) != 0;
  const size_t bytes = b16 ? head16_packed_bytes(M, c_in) : floats * sizeof(float);
  void *p = nullptr;
  if (dev::malloc_device(&p, bytes) != 0) {
    delete h;
    return fail(FGMM_ERR_NOMEM, "%zu bytes of device memory for the packed weights", bytes);
  }
  h->packed = static_cast<float *>(p);
  h->w.arith = b16 ? FGMM_HEAD_BF16X6 : 0;
  h->w.wp = p;
  h->w.bp = b16 ? reinterpret_cast<const float *>(static_cast<const char *>(p) + (bytes - bias_floats * sizeof(float))) : h->packed + (floats - bias_floats);
  int e = b16 ? launch_head16_pack(weight, bias, M, c_in, p, stream) : launch_head_pack(weight, bias, M, c_in, h->packed, h->packed + (floats - bias_floats), stream);
  if (!e) e = dev::stream_sync((dev::Stream)stream); // (the caller may free or overwrite its weights on return)
  if (e) {
    (void)dev::free_device(p);
    delete h;
    return fail(FGMM_ERR_HIP, "packing the head's weights: %s", dev::error_string(e));
  }
  *out = h;
  return FGMM_OK;
}

void fgmm_head_destroy(fgmm_head *h) {
  if (!h) return;
  {
    DeviceGuard g(h->device);
    (void)dev::free_device(h->packed);
    if (h->xs) (void)dev::free_device(h->xs);
  }
  delete h;
}

int fgmm_head_params_batch(fgmm_ctx *ctx, void *stream, const fgmm_head *head, const float *const *x, float *const *out, const int64_t *hw, int count) {
  if (!ctx || !head || count < 0 || (count && (!x || !out || !hw))) return fail(FGMM_ERR_INVALID, "bad argument");
  if (head->device != ctx->device) return fail(FGMM_ERR_INVALID, "head: made for device %d, context on %d", head->device, ctx->device);
  if (count == 0) return FGMM_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  if (!g.ok) return fail(FGMM_ERR_NO_DEVICE, "cannot select device %d", ctx->device);
  int rc;
  const size_t bytes = sizeof(HeadDesc) * (size_t)count;
  if ((rc = ctx->ensure_device(bytes)) || (rc = ctx->ensure_host(bytes))) return rc;
  HeadDesc *hd = reinterpret_cast<HeadDesc *>(ctx->h_ws);
  int64_t hw_max = 0;
  bool vec = true;
  for (int i = 0; i < count; ++i)
    if (hw[i] < 0 || (hw[i] && (!x[i] || !out[i]))) return fail(FGMM_ERR_INVALID, "item %d: null tensor / negative size", i);
  std::vector<const float *> xsplit;
  if (head->w.arith == FGMM_HEAD_BF16X6) {
    if ((rc = head16_split(head, stream, x, hw, count, xsplit))) return rc;
    x = xsplit.data();
  }
  for (int i = 0; i < count; ++i) {
    hd[i] = HeadDesc{x[i], out[i], hw[i]};
    hw_max = std::max(hw_max, hw[i]);
    vec = vec && (hw[i] & 3) == 0 && (reinterpret_cast<uintptr_t>(x[i]) & 15) == 0;
  }
  DEV_TRY(dev::copy_async(ctx->d_ws, hd, bytes, dev::kH2D, (dev::Stream)stream));
  if (head->w.arith == FGMM_HEAD_BF16X6) LAUNCH_TRY(launch_head16_params(reinterpret_cast<const HeadDesc *>(ctx->d_ws), head->w, count, hw_max, stream));
  else LAUNCH_TRY(launch_head_params(reinterpret_cast<const HeadDesc *>(ctx->d_ws), head->w, count, hw_max, vec, stream));
  DEV_TRY(dev::stream_sync((dev::Stream)stream)); // (the descriptors' staging area belongs to the next call)
  return FGMM_OK;
}

int fgmm_gmc_compress_head_batch(fgmm_ctx *ctx, void *stream, fgmm_item *items, const float *const *x, int count, const fgmm_head *head, int mode,
                                 int clamp_scales) {
  if (!head) return fail(FGMM_ERR_INVALID, "head == NULL");
  return compress_batch_impl(ctx, stream, items, count, mode, clamp_scales, head, x);
}
int fgmm_gmc_compress_head_batch_to(fgmm_ctx *ctx, void *stream, fgmm_item *items, const float *const *x, int count, const fgmm_head *head, int mode,
                                    int clamp_scales, const fgmm_sink *sink) {
  if (!head) return fail(FGMM_ERR_INVALID, "head == NULL");
  return compress_batch_impl(ctx, stream, items, count, mode, clamp_scales, head, x, sink);
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
  if (!g.ok) return fail(FGMM_ERR_NO_DEVICE, "cannot select device %d", ctx->device);
  std::vector<DecItem> v((size_t)count);
  for (int i = 0; i < count; ++i) {
    const fgmm_item &s = items[i];
    if (s.K != FGMM_K) return fail(FGMM_ERR_INVALID, "K = %d: the reference binds K = 4 only", s.K);
    if (s.M < 0 || s.hw < 0 || !s.bytes || (s.M && !s.zero_bitmap) ||
        (s.M * s.hw && (!s.yq_out || !s.params.scales || !s.params.means || !s.params.weights)))
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
  const int rc = decode_batch(ctx, (dev::Stream)stream, v, mode);
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
int stage_rows(fgmm_ctx *ctx, dev::Stream stream, const float *scales, const float *means, const float *weights,
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
    DEV_TRY(dev::malloc_device((void **)&d, span * sizeof(float) + 64));
    to_free.push_back(d);
    if (dense) {
      DEV_TRY(dev::copy_async(d, src[a], span * sizeof(float), dev::kH2D, stream));
    } else {
      std::vector<float> tmp((size_t)n * 4);
      for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < 4; ++k) tmp[(size_t)i * 4 + k] = src[a][i * stride_n + k * stride_k];
      DEV_TRY(dev::copy_sync(d, tmp.data(), tmp.size() * sizeof(float), dev::kH2D));
    }
    dst[a] = d;
  }
  *out = {dst[0], dst[1], dst[2], dense ? stride_n : 4, dense ? stride_k : 1};
  return FGMM_OK;
}

struct FreeList {
  std::vector<void *> v;
  ~FreeList() {
    for (void *p : v) (void)dev::free_device(p);
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
  if (!g.ok) return fail(FGMM_ERR_NO_DEVICE, "cannot select device %d", ctx->device);
  dev::Stream stream = nullptr;
  FreeList fl;
  StagedRows r;
  int rc = stage_rows(ctx, stream, scales, means, weights, n, stride_n, stride_k, memspace, fl.v, &r);
  if (rc) return rc;
  const int32_t *sym_dev = symbols;
  if (memspace == FGMM_HOST && n) {
    int32_t *d = nullptr;
    DEV_TRY(dev::malloc_device((void **)&d, sizeof(int32_t) * (size_t)n + 64));
    fl.v.push_back(d);
    DEV_TRY(dev::copy_async(d, symbols, sizeof(int32_t) * (size_t)n, dev::kH2D, stream));
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
  if (!g.ok) return fail(FGMM_ERR_NO_DEVICE, "cannot select device %d", ctx->device);
  dev::Stream stream = nullptr;
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
  DEV_TRY(dev::stream_sync((dev::Stream)stream));
  return FGMM_OK;
}

int fgmm_softmax4_hip(fgmm_ctx *ctx, void *stream, const float *logits, float *weights, int64_t n) {
  if (!ctx || n < 0 || (n && (!logits || !weights))) return fail(FGMM_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  LAUNCH_TRY(launch_softmax_probe(logits, weights, n, stream));
  DEV_TRY(dev::stream_sync((dev::Stream)stream));
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
  dev::Stream s = (dev::Stream)stream;
  DEV_TRY(dev::copy_async(ctx->d_ws, hd, sizeof *hd, dev::kH2D, s));
  DEV_TRY(dev::memset_async(ctx->d_ws + 1024, 0, meta_bytes, s));
  LAUNCH_TRY(launch_symtab(reinterpret_cast<const EncDesc *>(ctx->d_ws), 1, 1, n, n, false, mode, enc_vec4_ok(*hd, false) ? 4 : 1, false, false, s));
  DEV_TRY(dev::stream_sync(s));
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
  dev::Stream s = (dev::Stream)stream;
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
  DEV_TRY(dev::copy_async(ctx->d_ws, hd, sizeof *hd, dev::kH2D, s));
  DEV_TRY(dev::memset_async(ctx->d_ws + 1024, 0, 32, s));
  if (n) LAUNCH_TRY(launch_cdftab(reinterpret_cast<const DecDesc *>(ctx->d_ws), 1, 1, n, mode, (flags & FGMM_TAB_CLAMP) != 0, false, s));
  unsigned long long used[4] = {0, 0, 0, 0};
  DEV_TRY(dev::copy_async(used, ctx->d_ws + 1024, 32, dev::kD2H, s));
  DEV_TRY(dev::stream_sync(s));
  DEV_TRY(dev::copy_sync(pool_used, used, sizeof(uint64_t), dev::kH2D));
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
  if ((rc = ctx->ensure_device(1024 + align_up(kCounterBytes, 256))) || (rc = ctx->ensure_host(4096))) return rc;
  dev::Stream s = (dev::Stream)stream;
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
  unsigned long long cn[kTabCounters] = {};
  DEV_TRY(dev::copy_async(ctx->d_ws, hd, sizeof *hd, dev::kH2D, s));
  DEV_TRY(dev::memset_async(ctx->d_ws + 1024, 0, kCounterBytes, s));
  if (n) LAUNCH_TRY(launch_tab(reinterpret_cast<const DecDesc *>(ctx->d_ws), 1, (int)nblk, tl, cap_e, mode, (flags & FGMM_TAB_CLAMP) != 0, false, s));
  DEV_TRY(dev::copy_async(cn, ctx->d_ws + 1024, sizeof cn, dev::kD2H, s));
  DEV_TRY(dev::stream_sync(s));
  DEV_TRY(dev::copy_sync(rows_used, cn, sizeof(uint64_t), dev::kH2D));
  ctx->stat[3] = 0;
  for (int q = 0; q < kTabEdgeSlots; ++q) ctx->stat[3] += cn[4 + q];
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
  DEV_TRY(dev::memset_async(d, 0, 24, nullptr));
  LAUNCH_TRY(launch_fastmath_selftest(which, n, seed, d, nullptr));
  unsigned long long bad[3] = {0, 0, 0};
  DEV_TRY(dev::copy_sync(bad, d, 24, dev::kD2H));
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
  DEV_TRY(dev::memset_async(d, 0, 8, nullptr));
  LAUNCH_TRY(launch_saturation_selftest(mode, d, nullptr));
  unsigned long long bad = 0;
  DEV_TRY(dev::copy_sync(&bad, d, 8, dev::kD2H));
  *n_bad_out = bad;
  return FGMM_OK;
}

} // extern "C"

// stress_main.cpp — TEST INFRASTRUCTURE: randomized batches through the product's C ABI (include/flashgmm_amd.h) on the FAKE device
// (fake_device.cpp), meant to run under ThreadSanitizer / AddressSanitizer + UBSan (scripts/tsan_host.sh).  Every batch:
//   fgmm_gmc_compress_batch   -> each bitstream == the oracle's encoder on the same symbols and parameters (fgo_encode_gmm), bit for bit
//   fgmm_gmc_decompress_batch -> y_hat == round(y) for every item
// under random pipeline options (pieces, first launch, Elias-Fano threshold, encoder ways, whole / segmented encode
// tables, scatter rounds, checkpointed streams decoded in host segments / handed to the "GPU" and partly handed back, a tiny staging
// budget that forces overflow re-runs, an LDS budget that sends items to the generic kernels) and random worker counts; now and then a
// truncated bitstream, which must be refused, and a corrupted one, which must decode to what the oracle's decoder makes of it.
//   stress_main [seconds] [seed]
#include <atomic>
#include <thread>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <dirent.h>
#include <sched.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "../../include/flashgmm_amd.h"

extern "C" int fgo_encode_gmm(int mode, int64_t n, const int32_t *symbols, const float *scales, const float *means, const float *weights, int64_t sn, int64_t sk,
                              uint8_t **out, size_t *out_len, int64_t *n_bypass_out);
extern "C" int fgo_decode_gmm(int mode, const uint8_t *enc, size_t enc_len, int64_t n, const float *scales, const float *means, const float *weights, int64_t sn, int64_t sk,
                              int32_t max_bs, int32_t *out);
extern "C" void fgo_free(void *p);

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() {
  rng_state ^= rng_state << 13, rng_state ^= rng_state >> 7, rng_state ^= rng_state << 17;
  return rng_state;
}
static double uni() { return (double)(rnd() >> 11) / 9007199254740992.0; }
static double gauss() { return sqrt(-2.0 * log(uni() + 1e-300)) * cos(6.283185307179586 * uni()); }
static int64_t pick(int64_t lo, int64_t hi) { return lo + (int64_t)(rnd() % (uint64_t)(hi - lo + 1)); }

#define CHECK(cond, ...)                                     \
  do {                                                       \
    if (!(cond)) {                                           \
      fprintf(stderr, "STRESS FAILURE %s:%d: ", __FILE__, __LINE__); \
      fprintf(stderr, __VA_ARGS__);                          \
      fprintf(stderr, "  [%s]\n", fgmm_last_error());        \
      exit(1);                                               \
    }                                                        \
  } while (0)

struct Item {
  int M;
  int64_t hw;
  std::vector<float> y, sg, mu, pi, yq, yhat; // planes [K, M, hw]
  std::vector<int64_t> zb;
};

static void make_item(Item &it, int M, int64_t hw, double zero_frac) {
  it.M = M, it.hw = hw;
  const size_t n = (size_t)M * (size_t)hw;
  it.y.resize(n), it.yq.assign(n, -7.f), it.yhat.assign(n, -9.f), it.zb.assign((size_t)M, -1);
  it.sg.resize(4 * n), it.mu.resize(4 * n), it.pi.resize(4 * n);
  for (int c = 0; c < M; ++c) {
    const double e = exp(-3.0 + 4.5 * uni());
    const bool dead = uni() < zero_frac;
    for (int64_t p = 0; p < hw; ++p) {
      const size_t i = (size_t)c * hw + p;
      it.y[i] = dead ? (float)(0.4 * (uni() - 0.5)) : (float)(gauss() * 1.5 * e);
      double w[4], ws = 0;
      for (int k = 0; k < 4; ++k) w[k] = exp(gauss()), ws += w[k];
      for (int k = 0; k < 4; ++k) {
        it.mu[(size_t)k * n + i] = (float)(gauss() * e);
        it.sg[(size_t)k * n + i] = (float)((uni() * 2 + 0.05) * e); // un-clamped: the path clamps to [0.11, 256]
        it.pi[(size_t)k * n + i] = (float)(w[k] / ws * 0.999);      // (sum <= 1 after rounding)
      }
    }
  }
}

// one coded channel of an item through the raw (n, 4) boundary, host memory, the reference's view strides (1, n)
static void raw_boundary(fgmm_ctx *ctx, const Item &it, int mode) {
  const int64_t n = it.hw;
  const size_t n_all = (size_t)it.M * it.hw;
  const int c = (int)(rnd() % (uint64_t)it.M);
  std::vector<int32_t> sym((size_t)n);
  std::vector<float> s(4 * (size_t)n), m(4 * (size_t)n), w(4 * (size_t)n); // [k * n + i]: stride_n 1, stride_k n
  int32_t amax = 0;
  for (int64_t p = 0; p < n; ++p) {
    sym[(size_t)p] = (int32_t)nearbyintf(it.y[(size_t)c * it.hw + p]);
    amax = std::max(amax, std::abs(sym[(size_t)p]));
    for (int k = 0; k < 4; ++k) {
      s[(size_t)k * n + p] = fminf(fmaxf(it.sg[(size_t)k * n_all + (size_t)c * it.hw + p], 0.11f), 256.0f);
      m[(size_t)k * n + p] = it.mu[(size_t)k * n_all + (size_t)c * it.hw + p];
      w[(size_t)k * n + p] = it.pi[(size_t)k * n_all + (size_t)c * it.hw + p];
    }
  }
  uint8_t *want = nullptr, *got = nullptr;
  size_t want_len = 0, got_len = 0;
  CHECK(fgo_encode_gmm(mode, n, sym.data(), s.data(), m.data(), w.data(), 1, n, &want, &want_len, nullptr) == 0, "oracle");
  CHECK(fgmm_encode_with_indexes_gmm(ctx, sym.data(), s.data(), m.data(), w.data(), n, 1, n, 4, mode, FGMM_HOST, amax + 1, &got, &got_len) == FGMM_OK, "encode_with_indexes_gmm");
  CHECK(got_len == want_len && !memcmp(got, want, want_len), "raw boundary: bytes differ from the oracle's (%zu / %zu)", got_len, want_len);
  std::vector<int32_t> dec((size_t)n, 12345);
  CHECK(fgmm_decode_with_indexes_gmm(ctx, got, got_len, s.data(), m.data(), w.data(), n, 1, n, 4, mode, FGMM_HOST, amax + 1, dec.data()) == FGMM_OK, "decode_with_indexes_gmm");
  CHECK(dec == sym, "raw boundary: decoded symbols differ");
  fgmm_free(got);
  // BufferedRansEncoder: the channel in two appends, one flush
  fgmm_symbuf *b = nullptr;
  CHECK(fgmm_symbuf_create(&b) == FGMM_OK, "symbuf");
  const int64_t cut = n / 3;
  // (an append takes a contiguous run of rows: with stride (1, n) the second part starts `cut` elements into every plane)
  CHECK(fgmm_symbuf_append_gmm(ctx, b, sym.data(), s.data(), m.data(), w.data(), cut, 1, n, 4, mode, FGMM_HOST) == FGMM_OK, "symbuf append 1");
  CHECK(fgmm_symbuf_append_gmm(ctx, b, sym.data() + cut, s.data() + cut, m.data() + cut, w.data() + cut, n - cut, 1, n, 4, mode, FGMM_HOST) == FGMM_OK, "symbuf append 2");
  CHECK(fgmm_symbuf_flush(b, &got, &got_len) == FGMM_OK, "symbuf flush");
  CHECK(got_len == want_len && !memcmp(got, want, want_len), "buffered encoder: bytes differ from the oracle's");
  fgmm_free(got);
  fgmm_symbuf_destroy(b);
  fgo_free(want);
}

// the rows an item's bitstream codes, in coding order: (non-zero channel, position); parameters as (n, 4) rows, sigma clamped
static void coded_rows(const Item &it, std::vector<int32_t> &sym, std::vector<float> &s, std::vector<float> &m, std::vector<float> &w) {
  const size_t n_all = (size_t)it.M * it.hw;
  for (int c = 0; c < it.M; ++c) {
    bool nz = false;
    for (int64_t p = 0; p < it.hw; ++p) nz |= nearbyintf(it.y[(size_t)c * it.hw + p]) != 0.0f;
    if (!nz) continue;
    for (int64_t p = 0; p < it.hw; ++p) {
      const size_t at = (size_t)c * it.hw + p;
      sym.push_back((int32_t)nearbyintf(it.y[at]));
      for (int k = 0; k < 4; ++k) {
        s.push_back(fminf(fmaxf(it.sg[(size_t)k * n_all + at], 0.11f), 256.0f));
        m.push_back(it.mu[(size_t)k * n_all + at]);
        w.push_back(it.pi[(size_t)k * n_all + at]);
      }
    }
  }
}

static void set_opt(fgmm_ctx *ctx, const char *name, int64_t v) { CHECK(fgmm_ctx_set_option(ctx, name, v) == FGMM_OK, "option %s=%lld", name, (long long)v); }

// a caller's storage for the bitstreams of one call (fgmm_sink): a malloc of the exact size per item
struct Sunk {
  std::vector<void *> slot;
  std::vector<size_t> len;
  std::vector<std::atomic<int>> calls;
  int refuse = -1;
  std::thread::id owner = std::this_thread::get_id();
  void reset(int count) {
    for (void *p : slot) free(p);
    slot.assign((size_t)count, nullptr);
    len.assign((size_t)count, 0);
    calls = std::vector<std::atomic<int>>((size_t)count);
    refuse = -1;
  }
  ~Sunk() {
    for (void *p : slot) free(p);
  }
  static void *alloc(void *user, int item, size_t nbytes) { // (on the calling thread, inside the call)
    Sunk *s = static_cast<Sunk *>(user);
    if (item < 0 || (size_t)item >= s->slot.size() || std::this_thread::get_id() != s->owner) abort(); // (a wrong item; not the calling thread)
    if (s->calls[(size_t)item].fetch_add(1) != 0 || item == s->refuse) return nullptr;
    s->len[(size_t)item] = nbytes;
    return s->slot[(size_t)item] = malloc(nbytes ? nbytes : 1);
  }
};

int main(int argc, char **argv) {
  const double budget = argc > 1 ? atof(argv[1]) : 20.0;
  if (argc > 2) rng_state ^= (uint64_t)atoll(argv[2]) * 0x9E3779B97F4A7C15ull;
  fgmm_ctx *ctx = nullptr;
  if (getenv("FGMM_STRESS_NARROW_CALLER")) { // the creating thread keeps ONE CPU for itself before the context exists (a caller that has
    cpu_set_t have, one;                     // already moved to the CPUs it reserves: the workers' list is still the kernel's to grant)
    CHECK(sched_getaffinity(0, sizeof have, &have) == 0, "getaffinity");
    CPU_ZERO(&one);
    for (int c = 0; c < CPU_SETSIZE; ++c)
      if (CPU_ISSET(c, &have)) {
        CPU_SET(c, &one);
        break;
      }
    CHECK(sched_setaffinity(0, sizeof one, &one) == 0, "setaffinity");
  }
  CHECK(fgmm_ctx_create(0, 4, &ctx) == FGMM_OK, "ctx");
  if (const char *want = getenv("FGMM_STRESS_EXPECT_WORKER_CPUS")) { // (tests/test_fake_device_cpu.py: FGMM_WORKER_CPUS is honoured)
    char got[4096];
    CHECK(fgmm_ctx_worker_cpus(ctx, got, sizeof got) == FGMM_OK && !strcmp(got, want), "worker CPUs '%s', expected '%s'", got, want);
    CHECK(fgmm_ctx_set_threads(ctx, 6) == FGMM_OK, "threads"); // (a resized pool keeps the context's decision)
    // (the pool's threads name themselves when they start, after they have taken their mask: look until all six are there)
    int seen = 0;
    for (int attempt = 0; attempt < 400 && *want; ++attempt) {
      seen = 0;
      DIR *tasks = opendir("/proc/self/task");
      for (struct dirent *de; tasks && (de = readdir(tasks));) {
        const int tid = atoi(de->d_name);
        if (tid <= 0) continue;
        char path[64], line[256];
        snprintf(path, sizeof path, "/proc/self/task/%d/comm", tid);
        FILE *f = fopen(path, "r");
        if (!f) continue;
        const bool worker = fgets(line, sizeof line, f) && !strncmp(line, "fgmm-w", 6);
        fclose(f);
        if (!worker) continue;
        snprintf(path, sizeof path, "/proc/self/task/%d/status", tid);
        f = fopen(path, "r");
        CHECK(f != nullptr, "status of thread %d", tid);
        while (fgets(line, sizeof line, f))
          if (!strncmp(line, "Cpus_allowed_list:", 18)) {
            char *v = line + 18;
            while (*v == ' ' || *v == '\t') ++v;
            v[strcspn(v, "\n")] = 0;
            CHECK(!strcmp(v, want), "worker thread %d may run on '%s', expected '%s'", tid, v, want);
            ++seen;
          }
        fclose(f);
      }
      if (tasks) closedir(tasks);
      if (seen == 6) break;
      usleep(5000);
    }
    CHECK(!*want || seen == 6, "%d worker threads found, expected 6", seen);
    printf("worker CPUs: '%s' (%d threads checked)\n", got, seen);
  }
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(budget);
  long rounds = 0, refused = 0, streams = 0, symbols = 0, corrupted_same = 0, corrupted_refused = 0;
  static const int kThreads[] = {1, 2, 3, 5, 8, 16};
  while (std::chrono::steady_clock::now() < t_end) {
    // The first three rounds of every run are LARGE bitstreams under the AUTOMATIC piece plan (pieces = 0), which the random
    // items below - all under 65 536 latents, one piece - never reach: one bitstream of 131 072 latents on a pool of one worker and
    // on a pool of eight (one decoder: a small first piece, then growing ones - plan_pieces' lead_small), then two such bitstreams (the
    // general multi-piece plan, shrinking pieces).
    const bool big = rounds < 3;
    CHECK(fgmm_ctx_set_threads(ctx, big ? (rounds == 0 ? 1 : 8) : kThreads[rnd() % 6]) == FGMM_OK, "threads");
    set_opt(ctx, "pieces", big ? 0 : rnd() % 3 ? pick(1, 12) : 0);
    set_opt(ctx, "dec_first", pick(1, 6));
    set_opt(ctx, "ef_rows", pick(0, 2));
    set_opt(ctx, "ef_min", rnd() % 3 == 0 ? 14 : rnd() % 2 ? 33 : 49);
    set_opt(ctx, "enc_ways", pick(0, 4));
    set_opt(ctx, "enc_segs", pick(0, 2));
    set_opt(ctx, "scatter_rounds", rnd() % 2);
    set_opt(ctx, "ckpt_decode", pick(0, 2));
    set_opt(ctx, "gpu_decode", pick(0, 2));
    set_opt(ctx, "stage_max_mb", !big && rnd() % 4 == 0 ? 1 : 0); // a tiny staging budget: launches overflow and are re-run
    set_opt(ctx, "tab_cap_e", !big && rnd() % 5 == 0 ? 256 : 12288); // a tiny LDS budget: items take the generic two-pass kernels
    set_opt(ctx, "spin_lat", rnd() % 3 == 0 ? -1 : 400000);
    // one round in six: more bitstreams than workers under the automatic job plan with segments forced - the tables of the bitstreams
    // that are jobs of their own go in segments, the others whole (fgmm_encode.cpp: plan)
    const bool mixed_segs = !big && rounds % 6 == 5;
    if (mixed_segs) {
      CHECK(fgmm_ctx_set_threads(ctx, (int)pick(1, 3)) == FGMM_OK, "threads");
      set_opt(ctx, "enc_ways", 0);
      set_opt(ctx, "enc_segs", 2);
    }
    const int mode = (int)(rnd() % 3), clamp = 1;
    const int32_t stride = !big && rnd() % 2 ? (int32_t)(256 << (rnd() % 3)) : 0; // checkpointed streams
    const int count = big ? (rounds == 2 ? 2 : 1) : mixed_segs ? (int)pick(5, 12) : (int)pick(1, 12);
    std::vector<Item> its((size_t)count);
    std::vector<fgmm_item> fi((size_t)count);
    for (int i = 0; i < count; ++i) {
      static const int Ms[] = {1, 3, 8, 9, 17, 24};
      static const int64_t HWs[] = {1, 7, 64, 96, 192, 384};
      if (big) make_item(its[(size_t)i], 64, 2048, 0.0);
      else make_item(its[(size_t)i], Ms[rnd() % 6], rnd() % 40 == 0 ? 0 : HWs[rnd() % 6], rnd() % 5 == 0 ? 1.0 : 0.15); // (now and then an EMPTY item: hw = 0)
      Item &it = its[(size_t)i];
      // now and then an OUTLIER in a small item: a symbol far outside every component (bypass-coded, beyond int16: the wide symbol
      // paths), a half-width that needs 4- or 8-byte headers and the generic two-pass kernels
      if (it.hw > 0 && (size_t)it.M * it.hw <= 200 && rnd() % 6 == 0)
        it.y[rnd() % it.y.size()] = (float)((rnd() % 2 ? 1.0 : -1.0) * (double)pick(300, rnd() % 3 ? 3000 : 70000));
      fgmm_item &f = fi[(size_t)i];
      memset(&f, 0, sizeof f);
      f.y = it.y.data();
      f.params.scales = it.sg.data(), f.params.means = it.mu.data(), f.params.weights = it.pi.data();
      f.params.stride_k = (int64_t)it.M * it.hw, f.params.stride_c = it.hw, f.params.dtype = FGMM_F32;
      f.M = it.M, f.K = 4, f.hw = it.hw;
      f.yq_out = it.yq.data();
      f.zero_bitmap = it.zb.data();
      f.ckpt_stride = stride;
    }
    if (getenv("FGMM_STRESS_VERBOSE")) {
      fprintf(stderr, "[round %ld] threads %d mode %d stride %d count %d:", rounds, fgmm_ctx_threads(ctx), mode, stride, count);
      for (int i = 0; i < count; ++i) fprintf(stderr, " %dx%lld", its[(size_t)i].M, (long long)its[(size_t)i].hw);
      for (const char *nm : {"pieces", "dec_first", "ef_rows", "ef_min", "enc_ways", "enc_segs", "scatter_rounds", "ckpt_decode", "gpu_decode", "stage_max_mb", "tab_cap_e", "spin_lat"}) {
        int64_t v = 0;
        fgmm_ctx_get_option(ctx, nm, &v);
        fprintf(stderr, " %s=%lld", nm, (long long)v);
      }
      fprintf(stderr, "\n");
    }
    // One round in three through a SINK (fgmm_sink): the bitstreams go into storage handed out by the caller, of their exact size (a
    // malloc each: the sanitizer sees a byte too many), asked for once per item on the calling thread; now and then the sink
    // refuses an item first - the call must fail with FGMM_ERR_NOMEM and return no buffer - and the call is made again.
    Sunk sunk;
    const bool to_sink = rnd() % 3 == 0;
    if (to_sink) {
      sunk.reset(count);
      const fgmm_sink sink{&Sunk::alloc, &sunk};
      if (rnd() % 4 == 0) {
        sunk.refuse = (int)(rnd() % (uint64_t)count);
        CHECK(fgmm_gmc_compress_batch_to(ctx, nullptr, fi.data(), count, mode, clamp, &sink) == FGMM_ERR_NOMEM, "a sink that refuses item %d: the call went through", sunk.refuse);
        for (int i = 0; i < count; ++i) CHECK(!fi[(size_t)i].bytes && !fi[(size_t)i].ckpt, "a failed call returned a buffer (item %d)", i);
        sunk.reset(count);
      }
      CHECK(fgmm_gmc_compress_batch_to(ctx, nullptr, fi.data(), count, mode, clamp, &sink) == FGMM_OK, "compress_batch_to (count %d)", count);
      for (int i = 0; i < count; ++i)
        CHECK(sunk.calls[(size_t)i].load() == 1 && fi[(size_t)i].bytes == sunk.slot[(size_t)i] && fi[(size_t)i].bytes_len == sunk.len[(size_t)i],
              "sink: item %d asked %d times / bytes not where the sink said", i, sunk.calls[(size_t)i].load());
    } else {
      CHECK(fgmm_gmc_compress_batch(ctx, nullptr, fi.data(), count, mode, clamp) == FGMM_OK, "compress_batch (count %d)", count);
    }
    // ---- every bitstream against the oracle's encoder
    for (int i = 0; i < count; ++i) {
      Item &it = its[(size_t)i];
      const size_t n_all = (size_t)it.M * it.hw;
      std::vector<int32_t> sym;
      std::vector<float> s, m, w;
      for (int c = 0; c < it.M; ++c) {
        bool nz = false;
        for (int64_t p = 0; p < it.hw; ++p) {
          const float q = nearbyintf(it.y[(size_t)c * it.hw + p]);
          CHECK(it.yq[(size_t)c * it.hw + p] == q, "y_q item %d", i);
          nz |= q != 0.0f;
        }
        CHECK(it.zb[(size_t)c] == (nz ? 1 : 0), "zero bitmap item %d channel %d", i, c);
      }
      coded_rows(it, sym, s, m, w);
      uint8_t *want = nullptr;
      size_t want_len = 0;
      CHECK(fgo_encode_gmm(mode, (int64_t)sym.size(), sym.data(), s.data(), m.data(), w.data(), 4, 1, &want, &want_len, nullptr) == 0, "oracle");
      CHECK(fi[(size_t)i].bytes_len == want_len && !memcmp(fi[(size_t)i].bytes, want, want_len), "bitstream %d of %d differs from the oracle's (%zu / %zu bytes)", i,
            count, fi[(size_t)i].bytes_len, want_len);
      fgo_free(want);
      streams += 1, symbols += (long)sym.size();
    }
    // ---- decode: y_hat == y_q
    for (int i = 0; i < count; ++i) fi[(size_t)i].yq_out = its[(size_t)i].yhat.data();
    CHECK(fgmm_gmc_decompress_batch(ctx, nullptr, fi.data(), count, mode, clamp) == FGMM_OK, "decompress_batch (count %d)", count);
    for (int i = 0; i < count; ++i) CHECK(its[(size_t)i].yhat == its[(size_t)i].yq, "y_hat != y_q, item %d of %d", i, count);
    // ---- a truncated bitstream must be refused (and must not take the call down with it)
    if (rnd() % 3 == 0) {
      if (getenv("FGMM_STRESS_VERBOSE")) fprintf(stderr, "   truncated victim\n");
      int victim = -1;
      for (int i = 0; i < count; ++i)
        if (fi[(size_t)i].bytes_len > 64) victim = i;
      if (victim >= 0) {
        const fgmm_item keep = fi[(size_t)victim];
        fi[(size_t)victim].bytes_len = 16;
        if (rnd() % 2) fi[(size_t)victim].ckpt = nullptr, fi[(size_t)victim].n_ckpt = 0; // (with and without its notes)
        CHECK(fgmm_gmc_decompress_batch(ctx, nullptr, fi.data(), count, mode, clamp) != FGMM_OK, "a truncated bitstream decoded");
        fi[(size_t)victim] = keep;
        ++refused;
      }
    }
    // ---- a CORRUPTED bitstream (same length, bytes overwritten from somewhere on): no notes -> whatever the sequential decoder of the
    // reference makes of it (the oracle's float bisection on the same bytes), or the call refuses it when the desynchronised
    // decoder runs off the stream's end; with the good stream's notes -> the same: notes are verified, never trusted
    if (rnd() % 3 == 0) {
      int victim = -1;
      for (int i = 0; i < count; ++i)
        if (fi[(size_t)i].bytes_len > 64) victim = i;
      if (victim >= 0) {
        if (getenv("FGMM_STRESS_VERBOSE")) fprintf(stderr, "   corrupted victim %d\n", victim);
        Item &it = its[(size_t)victim];
        const fgmm_item keep = fi[(size_t)victim];
        std::vector<uint32_t> bad((keep.bytes_len + 3) / 4);
        memcpy(bad.data(), keep.bytes, keep.bytes_len);
        uint8_t *b8 = reinterpret_cast<uint8_t *>(bad.data());
        for (size_t k = (size_t)pick(0, (int64_t)keep.bytes_len - 1); k < keep.bytes_len; k += (size_t)pick(1, 9)) b8[k] = (uint8_t)rnd();
        fi[(size_t)victim].bytes = b8;
        if (rnd() % 2) fi[(size_t)victim].ckpt = nullptr, fi[(size_t)victim].n_ckpt = 0;
        std::fill(it.yhat.begin(), it.yhat.end(), -9.f);
        const int rc = fgmm_gmc_decompress_batch(ctx, nullptr, fi.data(), count, mode, clamp);
        if (rc == FGMM_OK) {
          std::vector<int32_t> sym, dec;
          std::vector<float> s, m, w;
          coded_rows(it, sym, s, m, w);
          dec.assign(sym.size(), 0);
          CHECK(fgo_decode_gmm(mode, b8, keep.bytes_len, (int64_t)sym.size(), s.data(), m.data(), w.data(), 4, 1, keep.abs_max + 1, dec.data()) == 0, "oracle decoder");
          size_t at = 0;
          for (int c = 0; c < it.M; ++c)
            for (int64_t p = 0; p < it.hw; ++p) {
              const float want = it.zb[(size_t)c] ? (float)dec[at++] : 0.0f;
              CHECK(it.yhat[(size_t)c * it.hw + p] == want, "corrupted bitstream: item %d channel %d position %lld decodes to %g, the oracle's decoder to %g", victim, c,
                    (long long)p, (double)it.yhat[(size_t)c * it.hw + p], (double)want);
            }
          ++corrupted_same;
        } else {
          CHECK(rc == FGMM_ERR_STREAM, "a corrupted bitstream: status %d", rc);
          ++corrupted_refused;
        }
        for (int i = 0; i < count; ++i)
          if (i != victim) CHECK(its[(size_t)i].yhat == its[(size_t)i].yq, "a corrupted neighbour changed item %d", i);
        fi[(size_t)victim] = keep;
      }
    }
    // ---- the reference's native boundary on one of the items' channels: (n, 4) rows in HOST memory, strided as the reference's views
    // (stride (1, n)), staged through the device by the library: RansEncoder / RansDecoder.*_with_indexes_gmm, and the buffered
    // form (BufferedRansEncoder: two appends, one flush == the concatenation's stream)
    if (getenv("FGMM_STRESS_VERBOSE")) fprintf(stderr, "   batch ok\n");
    if (rnd() % 2 == 0) {
      const Item &pick_ = its[(size_t)(rnd() % (uint64_t)count)];
      if (pick_.hw > 0) raw_boundary(ctx, pick_, mode);
    }
    // ---- buffers handed over in one native call (a binding that wants to own the bytes), the call log, trimming the context
    if (!to_sink && rnd() % 4 == 0) {
      std::vector<std::vector<uint8_t>> own((size_t)count);
      std::vector<void *> dst((size_t)count), src((size_t)count);
      std::vector<size_t> len((size_t)count);
      for (int i = 0; i < count; ++i) {
        own[(size_t)i].resize(fi[(size_t)i].bytes_len + 1);
        dst[(size_t)i] = own[(size_t)i].data(), src[(size_t)i] = fi[(size_t)i].bytes, len[(size_t)i] = fi[(size_t)i].bytes_len;
      }
      std::vector<std::vector<uint8_t>> want((size_t)count);
      for (int i = 0; i < count; ++i) want[(size_t)i].assign(fi[(size_t)i].bytes, fi[(size_t)i].bytes + fi[(size_t)i].bytes_len);
      CHECK(fgmm_ctx_take_buffers(ctx, dst.data(), src.data(), len.data(), count) == FGMM_OK, "take_buffers");
      for (int i = 0; i < count; ++i) {
        CHECK(!memcmp(own[(size_t)i].data(), want[(size_t)i].data(), want[(size_t)i].size()), "take_buffers item %d", i);
        fi[(size_t)i].bytes = nullptr; // (released by the library)
      }
      fgmm_call_marks log[8];
      int n_log = 0;
      CHECK(fgmm_ctx_call_log(ctx, log, 8, &n_log) == FGMM_OK && n_log >= 2, "call log");
      for (int k = 0; k < n_log; ++k) CHECK(log[k].count >= 1 && log[k].ms[5] >= log[k].ms[0] && log[k].ms[0] >= 0, "call log entry %d", k);
      if (rnd() % 2) CHECK(fgmm_ctx_trim(ctx) == FGMM_OK, "trim");
    }
    for (auto &f : fi) {
      if (!to_sink) fgmm_free(f.bytes); // (a sink's storage is the caller's: released with `sunk`)
      fgmm_free(f.ckpt);
    }
    ++rounds;
  }
  fgmm_ctx_destroy(ctx);
  printf("stress on the fake device: %ld batches, %ld bitstreams == oracle, %ld symbols, %ld truncated batches refused, %ld corrupted bitstreams == oracle's decoder (%ld ran off the end: refused)\n", rounds, streams, symbols, refused,
         corrupted_same, corrupted_refused);
  return 0;
}

// fgmm_encode.cpp — the batched encode: float work enqueued for ALL bitstreams of a call (quant_stats + symtab kernels, batched
// over items through device descriptors), the tables back by pinned async copies, one host rANS job per bitstream on the worker
// pool.  What it replaces: BufferedRansEncoder::encode_with_indexes_gmm + flush, rans_interface.cpp:458-585, and the tensor
// preparation of GaussianMixtureConditional.compress above it (entropy_models.py:833-867).
//   plan()            workspace layout, the bitstreams' order (largest first), whole tables or tail-first segments
//   enqueue()         descriptors, kernels, the copies and their events
//   side_info()       per bitstream: abs_max, zero bitmap, bypass count, wide symbols (host, after the kernels)
//   submit_jobs()     jobs of one bitstream - or of `ways` bitstreams coded in turn - onto the pool
#include <atomic>
#include <memory>

#include "fgmm_ctx.h"

namespace fgmm {
namespace {

// can item use the 16-B-per-lane symtab kernel?
bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
bool enc_vec4_ok(const EncDesc &d, bool f16) {
  const uintptr_t pm = f16 ? 7 : 15; // 4 parameters per load: 8 B (fp16) or 16 B (fp32)
  auto al = [pm](const void *p) { return (reinterpret_cast<uintptr_t>(p) & pm) == 0; };
  return d.stride_p == 1 && (d.hw & 3) == 0 && (d.stride_c & 3) == 0 && (d.stride_k & 3) == 0 && al(d.scales) && al(d.means) &&
         al(d.weights) && (d.y ? aligned16(d.y) : aligned16(d.sym)) && aligned16(d.packed);
}

// segmented tables: an encoder asks for a segment before it enters it and SLEEPS on the event of the copy that carries it
struct SegWaitArg {
  dev::Event *ev;       // the copy groups' events
  const int32_t *group; // EncItem::seg_group
  Trace *tr;
  double waited, last; // the time spent waiting, per job; when the last wait returned
};
int seg_wait(void *arg, int sg) {
  SegWaitArg *a = static_cast<SegWaitArg *>(arg);
  const double t0 = a->tr->ms();
  const bool ok = dev::event_sync(a->ev[a->group[sg]]) == 0;
  a->last = a->tr->ms();
  a->waited += a->last - t0;
  return ok ? FGMM_OK : FGMM_ERR_HIP;
}

// The sink's desk (include/flashgmm_amd.h: fgmm_sink).  The caller's alloc() hands out storage of a language runtime - Python `bytes`
// objects, under the interpreter's lock - so it is called on the CALLING thread, which has nothing else to do while the encoders run:
// an encoder whose bitstream is complete leaves (item, size) at the desk and waits for the address - a few microseconds of polling,
// then asleep - the calling thread serves the requests in the order they come and stays awake between requests that follow each
// other closely (the bitstreams of a call finish within tens of microseconds of each other).  Measured against the alternative, the
// workers calling alloc() themselves: 48 threads that want the interpreter's lock at once are woken one after the other, 0.1 ms at
// the end of a Kodak call (profiles/r06_README.md).
struct SinkDesk {
  explicit SinkDesk(const fgmm_sink *u, int count) : user(u), need((size_t)count, 0), ans(new Ans[(size_t)count]) {
    inner.alloc = &SinkDesk::ask;
    inner.user = this;
  }
  struct Ans {
    std::atomic<int> ready{0};
    void *p = nullptr;
  };
  const fgmm_sink *user;
  fgmm_sink inner; // what the encoders are given
  std::mutex m;
  std::condition_variable cv_req, cv_ans;
  std::vector<int> asked; // items whose bitstream is complete and waits for its storage
  std::vector<size_t> need;
  std::unique_ptr<Ans[]> ans;
  std::atomic<long> n_asked{0};
  std::atomic<int> jobs_left{0}; // (written under the lock; read without it by the calling thread's polling)
  int sleepers = 0;
  bool closed = false;

  static double us_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  }
  // an encoder (a host worker): where do item's nbytes go?  NULL: nowhere, the call fails
  static void *ask(void *self, int item, size_t nbytes) {
    SinkDesk *d = static_cast<SinkDesk *>(self);
    {
      std::lock_guard<std::mutex> l(d->m);
      if (d->closed) return nullptr;
      d->need[(size_t)item] = nbytes;
      d->asked.push_back(item);
      d->n_asked.fetch_add(1, std::memory_order_release);
    }
    d->cv_req.notify_one();
    Ans &a = d->ans[(size_t)item];
    const auto t0 = std::chrono::steady_clock::now();
    do {
      for (int k = 0; k < 32; ++k) {
        if (a.ready.load(std::memory_order_acquire)) return a.p;
        __builtin_ia32_pause();
      }
    } while (us_since(t0) < 25.0);
    std::unique_lock<std::mutex> l(d->m);
    ++d->sleepers;
    d->cv_ans.wait(l, [&] { return a.ready.load(std::memory_order_acquire) != 0; });
    --d->sleepers;
    return a.p;
  }
  void job_out() { // (before the job is submitted)
    std::lock_guard<std::mutex> l(m);
    ++jobs_left;
  }
  void job_done() {
    {
      std::lock_guard<std::mutex> l(m);
      --jobs_left;
    }
    cv_req.notify_one();
  }
  // the calling thread, until every job has ended
  void serve() {
    long served = 0;
    std::unique_lock<std::mutex> l(m);
    std::vector<int> take;
    for (;;) {
      if (asked.empty() && jobs_left > 0) {
        bool more = false;
        if (served > 0) { // the next request is a few microseconds away, as a rule: stay awake for it
          l.unlock();
          const auto t0 = std::chrono::steady_clock::now();
          do {
            for (int k = 0; k < 32 && !more; ++k) {
              more = n_asked.load(std::memory_order_acquire) != served || jobs_left.load(std::memory_order_acquire) == 0;
              __builtin_ia32_pause();
            }
          } while (!more && us_since(t0) < 60.0);
          l.lock();
        }
        if (!more) cv_req.wait(l, [&] { return !asked.empty() || jobs_left == 0; });
      }
      if (asked.empty()) {
        if (jobs_left == 0) return;
        continue;
      }
      take.clear();
      take.swap(asked);
      l.unlock();
      for (int item : take) {
        Ans &a = ans[(size_t)item];
        a.p = user->alloc(user->user, item, need[(size_t)item]);
        a.ready.store(1, std::memory_order_release);
      }
      served += (long)take.size();
      l.lock();
      if (sleepers) cv_ans.notify_all();
    }
  }
  // nobody serves any more (the call is on its way out, perhaps early): whoever asks or has asked gets no storage
  void close() {
    {
      std::lock_guard<std::mutex> l(m);
      closed = true;
      for (int item : asked) ans[(size_t)item].ready.store(1, std::memory_order_release); // (p stays NULL)
      asked.clear();
    }
    cv_ans.notify_all();
  }
};
struct DeskCloser {
  SinkDesk *d;
  ~DeskCloser() {
    if (d) d->close();
  }
};

struct EncodeCall {
  fgmm_ctx *ctx;
  dev::Stream stream;
  std::vector<EncItem> &items;
  int mode, count;
  const HeadW *head; // the parameters come out of the head's matrix product (fgmm_head.hip) instead of planes
  const fgmm_sink *sink; // the bitstreams go into the caller's storage (include/flashgmm_amd.h: fgmm_sink)
  std::unique_ptr<SinkDesk> desk; // ... asked for through the desk when the encoders run on the pool
  Trace tr;
  // plan
  Arena ar;
  size_t o_descs = 0, o_small = 0, small_bytes = 0;
  int M_max = 0;
  int64_t hw_max = 0;
  std::vector<int> order; // the bitstreams LARGEST FIRST (equal sizes: as given)
  int enc_T = 1, enc_ways = 1;
  bool segmented = false; // every table lies in segments (a worker per bitstream)
  int n_segd = 0;         // tables that do: all of them, or - more bitstreams than workers - those of the bitstreams that are a job of their own
  bool segd(const EncItem &it) const { return it.n_seg > 0; }
  // copies
  std::vector<int> group_of;
  int n_groups = 0, n_seg_groups = 0; // of whole tables (ctx->events, waited for by the calling thread) / of segments (ctx->sleep_events, by the encoders)
  size_t ev_meta = 0;
  // host side
  std::vector<std::vector<int32_t>> wide_syms; // only for bypass symbols beyond int16 (rare)
  std::vector<SegWaitArg> seg_args;
  std::vector<int> job_first, job_last; // by position in `order`
  std::vector<EncItem *> job_items;
  double marks[5] = {0, 0, 0, 0, 0}; // the call log: enqueued | kernels + side information here | jobs out | last table (segment) seen landed | last job done

  EncodeCall(fgmm_ctx *c, dev::Stream s, std::vector<EncItem> &it, int m, const HeadW *h, const fgmm_sink *k)
      : ctx(c), stream(s), items(it), mode(m), count((int)it.size()), head(h), sink(k), tr("encode", (int)c->opt.trace) {
    if (sink && count > 1) desk.reset(new SinkDesk(sink, count)); // (a one-item call codes on the calling thread: it asks the sink itself)
  }
  BytesTo to(const EncItem &e) const { return BytesTo{desk ? &desk->inner : sink, (int)(&e - items.data())}; }

  size_t table_bytes(const EncItem &it) const { return sizeof(uint32_t) * (size_t)it.M * (size_t)it.hw; }

  // ---- workspace: [descs][small: per item min|max|nz|list|meta][tables: per item packed, whole or in segments] -----------------
  int plan() {
    o_descs = ar.take(sizeof(EncDesc) * (size_t)count);
    o_small = ar.take(0);
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
    small_bytes = ar.off - o_small;
    // The bitstreams in the order of their size, LARGEST FIRST: their tables cross PCIe in that order and their host jobs are handed
    // out in that order - the long jobs start first and the short ones fill the workers' tails (ELIC's groups differ 12x in size)
    order.resize((size_t)count);
    for (int i = 0; i < count; ++i) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return (int64_t)items[a].M * items[a].hw > (int64_t)items[b].M * items[b].hw; });
    // A bitstream is encoded BACKWARDS (rANS), so its encoder needs the END of its table first - and with whole tables crossing PCIe
    // one after another the call ends a whole job (0.3-0.45 ms for a Kodak half) after the last table has landed.  When every
    // bitstream has a worker of its own, the tables are laid out in up to four SEGMENTS of compact channels each, LAST SEGMENT FIRST
    // across all bitstreams: the encoders start on the tails after an eighth of the transfer and follow the landing.
    enc_T = std::max(ctx->pool->size(), 1);
    // automatic: pairs as soon as there are more bitstreams than workers (48 bitstreams on 16 workers: pairs 1.78 ms per call,
    // threes 2.28, workers pulling one or two as the tables land 1.95-2.04)
    enc_ways = ctx->opt.enc_ways > 0 ? (int)ctx->opt.enc_ways : (count > enc_T ? 2 : 1);
    segmented = ctx->opt.enc_segs != 0 && count >= 2 && count <= enc_T && enc_ways == 1;
    size_t tables = 0;
    for (auto &it : items) {
      tables += table_bytes(it);
      segmented = segmented && it.latent && !it.symbuf && it.M >= 2 * kEncSegs && it.hw > 0;
    }
    segmented = segmented && (tables >= ((size_t)4 << 20) || ctx->opt.enc_segs == 2); // (smaller calls: the transfer is not what they wait for; 2: tests)
    auto in_segments = [&](EncItem &it) {
      it.cps = (it.M + kEncSegs - 1) / kEncSegs;
      it.n_seg = (it.M + it.cps - 1) / it.cps;
      ++n_segd;
    };
    if (segmented) {
      for (auto &it : items) in_segments(it);
    } else if (ctx->opt.enc_segs != 0 && ctx->opt.enc_ways == 0 && count > enc_T) {
      // More bitstreams than workers (ELIC: 160 per call of sixteen 4K images, the largest sixteen 3.5 M symbols = 12 ms of encoding
      // each): the bitstreams that are a worker's fair share by themselves (plan_jobs: jobs of their own) are the call's critical path -
      // the last of their whole tables landed 6 ms into the call.  Their tables go in segments too, tail first across all of them and
      // before every other table: their encoders start after a quarter of their transfer and follow the landing.
      int64_t n_total = 0;
      for (auto &it : items) n_total += (int64_t)it.M * it.hw;
      const int64_t big = n_total / (2 * (int64_t)enc_T) + 1;
      for (auto &it : items)
        if (it.latent && !it.symbuf && it.M >= 2 * kEncSegs && it.hw > 0 && (int64_t)it.M * it.hw >= big &&
            (table_bytes(it) >= ((size_t)1 << 20) || ctx->opt.enc_segs == 2))
          in_segments(it);
    }
    if (n_segd) {
      for (int sg = kEncSegs - 1; sg >= 0; --sg)
        for (int i : order) {
          EncItem &it = items[i];
          if (sg >= it.n_seg) continue;
          const int32_t ch = std::min(it.M, (sg + 1) * it.cps) - sg * it.cps;
          it.o_seg[sg] = ar.take(sizeof(uint32_t) * (size_t)ch * (size_t)it.hw + 64);
        }
      for (auto &it : items)
        if (segd(it)) it.o_packed = it.o_seg[0];
    }
    for (int i : order)
      if (!segd(items[i])) items[i].o_packed = ar.take(table_bytes(items[i]) + 64);
    int rc;
    if ((rc = ctx->ensure_device(ar.off)) || (rc = ctx->ensure_host(ar.off)) || (rc = ctx->ensure_events((size_t)count + 17, 16))) return rc;
    ev_meta = (size_t)count + 16; // (the copy groups of the tables use the events before it: at most count, or nine)
    return FGMM_OK;
  }

  void fill_desc(int i, EncDesc &d) const {
    const EncItem &it = items[i];
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
    d.x = it.x;
    d.meta_slots = (uint32_t)it.meta_count;
    d.yq = it.yq;
    d.chan_min = reinterpret_cast<float *>(ctx->d_ws + it.o_min);
    d.chan_max = reinterpret_cast<float *>(ctx->d_ws + it.o_max);
    d.chan_nz = it.latent ? reinterpret_cast<int32_t *>(ctx->d_ws + it.o_nz) : nullptr;
    d.chan_list = it.latent ? reinterpret_cast<int32_t *>(ctx->d_ws + it.o_list) : nullptr;
    d.packed = reinterpret_cast<uint32_t *>(ctx->d_ws + it.o_packed);
    d.seg_b[0] = d.seg_b[1] = d.seg_b[2] = INT32_MAX;
    d.packed_seg[0] = d.packed;
    if (segd(it)) {
      d.cps = it.cps;
      for (int sg = 0; sg < it.n_seg; ++sg) {
        d.packed_seg[sg] = reinterpret_cast<uint32_t *>(ctx->d_ws + it.o_seg[sg]);
        if (sg + 1 < it.n_seg) d.seg_b[sg] = (sg + 1) * it.cps;
      }
    }
    d.meta = reinterpret_cast<uint32_t *>(ctx->d_ws + it.o_meta);
  }

  // ---- descriptors, kernels -------------------------------------------------------------------------------------------------------
  int enqueue_kernels() {
    EncDesc *hd = reinterpret_cast<EncDesc *>(ctx->h_ws + o_descs);
    bool vec4 = true, vec8 = ctx->opt.enc_vec == 0 || ctx->opt.enc_vec == 8, any_y = false;
    for (int i = 0; i < count; ++i) {
      fill_desc(i, hd[i]);
      const EncDesc &d = hd[i];
      vec4 = vec4 && enc_vec4_ok(d, items[i].prm.dtype == FGMM_F16);
      // 8 positions per lane for fp16 planes (one 16-byte load per plane: +4-5 % over 8-byte loads on ELIC-4K batches,
      // profiles/r05_symtab_fp16_vec8_ab.txt): everything 16-byte aligned, rows of 8
      vec8 = vec8 && items[i].prm.dtype == FGMM_F16 && (d.hw & 7) == 0 && (d.stride_c & 7) == 0 && (d.stride_k & 7) == 0 && aligned16(d.scales) &&
             aligned16(d.means) && aligned16(d.weights);
      any_y = any_y || items[i].latent;
    }
    DEV_TRY(dev::copy_async(ctx->d_ws + o_descs, hd, sizeof(EncDesc) * (size_t)count, dev::kH2D, stream));
    DEV_TRY(dev::memset_async(ctx->d_ws + o_small, 0, small_bytes, stream));
    const EncDesc *dd = reinterpret_cast<const EncDesc *>(ctx->d_ws + o_descs);
    int rc;
    // a batch is homogeneous by construction: all latent-layout items (y given) or one raw (n,K) item
    if (any_y) {
      if ((rc = ctx->prof_begin(2, stream))) return rc;
      LAUNCH_TRY(launch_quant_stats(dd, count, M_max, stream));
      if ((rc = ctx->prof_end(2, stream))) return rc;
    }
    if ((rc = ctx->prof_begin(0, stream))) return rc;
    if (head) { // the fused head: matrix product + table entries in one kernel, no parameter planes
      bool vec = true;
      for (auto &it : items) vec = vec && (it.hw & 3) == 0 && aligned16(it.x);
      if (head->arith == FGMM_HEAD_BF16X6) LAUNCH_TRY(launch_head16_symtab(dd, *head, count, M_max, hw_max, mode, items[0].clamp != 0, stream));
      else LAUNCH_TRY(launch_head_symtab(dd, *head, count, M_max, hw_max, mode, items[0].clamp != 0, vec, stream));
      return ctx->prof_end(0, stream);
    }
    const int vec = vec4 ? (ctx->opt.enc_vec == 1 ? 1 : ctx->opt.enc_vec == 2 ? 2 : vec8 ? 8 : 4) : 1; // option "enc_vec" = 1, 2, 4: A/B narrower loads
    int64_t n_max = 0;
    bool linear = ctx->opt.enc_linear != 0; // option "enc_linear" = 0: A/B the per-channel grid
    for (auto &it : items) {
      n_max = std::max(n_max, (int64_t)it.M * it.hw);
      linear = linear && it.hw % (64 * vec) == 0;
    }
    LAUNCH_TRY(launch_symtab(dd, count, M_max, hw_max, n_max, linear, mode, vec, items[0].clamp != 0, items[0].prm.dtype == FGMM_F16, stream));
    return ctx->prof_end(0, stream);
  }

  // ---- tables back to the host: the small region first, then a handful of large copies, an event each --------------------------
  int enqueue_copies() {
    DEV_TRY(dev::copy_async(ctx->h_ws + o_small, ctx->d_ws + o_small, small_bytes, dev::kD2H, stream));
    DEV_TRY(dev::event_record(ctx->events[ev_meta], stream));
    group_of.assign((size_t)count, 0);
    if (n_segd) {
      // the segments in the order they were laid out (tails of all segmented bitstreams first), in about eight copies
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
      for (size_t c0 = 0; c0 < chunks.size(); ++n_seg_groups) {
        size_t c1 = c0, got = 0;
        do {
          got += chunks[c1].end - chunks[c1].beg;
          ++c1;
        } while (c1 < chunks.size() && got < per_group);
        const size_t beg = chunks[c0].beg, end = chunks[c1 - 1].end; // (laid out in this order: one contiguous range)
        if (end > beg) DEV_TRY(dev::copy_async(ctx->h_ws + beg, ctx->d_ws + beg, end - beg, dev::kD2H, stream));
        DEV_TRY(dev::event_record(ctx->sleep_events[(size_t)n_seg_groups], stream));
        for (size_t c = c0; c < c1; ++c) items[chunks[c].item].seg_group[chunks[c].sg] = n_seg_groups;
        c0 = c1;
      }
    }
    // the whole tables, in their order (after the segments)
    std::vector<int> rest;
    size_t tables = 0;
    for (int i : order)
      if (!segd(items[i])) rest.push_back(i), tables += table_bytes(items[i]);
    const int n_rest = (int)rest.size();
    const size_t per_group = n_rest >= 16 ? tables / 6 + 1 : 0; // (fewer than 16 bitstreams: a copy each)
    for (int p0 = 0; p0 < n_rest; ++n_groups) {
      int p1 = p0;
      size_t got = 0;
      do {
        got += table_bytes(items[rest[(size_t)p1]]);
        ++p1;
      } while (p1 < n_rest && got < per_group);
      const EncItem &a = items[rest[(size_t)p0]], &b = items[rest[(size_t)p1 - 1]];
      const size_t beg = a.o_packed, end = b.o_packed + table_bytes(b);
      if (end > beg) DEV_TRY(dev::copy_async(ctx->h_ws + beg, ctx->d_ws + beg, end - beg, dev::kD2H, stream));
      DEV_TRY(dev::event_record(ctx->events[(size_t)n_groups], stream));
      for (int p = p0; p < p1; ++p) group_of[(size_t)rest[(size_t)p]] = n_groups;
      p0 = p1;
    }
    return FGMM_OK;
  }

  // ---- per bitstream, on the host, once the small region is here: side information + what its job needs -----------------------
  int side_info(int i) {
    EncItem &it = items[i];
    int64_t n = (int64_t)it.M * it.hw;
    unsigned long long n_bypass = 0;
    for (size_t k = 0; k < it.meta_count; ++k) n_bypass += reinterpret_cast<const uint32_t *>(ctx->h_ws + it.o_meta)[k];
    const int32_t *syms_for_bypass = it.sym_host;
    if (it.latent) {
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
        // a bypassed symbol may not fit the 16 bits the table carries: fetch the GPU-rounded latents (y_q, written by
        // quant_stats_kernel) and convert them to the int32 symbols - an integer conversion, no arithmetic.  Without a y_q buffer
        // the raw latents are fetched and rounded to nearest-even here (rintf semantics).
        std::vector<float> yv((size_t)it.M * it.hw);
        DEV_TRY(dev::copy_sync(yv.data(), it.yq ? it.yq : it.y, sizeof(float) * yv.size(), dev::kD2H));
        wide_syms[(size_t)i].reserve((size_t)n);
        for (int c = 0; c < it.M; ++c)
          if (nz[c])
            for (int64_t p = 0; p < it.hw; ++p) {
              const float v = yv[(size_t)c * it.hw + p];
              wide_syms[(size_t)i].push_back((int32_t)(it.yq ? v : nearbyintf(v)));
            }
        syms_for_bypass = wide_syms[(size_t)i].data();
      }
    } else if (n_bypass && !it.sym_host) {
      wide_syms[(size_t)i].resize((size_t)n);
      DEV_TRY(dev::copy_sync(wide_syms[(size_t)i].data(), it.sym_dev, sizeof(int32_t) * (size_t)n, dev::kD2H));
      syms_for_bypass = wide_syms[(size_t)i].data();
    }
    it.job_syms = syms_for_bypass;
    it.job_n = n;
    it.job_bypass = (int64_t)n_bypass;
    return FGMM_OK;
  }

  // jobs: runs of up to `enc_ways` bitstreams adjacent in `order` (similar sizes), coded in turn by one worker; a bitstream that is
  // a worker's fair share by itself (>= 1 / (2 * workers) of the call) is a job of its own - sixteen large pairs on eight workers
  // would leave the other eight idle
  void plan_jobs() {
    job_first.assign((size_t)count, 0);
    job_last.assign((size_t)count, 0);
    int64_t n_total = 0;
    for (auto &it : items) n_total += (int64_t)it.M * it.hw;
    const int64_t big = ctx->opt.enc_ways > 0 ? INT64_MAX : n_total / (2 * (int64_t)enc_T) + 1;
    // ... and the streams that would form a last, half-empty round of pairs (48 on 16 workers: 16 pairs, then 8 pairs on 8 workers
    // while 8 idle) are singles instead: every worker gets a pair and a single
    const int tail = ctx->opt.enc_ways > 0 || enc_ways != 2 ? 0 : count % (2 * enc_T);
    const int first_single = tail <= enc_T ? count - tail : count;
    auto small = [&](int pos) {
      const EncItem &e = items[order[(size_t)pos]];
      return !e.symbuf && !segd(e) && (int64_t)e.M * e.hw < big;
    };
    for (int p = 0; p < count;) {
      int q = p + 1;
      if (small(p) && p < first_single)
        while (q < first_single && q - p < enc_ways && small(q)) ++q;
      for (int r = p; r < q; ++r) job_first[(size_t)r] = p, job_last[(size_t)r] = q - 1;
      p = q;
    }
    job_items.resize((size_t)count);
    for (int p = 0; p < count; ++p) job_items[(size_t)p] = &items[order[(size_t)p]];
  }

  static int alloc_ckpt(EncItem &e, int64_t stride, fgmm_ckpt **out) {
    *out = nullptr;
    const int64_t n_ck = stride > 0 && e.job_n > 0 ? (e.job_n - 1) / stride : 0;
    if (n_ck <= 0) return FGMM_OK;
    e.ckpt = static_cast<fgmm_ckpt *>(malloc(sizeof(fgmm_ckpt) * (size_t)n_ck));
    if (!e.ckpt) return FGMM_ERR_NOMEM;
    e.n_ckpt = n_ck;
    *out = e.ckpt;
    return FGMM_OK;
  }
  static void drop_ckpt(EncItem &e) {
    free(e.ckpt);
    e.ckpt = nullptr;
    e.n_ckpt = 0;
  }

  // one bitstream whose table lies in segments that land tail first: the encoder asks for each before it enters it
  void run_segmented(EncItem &e, SegWaitArg *arg) {
    e.t_start = tr.ms();
    SegTable t;
    t.n_seg = e.n_seg;
    t.seg_len = (int64_t)e.cps * e.hw;
    for (int sg = 0; sg < kEncSegs; ++sg) t.seg[sg] = sg < e.n_seg ? reinterpret_cast<const uint32_t *>(ctx->h_ws + e.o_seg[sg]) : nullptr;
    t.wait = seg_wait;
    t.arg = arg;
    fgmm_ckpt *ck = nullptr;
    int rc = alloc_ckpt(e, e.ckpt_stride, &ck);
    if (rc == FGMM_OK) rc = rans_encode_symtab_segs(t, e.job_syms, e.job_n, e.job_bypass, &e.bytes, &e.bytes_len, ck ? e.ckpt_stride : 0, ck, to(e));
    if (rc != FGMM_OK) drop_ckpt(e);
    e.status = rc;
    e.t_end = tr.ms();
    e.t_waited = arg->waited;
    e.t_lastland = arg->last;
  }

  // `n_in` bitstreams (whole tables on the host), coded in turn symbol by symbol
  void run_ways(EncItem *const *first, int n_in) {
    const double t_start = tr.ms();
    if (n_in == 1 && first[0]->symbuf) {
      EncItem &e = *first[0];
      e.t_start = t_start;
      e.status = fgmm_symbuf_append_symtab(e.symbuf, reinterpret_cast<const uint32_t *>(ctx->h_ws + e.o_packed), e.job_syms, e.job_n);
      e.t_end = tr.ms();
      return;
    }
    const uint32_t *packed[kMaxEncWays];
    const int32_t *syms[kMaxEncWays];
    int64_t n[kMaxEncWays], nb[kMaxEncWays];
    uint8_t **out[kMaxEncWays];
    size_t *len[kMaxEncWays];
    fgmm_ckpt *ck[kMaxEncWays];
    BytesTo dst[kMaxEncWays];
    const int64_t stride = first[0]->ckpt_stride; // one stride per call (checked at the boundary)
    int rc = FGMM_OK;
    for (int q = 0; q < n_in; ++q) {
      EncItem &e = *first[q];
      e.t_start = t_start;
      packed[q] = reinterpret_cast<const uint32_t *>(ctx->h_ws + e.o_packed);
      syms[q] = e.job_syms;
      n[q] = e.job_n;
      nb[q] = e.job_bypass;
      out[q] = &e.bytes;
      len[q] = &e.bytes_len;
      dst[q] = to(e);
      const int rq = alloc_ckpt(e, stride, &ck[q]);
      if (rq != FGMM_OK) rc = rq;
    }
    if (rc == FGMM_OK) rc = rans_encode_symtab_ways(n_in, packed, syms, n, nb, out, len, stride, ck, dst);
    const double t_end = tr.ms();
    for (int q = 0; q < n_in; ++q) {
      if (rc != FGMM_OK) drop_ckpt(*first[q]);
      first[q]->status = rc;
      first[q]->t_end = t_end;
    }
  }

  // per item side information, then one rANS job per item (or per run of items) as its tables are known to be on their way
  int submit_jobs() {
    wide_syms.resize((size_t)count);
    seg_args.resize((size_t)count);
    plan_jobs();
    for (int pos = 0; pos < count; ++pos) {
      const int i = order[(size_t)pos];
      int rc = side_info(i);
      if (rc) return rc;
      // More bitstreams than workers: the members of a job go to one worker, coded in turn symbol by symbol - two dependency chains
      // share a core: 1.5 instead of 2.4 ns/symbol.  The job is submitted with its last member (the later copies hold the smaller tables)
      if (pos < job_last[(size_t)pos]) continue;
      const int g_begin = job_first[(size_t)pos], n_in = pos - g_begin + 1;
      if (!segd(items[i])) {
        int last_group = 0;
        for (int r = g_begin; r <= pos; ++r) last_group = std::max(last_group, group_of[(size_t)order[(size_t)r]]);
        DEV_TRY(dev::event_sync(ctx->events[(size_t)last_group])); // copies complete in the order they were queued
        marks[3] = tr.ms();
      }
      EncItem *const *first = &job_items[(size_t)g_begin];
      const double t_sub = tr.ms();
      for (int q = 0; q < n_in; ++q) first[q]->t_sub = t_sub;
      std::function<void()> job;
      if (segd(items[i])) { // (a job of its own: plan_jobs)
        seg_args[(size_t)i] = SegWaitArg{ctx->sleep_events.data(), items[i].seg_group, &tr, 0.0, 0.0};
        SegWaitArg *arg = &seg_args[(size_t)i];
        job = [this, first, arg] { run_segmented(*first[0], arg); };
      } else {
        job = [this, first, n_in] { run_ways(first, n_in); };
      }
      if (count == 1) {
        job();
      } else if (desk) {
        desk->job_out();
        ctx->pool->submit([this, job = std::move(job)] {
          job();
          desk->job_done();
        });
      } else {
        ctx->pool->submit(std::move(job));
      }
    }
    return FGMM_OK;
  }

  int run() {
    int rc;
    if ((rc = plan()) || (rc = enqueue_kernels()) || (rc = enqueue_copies())) return rc;
    tr.mark("enqueued");
    if (tr.level > 0 && n_segd && !segmented) fprintf(stderr, "[fgmm encode]   %d of the %d tables in segments (the bitstreams that are jobs of their own)\n", n_segd, count);
    marks[0] = tr.ms();
    DEV_TRY(dev::event_sync(ctx->events[ev_meta]));
    tr.mark("kernels + meta landed");
    marks[1] = tr.ms();
    ctx->stat[0] = 0;
    for (auto &it : items) ctx->stat[0] += table_bytes(it);
    {
      PoolDrain drain{ctx->pool}; // on any return: wait for every job before the objects they use go away
      DeskCloser closer{desk.get()}; // ... which none of them does at a desk nobody serves any more (destroyed first)
      if ((rc = submit_jobs())) return rc;
      tr.mark(segmented ? "jobs out" : "all tables landed, jobs out");
      marks[2] = tr.ms();
      if (desk) desk->serve();
    }
    // Segmented tables: an encoder waits only for the segments it enters - a bitstream without a coded symbol enters none - so the last
    // copy group may still be in flight here, and the next call writes the pinned workspace it lands in (found on the fake device under
    // AddressSanitizer, round 5: descriptors of the following decode call overwritten by a stale table copy).  Copies complete in the
    // order they were queued: the last group's event covers them all; normally it has long fired.
    if (n_seg_groups > 0) DEV_TRY(dev::event_sync(ctx->sleep_events[(size_t)n_seg_groups - 1]));
    tr.mark("host rANS done");
    double busy = 0, wait = 0;
    for (auto &it : items) {
      marks[3] = std::max(marks[3], it.t_lastland);
      marks[4] = std::max(marks[4], it.t_end);
      busy += it.t_end - it.t_start - it.t_waited;
      wait += it.t_waited;
    }
    ctx->log_call(0, count, tr, marks, busy, wait);
    if (tr.level > 1)
      for (int i = 0; i < count; ++i)
        fprintf(stderr, "[fgmm encode]   item %2d  submitted %7.3f  job %7.3f .. %7.3f  (%.3f ms, %.3f of it waiting for its table's segments)\n", i,
                items[i].t_sub, items[i].t_start, items[i].t_end, items[i].t_end - items[i].t_start, items[i].t_waited);
    for (auto &it : items)
      if (it.status) return fail(it.status, "host rANS encode failed (%d)", it.status);
    return FGMM_OK;
  }
};

} // namespace

int encode_batch(fgmm_ctx *ctx, dev::Stream stream, std::vector<EncItem> &items, int mode, const HeadW *head, const fgmm_sink *sink) {
  if (items.empty()) return FGMM_OK;
  EncodeCall call(ctx, stream, items, mode, head, sink);
  const int rc = call.run();
  if (rc != FGMM_OK) (void)dev::stream_sync(stream); // (an early return: nothing of this call may still be writing the workspace the next one reuses)
  return rc;
}

} // namespace fgmm

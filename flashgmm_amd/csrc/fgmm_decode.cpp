// fgmm_decode.cpp — the batched decode through GPU-built edge tables (what replaces RansDecoder::decode_with_indexes_gmm,
// rans_interface.cpp:766-883, and GaussianMixtureConditional.decompress above it, entropy_models.py:872-910).
//
//   caller's stream : [H2D descriptors][tab_kernel unit 0][tab_kernel unit 1] ...          [y_hat scatter, round by round]
//   aux stream      : after unit u's kernel -> D2H of its counters (cursor = bytes of rows placed)
//   this thread     : unit u's size known -> a pinned range of exactly that size; copy stream: ONE copy per unit (headers + block
//                     offsets + rows); the unit's pieces are marked queued: the workers' (bitstream, piece) tasks become ready
//   host workers    : the earliest-landing ready task first; a worker SLEEPS on the landing event of the unit it needs WITHOUT taking
//                     the task, and takes it once the unit is known to have landed; the coder state (x, read pointer, position)
//                     travels with the bitstream from worker to worker between pieces
//   caller's stream : scatter kernels read the decoded symbols from pinned memory and write the float latent - round by round
//
// A launch unit is one ROUND of pieces: block range p of every item of the call (the first round is cut into small launches so
// that the first tables reach the host early).  Why pieces: a bitstream decodes sequentially (~9 ns/symbol), so whatever lands
// last leaves that much host work behind it; every item therefore crosses in shrinking pieces, piece-major, and what remains
// after the final copy is 1/36 of each bitstream.  Why a task is taken only AFTER its tables have landed (round 5): a worker
// that is woken late - on a shared host a woken thread can stand on a run queue for one or two scheduler ticks, 4-8 ms,
// profiles/r05_stall_diagnosis.md - then holds nothing; whoever is awake decodes the piece, and the late one finds it done.
// (Measured against taking the task first and sleeping on it, rounds 2-4's way: the same step times on quiet and on noisy
// boxes - the slow steps of a noisy box are mostly the CALLING thread's lost time slices, profiles/r05_hedge_ab_*.json.)
// Items whose half-width does not fit the single-pass kernel (tab_tl() == 0) take the generic two-pass kernels, one item at a
// time, synchronously (8-byte headers past max_bs 16382).
//
//   DecodeCall::configure / plan_items / plan_units / plan_staging / ensure_buffers / fill_host_side   the PLANNER (no GPU work)
//   launch_unit / launch_next / collect_unit / run_generic                                              launches, staging, copies
//   worker / take_* / run_piece / run_segment / mark_queued                                             the task queue
//   scatter_and_finish                                                                                  symbols back to the GPU
#include <array>

#include "fgmm_ctx.h"

namespace fgmm {
namespace {

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
};
// a task: piece `next_piece` of a sequentially decoded item (seg < 0), or segment `seg` of a checkpointed one; `unit` = the launch
// unit whose copy brings the last table piece the task needs (-1: nothing to wait for); the earliest-landing task first
struct Key {
  int unit, item, seg;
  bool operator>(const Key &o) const { return unit != o.unit ? unit > o.unit : (item != o.item ? item > o.item : seg > o.seg); }
};

constexpr int kLaunchAhead = 3; // launches enqueued ahead of the unit whose size the calling thread waits for (a launch costs it
                                // ~11 us of API calls: all up front, the first copy was 0.08 ms late)

struct DecodeCall {
  fgmm_ctx *ctx;
  dev::Stream stream;
  std::vector<DecItem> &items;
  const int mode, count;
  Trace tr;
  // ---- configuration of the call
  int cap_e = kTabCapE, np = 1, decoders = 1;
  bool clamped = false, f16 = false, spin = false;
  uint32_t ef_min = kTabNoEf;
  int64_t streams_of_work = 0;
  bool lead_small = false; // plan_pieces: a call with one decoder - a small first piece, then pieces that grow (lead_cum)
  double lead_cum[kMaxPieces + 1] = {0}; // ... the share of the bitstream's blocks before piece p
  // ---- plan
  std::vector<int> fast, generic;
  std::vector<Unit> units;
  int n_units = 0;
  Arena ar; // device workspace, mirrored in h_ws up to the counters
  size_t o_descs = 0, o_scat = 0, o_counters = 0, upload_bytes = 0, stage_total = 0, n_parts = 0;
  std::vector<size_t> unit_desc0;
  int n_round = 0, M_round = 0;
  int64_t hw_round = 0;
  bool dead_round = false;
  dev::Event *ev_kernel = nullptr, *ev_counters = nullptr, *ev_landed = nullptr;
  // ---- launches and copies
  TempDevice temp;
  int launched = 0;
  unsigned long long edges = 0;
  std::vector<std::array<double, 3>> unit_trace; // trace level 2: [queued at, bytes, landed at] per unit
  double head[3] = {0, 0, 0};                    // ... and the time before marks[1] in detail (fgmm_call_marks.head_ms)
  double marks[5] = {0, 0, 0, 0, 0};             // the call log: planned | first copy queued | last copy queued | last unit seen landed | last decoder done
  // ---- the task queue (everything below is guarded by `mu`)
  std::mutex mu;
  std::condition_variable work_cv, done_cv;
  bool abandon = false; // this call is returning early: workers must not wait for copies that will never be queued
  int unfinished = 0;
  std::priority_queue<Key, std::vector<Key>, std::greater<Key>> ready;
  std::vector<char> unit_landed; // a worker has seen the unit's copy complete (copies complete in the order they were queued)
  bool copy_failed = false;
  double wait_ms = 0; // summed over the workers: asleep on a landing event

  DecodeCall(fgmm_ctx *c, dev::Stream s, std::vector<DecItem> &it, int m)
      : ctx(c), stream(s), items(it), mode(m), count((int)it.size()), tr("decode", (int)c->opt.trace) {}

  // =================================================================================================================== planner
  int configure() {
    int rc;
    if ((rc = ctx->ensure_streams())) return rc;
    cap_e = (int)std::min<int64_t>(std::max<int64_t>(ctx->opt.tab_cap_e, 256), 32768) & ~31;
    if (ctx->opt.tab_cap_e == kTabCapE) // (the default: a half-width that only fits the wider budget gets it)
      for (const DecItem &it : items)
        if (!tab_tl(it.max_bs, cap_e) && tab_tl(it.max_bs, kTabCapEWide)) {
          cap_e = kTabCapEWide;
          break;
        }
    clamped = items[0].clamp != 0;
    f16 = items[0].prm.dtype == FGMM_F16;
    // Segments pay when a call has fewer bitstreams than workers (one image, ELIC's stages of a few images); a call with a bitstream
    // per worker or more keeps them all busy piece by piece and would only pay the segments' bookkeeping: the notes are ignored there
    const bool use_ckpt = ctx->opt.ckpt_decode == 1 || (ctx->opt.ckpt_decode == 0 && count < std::max(ctx->decoders()->size(), 1));
    for (auto &it : items) {
      if (!use_ckpt) it.ckpt = nullptr, it.n_ckpt = 0;
      streams_of_work += it.ckpt && it.n_ckpt > 0 ? it.n_ckpt + 1 : 1;
    }
    decoders = (int)std::min<int64_t>(std::max(ctx->decoders()->size(), 1), streams_of_work);
    // Elias-Fano rows (long rows: ef_min) are 18 % fewer bytes than uint16 rows and 40 % more nanoseconds to search (57.6 B and 12 ns
    // per latent against 70 B and 8.7 ns on the Kodak workload): with P decoders at work a latent costs max(bytes / 55.7 GB/s,
    // ns / P) - Elias-Fano rows pay when 12 / P < 70 B / 55.7 GB/s = 1.26 ns, P >= 10
    ef_min = ctx->opt.ef_rows == 1 || (ctx->opt.ef_rows == 0 && decoders >= 10) ? (uint32_t)ctx->opt.ef_min : kTabNoEf;
    return FGMM_OK;
  }

  // coded channels, header form, path
  int plan_items() {
    for (int i = 0; i < count; ++i) {
      DecItem &it = items[i];
      if (it.max_bs < 0 || it.max_bs > FGMM_MAX_BS) return fail(FGMM_ERR_UNSUPPORTED, "max_bs_value %d outside [0, %d]", it.max_bs, FGMM_MAX_BS);
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
    return FGMM_OK;
  }

  // Every item crosses in `np` pieces (block ranges), PIECE-MAJOR: piece 0 of every item, then piece 1 ...  Pieces shrink linearly
  // (4 pieces: 40, 30, 20, 10 % of the blocks).
  void plan_pieces() {
    np = (int)std::min<int64_t>(std::max<int64_t>(ctx->opt.pieces, 1), kMaxPieces);
    int64_t lat = 0, lat_max = 0;
    for (int i : fast) lat += items[i].n, lat_max = std::max(lat_max, items[i].n);
    if (ctx->opt.pieces > 0) return; // (an explicit number is taken as it is)
    // automatic: eight pieces leave a Kodak half's decoder at most 4 096 latents (1 / 36 of its bitstream, 40 us) behind the bus's last
    // byte; a bitstream of an ELIC-4K stage is up to 24 times as long - as many pieces as keep the last one at that size, at most 24
    // (ELIC-4K, 16 images: 231 -> 210 ms per step with 24, 32 no better; profiles/r04_elic_pieces_ab.txt)
    np = 8;
    while (np < kAutoPiecesMax && (int64_t)np * (np + 1) / 2 * 4096 < lat_max) ++np;
    if (lat < 65536) np = 1;                 // pieces only pay for rows that take a while to cross
    // One decoder (one image's half, the latency case): it is the bottleneck - 1.2 ms for a Kodak half against 0.14 ms of tables on the
    // bus, a latent takes 9-10 ns to decode and 1.0-1.3 ns to cross - so pieces only have to let it START early and never run dry: a small
    // first piece (an eighth of a Kodak half lands 0.09 ms sooner than the half that "3, 2, 1" made it), then pieces that grow FOURFOLD
    // (cumulatively): the next piece - three times what has landed so far - crosses in 3.1-3.9 ns per latent already here plus a round
    // trip of this thread (~60 us: launch, counter, copy), the decoder needs 7.2 ns for each of them: it never waits from 16 k latents
    // on.  Kodak half: 1/8, 3/8, 1/2 (two pieces - an eighth, the rest - left the decoder waiting 0.08-0.1 ms for the second: one image
    // 3.43 -> 3.28 ms, profiles/r06_README.md); an ELIC-4K stage of 3.5 M latents: 24 k, 72 k, 288 k, 1.2 M, the rest.
    if (decoders == 1 && np > 1) {
      lead_small = true;
      const double first = std::min<double>(std::max<double>((double)lat_max / 8, 16384), 24576);
      int k = 0;
      for (double cum = first; k + 1 < kAutoPiecesMax && cum < 0.6 * (double)lat_max; cum *= 4) lead_cum[++k] = cum / (double)lat_max;
      np = k + 1;
      lead_cum[np] = 1.0;
      return;
    }
    // every round costs this thread ~60 us of launch / counter / copy round trips: no more rounds than the tables' time on the bus
    // is worth
    np = (int)std::min<int64_t>(np, std::max<int64_t>(1, lat * 58 / 55700 / 60)); // lat * 58 B / 55.7 GB/s in units of 60 us
  }
  int64_t piece_bound(int64_t nblk, int p) const { // first block of piece p: weights np, np-1 ... 1
    if (lead_small) return p <= 0 ? 0 : p >= np ? nblk : std::min<int64_t>((int64_t)((double)nblk * lead_cum[p]), nblk);
    const int64_t tot = (int64_t)np * (np + 1) / 2, cum = (int64_t)p * (2 * np - p + 1) / 2;
    return (int64_t)((__int128)nblk * cum / tot);
  }

  // launch units over the fast items (in item order); per item: piece ends, segments, an aligned copy of a misaligned stream
  int plan_units() {
    const int n_fast = (int)fast.size();
    plan_pieces();
    for (int p = 0; p < np && n_fast; ++p) {
      // the first round is cut 2, 4, 8 ... so that the first tables land early; later rounds are one launch + one copy each
      int k = 0, sz = p == 0 && n_fast >= 8 ? (int)std::min<int64_t>(std::max<int64_t>(ctx->opt.dec_first, 1), n_fast) : n_fast;
      while (k < n_fast) {
        Unit u;
        const int k1 = std::min(k + sz, n_fast);
        for (; k < k1; ++k) {
          const DecItem &it = items[fast[(size_t)k]];
          u.parts.push_back(Part{fast[(size_t)k], piece_bound(it.nblk, p), piece_bound(it.nblk, p + 1), 0, 0, p});
        }
        units.push_back(std::move(u));
        sz = std::min(n_fast, sz * 2);
      }
    }
    n_units = (int)units.size();
    for (int i : fast) {
      DecItem &it = items[i];
      it.n_piece = np;
      for (int p = 0; p < np; ++p) it.piece_end[p] = std::min<int64_t>(piece_bound(it.nblk, p + 1) * it.tl, it.n);
      // segments: the notes must be exactly the ones an encoder writes for this many symbols (anything else: sequential)
      const bool seekable = it.ckpt && it.n_ckpt > 0 && it.ckpt_stride >= 256 && !(it.ckpt_stride & (it.ckpt_stride - 1)) &&
                            it.n_ckpt == (it.n - 1) / it.ckpt_stride && it.n_ckpt < (1 << 24);
      it.n_seg = seekable ? (int)it.n_ckpt + 1 : 0;
      it.segs_left.store(it.n_seg);
      // TabDecoder::begin copies a bitstream that is not 4-byte aligned (a C caller's; Python's bytes are aligned): once per item
      // here, not once per segment there
      if (it.n_seg && (reinterpret_cast<uintptr_t>(it.enc) & 3) && it.enc_len >= 8 && !(it.enc_len & 3)) {
        try {
          it.enc_aligned.resize(it.enc_len / 4);
        } catch (const std::bad_alloc &) {
          return fail(FGMM_ERR_NOMEM, "out of memory (%zu bytes of bitstream)", it.enc_len);
        }
        memcpy(it.enc_aligned.data(), it.enc, it.enc_len);
        it.enc = reinterpret_cast<const uint8_t *>(it.enc_aligned.data());
      }
      if (it.n_seg) it.view = TabView{it.ef_min, it.hdr_form, it.tl, it.n_piece, it.piece, nullptr, nullptr};
    }
    return FGMM_OK;
  }

  // each unit's range in the staging area: [headers | block offsets | rows provisioned for the worst case] when that fits the
  // budget, else every unit's row area shrinks by the same factor (a unit that then overflows is re-run with the exact size)
  void plan_staging() {
    size_t rows_worst_total = 0, fixed_total = 0;
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
      fixed_total += u.fixed + 512;
      n_parts += u.parts.size();
    }
    const size_t want = fixed_total + rows_worst_total + 256 * (size_t)n_units;
    const size_t budget = want <= ctx->d_stage_cap && ctx->opt.stage_max_mb <= 0 ? ctx->d_stage_cap : ctx->stage_budget(); // (the device is asked only when the area has to grow)
    if (want > budget && rows_worst_total) {
      const double f = budget > fixed_total ? (double)(budget - fixed_total) / (double)rows_worst_total : 0.0;
      for (auto &u : units) u.rows_cap = align_up(std::max<size_t>((size_t)((double)u.rows_cap * f), 4096), 256);
    }
    for (auto &u : units) {
      u.o_stage = stage_total;
      stage_total += align_up(u.fixed + u.rows_cap + 256, 256);
    }
  }

  // Symbols back to the GPU ROUND BY ROUND (ScatDesc): the sequentially decoded items of the single-pass path with a latent to write
  // (a checkpointed item's segments finish in any order, a generic item has one piece: those are scattered whole, as they finish)
  void plan_rounds() {
    if (count > 65535 || ctx->opt.scatter_rounds == 0) return;
    for (int i : fast) {
      DecItem &it = items[i];
      it.rounds = it.y_hat && it.n_seg == 0 && (int64_t)it.M * it.hw > 0 && it.M <= 65535;
      if (!it.rounds) continue;
      n_round = std::max(n_round, it.n_piece);
      M_round = std::max(M_round, it.M);
      hw_round = std::max(hw_round, it.hw);
      dead_round = dead_round || it.n_ch < it.M;
    }
  }

  // workspace, events, staging, pinned output areas, the int32 symbol scratch
  int ensure_buffers() {
    o_descs = ar.take(sizeof(DecDesc) * std::max<size_t>(n_parts, 1));
    o_scat = ar.take(sizeof(ScatDesc) * (size_t)count, 16);
    o_counters = ar.take(kCounterBytes * (size_t)std::max(n_units, 1), 256);
    upload_bytes = o_counters;
    // Workers SLEEP on the copies' events - except in a small call (one image: a few hundred microseconds in all), where being woken by
    // an interrupt costs as much as the work itself: there they wait on plain events, which the runtime polls
    int64_t lat_total = 0;
    for (auto &it : items) lat_total += it.n;
    spin = ctx->opt.spin_lat < 0 ? false : lat_total <= ctx->opt.spin_lat;
    int rc;
    if ((rc = ctx->ensure_device(ar.off)) || (rc = ctx->ensure_host(ar.off)) ||
        (rc = ctx->ensure_events((spin ? 3 : 2) * (size_t)std::max(n_units, 1) + 2, (size_t)n_units + 2)) || (rc = ctx->ensure_stage(stage_total)))
      return rc;
    ctx->chunks_reset();
    ev_kernel = ctx->events.data();
    ev_counters = ev_kernel + n_units;
    ev_landed = spin ? ev_counters + n_units : ctx->sleep_events.data();
    size_t out_total = 256, need = 0;
    for (auto &it : items) {
      out_total += align_up(sizeof(int32_t) * (size_t)std::max<int64_t>(it.n, 1), 256);
      need += it.sym_host_out ? 0 : (size_t)std::max<int64_t>(it.n, 1);
    }
    char *h_outs = nullptr;
    if ((rc = ctx->chunk_alloc(out_total, &h_outs))) return rc;
    try {
      if (ctx->h_sym.size() < need) ctx->h_sym.resize(need);
    } catch (const std::bad_alloc &) {
      return fail(FGMM_ERR_NOMEM, "out of memory (%zu decoded symbols)", need);
    }
    size_t o = 0, at = 0;
    for (auto &it : items) {
      it.h_out = h_outs + o;
      o += align_up(sizeof(int32_t) * (size_t)std::max<int64_t>(it.n, 1), 256);
      it.sym = it.sym_host_out ? it.sym_host_out : ctx->h_sym.data() + at;
      if (!it.sym_host_out) at += (size_t)std::max<int64_t>(it.n, 1);
    }
    unit_landed.assign((size_t)n_units + 1, 0);
    unfinished = count;
    return FGMM_OK;
  }

  DecDesc base_desc(const DecItem &it) const {
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
  }
  void fill_unit_descs(int u) {
    const Unit &un = units[(size_t)u];
    DecDesc *hd = reinterpret_cast<DecDesc *>(ctx->h_ws + o_descs);
    for (size_t k = 0; k < un.parts.size(); ++k) {
      const Part &p = un.parts[k];
      DecDesc &d = hd[unit_desc0[(size_t)u] + k];
      d = base_desc(items[p.item]);
      d.blk_begin = (int32_t)p.blk_begin;
      d.blk_end = (int32_t)p.blk_end;
      d.hdr_out = un.d_range + p.o_hdr;
      d.blkoff_out = reinterpret_cast<uint32_t *>(un.d_range + p.o_blkoff);
      d.rows = reinterpret_cast<uint8_t *>(un.d_range + un.fixed);
      d.rows_cap = un.rows_cap;
      d.counters = reinterpret_cast<unsigned long long *>(ctx->d_ws + o_counters + kCounterBytes * (size_t)u);
    }
  }

  // channel lists and ranks, scatter descriptors, table descriptors: everything the upload carries
  void fill_host_side() {
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
      if (!it.rounds) continue;
      sd.sym = reinterpret_cast<const int16_t *>(it.h_out);
      sd.chan_list = reinterpret_cast<const int32_t *>(ctx->d_ws + it.o_list);
      sd.rank = reinterpret_cast<const int32_t *>(ctx->d_ws + it.o_rank);
      sd.y_hat = it.y_hat;
      sd.hw = it.hw;
      sd.M = it.M;
      for (int p = 0; p < kMaxPieces; ++p) sd.bound[p + 1] = p < it.n_piece ? it.piece_end[p] : it.n;
    }
    unit_desc0.assign((size_t)n_units + 1, 0);
    for (int u = 0; u < n_units; ++u) {
      unit_desc0[(size_t)u + 1] = unit_desc0[(size_t)u] + units[(size_t)u].parts.size();
      units[(size_t)u].d_range = ctx->d_stage + units[(size_t)u].o_stage;
      fill_unit_descs(u);
    }
  }

  // ======================================================================================================== launches and copies
  int launch_unit(int u) {
    const Unit &un = units[(size_t)u];
    int64_t blocks_max = 0;
    int tl_max = 16;
    for (auto &p : un.parts) {
      blocks_max = std::max(blocks_max, p.blk_end - p.blk_begin);
      tl_max = std::max(tl_max, (int)items[p.item].tl);
    }
    const DecDesc *dd = reinterpret_cast<const DecDesc *>(ctx->d_ws + o_descs);
    LAUNCH_TRY(launch_tab(dd + unit_desc0[(size_t)u], (int)un.parts.size(), (int)blocks_max, tl_max, cap_e, mode, clamped, f16, stream));
    return FGMM_OK;
  }
  int launch_next() {
    const int u = launched;
    int rc = launch_unit(u);
    if (rc) return rc;
    unsigned long long *h_counters = reinterpret_cast<unsigned long long *>(ctx->h_ws + o_counters);
    DEV_TRY(dev::event_record(ev_kernel[u], stream));
    DEV_TRY(dev::stream_wait_event(ctx->aux_stream, ev_kernel[u]));
    DEV_TRY(dev::copy_async(h_counters + kTabCounters * (size_t)u, ctx->d_ws + o_counters + kCounterBytes * (size_t)u, kCounterBytes, dev::kD2H, ctx->aux_stream));
    DEV_TRY(dev::event_record(ev_counters[u], ctx->aux_stream));
    ++launched;
    if (launched == n_units) return ctx->prof_end(1, stream); // brackets every table kernel of the call
    return FGMM_OK;
  }
  int upload_and_prime() {
    DEV_TRY(dev::copy_async(ctx->d_ws, ctx->h_ws, upload_bytes, dev::kH2D, stream));
    DEV_TRY(dev::memset_async(ctx->d_ws + o_counters, 0, kCounterBytes * (size_t)std::max(n_units, 1), stream));
    const ScatDesc *d_scat = reinterpret_cast<const ScatDesc *>(ctx->d_ws + o_scat);
    if (n_round && dead_round) LAUNCH_TRY(launch_yhat_zero_dead(d_scat, count, M_round, hw_round, stream)); // (channels without a coded symbol)
    int rc;
    if ((rc = ctx->prof_begin(1, stream))) return rc;
    if (n_units == 0 && (rc = ctx->prof_end(1, stream))) return rc;
    while (launched < std::min(n_units, kLaunchAhead))
      if ((rc = launch_next())) return rc;
    return FGMM_OK;
  }

  // The provisioned row area of unit u was too small (its cursor says how much it takes): once more into an area of exactly that size
  int rerun_unit(int u, unsigned long long *cn) {
    Unit &un = units[(size_t)u];
    for (int attempt = 0; cn[1] && attempt < 3; ++attempt) {
      const size_t need = align_up((size_t)cn[0], 256);
      char *d_new = nullptr;
      int rc = temp.alloc(un.fixed + need + 256, &d_new);
      if (rc) return rc;
      un.d_range = d_new;
      un.rows_cap = need;
      fill_unit_descs(u);
      const DecDesc *hd = reinterpret_cast<const DecDesc *>(ctx->h_ws + o_descs);
      DEV_TRY(dev::copy_async(ctx->d_ws + o_descs + sizeof(DecDesc) * unit_desc0[(size_t)u], hd + unit_desc0[(size_t)u], sizeof(DecDesc) * un.parts.size(),
                              dev::kH2D, stream));
      DEV_TRY(dev::memset_async(ctx->d_ws + o_counters + kCounterBytes * (size_t)u, 0, kCounterBytes, stream));
      if ((rc = launch_unit(u))) return rc;
      DEV_TRY(dev::event_record(ev_kernel[u], stream));
      DEV_TRY(dev::copy_async(cn, ctx->d_ws + o_counters + kCounterBytes * (size_t)u, kCounterBytes, dev::kD2H, stream));
      DEV_TRY(dev::stream_sync(stream));
    }
    if (cn[1]) return fail(FGMM_ERR_HIP, "decode tables could not be placed (unit %d: flags %llu, %llu of %zu bytes)", u, cn[1], cn[0], un.rows_cap);
    return FGMM_OK;
  }

  // unit u: its size is known -> a pinned range of exactly that size, ONE copy; the workers are told
  int collect_unit(int u) {
    Unit &un = units[(size_t)u];
    int rc;
    while (launched < std::min(n_units, u + 1 + kLaunchAhead))
      if ((rc = launch_next())) return rc;
    DEV_TRY(dev::event_sync(ev_counters[u]));
    if (u == 0) head[2] = tr.ms();
    unsigned long long *cn = reinterpret_cast<unsigned long long *>(ctx->h_ws + o_counters) + kTabCounters * (size_t)u;
    if (cn[1] && (rc = rerun_unit(u, cn))) return rc;
    const size_t used = (size_t)cn[0];
    for (int q = 0; q < kTabEdgeSlots; ++q) edges += cn[4 + q];
    char *h_range = nullptr;
    if ((rc = ctx->chunk_alloc(un.fixed + used + 256, &h_range))) return rc;
    memset(h_range + un.fixed + used, 0, 256); // slack: the host's SIMD search reads a little past a row
    // the unit's kernel is complete (its counters are here): the copy depends on nothing else
    DEV_TRY(dev::stream_wait_event(ctx->copy_stream, ev_kernel[u]));
    if (un.fixed + used) DEV_TRY(dev::copy_async(h_range, un.d_range, un.fixed + used, dev::kD2H, ctx->copy_stream));
    DEV_TRY(dev::event_record(ev_landed[u], ctx->copy_stream));
    for (auto &p : un.parts) {
      DecItem &it = items[p.item];
      TabPiece &pc = it.piece[p.piece];
      pc.hdr = h_range + p.o_hdr;
      pc.blk_off = reinterpret_cast<const uint32_t *>(h_range + p.o_blkoff);
      pc.rows = reinterpret_cast<const uint8_t *>(h_range + un.fixed);
      pc.rows_len = used + 256;
      pc.end = std::min<int64_t>(p.blk_end * it.tl, it.n);
      it.piece_unit[p.piece] = u;
      const int64_t lat = pc.end - std::min<int64_t>(p.blk_begin * it.tl, it.n);
      it.table_bytes += (uint64_t)it.hdr_form * (uint64_t)lat + sizeof(uint32_t) * (uint64_t)(p.blk_end - p.blk_begin);
      mark_queued(p.item, p.piece + 1); // pieces reach an item in order: rounds are piece-major
    }
    if (!un.parts.empty()) items[un.parts[0].item].table_bytes += used; // rows are shared by the unit's items: accounted once
    marks[2] = tr.ms();
    if (u == 0) marks[1] = marks[2];
    if (tr.level > 1) unit_trace.push_back({tr.ms(), (double)(un.fixed + used), 0.0});
    return FGMM_OK;
  }

  // an item too wide for the single-pass kernel: the generic two-pass kernels, synchronously; its one piece needs no wait
  int queue_generic_as_landed(int i, const TabPiece &pc) {
    DecItem &it = items[i];
    it.n_piece = 1;
    it.piece[0] = pc;
    it.piece_unit[0] = -1;
    mark_queued(i, 1);
    return FGMM_OK;
  }
  int run_generic(int i) {
    DecItem &it = items[i];
    if (it.n == 0) return queue_generic_as_landed(i, TabPiece{ctx->h_ws, nullptr, reinterpret_cast<const uint8_t *>(ctx->h_ws), 0, 0});
    const int32_t tiles = (int32_t)((it.hw + 255) / 256);
    const size_t nblk = (size_t)it.n_ch * (size_t)tiles;
    const size_t hdr_bytes = align_up((size_t)(it.hdr_form == 8 ? 8 : 4) * (size_t)it.n, 256);
    Arena ga;
    const size_t g_desc = ga.take(sizeof(DecDesc)), g_used = ga.take(64), g_hdr = ga.take(hdr_bytes), g_bsum = ga.take(4 * nblk + 64),
                 g_boff = ga.take(8 * nblk + 64);
    char *d_g = nullptr;
    int rc;
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
    DEV_TRY(dev::memset_async(d_g + g_used, 0, 64, stream));
    DEV_TRY(dev::copy_async(d_g + g_desc, &d, sizeof d, dev::kH2D, stream));
    LAUNCH_TRY(launch_cdftab_count(reinterpret_cast<const DecDesc *>(d_g + g_desc), 1, it.n_ch, it.hw, mode, clamped, f16, stream));
    unsigned long long used4[4] = {0, 0, 0, 0};
    DEV_TRY(dev::copy_async(used4, d_g + g_used, sizeof used4, dev::kD2H, stream));
    DEV_TRY(dev::stream_sync(stream));
    if (used4[3]) {
      it.status = FGMM_ERR_UNSUPPORTED;
      return queue_generic_as_landed(i, TabPiece{ctx->h_ws, nullptr, reinterpret_cast<const uint8_t *>(ctx->h_ws), 0, 0});
    }
    const size_t pool_bytes = (size_t)used4[0];
    char *d_pool = nullptr, *h_range = nullptr;
    if ((rc = temp.alloc(pool_bytes + 256, &d_pool)) || (rc = ctx->chunk_alloc(hdr_bytes + pool_bytes + 256, &h_range))) return rc;
    d.pool = reinterpret_cast<uint8_t *>(d_pool);
    DEV_TRY(dev::copy_async(d_g + g_desc, &d, sizeof d, dev::kH2D, stream));
    LAUNCH_TRY(launch_cdftab_fill(reinterpret_cast<const DecDesc *>(d_g + g_desc), 1, it.n_ch, it.hw, mode, clamped, f16, stream));
    DEV_TRY(dev::copy_async(h_range, d_g + g_hdr, hdr_bytes, dev::kD2H, stream));
    if (pool_bytes) DEV_TRY(dev::copy_async(h_range + hdr_bytes, d_pool, pool_bytes, dev::kD2H, stream));
    memset(h_range + hdr_bytes + pool_bytes, 0, 256);
    DEV_TRY(dev::stream_sync(stream)); // nothing left to wait for
    it.table_bytes = hdr_bytes + pool_bytes;
    return queue_generic_as_landed(i, TabPiece{h_range, nullptr, reinterpret_cast<const uint8_t *>(h_range + hdr_bytes), pool_bytes + 256, it.n});
  }

  // ================================================================================================================ task queue
  int seg_last_piece(const DecItem &it, int sg) const { // the piece that holds the last latent of segment sg
    const int64_t hi = sg + 1 == it.n_seg ? it.n : (int64_t)(sg + 1) * it.ckpt_stride;
    int p = 0;
    while (p + 1 < it.n_piece && it.piece_end[p] < hi) ++p;
    return p;
  }
  int push_if_ready(int i) { // under mu -> tasks pushed
    DecItem &it = items[i];
    if (it.n_seg) { // every segment whose tables are queued by now; segments are independent of one another
      int pushed = 0;
      while (it.next_seg_push < it.n_seg) {
        const int lp = seg_last_piece(it, it.next_seg_push);
        if (lp >= it.queued) break;
        ready.push(Key{it.piece_unit[lp], i, it.next_seg_push++});
        ++pushed;
      }
      return pushed;
    }
    if (!it.busy && !it.in_ready && !it.done.load() && it.next_piece < it.queued) {
      it.in_ready = true;
      ready.push(Key{it.piece_unit[it.next_piece], i, -1});
      return 1;
    }
    return 0;
  }
  void mark_queued(int i, int pieces) { // the calling thread: the copy that carries the item's piece `pieces - 1` is queued
    int pushed;
    {
      std::lock_guard<std::mutex> l(mu);
      items[i].queued = pieces;
      pushed = push_if_ready(i);
    }
    if (pushed > 1) work_cv.notify_all(); else if (pushed == 1) work_cv.notify_one();
  }
  void item_finished(int i) { // under mu
    items[i].done.store(1);
    if (--unfinished == 0) work_cv.notify_all();
    done_cv.notify_all();
  }

  // the decoded symbols [k0, k1) of an item -> pinned memory for the scatter kernel: int16 unless some (bypass-coded) symbol does not fit
  static bool narrow(DecItem &it, int64_t k0, int64_t k1) {
    int16_t *s16 = reinterpret_cast<int16_t *>(it.h_out);
    int32_t acc = 0;
    for (int64_t k = k0; k < k1; ++k) {
      const int32_t v = it.sym[k];
      s16[k] = (int16_t)v;
      acc |= v ^ (int32_t)(int16_t)v;
    }
    return acc != 0;
  }

  // piece p of a sequentially decoded item, on this thread, no lock held; its tables have landed -> true: the item is finished
  bool run_piece(DecItem &it, int p) {
    const double t0 = tr.ms();
    if (p == 0) {
      it.t_taken = it.t_start = t0;
      it.view = TabView{it.ef_min, it.hdr_form, it.tl, it.n_piece, it.piece, nullptr, nullptr};
      if (it.status == FGMM_OK) it.status = it.dec.begin(it.enc, it.enc_len, &it.view, it.n, it.max_bs, it.sym);
    }
    if (it.status == FGMM_OK) it.status = it.dec.piece(p);
    if (it.status == FGMM_OK && it.y_hat) {
      const int64_t k1 = std::min<int64_t>(it.dec.i, it.n);
      it.wide |= narrow(it, it.narrowed, k1);
      it.narrowed = k1;
    }
    it.t_work += tr.ms() - t0;
    if (it.status == FGMM_OK && p + 1 < it.n_piece) return false;
    const int rf = it.dec.finish();
    if (it.status == FGMM_OK) it.status = rf;
    // (an item scattered round by round: its int16 symbols may still be being read - the calling thread's final loop redoes a wide one)
    if (it.status == FGMM_OK && it.y_hat && it.wide && !it.rounds) memcpy(it.h_out, it.sym, sizeof(int32_t) * (size_t)it.n);
    it.t_end = tr.ms();
    return true;
  }

  // One segment of a checkpointed bitstream, start to end on this thread (no lock held; the table pieces it touches have landed): from
  // its checkpoint - the stream's own head for segment 0 - to the next one, which it must hit exactly.  The last segment to finish
  // closes the item; if any segment missed its checkpoint the whole bitstream is decoded sequentially then (the notes were wrong:
  // nothing of what the segments wrote is kept).  -> true: the item is finished
  bool run_segment(DecItem &it, int sg) {
    const int64_t lo = (int64_t)sg * it.ckpt_stride, hi = sg + 1 == it.n_seg ? it.n : (int64_t)(sg + 1) * it.ckpt_stride;
    const double t0 = tr.ms();
    bool ok = !it.ckpt_bad.load(std::memory_order_relaxed);
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
    if (!ok) it.ckpt_bad.store(1);
    else if (it.y_hat && narrow(it, lo, hi)) it.wide_any.store(1);
    {
      std::lock_guard<std::mutex> l(mu);
      it.t_work += tr.ms() - t0;
      if (sg == 0) it.t_taken = it.t_start = t0;
    }
    if (it.segs_left.fetch_sub(1) != 1) return false;
    // the last segment: close the item
    if (it.ckpt_bad.load()) { // sequential decode of the whole bitstream (every piece has landed: the last segment needed the last one)
      it.status = it.dec.begin(it.enc, it.enc_len, &it.view, it.n, it.max_bs, it.sym);
      for (int p = 0; p < it.n_piece && it.status == FGMM_OK; ++p) it.status = it.dec.piece(p);
      const int rf = it.dec.finish();
      if (it.status == FGMM_OK) it.status = rf;
      if (it.status == FGMM_OK && it.y_hat) it.wide_any.store(narrow(it, 0, it.n) ? 1 : 0);
    }
    it.wide = it.wide_any.load();
    if (it.status == FGMM_OK && it.y_hat && it.wide) memcpy(it.h_out, it.sym, sizeof(int32_t) * (size_t)it.n);
    it.t_end = tr.ms();
    return true;
  }

  // under mu (released while asleep): sleep on unit u's landing event; afterwards every unit up to u is known to have landed
  void await_unit(std::unique_lock<std::mutex> &l, int u) {
    l.unlock();
    const double t0 = tr.ms();
    const bool ok = dev::event_sync(ev_landed[u]) == 0;
    const double t1 = tr.ms();
    l.lock();
    wait_ms += t1 - t0;
    marks[3] = std::max(marks[3], t1);
    if (!ok) copy_failed = true;
    for (int v = u; v >= 0 && !unit_landed[(size_t)v]; --v) unit_landed[(size_t)v] = 1; // (copies complete in the order they were queued)
    if (tr.level > 1 && (size_t)u < unit_trace.size() && unit_trace[(size_t)u][2] == 0.0) unit_trace[(size_t)u][2] = t1;
  }

  void worker() {
    std::unique_lock<std::mutex> l(mu);
    for (;;) {
      if (unfinished == 0) return;
      if (ready.empty()) {
        if (abandon) return;
        work_cv.wait(l);
        continue;
      }
      const Key k = ready.top();
      const bool here = k.unit < 0 || unit_landed[(size_t)k.unit];
      if (!here) {
        // The tables of the earliest task are not known to be on the host yet: sleep on that unit's copy WITHOUT taking the task.
        // Whoever is awake when it lands takes it; a worker that wakes late finds it gone and holds nothing up.
        if (abandon) return;
        await_unit(l, k.unit);
        continue;
      }
      ready.pop();
      DecItem &it = items[k.item];
      if (k.seg >= 0) { // a segment of a checkpointed bitstream: independent of every other task
        if (copy_failed) it.ckpt_bad.store(1), it.status = FGMM_ERR_HIP;
        l.unlock();
        const bool fin = run_segment(it, k.seg);
        l.lock();
        if (fin) item_finished(k.item);
        continue;
      }
      it.in_ready = false;
      it.busy = true;
      const int p = it.next_piece;
      if (copy_failed && it.status == FGMM_OK) it.status = FGMM_ERR_HIP;
      l.unlock();
      const bool fin = run_piece(it, p);
      l.lock();
      it.busy = false;
      it.next_piece = p + 1;
      if (fin) item_finished(k.item);
      else {
        if (push_if_ready(k.item) && ready.size() > 1) work_cv.notify_one(); // (this worker takes one task itself: another one for the rest)
        if (it.rounds) done_cv.notify_all(); // (a piece's symbols are in pinned memory: its round may be complete)
      }
    }
  }

  // ============================================================================================== symbols back to the GPU, the end
  // Round r (piece r of every item that is decoded piece by piece) goes as soon as every such item has decoded it: one launch for
  // all of them, while the later pieces are still on the bus - what is left after the last decoder is the last, smallest piece
  int scatter_and_finish() {
    const ScatDesc *d_scat = reinterpret_cast<const ScatDesc *>(ctx->d_ws + o_scat);
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
        DEV_TRY(dev::stream_sync(stream));
        memcpy(it.h_out, it.sym, sizeof(int32_t) * (size_t)it.n);
      }
      LAUNCH_TRY(launch_yhat_scatter(it.h_out, it.wide, reinterpret_cast<const int32_t *>(ctx->d_ws + it.o_rank), it.y_hat, it.M, it.hw, stream));
    }
    tr.mark("host rANS done");
    DEV_TRY(dev::stream_sync(stream));
    tr.mark("y_hat written");
    if (first_err)
      return fail(first_err, "host rANS decode failed (%d)%s", first_err,
                  first_err == FGMM_ERR_STREAM ? ": bitstream too short"
                  : first_err == FGMM_ERR_UNSUPPORTED ? ": an evaluation window beyond 2^20 edges (see FGMM_MAX_BS)" : "");
    return FGMM_OK;
  }

  void account() {
    ctx->stat[1] = ctx->stat[2] = 0;
    ctx->stat[3] = edges;
    double busy = 0;
    for (auto &it : items) {
      ctx->stat[1] += it.table_bytes;
      ctx->stat[2] += (unsigned long long)it.n;
      marks[4] = std::max(marks[4], it.t_end);
      busy += it.t_work;
    }
    std::lock_guard<std::mutex> l(mu);
    ctx->log_call(1, count, tr, marks, busy, wait_ms, head);
    if (tr.level > 1) {
      for (size_t u = 0; u < unit_trace.size(); ++u)
        fprintf(stderr, "[fgmm decode]   unit %2zu  %2zu parts  %9.0f bytes  queued %7.3f  first seen landed %7.3f\n", u, units[u].parts.size(), unit_trace[u][1],
                unit_trace[u][0], unit_trace[u][2]);
      for (int i = 0; i < count; ++i)
        fprintf(stderr, "[fgmm decode]   item %2d  pieces %d  taken %7.3f  job %7.3f .. %7.3f  (decoding %.3f ms)\n", i, items[i].n_piece, items[i].t_taken,
                items[i].t_start, items[i].t_end, items[i].t_work);
    }
  }

  int run() {
    int rc;
    if ((rc = configure()) || (rc = plan_items()) || (rc = plan_units())) return rc;
    plan_staging();
    plan_rounds();
    if ((rc = ensure_buffers())) return rc;
    tr.mark("planned, buffers ensured");
    marks[0] = tr.ms();
    fill_host_side();
    tr.mark("descriptors built");
    head[0] = tr.ms();
    struct Abandon { // any return: release workers that wait for work (before PoolDrain waits for the workers)
      DecodeCall *c;
      ~Abandon() {
        {
          std::lock_guard<std::mutex> l(c->mu);
          c->abandon = true;
        }
        c->work_cv.notify_all();
      }
    };
    // Any return, last of all (after the workers have gone): nothing of this call may still be in flight on the device - a bitstream
    // that fails early (truncated input) finishes its item while the copies of its later pieces are still queued, and the NEXT call
    // would hand the same staging area to its kernels and the same pinned ranges to its copies (found by ThreadSanitizer on the fake
    // device, round 5).  Free on the normal path: the last unit is known to have landed, the caller's stream has been synchronised.
    struct Quiesce {
      DecodeCall *c;
      bool clean = false;
      ~Quiesce() {
        bool landed_all;
        {
          std::lock_guard<std::mutex> l(c->mu);
          landed_all = c->n_units == 0 || c->unit_landed[(size_t)c->n_units - 1];
        }
        if (!clean) (void)dev::stream_sync(c->stream), (void)dev::stream_sync(c->ctx->aux_stream);
        if (!clean || !landed_all || c->launched < c->n_units) (void)dev::stream_sync(c->ctx->copy_stream);
      }
    } quiesce{this};
    PoolDrain drain{ctx->decoders()}; // on any return: wait for every job before the objects they use go away
    Abandon abandon_on_exit{this};
    if ((rc = upload_and_prime())) return rc;
    tr.mark("first launches enqueued");
    // (a single bitstream too: its decoder starts on piece 0 while this thread is still queuing the later pieces' copies)
    const int n_workers = (int)std::min<int64_t>(std::max(ctx->decoders()->size(), 1), streams_of_work);
    for (int j = 0; j < n_workers; ++j) ctx->decoders()->submit([this] { worker(); });
    head[1] = tr.ms();
    for (int u = 0; u < n_units; ++u)
      if ((rc = collect_unit(u))) return rc;
    tr.mark("sizes known, copies queued");
    for (int i : generic)
      if ((rc = run_generic(i))) return rc;
    rc = scatter_and_finish();
    // (only a call that finished: on an early error return the workers are still decoding - Abandon and PoolDrain run when this
    // scope is left - and would be writing the fields account() reads; a failed call is not in the call log)
    if (rc == FGMM_OK) account();
    quiesce.clean = rc == FGMM_OK;
    return rc;
  }
};

// Is the GPU the faster decoder for this call's checkpointed bitstreams?  A segment's waves decode it at ~0.65 us per symbol however
// empty the chip is, and the chip as a whole at ~0.35 ns per symbol (Kodak-like latents); the host decodes at ~12 ns per symbol and
// worker and is fed at 58 B per latent over PCIe.  Many segments (a batch, a 4K image's group): the GPU, by 2-4x; one Kodak half in a
// few hundred long segments: the host workers.  The rates are those of the boxes this was measured on (MI355X + EPYC 9575F, PCIe 5
// x16): another host overrides the choice with option "gpu_decode" = 1 / 2.
bool gpu_is_faster(const fgmm_ctx *ctx, const std::vector<DecItem> &items, const std::vector<int> &gpu) {
  double syms = 0, stride_max = 0, work = 0;
  for (int i : gpu) {
    syms += (double)items[i].n;
    stride_max = std::max(stride_max, (double)std::min<int64_t>(items[i].ckpt_stride, items[i].n));
    work += (double)(items[i].n_ckpt + 1);
  }
  // (rates measured on MI355X + EPYC 9575F.  A version that re-fitted them from the context's own calls was built and dropped in round 5:
  // small or unusual calls - the fixed costs of a 200 k-symbol GPU call read as throughput - walked the estimate far enough to flip
  // the choice for a Kodak batch, a 3x margin; on another platform the caller sets the option to 1 or 2)
  const double t_gpu = std::max(stride_max * 0.65, syms * 0.00035) + 100.0;
  const double workers = std::min<double>(std::max(ctx->decoders()->size(), 1), work);
  const double t_host = std::max(syms * 0.012 / workers, syms * 58.0 / 55700.0) + 450.0 + 3.0 * work / workers; // + a segment's set-up
  return t_gpu < t_host;
}

// checkpointed bitstreams to the GPU's segment decoder; whatever it does not take or cannot finish goes through the tables, as a
// batch of its own (the notes of a bitstream that failed them are dropped) -> true: the call is complete (*rc_out its status)
bool decode_on_gpu_first(fgmm_ctx *ctx, dev::Stream stream, std::vector<DecItem> &items, int mode, int *rc_out) {
  const int count = (int)items.size();
  std::vector<int> gpu, rest;
  for (int i = 0; i < count; ++i) {
    DecItem &it = items[i];
    int n_ch = 0;
    for (int c = 0; c < it.M; ++c) n_ch += it.zero_bitmap ? (it.zero_bitmap[c] != 0) : 1;
    it.n_ch = n_ch;
    it.n = (int64_t)n_ch * it.hw;
    (gpu_decodable(it, it.n) && it.clamp == items[0].clamp && it.prm.dtype == items[0].prm.dtype ? gpu : rest).push_back(i);
  }
  if (gpu.empty() || (ctx->opt.gpu_decode == 0 && !gpu_is_faster(ctx, items, gpu))) return false;
  std::vector<int> redo;
  int rc = decode_batch_gpu(ctx, stream, items, gpu, mode, redo);
  *rc_out = rc;
  if (rc) return true;
  for (int i : redo) rest.push_back(i);
  if (rest.empty()) return true;
  std::vector<DecItem> sub(rest.size());
  for (size_t k = 0; k < rest.size(); ++k) {
    const DecItem &s0 = items[rest[k]];
    DecItem &t = sub[k];
    t.enc = s0.enc, t.enc_len = s0.enc_len, t.prm = s0.prm, t.stride_p = s0.stride_p, t.M = s0.M, t.hw = s0.hw, t.clamp = s0.clamp;
    t.max_bs = s0.max_bs, t.zero_bitmap = s0.zero_bitmap, t.y_hat = s0.y_hat, t.sym_host_out = s0.sym_host_out;
    const bool failed = std::find(redo.begin(), redo.end(), rest[k]) != redo.end();
    if (!failed) t.ckpt = s0.ckpt, t.n_ckpt = s0.n_ckpt, t.ckpt_stride = s0.ckpt_stride;
  }
  DecodeCall call(ctx, stream, sub, mode);
  *rc_out = call.run();
  for (size_t k = 0; k < rest.size(); ++k) items[rest[k]].status = sub[k].status;
  return true;
}

} // namespace

int decode_batch(fgmm_ctx *ctx, dev::Stream stream, std::vector<DecItem> &items, int mode) {
  if (items.empty()) return FGMM_OK;
  // checkpointed bitstreams go to the GPU's segment decoder when that is the faster one (option "gpu_decode": 0 estimate, 1 whenever
  // possible, 2 never); everything else takes the table path
  int rc = FGMM_OK;
  if (ctx->opt.gpu_decode != 2 && decode_on_gpu_first(ctx, stream, items, mode, &rc)) return rc;
  DecodeCall call(ctx, stream, items, mode);
  return call.run();
}

} // namespace fgmm

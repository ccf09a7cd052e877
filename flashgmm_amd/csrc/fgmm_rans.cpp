// fgmm_rans.cpp — host side of the path: the integer rANS state machine, fed by GPU-built tables.
//
// Nothing in this file touches floating point.  It consumes
//   * the encode-side symbol table (uint32 start | range<<16 per symbol, range == 0 = bypass escape) and
//   * the decode-side trimmed edge tables (4-byte header + row per latent, rows in latent order: uint16 entries,
//     or Elias-Fano coded when wide and monotone)
// produced by fgmm_kernels.hip and reproduces, bit for bit, the streams / symbols of
//   BufferedRansEncoder::flush            compressai/cpp_exts/rans/rans_interface.cpp:557-585
//   the bypass escape                     compressai/cpp_exts/rans/rans_interface.cpp:513-552, :808-824
//   RansDecoder::decode_with_indexes_gmm  compressai/cpp_exts/rans/rans_interface.cpp:798-881 (bisection on F)
//   Rans64Enc/Dec primitives              third_party/ryg_rans/rans64.h:65-142, bit put/get rans_interface.cpp:295-331
//
// Encoder: the per-symbol 64-bit division of Rans64EncPut is replaced by a multiply with an exact
// Alverson reciprocal from a 65 536-entry table (state-identical by construction for every state the
// encoder can reach; the same identity ryg's Rans64EncPutSymbol relies on).
// Decoder: monotone rows are searched with one or two AVX2 compares (uint16 rows) or a select on the unary high
// parts (Elias-Fano rows); everything else (flagged non-monotone rows, a cum_freq no interval contains) goes
// through a literal replay of the reference's bisection over the virtual table, so the result is the
// reference's result in every case.
#include <immintrin.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <mutex>
#include <vector>

#include "../../include/flashgmm_amd.h"
#include "fgmm_internal.h"

namespace fgmm {

namespace {

constexpr uint64_t kRansL = 1ull << 31; // rans64.h:59
constexpr uint32_t kPrec = 16;          // rans_interface.cpp:55
constexpr uint32_t kMaxCdf = 65535;     // rans_interface.cpp:56
constexpr uint32_t kBypassBits = 4;     // rans_interface.cpp:58
constexpr uint32_t kMaxBypassVal = 15;  // rans_interface.cpp:59

struct Rcp {
  uint64_t rcp;      // fixed-point reciprocal
  uint32_t shift;    // post-shift
  uint32_t bias_add; // 65535 for freq == 1 (see below), else 0
};
Rcp g_rcp[65536];
std::once_flag g_rcp_once;

// Alverson, "Integer division using reciprocals": for 2 <= freq < 2^16 and x < 2^63,
//   floor(x / freq) == mulhi(x, rcp) >> shift   with shift = ceil(log2 freq) - 1,
//   rcp = ceil(2^(shift+64) / freq).
// freq == 1 cannot be expressed (reciprocal 1.0); with rcp = 2^64-1, shift = 0 the quotient comes out as
// x - 1, which the update x + bias + q*(2^16 - freq) absorbs by bias += 2^16 - 1.
void init_rcp() {
  g_rcp[0] = {0, 0, 0};
  g_rcp[1] = {~0ull, 0, 65535};
  for (uint32_t freq = 2; freq < 65536; ++freq) {
    uint32_t shift = 0;
    while (freq > (1u << shift)) shift++;
    // ceil(2^(shift+63) / freq) by a 128/64 division
    const unsigned __int128 num = ((unsigned __int128)1 << (shift + 63)) + freq - 1;
    g_rcp[freq] = {(uint64_t)(num / freq), shift - 1, 0};
  }
}

inline uint64_t mulhi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }

struct Enc {
  uint64_t x;
  uint32_t *ptr;
  inline void put(uint32_t start, uint32_t freq) { // Rans64EncPut(start, freq, 16)
    const uint64_t x_max = (uint64_t)freq << 47;   // ((RANS64_L >> 16) << 32) * freq
    uint64_t xx = x;
    // renormalisation without a branch (it is taken for ~1 symbol in 10, unpredictably): the word is always stored
    // below the write pointer — there is always room, the buffers are sized for one word per entry — and the pointer
    // moves only if it was due.  2.8 -> 2.4 ns/symbol on Zen 5, and what lets two streams share a core (below).
    const bool renorm = xx >= x_max;
    ptr[-1] = (uint32_t)xx;
    ptr -= renorm;
    xx = renorm ? (xx >> 32) : xx;
    const Rcp &r = g_rcp[freq];
    const uint64_t q = mulhi(xx, r.rcp) >> r.shift;
    x = xx + start + r.bias_add + q * (65536u - freq); // == ((xx / freq) << 16) + xx % freq + start
  }
  inline void put_bits(uint32_t val) {             // Rans64EncPutBits(val, 4)
    const uint64_t x_max = 1ull << 59;             // ((RANS64_L >> 16) << 32) * (1 << 12)
    uint64_t xx = x;
    if (xx >= x_max) {
      *--ptr = (uint32_t)xx;
      xx >>= 32;
    }
    x = (xx << kBypassBits) | val;
  }
};

} // namespace

// one symbol of the reversed walk over a GPU-built table (rans_interface.cpp:569-583)
static inline void encode_entry(Enc &e, uint32_t ent, const int32_t *symbols, int64_t i) {
  const uint32_t freq = ent >> 16;
  if (__builtin_expect(freq != 0, 1)) {
    e.put(ent & 0xFFFFu, freq);
    return;
  }
  // bypass escape; forward order was [sentinel {65535,1}] [count] [nibble 0 .. nibble k-1]  (:519-551)
  const int32_t value = symbols ? symbols[i] : (int32_t)(int16_t)(uint16_t)(ent & 0xFFFFu);
  const uint32_t raw = (uint32_t)value;
  int nn = 0;
  for (uint32_t t = raw; t != 0; t >>= kBypassBits) ++nn; // <= 8
  for (int j = nn - 1; j >= 0; --j) e.put_bits((raw >> (j * kBypassBits)) & kMaxBypassVal);
  e.put_bits((uint32_t)nn); // nn <= 8 < 15: the count is always a single nibble (:538-543)
  e.put(kMaxCdf, 1);
}
static int64_t count_bypass(const uint32_t *packed, int64_t n) {
  int64_t nb = 0;
  for (int64_t i = 0; i < n; ++i) nb += (packed[i] >> 16) == 0;
  return nb;
}
// every entry of the reference's _syms emits at most one 32-bit word; a bypassed symbol is 1 + 1 + <=8 entries;
// +16: the flush words and the word the branch-free renormalisation always stores below the pointer
static inline size_t encode_words(int64_t n, int64_t nb) { return (size_t)n + (size_t)nb * 10 + 16; }
// ---- checkpoints (seekable streams) ----------------------------------------------------------------------
// The decoder's state before it decodes symbol i IS the encoder's state after it has encoded symbol i on its reversed walk
// (decoding inverts encoding step by step, renormalisation included), and the words the decoder has consumed by then are
// the ones the encoder emits AFTER that point.  So the encoder can note, every `stride` symbols, (state, words emitted so far)
// - out of band: the bitstream stays the reference's, byte for byte - and a decoder that is handed the notes may start at
// any of them.  A segment that ends exactly in the next note's (state, position) is the sequential decoder's work on that
// range, by induction from the stream's head: the notes are VERIFIED by the decoder, never trusted.
struct CkRec { // while encoding: words emitted so far; finish_ckpt turns that into the decoder's word position
  fgmm_ckpt *out = nullptr;
  int64_t stride = 0, n_out = 0;
  inline void note(const Enc &e, const uint32_t *end, int64_t i) { // the encoder has just encoded symbol i (i % stride == 0, i > 0)
    fgmm_ckpt &c = out[i / stride - 1];
    c.x = e.x;
    c.pos = (uint64_t)(end - e.ptr);
  }
  inline void finish(const Enc &e, const uint32_t *end) { // before the flush: e.ptr = start of the renormalisation words
    const uint64_t total = (uint64_t)(end - e.ptr);
    for (int64_t k = 0; k < n_out; ++k) out[k].pos = total - out[k].pos; // words the decoder has read before that symbol
  }
};
static inline int64_t ckpt_count(int64_t n, int64_t stride) { return stride > 0 && n > 0 ? (n - 1) / stride : 0; }

static int finish_stream(Enc &e, uint32_t *end, uint8_t **out, size_t *out_len, BytesTo to = {}) { // Rans64EncFlush + copy out
  e.ptr -= 2;
  e.ptr[0] = (uint32_t)(e.x >> 0);
  e.ptr[1] = (uint32_t)(e.x >> 32);
  const size_t nbytes = (size_t)(end - e.ptr) * sizeof(uint32_t);
  uint8_t *o = (uint8_t *)(to.sink ? to.sink->alloc(to.sink->user, to.item, nbytes) : malloc(nbytes)); // (the caller's storage: fgmm_sink)
  if (!o) return FGMM_ERR_NOMEM;
  memcpy(o, e.ptr, nbytes);
  *out = o;
  *out_len = nbytes;
  return FGMM_OK;
}

int rans_encode_symtab(const uint32_t *packed, const int32_t *symbols, int64_t n, int64_t n_bypass_hint,
                       uint8_t **out, size_t *out_len) {
  return rans_encode_symtab_ckpt(packed, symbols, n, n_bypass_hint, out, out_len, 0, nullptr);
}

// the same, noting a checkpoint every `stride` symbols (a power of two; 0: none) into ckpt[ckpt_count(n, stride)]
int rans_encode_symtab_ckpt(const uint32_t *packed, const int32_t *symbols, int64_t n, int64_t n_bypass_hint, uint8_t **out,
                            size_t *out_len, int64_t stride, fgmm_ckpt *ckpt, BytesTo to) {
  std::call_once(g_rcp_once, init_rcp);
  if (n < 0 || !out || !out_len || (n > 0 && !packed) || stride < 0 || (stride & (stride - 1)) || (ckpt_count(n, stride) && !ckpt))
    return FGMM_ERR_INVALID;
  const size_t nwords = encode_words(n, n_bypass_hint < 0 ? count_bypass(packed, n) : n_bypass_hint);
  // worst-case sized scratch, kept per thread: a fresh 600 KB malloc per stream is an mmap + page faults + munmap
  static thread_local std::vector<uint32_t> scratch;
  try {
    if (scratch.size() < nwords) scratch.resize(nwords);
  } catch (const std::bad_alloc &) {
    return FGMM_ERR_NOMEM;
  }
  uint32_t *const end = scratch.data() + nwords;
  Enc e{kRansL, end}; // Rans64EncInit
  CkRec ck{ckpt, stride, ckpt_count(n, stride)};
  const int64_t seg = ck.n_out ? stride : (n > 0 ? n : 1); // the walk in runs that end on a checkpoint
  for (int64_t hi = n; hi > 0;) {
    const int64_t lo = (hi - 1) / seg * seg;
    for (int64_t i = hi - 1; i >= lo; --i) { // reversed _syms (rans_interface.cpp:569)
      if ((i & 15) == 15) __builtin_prefetch(packed + i - 512); // the table was just DMA-written: not in any cache
      encode_entry(e, packed[i], symbols, i);
    }
    if (ck.n_out && lo > 0) ck.note(e, end, lo);
    hi = lo;
  }
  ck.finish(e, end);
  return finish_stream(e, end, out, out_len, to);
}

// the same from a table in segments (SegTable): the walk is backwards, so it starts in the last segment and asks for each one
// before it enters it - the batched encoder's tables cross PCIe tail first and the encoders follow the landing
int rans_encode_symtab_segs(const SegTable &t, const int32_t *symbols, int64_t n, int64_t n_bypass_hint, uint8_t **out, size_t *out_len,
                            int64_t stride, fgmm_ckpt *ckpt, BytesTo to) {
  std::call_once(g_rcp_once, init_rcp);
  if (n < 0 || !out || !out_len || n_bypass_hint < 0 || stride < 0 || (stride & (stride - 1)) || (ckpt_count(n, stride) && !ckpt) ||
      t.n_seg < 1 || t.n_seg > kEncSegs || t.seg_len < 1 || n > (int64_t)t.n_seg * t.seg_len)
    return FGMM_ERR_INVALID;
  const size_t nwords = encode_words(n, n_bypass_hint);
  static thread_local std::vector<uint32_t> scratch;
  try {
    if (scratch.size() < nwords) scratch.resize(nwords);
  } catch (const std::bad_alloc &) {
    return FGMM_ERR_NOMEM;
  }
  uint32_t *const end = scratch.data() + nwords;
  Enc e{kRansL, end}; // Rans64EncInit
  CkRec ck{ckpt, stride, ckpt_count(n, stride)};
  const uint64_t smask = ck.n_out ? (uint64_t)stride - 1 : ~0ull;
  for (int64_t hi = n; hi > 0;) {
    const int sg = (int)((hi - 1) / t.seg_len);
    const int64_t lo = (int64_t)sg * t.seg_len;
    if (!t.seg[sg]) return FGMM_ERR_INVALID;
    if (t.wait) {
      const int rc = t.wait(t.arg, sg);
      if (rc != FGMM_OK) return rc;
    }
    const uint32_t *const p = t.seg[sg] - lo; // p[i] for lo <= i < hi
    for (int64_t i = hi - 1; i >= lo; --i) {  // reversed _syms (rans_interface.cpp:569)
      if ((i & 15) == 15 && i - 512 >= lo) __builtin_prefetch(p + i - 512); // the table was just DMA-written: not in any cache
      encode_entry(e, p[i], symbols, i);
      if (__builtin_expect(((uint64_t)i & smask) == 0 && i > 0, 0)) ck.note(e, end, i);
    }
    hi = lo;
  }
  ck.finish(e, end);
  return finish_stream(e, end, out, out_len, to);
}

// Up to four independent bitstreams coded by one thread, symbol by symbol in turn.  A stream's state update is a chain
// of ~11 dependent cycles per symbol; several chains fill the core's issue slots: 1.47 ns/symbol with two against 2.4 for
// one stream alone (Zen 5, scripts/enc_ilp.cpp).  Each stream's output is exactly what rans_encode_symtab gives.
template <int N>
static int encode_ways(const uint32_t *const *packed, const int32_t *const *symbols, const int64_t *n, const int64_t *n_bypass_hint,
                       uint8_t ***out, size_t **out_len, int64_t stride, fgmm_ckpt *const *ckpt, const BytesTo *to) {
  size_t nwords[N], total = 0;
  for (int k = 0; k < N; ++k) {
    if (n[k] < 0 || !out[k] || !out_len[k] || (n[k] > 0 && !packed[k])) return FGMM_ERR_INVALID;
    nwords[k] = encode_words(n[k], n_bypass_hint[k] < 0 ? count_bypass(packed[k], n[k]) : n_bypass_hint[k]);
    total += nwords[k];
  }
  static thread_local std::vector<uint32_t> scratch;
  try {
    if (scratch.size() < total) scratch.resize(total);
  } catch (const std::bad_alloc &) {
    return FGMM_ERR_NOMEM;
  }
  uint32_t *end[N];
  Enc e[N];
  CkRec ck[N];
  int64_t i[N], common = INT64_MAX;
  {
    uint32_t *at = scratch.data();
    for (int k = 0; k < N; ++k) {
      at += nwords[k];
      end[k] = at;
      e[k] = Enc{kRansL, at};
      i[k] = n[k] - 1;
      common = std::min(common, n[k]);
      ck[k] = CkRec{ckpt ? ckpt[k] : nullptr, stride, ckpt && ckpt[k] ? ckpt_count(n[k], stride) : 0};
    }
  }
  // (a checkpoint is due when a stream has just encoded a symbol whose index is a non-zero multiple of the stride: one
  // well-predicted test per symbol; the stride is a power of two)
  const uint64_t smask = stride > 0 ? (uint64_t)stride - 1 : ~0ull;
  for (int64_t s = 0; s < common; ++s) { // all streams have a symbol left
    if ((s & 15) == 0)
      for (int k = 0; k < N; ++k) __builtin_prefetch(packed[k] + i[k] - 512); // the tables were just DMA-written
#pragma unroll
    for (int k = 0; k < N; ++k) {
      encode_entry(e[k], packed[k][i[k]], symbols[k], i[k]);
      if (__builtin_expect(((uint64_t)i[k] & smask) == 0 && ck[k].n_out && i[k] > 0, 0)) ck[k].note(e[k], end[k], i[k]);
      --i[k];
    }
  }
  for (int k = 0; k < N; ++k) // the longer streams' remainders
    for (; i[k] >= 0; --i[k]) {
      encode_entry(e[k], packed[k][i[k]], symbols[k], i[k]);
      if (__builtin_expect(((uint64_t)i[k] & smask) == 0 && ck[k].n_out && i[k] > 0, 0)) ck[k].note(e[k], end[k], i[k]);
    }
  for (int k = 0; k < N; ++k) ck[k].finish(e[k], end[k]);
  int rc = FGMM_OK;
  for (int k = 0; k < N && rc == FGMM_OK; ++k)
    if ((rc = finish_stream(e[k], end[k], out[k], out_len[k], to ? to[k] : BytesTo{})) != FGMM_OK)
      for (int q = 0; q < k; ++q) {
        if (!to || !to[q].sink) free(*out[q]); // (a sink's storage is the caller's)
        *out[q] = nullptr;
      }
  return rc;
}

int rans_encode_symtab_ways(int ways, const uint32_t *const *packed, const int32_t *const *symbols, const int64_t *n,
                            const int64_t *n_bypass_hint, uint8_t ***out, size_t **out_len, int64_t stride, fgmm_ckpt *const *ckpt, const BytesTo *to) {
  std::call_once(g_rcp_once, init_rcp);
  if (stride < 0 || (stride & (stride - 1))) return FGMM_ERR_INVALID;
  switch (ways) {
  case 1: return rans_encode_symtab_ckpt(packed[0], symbols[0], n[0], n_bypass_hint[0], out[0], out_len[0], ckpt && ckpt[0] ? stride : 0, ckpt ? ckpt[0] : nullptr, to ? to[0] : BytesTo{});
  case 2: return encode_ways<2>(packed, symbols, n, n_bypass_hint, out, out_len, stride, ckpt, to);
  case 3: return encode_ways<3>(packed, symbols, n, n_bypass_hint, out, out_len, stride, ckpt, to);
  case 4: return encode_ways<4>(packed, symbols, n, n_bypass_hint, out, out_len, stride, ckpt, to);
  default: return FGMM_ERR_INVALID;
  }
}

int rans_encode_symtab2(const uint32_t *const packed[2], const int32_t *const symbols[2], const int64_t n[2],
                        const int64_t n_bypass_hint[2], uint8_t **out[2], size_t *out_len[2]) {
  return rans_encode_symtab_ways(2, packed, symbols, n, n_bypass_hint, out, out_len, 0, nullptr);
}

namespace {

struct Dec {
  uint64_t x;
  const uint32_t *ptr;
  const uint32_t *end;
  bool underrun = false;
  inline uint32_t next_word() {
    if (__builtin_expect(ptr >= end, 0)) {
      underrun = true;
      return 0;
    }
    return *ptr++;
  }
  inline void advance(uint32_t start, uint32_t freq) { // Rans64DecAdvance(start, freq, 16)
    uint64_t xx = (uint64_t)freq * (x >> kPrec) + (x & 0xFFFFu) - start;
    if (xx < kRansL) xx = (xx << 32) | next_word();
    x = xx;
  }
  inline uint32_t get_bits() { // Rans64DecGetBits(4)
    const uint32_t val = (uint32_t)(x & kMaxBypassVal);
    uint64_t xx = x >> kBypassBits;
    if (xx < kRansL) xx = (xx << 32) | next_word();
    x = xx;
    return val;
  }
  inline int32_t bypass() { // rans_interface.cpp:809-824
    advance(kMaxCdf, 1);
    int32_t val = (int32_t)get_bits();
    int32_t nn = val;
    while (val == (int32_t)kMaxBypassVal && !underrun) {
      val = (int32_t)get_bits();
      nn += val;
    }
    uint32_t raw = 0;
    for (int j = 0; j < nn && !underrun; ++j) raw |= get_bits() << ((j * kBypassBits) & 31);
    return (int32_t)raw;
  }
};

// virtual full table F[v], v in [-max_bs, max_bs+1], rebuilt from a row held as plain uint16 entries
struct Row {
  const uint16_t *row;
  int32_t a;
  int32_t cnt;
  inline uint32_t F(int32_t v) const {
    const int32_t idx = v - a;
    if (idx < 0) return 0;
    return row[idx >= cnt ? cnt - 1 : idx];
  }
};

// literal replay of rans_interface.cpp:826-877 on F
inline int32_t bisect_reference(const Row &r, uint32_t cum_freq, int32_t max_bs, uint32_t *start, uint32_t *freq) {
  int32_t s_bs = -max_bs, e_bs = max_bs, mid = 0;
  uint32_t c1 = 0, c2 = 0;
  while (s_bs <= e_bs) {
    mid = s_bs + (e_bs - s_bs) / 2;
    c1 = r.F(mid);
    c2 = r.F(mid + 1);
    if (c1 <= cum_freq && c2 > cum_freq) break;
    else if (c1 > cum_freq) e_bs = mid - 1;
    else s_bs = mid + 1;
  }
  c1 = r.F(mid);
  c2 = r.F(mid + 1);
  uint32_t pmf = (c2 - c1) & 0xFFFFu;
  if (pmf == 0) pmf = 1; // :866-868 (the start adjustment at :869 cannot trigger: c1 <= 65535)
  *start = c1;
  *freq = pmf;
  return mid;
}

// first index j in [0, cnt) with row[j] > cf, or cnt; row is non-decreasing; reads up to 30 bytes past the row.
// One vector at a time with an early exit: measured faster than four vectors at a time without a data-dependent branch
// (the compares of the vectors are independent and run ahead anyway; rounds 1 and 2, scripts/host_bench.py).
// Rows beyond 256 entries (wide items of the generic path, up to 2^20 entries: max_bs comes out of a bitstream's side
// information) are first narrowed by bisection to a window of at most 256 entries: the cost per symbol stays logarithmic in
// the row length whatever a stream announces.
__attribute__((target("avx2"))) inline int32_t upper_bound_u16(const uint16_t *row, int32_t cnt, uint32_t cf) {
  const __m256i bias = _mm256_set1_epi16((short)0x8000);
  const __m256i key = _mm256_set1_epi16((short)(cf ^ 0x8000u));
  int32_t k0 = 0;
  if (__builtin_expect(cnt > 256, 0)) {
    int32_t lo = 0, hi = cnt; // every entry before lo is <= cf, the first entry > cf (if any) lies in [lo, hi)
    while (hi - lo > 256) {
      const int32_t mid = lo + ((hi - lo) >> 1);
      if (row[mid] <= cf) lo = mid + 1; else hi = mid + 1;
    }
    k0 = lo;
  }
  for (int32_t k = k0; k < cnt; k += 16) {
    const __m256i v = _mm256_xor_si256(_mm256_loadu_si256((const __m256i *)(row + k)), bias);
    const uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpgt_epi16(v, key));
    if (m) {
      const int32_t j = k + (int32_t)(__builtin_ctz(m) >> 1);
      return j < cnt ? j : cnt;
    }
  }
  return cnt;
}

// ---- Elias-Fano rows (format v5): one bit string: HB = cnt + (65536 >> l) bits of unary high parts (bit (E_j >> l) + j
// set for entry j), then, from the next byte boundary LB on, the low l bits of every entry.  Bucket b (the entries whose
// high part is b) is a run of ones followed by one zero; zero #b sits at bit b + (number of entries with high part <= b).
// The unary part is read 64 bits at a time (rows are only 2-byte aligned); bits from HB on - the low parts, the next row -
// read as ones ("no zero there").
struct EfRow {
  const uint8_t *p;
  int32_t cnt;
  uint32_t l, HB, LB;
  inline int32_t words() const { return (int32_t)((HB + 63u) >> 6); }
  inline uint64_t operator[](int64_t w) const {
    uint64_t v;
    memcpy(&v, p + 8 * w, 8);
    const int64_t rem = (int64_t)HB - 64 * w; // valid bits of this word
    if (__builtin_expect(rem < 64, 0)) v |= rem <= 0 ? ~0ull : (~0ull << rem);
    return v;
  }
  inline uint32_t low(int32_t j) const {
    const uint32_t b = LB + (uint32_t)j * l;
    uint32_t v;
    memcpy(&v, p + (b >> 3), 4);
    return (v >> (b & 7u)) & ((1u << l) - 1u);
  }
};
// position of the k-th (0-based) ZERO bit of the unary part, k < 65536 >> l (exists in a well-formed row); -1 if not
__attribute__((target("bmi2,popcnt"))) inline int32_t ef_select0(const EfRow &r, uint32_t k) {
  uint32_t wsel = 0, zbase = 0, acc = 0;
  const int32_t U = r.words();
  for (int32_t w = 0; w < U; ++w) {
    acc += (uint32_t)__builtin_popcountll(~r[w]);
    const bool le = acc <= k;
    wsel += le;
    zbase = le ? acc : zbase;
  }
  if (__builtin_expect((int32_t)wsel >= U, 0)) return -1; // malformed row
  return (int32_t)(wsel * 64u + (uint32_t)__builtin_ctzll(_pdep_u64(1ull << (k - zbase), ~r[wsel])));
}
// position of the k-th (0-based) ONE bit, k < cnt (only used to expand a row for the bisection replay); -1 if not found
__attribute__((target("bmi2,popcnt"))) inline int32_t ef_select1(const EfRow &r, uint32_t k) {
  int32_t base = 0;
  const int32_t U = r.words();
  for (int32_t w = 0; w < U; ++w, base += 64) {
    uint64_t o = r[w];
    const int64_t rem = (int64_t)r.HB - base;
    if (rem < 64) o &= (1ull << rem) - 1ull; // the forced ones are not entries (rem >= 1 here)
    const uint32_t c = (uint32_t)__builtin_popcountll(o);
    if (k < c) return base + (int32_t)__builtin_ctzll(_pdep_u64(1ull << k, o));
    k -= c;
  }
  return -1;
}
// Bracket search: 1 with (*j, *start, *freq) such that E[j-1] = start <= cf < E[j]; 0 when no entry pair brackets cf
// (j would be 0 or cnt) - the caller replays the reference's bisection; -1 for a malformed row (hostile table).
// zero_before: an (unstored) zero edge precedes the row, so cf below the first entry is the interval [0, E_0): j = 0.
// Rows of up to 48 entries (12 low bits, 16 buckets): the unary part is ONE 64-bit word and everything between cf and
// (start, freq) is straight-line code on it - the decoder's dependency chain runs through this search, a mispredicted
// branch costs as much as the whole of it, and so do the extra instructions of a general search.
__attribute__((target("bmi2,popcnt,lzcnt"))) inline int ef_bracket64(const EfRow &r, uint32_t cf, bool zero_before, int32_t *jout,
                                                                    uint32_t *start, uint32_t *freq) {
  constexpr uint32_t l = 12, m = 0xFFFu;
  uint64_t w;
  memcpy(&w, r.p, 8);
  const uint32_t HB = r.HB; // 30 .. 64
  const uint64_t vm = _bzhi_u64(~0ull, HB), O = w & vm, Z = ~w & vm; // valid bits; entries; the zeros that close the buckets
  const uint32_t h = cf >> l, lcf = cf & m;
  // first bit of bucket h = one past the zero that closes bucket h - 1; with a zero imagined at bit -1 that is zero #h of
  // Z << 1 | 1 (the zero that closes the last bucket is shifted out: never asked for)
  const uint32_t s = (uint32_t)_tzcnt_u64(_pdep_u64(1ull << h, (Z << 1) | 1ull)); // 64 when the row is malformed
  if (__builtin_expect(s >= HB, 0)) return -1;
  const int32_t lo = (int32_t)s - (int32_t)h;                       // entries with a high part < h
  const int32_t run = (int32_t)_tzcnt_u64(~((w | ~vm) >> s));        // size of bucket h (the shifted-in bits end any run)
  if (__builtin_expect(lo < 0 || lo + run > r.cnt || s + (uint32_t)run >= HB, 0)) return -1;
  // entries of bucket h whose low part is <= lcf (low parts ascend inside a bucket): four fields at once
  const uint32_t b = r.LB + (uint32_t)lo * l;
  uint64_t v;
  memcpy(&v, r.p + (b >> 3), 8);
  v >>= (b & 7u);
  // (no && here: the compiler turns those into branches on run, which the predictor cannot learn)
  const uint32_t le = (uint32_t)(((uint32_t)v & m) <= lcf) | ((uint32_t)(((uint32_t)(v >> l) & m) <= lcf) << 1) |
                      ((uint32_t)(((uint32_t)(v >> (2 * l)) & m) <= lcf) << 2) | ((uint32_t)(((uint32_t)(v >> (3 * l)) & m) <= lcf) << 3);
  int32_t c = __builtin_popcount(le & _bzhi_u32(15u, (uint32_t)run)); // among the first min(run, 4) entries of the bucket
  if (__builtin_expect(run > 4 && c == 4, 0)) { // the two tail buckets of a row hold many entries and are rarely asked for
    int32_t hi = run; // low(lo + c - 1) <= lcf < low(lo + hi)
    while (c < hi) {
      const int32_t mid = (c + hi) >> 1;
      if (r.low(lo + mid) <= lcf) c = mid + 1; else hi = mid;
    }
  }
  const int32_t j = lo + c;
  if (__builtin_expect((j < 1 && !zero_before) || j >= r.cnt, 0)) return 0;
  // Entries j - 1 and j in full.  Their high parts are h when they lie in bucket h, else those of the entries next to the
  // bucket - which depend on (s, run) only and are ready by the time c is; their low parts are fields of v or the one before.
  const uint32_t pos_prev = 63u - (uint32_t)_lzcnt_u64(_bzhi_u64(O, s));                         // entry lo - 1 (garbage if none: unused)
  const uint32_t pos_next = s + (uint32_t)run + 1u + (uint32_t)_tzcnt_u64((O >> (s + (uint32_t)run)) >> 1); // entry lo + run
  const uint32_t hp_prev = pos_prev - (uint32_t)(lo - 1), hp_next = pos_next - (uint32_t)(lo + run);
  uint32_t low_j, low_f;
  if (__builtin_expect(c <= 4, 1)) {
    low_j = c < 4 ? (uint32_t)(v >> ((uint32_t)c * l)) & m : r.low(j);
    low_f = (uint32_t)(v >> (((uint32_t)(c - 1) & 3u) * l)) & m; // entry j - 1 when it is one of the fields
  } else {
    low_j = r.low(j);
    low_f = r.low(j - 1);
  }
  const uint32_t low_before = r.low(lo > 0 ? lo - 1 : 0); // entry lo - 1 (loaded whether needed or not)
  const uint32_t in0 = 0u - (uint32_t)(c != 0), in1 = 0u - (uint32_t)(c < run); // masks: entry j - 1 / j lies in bucket h
  const uint32_t e1 = (((h & in1) | (hp_next & ~in1)) << l) | low_j;
  const uint32_t e0 = ((((h & in0) | (hp_prev & ~in0)) << l) | ((low_f & in0) | (low_before & ~in0))) & (0u - (uint32_t)(j >= 1)); // j = 0: the implied zero edge
  *jout = j;
  *start = e0;
  *freq = (e1 - e0) & 0xFFFFu;
  return 1;
}

// (Rows of 49 .. 256 entries - 8 low bits, an unary part of up to eight words - were also given a straight-line search on the
// popcount prefixes of those words, select0 / select1 by one vector compare + pdep each: correct, and SLOWER than the general
// search below on the decode hosts' EPYC 9575F - 14.5 against 12.7 ns/symbol over a Kodak half at ef_min 49, 16.5 against 14.6
// at 14: twice the instructions for the two mispredictions it saves.  Commit ee79904, profiles/r03_host_decoder_epyc9575f.txt.)
__attribute__((target("bmi2,popcnt,sse4.1"))) inline int ef_bracket(const EfRow &r, uint32_t cf, bool zero_before, int32_t *jout,
                                                                   uint32_t *start, uint32_t *freq) {
  if (r.l == 12 && r.HB <= 64) return ef_bracket64(r, cf, zero_before, jout, start, freq);
  const uint32_t h = cf >> r.l, lcf = cf & ((1u << r.l) - 1u);
  const int32_t nbits = (int32_t)r.HB;
  int32_t p_prev = -1; // zero that closes bucket h-1
  if (h) {
    p_prev = ef_select0(r, h - 1);
    if (__builtin_expect(p_prev < 0 || p_prev >= nbits, 0)) return -1;
  }
  const int32_t lo = p_prev + 1 - (int32_t)h; // entries with a high part < h
  if (__builtin_expect(lo < 0 || lo > r.cnt, 0)) return -1;
  // length of the run of ones that starts at bit p_prev + 1  (= size of bucket h)
  int32_t run = 0;
  {
    uint32_t s = (uint32_t)(p_prev + 1);
    for (;;) {
      if (__builtin_expect((int32_t)s >= nbits, 0)) return -1; // no closing zero: malformed
      const uint32_t b = s & 63u;
      const uint64_t t = ~(r[s >> 6] >> b); // shifted-in zeros become ones: the run ends at the word end at the latest
      const int32_t len = t ? (int32_t)__builtin_ctzll(t) : 64;
      run += len;
      if (__builtin_expect(len < (int32_t)(64u - b), 1)) break;
      s += (uint32_t)len; // the run continues in the next word (rare)
    }
  }
  if (__builtin_expect(lo + run > r.cnt || p_prev + 1 + run >= nbits, 0)) return -1;
  // entries of bucket h whose low part is <= lcf (low parts ascend inside a bucket)
  int32_t c;
  if (r.l == 8 && __builtin_expect(run <= 16, 1)) { // the low parts are bytes
    const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(r.p + (r.LB >> 3) + lo));
    const __m128i key = _mm_set1_epi8((char)lcf);
    const uint32_t le = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_min_epu8(v, key), v));
    c = __builtin_popcount(le & ((1u << run) - 1u));
  } else if (run <= 4) {
    c = 0;
    while (c < run && r.low(lo + c) <= lcf) ++c;
  } else {
    c = 0;
    int32_t hi = run; // low(lo + c - 1) <= lcf < low(lo + hi)
    while (c < hi) {
      const int32_t mid = (c + hi) >> 1;
      if (r.low(lo + mid) <= lcf) c = mid + 1; else hi = mid;
    }
  }
  const int32_t j = lo + c;
  if (__builtin_expect((j < 1 && !zero_before) || j >= r.cnt, 0)) return 0;
  uint32_t e0, e1;
  if (j < 1) {
    e0 = 0; // the implied zero edge
  } else if (c > 0) {
    e0 = (h << r.l) | r.low(j - 1);
  } else { // previous entry lives in an earlier bucket: the highest one below bit p_prev
    int32_t w = p_prev >> 6;
    uint64_t m = r[w] & ((1ull << (p_prev & 63)) - 1ull);
    while (!m) {
      if (__builtin_expect(--w < 0, 0)) return -1;
      m = r[w];
    }
    const int32_t pos = w * 64 + 63 - (int32_t)__builtin_clzll(m);
    e0 = ((uint32_t)(pos - (j - 1)) << r.l) | r.low(j - 1);
  }
  if (c < run) {
    e1 = (h << r.l) | r.low(j);
  } else { // next entry lives in a later bucket: the lowest one above the zero that closes bucket h
    const int32_t pz = p_prev + 1 + run;
    int32_t w = (pz + 1) >> 6;
    if (__builtin_expect(pz + 1 >= nbits, 0)) return -1;
    uint64_t m = r[w] & ~((1ull << ((pz + 1) & 63)) - 1ull);
    for (;;) {
      const int64_t rem = (int64_t)nbits - 64 * (int64_t)w;
      if (rem < 64) m &= (1ull << rem) - 1ull; // not the forced ones past the unary part
      if (m) break;
      if (__builtin_expect(++w >= r.words(), 0)) return -1;
      m = r[w];
    }
    const int32_t pos = w * 64 + (int32_t)__builtin_ctzll(m);
    e1 = ((uint32_t)(pos - j) << r.l) | r.low(j);
  }
  *jout = j;
  *start = e0;
  *freq = (e1 - e0) & 0xFFFFu;
  return 1;
}

} // namespace

// The decoder proper, resumable piece by piece (a host worker may follow several bitstreams as their tables land):
// walks the latents in order; per latent the header gives (a, cnt, nonmono), the row follows in the pool.  Every row
// extent is checked against the piece's row area before it is touched (a table is trusted to be well-formed only as far
// as memory safety does not depend on it).
int TabDecoder::begin(const uint8_t *enc, size_t enc_len, const TabView *view, int64_t n_, int32_t max_bs_, int32_t *out_) {
  tv = view;
  n = n_;
  max_bs = max_bs_;
  out = out_;
  i = 0;
  next_piece = 0;
  rc = FGMM_OK;
  copy = nullptr;
  scratch = nullptr;
  scratch_cap = 0;
  if (n < 0 || !enc || (n > 0 && (!out || !tv || !tv->piece || tv->npiece < 1))) return rc = FGMM_ERR_INVALID;
  if (tv && tv->hdr_form != 2 && tv->hdr_form != 4 && tv->hdr_form != 8) return rc = FGMM_ERR_INVALID;
  if (enc_len < 8 || (enc_len & 3)) return rc = FGMM_ERR_STREAM;
  const uint32_t *words;
  if (reinterpret_cast<uintptr_t>(enc) & 3) { // Python bytes are aligned in practice; stay safe
    copy = (uint32_t *)malloc(enc_len);
    if (!copy) return rc = FGMM_ERR_NOMEM;
    memcpy(copy, enc, enc_len);
    words = copy;
  } else {
    words = reinterpret_cast<const uint32_t *>(enc);
  }
  x = (uint64_t)words[0] | ((uint64_t)words[1] << 32); // Rans64DecInit
  ptr = base_ = words + 2;
  end_ = words + enc_len / 4;
  return FGMM_OK;
}

#ifndef FGMM_PREFETCH_AHEAD
#define FGMM_PREFETCH_AHEAD 1024 // bytes: the rows were just DMA-written, they are in no cache
#endif
namespace {

// One decoder's walk through one piece, a latent per step().
struct PieceRun {
  TabDecoder *td = nullptr;
  const TabPiece *pc = nullptr;
  Dec dec{};
  int64_t i = 0, i_beg = 0, i_end = 0, b_end = 0, blk = 0, tl = 1, W = 0;
  const uint8_t *rowp = nullptr, *rows_end = nullptr;
  int32_t max_bs = 0;
  int hdr_form = 4;
  uint32_t ef_min = kTabEfMin;
  int rc = FGMM_OK;

  // false: nothing to do (rc says whether that is an error)
  bool begin(TabDecoder &d, int k) {
    td = &d;
    rc = d.rc;
    if (rc != FGMM_OK) return false;
    if (k != d.next_piece || !d.tv || k >= d.tv->npiece) {
      rc = d.rc = FGMM_ERR_INVALID;
      return false;
    }
    ++d.next_piece;
    pc = &d.tv->piece[k];
    i = i_beg = d.i;
    i_end = pc->end < d.n ? pc->end : d.n;
    if (i >= i_end) return false;
    dec.x = d.x;
    dec.ptr = d.ptr;
    dec.end = d.end_;
    max_bs = d.max_bs;
    W = 2 * (int64_t)max_bs + 2;
    hdr_form = d.tv->hdr_form;
    ef_min = d.tv->ef_min;
    // a search may read up to 32 bytes past its row: rows must end that far before the end of the area
    rows_end = pc->rows + (pc->rows_len >= 32 ? pc->rows_len - 32 : 0);
    rowp = pc->rows; // sequential placement: a running sum, never stored
    tl = pc->blk_off ? (d.tv->tl > 0 ? d.tv->tl : 1) : i_end - i_beg;
    blk = 0;
    b_end = i; // the first step opens block 0
    return true;
  }
  // The same for a SEGMENT of a checkpointed stream: piece k from latent d.i on (anywhere inside the piece), at most up to
  // i_stop; d.x / d.ptr hold the checkpoint's coder state.  Rows inside a block are sequential, so the rows of the block's
  // latents before d.i are skipped header by header (fewer than tl of them).  Block-placed tables only.
  bool begin_at(TabDecoder &d, int k, int64_t i_stop) {
    td = &d;
    rc = d.rc;
    if (rc != FGMM_OK) return false;
    if (!d.tv || k < 0 || k >= d.tv->npiece || !d.tv->piece[k].blk_off || d.tv->tl <= 0) {
      rc = d.rc = FGMM_ERR_INVALID;
      return false;
    }
    pc = &d.tv->piece[k];
    i_beg = k ? d.tv->piece[k - 1].end : 0;
    i_end = std::min(std::min<int64_t>(pc->end, d.n), i_stop);
    if (d.i < i_beg || d.i >= i_end) return false;
    dec.x = d.x;
    dec.ptr = d.ptr;
    dec.end = d.end_;
    max_bs = d.max_bs;
    W = 2 * (int64_t)max_bs + 2;
    hdr_form = d.tv->hdr_form;
    ef_min = d.tv->ef_min;
    rows_end = pc->rows + (pc->rows_len >= 32 ? pc->rows_len - 32 : 0);
    tl = d.tv->tl;
    blk = (d.i - i_beg) / tl;
    i = b_end = i_beg + blk * tl; // the first step / skip opens this block
    while (rc == FGMM_OK && i < d.i) skip();
    return rc == FGMM_OK;
  }
  // passes latent i without decoding it: its row's extent from its header
  void skip() {
    if (i == b_end) {
      open_block();
      if (rc != FGMM_OK) return;
    }
    const int64_t li = i - i_beg;
    int64_t cnt;
    uint32_t nonmono = 0;
    if (hdr_form == 2) {
      cnt = static_cast<const uint16_t *>(pc->hdr)[li] >> 8;
      if (cnt == kHdr2Escape) {
        if (rows_end - rowp < 4) { rc = FGMM_ERR_INVALID; return; }
        uint32_t h;
        memcpy(&h, rowp, 4);
        rowp += 4;
        cnt = tab_hdr_cnt(h);
        nonmono = tab_hdr_nonmono(h);
      }
    } else if (hdr_form == 4) {
      const uint32_t h = static_cast<const uint32_t *>(pc->hdr)[li];
      cnt = tab_hdr_cnt(h);
      nonmono = tab_hdr_nonmono(h);
    } else {
      const uint64_t h = static_cast<const uint64_t *>(pc->hdr)[li];
      cnt = (int64_t)((h >> 32) & 0x7FFFFFFFu);
      nonmono = (uint32_t)(h >> 63);
    }
    const uint64_t rbytes = tab_row_bytes((uint32_t)cnt, nonmono, ef_min);
    if (cnt < 1 || (uint64_t)(rows_end - rowp) < rbytes) { rc = FGMM_ERR_INVALID; return; }
    rowp += rbytes;
    ++i;
  }
  inline bool active() const { return rc == FGMM_OK && i < i_end; }
  void finish() { // the coder state goes back to the decoder (the next piece may run on another thread)
    td->x = dec.x;
    td->ptr = dec.ptr;
    td->i = i;
    if (rc != FGMM_OK) td->rc = rc;
  }

  inline void open_block() {
    b_end = i + tl < i_end ? i + tl : i_end;
    if (pc->blk_off) {
      rowp = pc->rows + 4 * (size_t)pc->blk_off[blk];
      if (__builtin_expect(rowp > rows_end, 0)) {
        rc = FGMM_ERR_INVALID;
        return;
      }
      // blocks lie in no particular order: the running prefetch of step() (1 KiB ahead of the row being searched) runs off
      // the end of this block into someone else's rows, so the head of the NEXT block is fetched here, a block ahead
      if (b_end < i_end) {
        const uint8_t *nx = pc->rows + 4 * (size_t)pc->blk_off[blk + 1];
        if (nx + FGMM_PREFETCH_AHEAD <= rows_end)
          for (int q = 0; q < FGMM_PREFETCH_AHEAD; q += 64) __builtin_prefetch(nx + q);
      }
    }
    ++blk;
  }

  // decodes latent i: the header gives (a, cnt, nonmono), the row follows at rowp
  __attribute__((always_inline)) inline void step() {
    if (__builtin_expect(i == b_end, 0)) {
      open_block();
      if (rc != FGMM_OK) return;
    }
    __builtin_prefetch(rowp + FGMM_PREFETCH_AHEAD);
    __builtin_prefetch(rowp + FGMM_PREFETCH_AHEAD + 64);
    int64_t a;
    int64_t cnt;
    uint32_t nonmono;
    const int64_t li = i - i_beg;
    if (hdr_form == 2) { // (a + max_bs) | cnt << 8; cnt 255: the row carries a 4-byte header
      const uint32_t c = static_cast<const uint16_t *>(pc->hdr)[li];
      a = (int64_t)(c & 0xFFu) - max_bs;
      cnt = c >> 8;
      nonmono = 0;
      if (__builtin_expect(cnt == kHdr2Escape, 0)) {
        if (rows_end - rowp < 4) { rc = FGMM_ERR_INVALID; return; }
        uint32_t h;
        memcpy(&h, rowp, 4);
        rowp += 4;
        a = tab_hdr_a(h);
        cnt = tab_hdr_cnt(h);
        nonmono = tab_hdr_nonmono(h);
      }
    } else if (hdr_form == 4) {
      const uint32_t h = static_cast<const uint32_t *>(pc->hdr)[li];
      a = tab_hdr_a(h);
      cnt = tab_hdr_cnt(h);
      nonmono = tab_hdr_nonmono(h);
    } else {
      const uint64_t h = static_cast<const uint64_t *>(pc->hdr)[li];
      a = (int32_t)(uint32_t)h;
      cnt = (int64_t)((h >> 32) & 0x7FFFFFFFu);
      nonmono = (uint32_t)(h >> 63);
    }
    // a row covers indices [a + max_bs, a + max_bs + cnt) of the W-entry virtual table
    if (__builtin_expect(cnt < 1 || a < -(int64_t)max_bs || a + max_bs + cnt > W, 0)) { rc = FGMM_ERR_INVALID; return; }
    const bool is_ef = tab_row_is_ef((uint32_t)cnt, nonmono, ef_min);
    const uint64_t rbytes = tab_row_bytes((uint32_t)cnt, nonmono, ef_min);
    if (__builtin_expect((uint64_t)(rows_end - rowp) < rbytes, 0)) { rc = FGMM_ERR_INVALID; return; }
    const bool zero_before = a > -(int64_t)max_bs; // rows start at their first non-zero edge; F[v < a] = 0 (v >= -max_bs exists)
    const uint8_t *row_bytes = rowp;
    rowp += rbytes;

    const uint32_t cf = (uint32_t)(dec.x & 0xFFFFu); // Rans64DecGet
    int32_t value;
    if (__builtin_expect(cf == kMaxCdf, 0)) {
      value = dec.bypass();
    } else {
      uint32_t start = 0, freq = 0;
      bool done = false;
      if (!is_ef) {
        const uint16_t *row = reinterpret_cast<const uint16_t *>(row_bytes);
        if (__builtin_expect(!nonmono, 1)) {
          const int32_t j = upper_bound_u16(row, (int32_t)cnt, cf);
          if (__builtin_expect(j >= 1 && j < cnt, 1)) { // row[j-1] <= cf < row[j]: the unique bracket
            start = row[j - 1];
            freq = (uint32_t)(row[j] - start) & 0xFFFFu;
            value = (int32_t)(a + j - 1);
            done = true;
          } else if (j == 0 && zero_before) { // 0 <= cf < row[0]: the symbol whose lower edge is the implied zero
            start = 0;
            freq = row[0];
            value = (int32_t)(a - 1);
            done = true;
          }
        }
        if (!done) value = bisect_reference(Row{row, (int32_t)a, (int32_t)cnt}, cf, max_bs, &start, &freq);
      } else {
        const uint32_t efl = tab_ef_l((uint32_t)cnt);
        const EfRow r{row_bytes, (int32_t)cnt, efl, tab_ef_hb((uint32_t)cnt, efl), tab_ef_lb((uint32_t)cnt, efl)};
        int32_t j;
        const int br = ef_bracket(r, cf, zero_before, &j, &start, &freq);
        if (__builtin_expect(br > 0, 1)) {
          value = (int32_t)(a + j - 1);
        } else if (br < 0) {
          rc = FGMM_ERR_INVALID;
          return;
        } else { // no interval contains cf: expand the row and replay the reference's bisection
          value = expand_and_bisect(r, a, cf, &start, &freq);
          if (rc != FGMM_OK) return;
        }
      }
      if (__builtin_expect(freq == 0, 0)) { rc = FGMM_ERR_INVALID; return; } // cannot come out of a well-formed row
      dec.advance(start, freq);
    }
    td->out[i] = value;
    if (__builtin_expect(dec.underrun, 0)) { rc = FGMM_ERR_STREAM; return; }
    ++i;
  }

  __attribute__((noinline)) int32_t expand_and_bisect(const EfRow &r, int64_t a, uint32_t cf, uint32_t *start, uint32_t *freq) {
    const int32_t cnt = r.cnt;
    if ((size_t)cnt > td->scratch_cap) {
      free(td->scratch);
      td->scratch_cap = (size_t)cnt + 64;
      td->scratch = (uint16_t *)malloc(td->scratch_cap * sizeof(uint16_t));
      if (!td->scratch) {
        td->scratch_cap = 0;
        rc = FGMM_ERR_NOMEM;
        return 0;
      }
    }
    for (int32_t q = 0; q < cnt; ++q) {
      const int32_t pos = ef_select1(r, (uint32_t)q);
      if (pos < q) {
        rc = FGMM_ERR_INVALID;
        return 0;
      }
      td->scratch[q] = (uint16_t)((((uint32_t)(pos - q)) << r.l) | r.low(q));
    }
    return bisect_reference(Row{td->scratch, (int32_t)a, cnt}, cf, max_bs, start, freq);
  }
};

} // namespace

// decodes the latents of piece k (pieces before it are done, its tables are on the host)
int TabDecoder::piece(int k) {
  PieceRun r;
  if (!r.begin(*this, k)) return r.rc;
  while (r.active()) r.step();
  r.finish();
  return rc;
}

// Latents [lo, hi) of a checkpointed stream, from the checkpoint (x0 = coder state before symbol lo, pos0 = renormalisation
// words read by then); every piece the range touches has landed.  *x1 / *pos1: where the coder stands after symbol hi - 1 -
// the caller compares that with the NEXT checkpoint: a segment that ends exactly there did the sequential decoder's work.
int TabDecoder::segment(int64_t lo, int64_t hi, uint64_t x0, uint64_t pos0, uint64_t *x1, uint64_t *pos1) {
  if (rc != FGMM_OK) return rc;
  if (!tv || lo < 0 || hi > n || lo > hi || pos0 > (uint64_t)(end_ - base_)) return rc = FGMM_ERR_INVALID;
  x = x0;
  ptr = base_ + pos0;
  i = lo;
  int k = 0;
  while (k < tv->npiece && tv->piece[k].end <= lo) ++k;
  for (; i < hi && k < tv->npiece && rc == FGMM_OK; ++k) {
    PieceRun r;
    if (!r.begin_at(*this, k, hi)) {
      if (r.rc != FGMM_OK) rc = r.rc;
      break;
    }
    while (r.active()) r.step();
    r.finish();
  }
  if (rc == FGMM_OK && i < hi) rc = FGMM_ERR_INVALID; // the pieces do not cover the range
  *x1 = x;
  *pos1 = (uint64_t)(ptr - base_);
  return rc;
}

// after the last piece: FGMM_OK only if every latent was decoded
int TabDecoder::finish() {
  if (rc == FGMM_OK && i < n) rc = FGMM_ERR_INVALID; // the pieces do not cover the latents
  free(copy);
  free(scratch);
  copy = nullptr;
  scratch = nullptr;
  return rc;
}

int rans_decode_tab(const uint8_t *enc, size_t enc_len, const TabView &tv, int64_t n, int32_t max_bs, int32_t *out) {
  TabDecoder td;
  int rc = td.begin(enc, enc_len, &tv, n, max_bs, out);
  for (int k = 0; rc == FGMM_OK && k < tv.npiece; ++k) {
    if (k > 0 && tv.wait && (rc = tv.wait(tv.arg, k)) != FGMM_OK) break;
    rc = td.piece(k);
  }
  const int rf = td.finish();
  return rc != FGMM_OK ? rc : rf;
}


// ============================================================================================================
// Table path — CompressAI's original table rANS, used for the `z` hyper-latent:
//   BufferedRansEncoder::encode_with_indexes (lists)   rans_interface.cpp:334-399
//   RansDecoder::decode_with_indexes / decode_stream   rans_interface.cpp:619-688, :894-956
// plus the buffered form shared with the GMM tables (BufferedRansEncoder::_syms, rans_interface.hpp:85).
// Integer only.  Unlike the reference (asserts compiled out) every index is validated.
// ============================================================================================================
struct SymBuf {
  // one entry of the reference's _syms: start | range << 16 | bypass << 32
  std::vector<uint64_t> e;
  inline void push(uint32_t start, uint32_t range, bool bypass) {
    e.push_back((uint64_t)(start & 0xFFFFu) | ((uint64_t)(range & 0xFFFFu) << 16) | ((uint64_t)bypass << 32));
  }
};

static int table_args_ok(const int32_t *cdfs, int64_t cdf_stride, int32_t n_cdfs, const int32_t *cdfs_sizes,
                         const int32_t *offsets) {
  if (!cdfs || !cdfs_sizes || !offsets || n_cdfs <= 0 || cdf_stride < 2) return 0;
  for (int32_t k = 0; k < n_cdfs; ++k)
    if (cdfs_sizes[k] < 2 || cdfs_sizes[k] > cdf_stride) return 0;
  return 1;
}

// Escape code of the table coder (rans_interface.cpp:373-397): the magnitude travels as raw 4-bit digits, least
// significant first, behind their count; a count of 15 or more continues in further digits of 15.
static void push_escape_digits(SymBuf &sb, uint32_t magnitude) {
  int digits = 0;
  for (uint32_t rest = magnitude; rest != 0 && digits < 8; rest >>= kBypassBits) ++digits;
  int count = digits;
  for (; count >= (int)kMaxBypassVal; count -= (int)kMaxBypassVal) sb.push(kMaxBypassVal, kMaxBypassVal + 1, true);
  sb.push((uint32_t)count, (uint32_t)count + 1, true);
  for (int d = 0; d < digits; ++d) {
    const uint32_t digit = (magnitude >> (d * kBypassBits)) & kMaxBypassVal;
    sb.push(digit, digit + 1, true);
  }
}

int symbuf_append_table(SymBuf &sb, const int32_t *symbols, const int32_t *indexes, int64_t n, const int32_t *cdfs,
                        int64_t cdf_stride, int32_t n_cdfs, const int32_t *cdfs_sizes, const int32_t *offsets) {
  if (n < 0 || (n > 0 && (!symbols || !indexes))) return FGMM_ERR_INVALID;
  if (n == 0) return FGMM_OK;
  if (!table_args_ok(cdfs, cdf_stride, n_cdfs, cdfs_sizes, offsets)) return FGMM_ERR_INVALID;
  sb.e.reserve(sb.e.size() + (size_t)n + 16);
  for (int64_t i = 0; i < n; ++i) {
    const int32_t t = indexes[i];
    if (t < 0 || t >= n_cdfs) return FGMM_ERR_INVALID;
    const int32_t *row = cdfs + (int64_t)t * cdf_stride;
    // Slots 0 .. last-1 of a table code in-range symbols; the last slot announces an escape.  An out-of-range symbol is
    // folded into one unsigned magnitude, odd for the ones below the table, even for the ones above (:350-363; the
    // reference computes in int32: identical while |symbol - offset| < 2^30).
    const int64_t last = (int64_t)cdfs_sizes[t] - 2, rel = (int64_t)symbols[i] - offsets[t];
    const bool below = rel < 0, above = rel >= last;
    const int64_t slot = (below || above) ? last : rel;
    sb.push((uint32_t)row[slot], (uint32_t)(row[slot + 1] - row[slot]), false); // :368-370
    if (below) push_escape_digits(sb, (uint32_t)(-2 * rel - 1));
    else if (above) push_escape_digits(sb, (uint32_t)(2 * (rel - last)));
  }
  return FGMM_OK;
}

// GMM symbol table (start | range << 16, range == 0 = escape) -> _syms entries, rans_interface.cpp:509-552
int symbuf_append_symtab(SymBuf &sb, const uint32_t *packed, const int32_t *symbols, int64_t n) {
  if (n < 0 || (n > 0 && !packed)) return FGMM_ERR_INVALID;
  sb.e.reserve(sb.e.size() + (size_t)n + 16);
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t ent = packed[i], freq = ent >> 16;
    if (freq) {
      sb.push(ent & 0xFFFFu, freq, false);
      continue;
    }
    const int32_t value = symbols ? symbols[i] : (int32_t)(int16_t)(uint16_t)(ent & 0xFFFFu);
    const uint32_t raw = (uint32_t)value;
    int nn = 0;
    for (uint32_t t = raw; t != 0; t >>= kBypassBits) ++nn;
    sb.push(kMaxCdf, 1, false);
    sb.push((uint32_t)nn, (uint32_t)nn + 1, true);
    for (int j = 0; j < nn; ++j) {
      const uint32_t nib = (raw >> (j * kBypassBits)) & kMaxBypassVal;
      sb.push(nib, nib + 1, true);
    }
  }
  return FGMM_OK;
}

// BufferedRansEncoder::flush, rans_interface.cpp:557-585
int symbuf_flush(SymBuf &sb, uint8_t **out, size_t *out_len) {
  std::call_once(g_rcp_once, init_rcp);
  if (!out || !out_len) return FGMM_ERR_INVALID;
  const size_t nwords = sb.e.size() + 16;
  uint32_t *buf = (uint32_t *)malloc(nwords * sizeof(uint32_t));
  if (!buf) return FGMM_ERR_NOMEM;
  uint32_t *const end = buf + nwords;
  Enc e{kRansL, end};
  for (size_t i = sb.e.size(); i-- > 0;) {
    const uint64_t v = sb.e[i];
    const uint32_t start = (uint32_t)(v & 0xFFFFu), range = (uint32_t)((v >> 16) & 0xFFFFu);
    if (v >> 32) e.put_bits(start);
    else if (range) e.put(start, range);
    else { // a zero-width table entry cannot be coded (the reference divides by zero): refuse
      free(buf);
      sb.e.clear();
      return FGMM_ERR_INVALID;
    }
  }
  sb.e.clear();
  e.ptr -= 2;
  e.ptr[0] = (uint32_t)(e.x >> 0);
  e.ptr[1] = (uint32_t)(e.x >> 32);
  const size_t nbytes = (size_t)(end - e.ptr) * sizeof(uint32_t);
  uint8_t *o = (uint8_t *)malloc(nbytes);
  if (!o) {
    free(buf);
    return FGMM_ERR_NOMEM;
  }
  memcpy(o, e.ptr, nbytes);
  free(buf);
  *out = o;
  *out_len = nbytes;
  return FGMM_OK;
}

struct DecStream { // RansDecoder's streaming state: _rans, _stream, _ptr (rans_interface.hpp:150-153)
  std::vector<uint32_t> words;
  Dec d;
};

int decstream_init(DecStream &ds, const uint8_t *enc, size_t enc_len) {
  if (!enc || enc_len < 8 || (enc_len & 3)) return FGMM_ERR_STREAM;
  ds.words.resize(enc_len / 4);
  memcpy(ds.words.data(), enc, enc_len);
  ds.d = Dec();
  ds.d.x = (uint64_t)ds.words[0] | ((uint64_t)ds.words[1] << 32);
  ds.d.ptr = ds.words.data() + 2;
  ds.d.end = ds.words.data() + ds.words.size();
  return FGMM_OK;
}

int decstream_decode(DecStream &ds, const int32_t *indexes, int64_t n, const int32_t *cdfs, int64_t cdf_stride,
                     int32_t n_cdfs, const int32_t *cdfs_sizes, const int32_t *offsets, int32_t *out) {
  if (n < 0 || (n > 0 && (!indexes || !out))) return FGMM_ERR_INVALID;
  if (n == 0) return FGMM_OK;
  if (!table_args_ok(cdfs, cdf_stride, n_cdfs, cdfs_sizes, offsets)) return FGMM_ERR_INVALID;
  Dec &d = ds.d;
  for (int64_t i = 0; i < n; ++i) {
    const int32_t k = indexes[i];
    if (k < 0 || k >= n_cdfs) return FGMM_ERR_INVALID;
    const int32_t *cdf = cdfs + (int64_t)k * cdf_stride;
    const int32_t size = cdfs_sizes[k], max_value = size - 2;
    const uint32_t cf = (uint32_t)(d.x & 0xFFFFu); // :648
    // std::lower_bound(cdf, cdf + size, cf + 1) - 1: the last s with cdf[s] <= cf   (:651-653)
    int32_t lo = 0, hi = size;
    while (lo < hi) {
      const int32_t mid = lo + (hi - lo) / 2;
      if ((uint32_t)cdf[mid] < cf + 1) lo = mid + 1; else hi = mid;
    }
    const int32_t s = lo - 1;
    if (s < 0 || s + 1 >= size) return FGMM_ERR_INVALID; // malformed table (cdf[0] != 0 or not reaching 2^16)
    d.advance((uint32_t)cdf[s], (uint32_t)(cdf[s + 1] - cdf[s])); // :656
    int32_t value = s;
    if (value == max_value) { // :660-682
      int32_t val = (int32_t)d.get_bits();
      int32_t n_bypass = val;
      while (val == (int32_t)kMaxBypassVal && !d.underrun) {
        val = (int32_t)d.get_bits();
        n_bypass += val;
      }
      int32_t raw_val = 0;
      for (int j = 0; j < n_bypass && !d.underrun; ++j)
        raw_val |= (int32_t)(d.get_bits() << ((j * kBypassBits) & 31));
      value = raw_val >> 1;
      if (raw_val & 1) value = -value - 1; else value += max_value;
    }
    out[i] = value + offsets[k]; // :684
    if (d.underrun) return FGMM_ERR_STREAM;
  }
  return FGMM_OK;
}

// compressai._CXX.pmf_to_quantized_cdf (ops.cpp:40-109; runs once per model; float -> integer, host), restated on
// FREQUENCIES: the reference edits the cumulative array in place (a symbol left without a count shifts every entry
// between itself and the donor by one); shifting cdf[a+1 .. b] by one is moving one count from symbol a to symbol b, so
// the same result is: scale the rounded masses to 2^precision (each floored, the last symbol takes what is left),
// then, symbol by symbol, give every empty one a count from the rarest symbol that can spare it (the first such on
// ties), and only then accumulate.
int pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf) {
  if (n <= 0 || !pmf || !cdf || precision < 1 || precision > 16) return FGMM_ERR_INVALID;
  const uint64_t full = 1ull << precision;
  std::vector<uint64_t> mass((size_t)n);
  uint64_t total = 0;
  for (int i = 0; i < n; ++i) {
    if (!(pmf[i] >= 0) || !std::isfinite(pmf[i])) return FGMM_ERR_INVALID;
    mass[(size_t)i] = (uint64_t)std::round(pmf[i] * (float)full);
    total += mass[(size_t)i];
  }
  if (total == 0) return FGMM_ERR_INVALID;
  std::vector<uint32_t> freq((size_t)n);
  uint64_t used = 0;
  for (int i = 0; i + 1 < n; ++i) {
    freq[(size_t)i] = (uint32_t)(full * mass[(size_t)i] / total);
    used += freq[(size_t)i];
  }
  freq[(size_t)n - 1] = (uint32_t)(full - used); // the reference pins the last cumulative entry to 2^precision
  for (int i = 0; i < n; ++i) {
    if (freq[(size_t)i] != 0) continue;
    int donor = -1;
    for (int j = 0; j < n; ++j)
      if (freq[(size_t)j] > 1 && (donor < 0 || freq[(size_t)j] < freq[(size_t)donor])) donor = j;
    if (donor < 0) return FGMM_ERR_INVALID; // more symbols than 2^precision counts (the reference asserts)
    --freq[(size_t)donor];
    ++freq[(size_t)i];
  }
  cdf[0] = 0;
  for (int i = 0; i < n; ++i) cdf[i + 1] = cdf[i] + freq[(size_t)i];
  return FGMM_OK;
}

} // namespace fgmm

extern "C" {

int fgmm_rans_encode_symtab(const uint32_t *packed, const int32_t *symbols_or_null, int64_t n, uint8_t **out,
                            size_t *out_len) {
  return fgmm::rans_encode_symtab(packed, symbols_or_null, n, -1, out, out_len);
}
int64_t fgmm_ckpt_count(int64_t n, int64_t stride) { return stride > 0 && n > 0 ? (n - 1) / stride : 0; }
int fgmm_rans_encode_symtab_ckpt(const uint32_t *packed, const int32_t *symbols_or_null, int64_t n, int64_t stride, uint8_t **out,
                                 size_t *out_len, fgmm_ckpt *ckpt_out) {
  if (stride < 0 || (stride & (stride - 1))) return FGMM_ERR_INVALID;
  return fgmm::rans_encode_symtab_ckpt(packed, symbols_or_null, n, -1, out, out_len, stride, ckpt_out);
}
// Segment by segment on the calling thread (the batched decoder runs the segments on its workers): every segment starts from
// its checkpoint and must end in the next one's (state, position); the first mismatch makes the whole stream a sequential decode.
int fgmm_rans_decode_tab_ckpt(const uint8_t *encoded, size_t encoded_len, const void *hdr, int hdr_form, const uint32_t *blk_off,
                              int32_t tl, const uint8_t *rows, uint64_t rows_len, int64_t n, int32_t max_bs, int flags,
                              const fgmm_ckpt *ckpt, int64_t n_ckpt, int64_t stride, int32_t *out_symbols, int32_t *verified_out) {
  if (verified_out) *verified_out = 0;
  if (!ckpt || n_ckpt == 0 || !blk_off || stride <= 0 || (stride & (stride - 1)) || n_ckpt != fgmm_ckpt_count(n, stride))
    return fgmm_rans_decode_tab(encoded, encoded_len, hdr, hdr_form, blk_off, tl, rows, rows_len, n, max_bs, flags, out_symbols);
  if (n > 0 && (!hdr || !rows)) return FGMM_ERR_INVALID;
  if (max_bs < 0 || max_bs > FGMM_MAX_BS || tl < 1) return FGMM_ERR_INVALID;
  if (hdr_form == 2 && !fgmm::tab_hdr_fits16(max_bs)) return FGMM_ERR_INVALID;
  if (hdr_form == 4 && max_bs > FGMM_MAX_BS_H4) return FGMM_ERR_INVALID;
  const fgmm::TabPiece pc{hdr, blk_off, rows, (size_t)rows_len, n};
  const fgmm::TabView tv{(flags & FGMM_TAB_RAW_ROWS) ? fgmm::kTabNoEf : fgmm::kTabEfMin, hdr_form, tl, 1, &pc, nullptr, nullptr};
  bool ok = true;
  int hard = FGMM_OK; // an error that is not the notes' fault
  // ONE decoder for all segments: begin() copies a misaligned bitstream whole, which must not happen once per segment
  fgmm::TabDecoder td;
  int rc = td.begin(encoded, encoded_len, &tv, n, max_bs, out_symbols);
  if (rc != FGMM_OK) {
    td.finish();
    return rc;
  }
  const uint64_t x_head = td.x;
  for (int64_t sgm = 0; sgm <= n_ckpt && ok; ++sgm) {
    const int64_t lo = sgm * stride, hi = sgm == n_ckpt ? n : (sgm + 1) * stride;
    const uint64_t x0 = sgm ? ckpt[sgm - 1].x : x_head, pos0 = sgm ? ckpt[sgm - 1].pos : 0;
    uint64_t x1 = 0, pos1 = 0;
    rc = td.segment(lo, hi, x0, pos0, &x1, &pos1);
    td.rc = FGMM_OK; // (the next segment starts from its own note; finish() reports a range as "not covered" otherwise)
    if (rc != FGMM_OK) ok = false, hard = sgm == 0 ? rc : hard; // segment 0 starts from the stream's own head: its errors are real
    else if (sgm < n_ckpt && (x1 != ckpt[sgm].x || pos1 != ckpt[sgm].pos)) ok = false;
  }
  td.i = n;
  td.finish();
  if (ok) {
    if (verified_out) *verified_out = 1;
    return FGMM_OK;
  }
  if (hard != FGMM_OK) return hard;
  return fgmm_rans_decode_tab(encoded, encoded_len, hdr, hdr_form, blk_off, tl, rows, rows_len, n, max_bs, flags, out_symbols);
}

int fgmm_rans_encode_symtab_segs(const uint32_t *const *seg, int n_seg, int64_t seg_len, const int32_t *symbols_or_null, int64_t n,
                                 int64_t stride, uint8_t **out, size_t *out_len, fgmm_ckpt *ckpt_out) {
  if (!seg || n_seg < 1 || n_seg > fgmm::kEncSegs || seg_len < 1 || n < 0 || n > (int64_t)n_seg * seg_len || stride < 0 || (stride & (stride - 1)))
    return FGMM_ERR_INVALID;
  fgmm::SegTable t{};
  t.seg_len = seg_len;
  t.n_seg = n_seg;
  int64_t nb = 0;
  for (int s = 0; s < n_seg; ++s) {
    t.seg[s] = seg[s];
    const int64_t lo = (int64_t)s * seg_len, cnt = std::min(seg_len, n - lo);
    if (cnt > 0 && !seg[s]) return FGMM_ERR_INVALID;
    for (int64_t i = 0; i < cnt; ++i) nb += (seg[s][i] >> 16) == 0;
  }
  return fgmm::rans_encode_symtab_segs(t, symbols_or_null, n, nb, out, out_len, stride, ckpt_out);
}

int fgmm_rans_encode_symtab2(const uint32_t *packed0, const int32_t *symbols0_or_null, int64_t n0, const uint32_t *packed1,
                             const int32_t *symbols1_or_null, int64_t n1, uint8_t **out0, size_t *out0_len, uint8_t **out1,
                             size_t *out1_len) {
  const uint32_t *const packed[2] = {packed0, packed1};
  const int32_t *const syms[2] = {symbols0_or_null, symbols1_or_null};
  const int64_t n[2] = {n0, n1}, nb[2] = {-1, -1};
  uint8_t **out[2] = {out0, out1};
  size_t *len[2] = {out0_len, out1_len};
  return fgmm::rans_encode_symtab2(packed, syms, n, nb, out, len);
}

int fgmm_rans_encode_symtab_n(int ways, const uint32_t *const *packed, const int32_t *const *symbols_or_null, const int64_t *n,
                              uint8_t **out, size_t *out_len) {
  if (ways < 1 || ways > fgmm::kMaxEncWays || !packed || !n || !out || !out_len) return FGMM_ERR_INVALID;
  const int32_t *syms[fgmm::kMaxEncWays];
  int64_t nb[fgmm::kMaxEncWays];
  uint8_t **o[fgmm::kMaxEncWays];
  size_t *l[fgmm::kMaxEncWays];
  for (int k = 0; k < ways; ++k) {
    syms[k] = symbols_or_null ? symbols_or_null[k] : nullptr;
    nb[k] = -1;
    out[k] = nullptr;
    o[k] = &out[k];
    l[k] = &out_len[k];
  }
  return fgmm::rans_encode_symtab_ways(ways, packed, syms, n, nb, o, l);
}

int fgmm_rans_decode_cdftab(const uint8_t *encoded, size_t encoded_len, const uint32_t *hdr, const uint8_t *pool,
                            uint64_t pool_len, int64_t n, int32_t max_bs, int flags, int32_t *out_symbols) {
  if (n > 0 && (!hdr || !pool)) return FGMM_ERR_INVALID;
  if (max_bs < 0 || max_bs > FGMM_MAX_BS_H4) return FGMM_ERR_UNSUPPORTED; // this entry point takes 4-byte headers
  const fgmm::TabPiece pc{hdr, nullptr, pool, (size_t)pool_len, n};
  const fgmm::TabView tv{(flags & FGMM_TAB_RAW_ROWS) ? fgmm::kTabNoEf : fgmm::kTabEfMin, 4, 0, 1, &pc, nullptr, nullptr};
  return fgmm::rans_decode_tab(encoded, encoded_len, tv, n, max_bs, out_symbols);
}

int fgmm_rans_decode_tab(const uint8_t *encoded, size_t encoded_len, const void *hdr, int hdr_form, const uint32_t *blk_off,
                         int32_t tl, const uint8_t *rows, uint64_t rows_len, int64_t n, int32_t max_bs, int flags,
                         int32_t *out_symbols) {
  if (n > 0 && (!hdr || !rows)) return FGMM_ERR_INVALID;
  if (max_bs < 0 || max_bs > FGMM_MAX_BS || (blk_off && tl < 1)) return FGMM_ERR_INVALID;
  if (hdr_form == 2 && !fgmm::tab_hdr_fits16(max_bs)) return FGMM_ERR_INVALID;
  if (hdr_form == 4 && max_bs > FGMM_MAX_BS_H4) return FGMM_ERR_INVALID;
  const fgmm::TabPiece pc{hdr, blk_off, rows, (size_t)rows_len, n};
  const fgmm::TabView tv{(flags & FGMM_TAB_RAW_ROWS) ? fgmm::kTabNoEf : fgmm::kTabEfMin, hdr_form, tl, 1, &pc, nullptr, nullptr};
  return fgmm::rans_decode_tab(encoded, encoded_len, tv, n, max_bs, out_symbols);
}

void fgmm_free(void *p) { free(p); }

struct fgmm_symbuf {
  fgmm::SymBuf sb;
};
struct fgmm_decstream {
  fgmm::DecStream ds;
};

int fgmm_symbuf_create(fgmm_symbuf **out) {
  if (!out) return FGMM_ERR_INVALID;
  *out = new (std::nothrow) fgmm_symbuf;
  return *out ? FGMM_OK : FGMM_ERR_NOMEM;
}
void fgmm_symbuf_destroy(fgmm_symbuf *b) { delete b; }
int64_t fgmm_symbuf_size(const fgmm_symbuf *b) { return b ? (int64_t)b->sb.e.size() : -1; }
int fgmm_symbuf_append_table(fgmm_symbuf *b, const int32_t *symbols, const int32_t *indexes, int64_t n,
                             const int32_t *cdfs, int64_t cdf_stride, int32_t n_cdfs, const int32_t *cdfs_sizes,
                             const int32_t *offsets) {
  if (!b) return FGMM_ERR_INVALID;
  return fgmm::symbuf_append_table(b->sb, symbols, indexes, n, cdfs, cdf_stride, n_cdfs, cdfs_sizes, offsets);
}
int fgmm_symbuf_append_symtab(fgmm_symbuf *b, const uint32_t *packed, const int32_t *symbols_or_null, int64_t n) {
  if (!b) return FGMM_ERR_INVALID;
  return fgmm::symbuf_append_symtab(b->sb, packed, symbols_or_null, n);
}
int fgmm_symbuf_flush(fgmm_symbuf *b, uint8_t **out, size_t *out_len) {
  if (!b) return FGMM_ERR_INVALID;
  return fgmm::symbuf_flush(b->sb, out, out_len);
}

int fgmm_encode_with_indexes(const int32_t *symbols, const int32_t *indexes, int64_t n, const int32_t *cdfs,
                             int64_t cdf_stride, int32_t n_cdfs, const int32_t *cdfs_sizes, const int32_t *offsets,
                             uint8_t **out, size_t *out_len) {
  fgmm::SymBuf sb;
  const int rc = fgmm::symbuf_append_table(sb, symbols, indexes, n, cdfs, cdf_stride, n_cdfs, cdfs_sizes, offsets);
  return rc ? rc : fgmm::symbuf_flush(sb, out, out_len);
}

int fgmm_decstream_create(const uint8_t *encoded, size_t encoded_len, fgmm_decstream **out) {
  if (!out) return FGMM_ERR_INVALID;
  fgmm_decstream *d = new (std::nothrow) fgmm_decstream;
  if (!d) return FGMM_ERR_NOMEM;
  const int rc = fgmm::decstream_init(d->ds, encoded, encoded_len);
  if (rc) {
    delete d;
    return rc;
  }
  *out = d;
  return FGMM_OK;
}
void fgmm_decstream_destroy(fgmm_decstream *d) { delete d; }
int fgmm_decstream_decode(fgmm_decstream *d, const int32_t *indexes, int64_t n, const int32_t *cdfs, int64_t cdf_stride,
                          int32_t n_cdfs, const int32_t *cdfs_sizes, const int32_t *offsets, int32_t *out_symbols) {
  if (!d) return FGMM_ERR_INVALID;
  return fgmm::decstream_decode(d->ds, indexes, n, cdfs, cdf_stride, n_cdfs, cdfs_sizes, offsets, out_symbols);
}

int fgmm_decode_with_indexes(const uint8_t *encoded, size_t encoded_len, const int32_t *indexes, int64_t n,
                             const int32_t *cdfs, int64_t cdf_stride, int32_t n_cdfs, const int32_t *cdfs_sizes,
                             const int32_t *offsets, int32_t *out_symbols) {
  fgmm::DecStream ds;
  const int rc = fgmm::decstream_init(ds, encoded, encoded_len);
  return rc ? rc : fgmm::decstream_decode(ds, indexes, n, cdfs, cdf_stride, n_cdfs, cdfs_sizes, offsets, out_symbols);
}

int fgmm_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf_out) {
  return fgmm::pmf_to_quantized_cdf(pmf, n, precision, cdf_out);
}

} // extern "C"

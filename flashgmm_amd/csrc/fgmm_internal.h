// fgmm_internal.h — structures shared by the HIP kernels (fgmm_kernels.hip), the host rANS coder
// (fgmm_rans.cpp) and the C-ABI glue (fgmm_capi.cpp).  Not part of the public ABI.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/flashgmm_amd.h"

#if defined(__HIPCC__)
#define FGMM_HD __host__ __device__
#else
#define FGMM_HD
#endif

namespace fgmm {

// ---- per-item descriptors of the batched kernels (one item = one bitstream: an image half / a channel group) --

struct EncDesc {
  // inputs (device)
  const float *y;        // [M*hw] latents, rounded by the kernel;   null when `sym` is given
  const int32_t *sym;    // raw-boundary form: symbols given, [hw] with M == 1
  const void *scales, *means, *weights; // float32 or float16 planes (the launcher picks the kernel)
  int64_t stride_k, stride_c, stride_p; // elements
  int64_t hw;
  int32_t M;
  int32_t clamp;
  int32_t logits;        // the weights planes hold LOGITS: pi = softmax4 over K in the kernel (fgmm_math.h)
  uint32_t meta_slots;   // entries of `meta`
  const float *x;        // fused parameter head (fgmm_head.hip): the head's input features [c_in, hw]; the planes above are unused
  // outputs (device)
  float *yq;             // [M*hw] round(y), or null
  float *chan_min;       // [M]  min over the channel of y
  float *chan_max;       // [M]
  int32_t *chan_nz;      // [M]  any round(y) != 0
  int32_t *chan_list;    // [M+1] compact index -> channel; [M] = number of non-zero channels (null: identity)
  uint32_t *packed;      // [n_nz*hw] start | range<<16, channels compacted - when the table is ONE piece (seg_b[0] = INT32_MAX)
  // The table in up to kEncSegs SEGMENTS of compact channels: segment s holds the compact channels [s * cps, (s + 1) * cps)
  // contiguously at packed_seg[s]; seg_b[s] = (s + 1) * cps for the segments in use, INT32_MAX beyond.  The batched encoder
  // lays the segments of a call out LAST SEGMENT FIRST across all items: a bitstream is encoded backwards, so the tables'
  // tails cross PCIe first and the host encoders follow the landing instead of waiting for whole tables (fgmm_capi.cpp).
  uint32_t *packed_seg[4];
  int32_t seg_b[3];
  int32_t cps;
  uint32_t *meta;        // [4 * blocks] bypass symbols seen by each wave (zeroed by the host, summed by the host:
                         // device-scope atomics on one counter serialise across the 8 XCDs, ~0.1 ms for a 6 M-symbol item)
};

struct DecDesc {
  const void *scales, *means, *weights; // float32 or float16 planes
  int64_t stride_k, stride_c, stride_p;
  int64_t hw;
  int64_t n;                     // latents = n_ch * hw
  const int32_t *chan_list;      // device [n_ch] source channel of compact channel j; null: identity
  int32_t n_ch;
  int32_t max_bs;
  int32_t clamp;
  int32_t logits;                // the weights planes hold logits: pi = softmax4 over K in the kernel (fgmm_math.h)
  int32_t prune;                 // 1: skip the saturated tails (exact, see tab_window); 0: evaluate all of F
  int32_t hdr_form;              // bytes per header as the host gets them: 2, 4 or 8 (format v5 below)
  uint32_t ef_min;               // rows with at least this many entries are Elias-Fano coded (kTabEfMin / kTabNoEf)
  int32_t tl;                    // tab_kernel: latents per block (rows of a block are contiguous, blocks are placed by a cursor)
  // ---- tab_kernel (single pass): the blocks [blk_begin, blk_end) of this item, into one launch's range
  int32_t blk_begin, blk_end;
  void *hdr_out;                 // headers of latent blk_begin * tl onward
  uint32_t *blkoff_out;          // [blk_end - blk_begin] byte offset / 4 of each block's first row, from `rows`
  uint8_t *rows;                 // the launch's row area (device staging)
  unsigned long long rows_cap;   // bytes
  unsigned long long *counters;  // shared by the launch (kTabCounters words): [0] bytes of rows placed - the cursor: ONE returning
                                 // atomic add per block (blocks lie in arrival order), [1] overflow (a block did not fit: the host
                                 // re-runs the launch with [0] bytes), [2] unused, [3] blocks with a non-monotone row,
                                 // [4 .. 4 + kTabEdgeSlots) edges evaluated, summed by the host (only when count_edges; spread over
                                 // slots: same-address atomics serialise chip-wide)
  int32_t count_edges, pad2_;
  // ---- generic two-pass path (cdftab_count / scan / fill): any half-width, rows sequential in latent order
  void *hdr;                     // [n] headers, 4-byte form (8-byte form when hdr_form == 8)
  int32_t tiles;                 // blocks per channel = ceil(hw / 256)
  int32_t pad_;
  uint8_t *pool;                 // rows, latent order
  unsigned long long pool_cap;   // bytes
  unsigned long long *pool_used; // [0] bytes used, [1] overflow flag, [2] some row is non-monotone
  uint32_t *blk_sums;            // [n_ch*tiles] row bytes per block
  unsigned long long *blk_off;   // [n_ch*tiles] byte offset of each block's first row
};

// ---- the parameter head's last layer on the matrix cores (fgmm_head.hip) ----------------------------------------
constexpr int kHeadCG = 16; // latent channels per block: the packed weights are laid out in groups of this many
constexpr int kHeadBK = 32; // input channels per LDS tile
struct HeadW {              // the PACKED weights of a head (device): see head_pack_kernel / head16_pack_kernel
  const void *wp;           // exact form: float [n_cg][n_kt][12 * kHeadCG][kHeadBK]; bf16x6: bf16 [n_cg][n_kt][3][12 * kHeadCG][kHeadBK]
  const float *bp;          // [n_cg][12 * kHeadCG]
  int32_t M, c_in, n_cg, n_kt;
  int32_t arith, pad_;      // 0: binary32 on v_mfma_f32_32x32x2_f32 (fgmm_head.hip); FGMM_HEAD_BF16X6: fgmm_head16.hip
};
struct HeadDesc { // one item of the un-fused form
  const float *x; // device [c_in, hw]
  float *out;     // device [3 * 4 * M, hw]: scales | means | logits planes, channel k * M + c
  int64_t hw;
};
static inline size_t head_packed_floats(int M, int c_in) {
  const size_t n_cg = ((size_t)M + kHeadCG - 1) / kHeadCG, n_kt = ((size_t)c_in + kHeadBK - 1) / kHeadBK;
  return n_cg * n_kt * 12 * kHeadCG * kHeadBK + n_cg * 12 * kHeadCG;
}
int launch_head_pack(const float *w, const float *bias_or_null, int M, int c_in, float *wp, float *bp, void *stream);
// the bf16x6 form (fgmm_head16.hip): `packed` holds the three bf16 parts of the weights, then the bias
size_t head16_packed_bytes(int M, int c_in);
int launch_head16_pack(const float *w, const float *bias_or_null, int M, int c_in, void *packed, void *stream);
// the features' split copy: head16_split_elems bf16 per item, written by launch_head16_split; the kernels' descriptors then carry THAT
// pointer as `x`
size_t head16_split_elems(int c_in, int64_t hw);
int launch_head16_split(const float *x0, void *xs0, int64_t hw, int c_in, int count, int64_t x_stride, int64_t xs_stride, void *stream);
int launch_head16_params(const HeadDesc *d_descs, const HeadW &w, int count, int64_t hw_max, void *stream);
int launch_head16_symtab(const EncDesc *d_descs, const HeadW &w, int count, int M_max, int64_t hw_max, int mode, bool clamped, void *stream);
int launch_head_params(const HeadDesc *d_descs, const HeadW &w, int count, int64_t hw_max, bool vec, void *stream); // vec: every hw % 4 == 0, x 16-byte aligned
int launch_head_symtab(const EncDesc *d_descs, const HeadW &w, int count, int M_max, int64_t hw_max, int mode, bool clamped, bool vec, void *stream);

// ---- GPU-side decode of CHECKPOINTED bitstreams (segdec_kernel): one wave per segment ---------------------------
struct SegDesc {
  const void *scales, *means, *weights; // float32 or float16 planes
  int64_t stride_k, stride_c, stride_p;
  int64_t hw;
  int64_t n;                     // coded latents = n_ch * hw
  const int32_t *chan_list;      // device [n_ch] source channel of compact channel j; null: identity
  int32_t max_bs, clamp, logits, pad_;
  const uint32_t *words;         // device copy of the bitstream: words[0..1] the coder's initial state, then the renormalisation words
  int64_t n_words;               // all of them (>= 2)
  const fgmm_ckpt *ckpt;         // device [n_ckpt]
  int64_t n_ckpt, stride;
  float *y_hat;                  // device [M * hw]: decoded symbols of the coded channels as floats; the others: zero (segzero_kernel)
  uint32_t *status;              // device [n_ckpt + 1], per segment: 0 = decoded and ended in the next checkpoint; else why not
  const int32_t *dead_list;      // device [n_dead] channels without a coded symbol
  int64_t n_dead;
};
struct SegRef { // one wave's work
  int32_t item, seg;
};
enum : uint32_t { kSegOk = 0, kSegMismatch = 1, kSegHard = 2, kSegStream = 3 }; // hard: a row the wave does not decode itself
int launch_segdec(const SegDesc *d_descs, const SegRef *d_segs, int64_t n_segs, int mode, bool clamped, bool f16, void *stream);
int launch_segzero(const SegDesc *d_descs, int count, int64_t max_dead, void *stream); // zeroes every item's dead channels in y_hat

// ---- decode-side table format v5 (documented in include/flashgmm_amd.h) -------------------------------------
//   header of latent i, one of three forms (per item):
//     2 bytes: (a + max_bs) | cnt << 8, cnt in [1, 254]                    items with 2*max_bs + 2 <= 254
//              cnt field 255 = escape: the row starts with a 4-byte header of the next form (non-monotone rows)
//     4 bytes: int16 a | cnt << 16 (15 bits) | nonmono << 31                 items with max_bs <= 16382
//     8 bytes: int32 a ; cnt (31 bits) | nonmono << 31                       any half-width
//   row i = F_i[a .. a+cnt), starting at the first non-zero edge (F_i[v < a] = 0, F_i[v >= a+cnt] = last entry);
//   rows 2-byte aligned:
//     raw (cnt < kTabEfMin or nonmono): uint16[cnt]
//     EF  (cnt >= kTabEfMin, monotone): Elias-Fano with l = tab_ef_l(cnt) low bits (12 up to 48 entries, else 8), ONE little-endian bit string (bit b of
//                                       the row = bit (b & 7) of byte b >> 3), rounded up to 16 bits:
//                                         bits [0, HB), HB = cnt + (65536 >> l): bit ((E_j >> l) + j) set for every entry j
//                                         bits [LB + j * l, LB + (j + 1) * l), LB = HB rounded up to 8: the low l bits of E_j
//   placement: sequential in latent order (generic path, the building-block API), or per block of `tl` latents at
//   rows + 4 * blk_off[block] (tab_kernel: blocks are placed by an atomic cursor, in no particular order)
constexpr int kMaxPieces = 32; // FGMM_MAX_PIECES
constexpr int kAutoPiecesMax = 24; // what the automatic choice goes up to (option "pieces" 0)
constexpr int kEncSegs = 4;    // EncDesc::packed_seg
constexpr int kTabEdgeSlots = 64;               // DecDesc::counters
constexpr int kTabCounters = 4 + kTabEdgeSlots;
#ifndef FGMM_EF_MIN
#define FGMM_EF_MIN 14
#endif
constexpr uint32_t kTabEfMin = FGMM_EF_MIN; // rows with at least this many entries MAY be Elias-Fano coded (smaller than raw from here on)
constexpr uint32_t kTabEfDefault = 33;     // ... and are, by default, in the batched decoder ("ef_min" option): see DESIGN.md section 5 (49 until
                                           // round 6: with the decoders on a pool of their own, one hardware thread per core, the rows of 33-48 entries
                                           // pay as Elias-Fano too - 57.6 -> 56.6 B/latent, step median 8.83 -> 8.72 ms, CPU per step +2 %)
constexpr uint32_t kHdr2Escape = 255;
FGMM_HD static inline uint32_t tab_hdr_pack(int32_t a, uint32_t cnt, uint32_t nonmono) {
  return (uint32_t)(uint16_t)(int16_t)a | ((cnt & 0x7FFFu) << 16) | (nonmono << 31);
}
FGMM_HD static inline int32_t tab_hdr_a(uint32_t h) { return (int32_t)(int16_t)(uint16_t)(h & 0xFFFFu); }
FGMM_HD static inline uint32_t tab_hdr_cnt(uint32_t h) { return (h >> 16) & 0x7FFFu; }
FGMM_HD static inline uint32_t tab_hdr_nonmono(uint32_t h) { return h >> 31; }
FGMM_HD static inline unsigned long long tab_hdr8_pack(int32_t a, uint32_t cnt, uint32_t nonmono) {
  return (unsigned long long)(uint32_t)a | ((unsigned long long)((cnt & 0x7FFFFFFFu) | (nonmono << 31)) << 32);
}
// ef_min: rows with at least this many entries are Elias-Fano coded; kTabEfMin when PCIe is the bottleneck, kTabNoEf (no
// such row) when a call is bound by its host decoders: a uint16 row is searched faster, an Elias-Fano row is smaller
constexpr uint32_t kTabNoEf = 0x7FFFFFFFu;
FGMM_HD static inline bool tab_row_is_ef(uint32_t cnt, uint32_t nonmono, uint32_t ef_min) { return cnt >= ef_min && !nonmono; }
// low bits of an Elias-Fano row: 12 up to 48 entries (16 buckets: the unary part is one 64-bit word for the host's
// search), 8 beyond (256 buckets, low parts are bytes).  Within 2 % of the best choice per row on Kodak-like tables, and
// below the 2 * cnt bytes of the raw form from cnt = 14 on.
constexpr uint32_t kTabEf12Max = 48;
FGMM_HD static inline uint32_t tab_ef_l(uint32_t cnt) { return cnt <= kTabEf12Max ? 12u : 8u; }
FGMM_HD static inline uint32_t tab_ef_hb(uint32_t cnt, uint32_t l) { return cnt + (65536u >> l); } // bits of the unary high part
FGMM_HD static inline uint32_t tab_ef_lb(uint32_t cnt, uint32_t l) { return (tab_ef_hb(cnt, l) + 7u) & ~7u; }       // first bit of the low parts
FGMM_HD static inline unsigned long long tab_ef_bits(uint32_t cnt, uint32_t l) { return (unsigned long long)cnt * l + tab_ef_lb(cnt, l); }
FGMM_HD static inline unsigned long long tab_row_bytes(uint32_t cnt, uint32_t nonmono, uint32_t ef_min) {
  return tab_row_is_ef(cnt, nonmono, ef_min) ? 2ull * ((tab_ef_bits(cnt, tab_ef_l(cnt)) + 15ull) >> 4) : 2ull * (unsigned long long)cnt;
}
FGMM_HD static inline bool tab_hdr_fits16(int32_t max_bs) { return 2 * (int64_t)max_bs + 2 <= 254; }
FGMM_HD static inline int tab_hdr_form(int32_t max_bs) { return tab_hdr_fits16(max_bs) ? 2 : (max_bs <= 16382 ? 4 : 8); }
// tab_kernel: entries of evaluated edges one block may keep in LDS, and the latents per block that guarantees it
constexpr int kTabCapE = 12288; // default of the "tab_cap_e" option (upper limit 32768: 16-bit prefix sums in the kernel): 80 latents of
                                // Kodak's half-width per block, four blocks per CU - the kernel alone 1.29 ms against 1.44 at 16384
constexpr int kTabCapEWide = 16384; // ... and what a call gets by default when an item's window only fits that (383 < max_bs <= 511)
constexpr int kTabMaxTl = 256;
FGMM_HD static inline int tab_tl(int32_t max_bs, int cap_e) { // 0: the item does not fit the single-pass kernel
  const int64_t W = 2 * (int64_t)max_bs + 2;
  const int64_t t = (cap_e / W) & ~15ll;
  return t >= 16 ? (int)(t > kTabMaxTl ? kTabMaxTl : t) : 0;
}

// ---- kernel launchers (fgmm_kernels.hip); stream is a hipStream_t; all return hipError_t as int ----------
int launch_quant_stats(const EncDesc *d_descs, int count, int M_max, void *stream);
// M_max, hw_max, n_max: the largest M, hw and M * hw of the batch.  linear: every hw is a multiple of 64 * vec, waves
// take consecutive coded symbols across channels (all waves full); else one block per (tile, channel).
int launch_symtab(const EncDesc *d_descs, int count, int M_max, int64_t hw_max, int64_t n_max, bool linear, int mode, int vec,
                  bool clamped, bool f16, void *stream);
int launch_cdf_pair(const int32_t *v, const float *scales, const float *means, const float *weights, int64_t n,
                    int64_t stride_n, int64_t stride_k, int mode, float *c1, float *c2, void *stream);
// generic path: count + scan (sizes and offsets), then fill (rows); launch_cdftab = both, back to back on one stream
int launch_cdftab_count(const DecDesc *d_descs, int count, int n_ch_max, int64_t hw_max, int mode, bool clamped, bool f16,
                        void *stream);
int launch_cdftab_fill(const DecDesc *d_descs, int count, int n_ch_max, int64_t hw_max, int mode, bool clamped, bool f16,
                       void *stream);
int launch_cdftab(const DecDesc *d_descs, int count, int n_ch_max, int64_t hw_max, int mode, bool clamped, bool f16,
                  void *stream);
// single pass: window, evaluation (flattened over pairs of edges, parameters staged in LDS), trimming, rows.
// blocks_max = largest blk_end - blk_begin of the launch; tl_max = largest tl; cap_e as passed to tab_tl().
int launch_tab(const DecDesc *d_descs, int count, int blocks_max, int tl_max, int cap_e, int mode, bool clamped, bool f16,
               void *stream);
// y_hat[c, p] = rank[c] < 0 ? 0 : (float)sym[rank[c] * hw + p]; sym is int16 (wide = 0) or int32 and may live in pinned
// host memory (read over PCIe)   (entropy_models.py:903-908)
int launch_softmax_probe(const float *logits, float *pi, int64_t n, void *stream);
int launch_yhat_scatter(const void *sym, int wide, const int32_t *rank, float *y_hat, int M, int64_t hw, void *stream);
// The decoded symbols of a call's items back into their latents ROUND BY ROUND (round = piece index of the table pipeline): the
// items' symbols of pieces that every decoder has finished are scattered while the later pieces are still being decoded, so
// that what is left after the last decoder is the last, smallest piece (decode_batch).
struct ScatDesc {
  const int16_t *sym;        // pinned host: the item's compact symbols, int16 (an item that turns out wide is redone whole)
  const int32_t *chan_list;  // device [n_ch]: compact channel -> channel
  const int32_t *rank;       // device [M]: channel -> compact channel, -1 = no coded symbol (zero in y_hat)
  float *y_hat;              // device [M * hw]; null: the item takes no part
  int64_t hw;
  int32_t M, pad;
  int64_t bound[kMaxPieces + 1]; // round r scatters the compact symbols [bound[r], bound[r + 1])
};
int launch_yhat_scatter_round(const ScatDesc *d_descs, int count, int round, int64_t max_range, void *stream);
int launch_yhat_zero_dead(const ScatDesc *d_descs, int count, int M_max, int64_t hw_max, void *stream);
// checkerboard split (embed = false: [planes,h,w] -> [2,planes,h,w/2]) / merge (embed = true); w even, elem_bytes 2 or 4
int launch_ckbd(const void *src, void *dst, int64_t planes, int64_t h, int64_t w, int elem_bytes, int anchor_odd, bool embed,
                void *stream);
int launch_fastmath_selftest(int which, unsigned long long n, unsigned long long seed, unsigned long long *n_bad, void *stream);
// exhaustive check of the saturation lemmas behind the pruning; *n_bad (device) receives the number of violations
int launch_saturation_selftest(int mode, unsigned long long *n_bad, void *stream);

// ---- host rANS (fgmm_rans.cpp), integer only --------------------------------------------------------------
// Where a finished bitstream goes.  Default: a malloc'ed buffer (fgmm_free).  With a sink (include/flashgmm_amd.h: fgmm_sink) the
// encoder asks it for storage of the stream's exact size once that is known and copies the stream there out of its scratch: the
// caller's storage (a Python bytes object, say) is filled by the flush, with no buffer of the library's in between.  (The batched
// encoder gives its workers a sink of its own, which passes the question on to the calling thread: fgmm_encode.cpp, SinkDesk.)
struct BytesTo {
  const fgmm_sink *sink = nullptr;
  int item = 0;
};
// returns 0 or an fgmm_status; *out malloc'ed
int rans_encode_symtab(const uint32_t *packed, const int32_t *symbols_or_null, int64_t n, int64_t n_bypass_hint,
                       uint8_t **out, size_t *out_len);
// ... noting a checkpoint (include/flashgmm_amd.h: fgmm_ckpt) every `stride` symbols (a power of two; 0: none):
// ckpt[(n - 1) / stride] entries, entry k for symbol (k + 1) * stride
int rans_encode_symtab_ckpt(const uint32_t *packed, const int32_t *symbols_or_null, int64_t n, int64_t n_bypass_hint, uint8_t **out,
                            size_t *out_len, int64_t stride, fgmm_ckpt *ckpt, BytesTo to = {});
// ... from a table that lies in `n_seg` segments of `seg_len` entries (the last one may be shorter): seg[s] holds the entries
// [s * seg_len, (s + 1) * seg_len).  The walk runs backwards, so the LAST segment is needed first; wait(arg, s) (may be null)
// returns once segment s may be read (FGMM_OK) - the batched encoder's tables land tail first.
struct SegTable {
  const uint32_t *seg[kEncSegs];
  int64_t seg_len;
  int n_seg;
  int (*wait)(void *arg, int s);
  void *arg;
};
int rans_encode_symtab_segs(const SegTable &t, const int32_t *symbols_or_null, int64_t n, int64_t n_bypass_hint, uint8_t **out,
                            size_t *out_len, int64_t stride, fgmm_ckpt *ckpt, BytesTo to = {});
// two streams by one thread, interleaved (each output identical to rans_encode_symtab's)
constexpr int kMaxEncWays = 4;
// `ways` (1..4) tables -> bitstreams, coded by the calling thread symbol by symbol in turn
int rans_encode_symtab_ways(int ways, const uint32_t *const *packed, const int32_t *const *symbols, const int64_t *n,
                            const int64_t *n_bypass_hint, uint8_t ***out, size_t **out_len, int64_t ckpt_stride = 0,
                            fgmm_ckpt *const *ckpt = nullptr, const BytesTo *to = nullptr);
int rans_encode_symtab2(const uint32_t *const packed[2], const int32_t *const symbols[2], const int64_t n[2],
                        const int64_t n_bypass_hint[2], uint8_t **out[2], size_t *out_len[2]);
// Decode-side tables as the host decoder sees them: `npiece` pieces in latent order; piece k holds the latents
// [k ? piece[k-1].end : 0, piece[k].end) and begins on a block boundary (a multiple of tl latents).  Piece 0 has landed
// before the decoder is called; wait(arg, k) blocks until piece k (k >= 1) has.  The decoder walks the latents in
// order, so it only ever waits for the next piece.
struct TabPiece {
  const void *hdr;         // headers of the piece's latents (hdr_form bytes each)
  const uint32_t *blk_off; // per block of tl latents: offset / 4 of its first row from `rows`; null: rows are sequential
  const uint8_t *rows;
  size_t rows_len;         // bytes of the row area that may be read (rows lie inside; checked before every access)
  int64_t end;             // one past the piece's last latent
};
struct TabView {
  uint32_t ef_min; // as the kernels were told (DecDesc::ef_min)
  int hdr_form; // 2, 4, 8
  int tl;       // latents per block (blk_off granularity); ignored when blk_off is null
  int npiece;
  const TabPiece *piece;
  void *arg;
  int (*wait)(void *arg, int k); // FGMM_OK or an error status; null when npiece == 1
};
int rans_decode_tab(const uint8_t *enc, size_t enc_len, const TabView &tv, int64_t n, int32_t max_bs, int32_t *out);
// the same, one piece at a time: begin(), then piece(0), piece(1) ... each once its tables have landed, then finish()
struct TabDecoder {
  const TabView *tv = nullptr;
  int64_t n = 0, i = 0;
  int32_t max_bs = 0;
  int32_t *out = nullptr;
  int next_piece = 0, rc = 0;
  uint64_t x = 0;            // rANS state
  const uint32_t *ptr = nullptr, *end_ = nullptr, *base_ = nullptr; // base_: the first renormalisation word
  uint32_t *copy = nullptr;  // aligned copy of a misaligned bitstream
  uint16_t *scratch = nullptr;
  size_t scratch_cap = 0;
  int begin(const uint8_t *enc, size_t enc_len, const TabView *view, int64_t n, int32_t max_bs, int32_t *out);
  int piece(int k);
  int segment(int64_t lo, int64_t hi, uint64_t x0, uint64_t pos0, uint64_t *x1, uint64_t *pos1); // checkpointed streams
  int finish();
};

} // namespace fgmm

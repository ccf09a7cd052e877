// fgmm_internal.h — structures shared by the HIP kernels (fgmm_kernels.hip), the host rANS coder
// (fgmm_rans.cpp) and the C-ABI glue (fgmm_capi.cpp).  Not part of the public ABI.
#pragma once
#include <stddef.h>
#include <stdint.h>

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
  // outputs (device)
  float *yq;             // [M*hw] round(y), or null
  float *chan_min;       // [M]  min over the channel of y
  float *chan_max;       // [M]
  int32_t *chan_nz;      // [M]  any round(y) != 0
  int32_t *chan_list;    // [M+1] compact index -> channel; [M] = number of non-zero channels (null: identity)
  uint32_t *packed;      // [n_nz*hw] start | range<<16, channels compacted
  uint32_t *meta;        // [4 * blocks] bypass symbols seen by each wave (zeroed by the host, summed by the host:
                         // device-scope atomics on one counter serialise across the 8 XCDs, ~0.1 ms for a 6 M-symbol item)
};

struct DecDesc {
  const void *scales, *means, *weights; // float32 or float16 planes
  int64_t stride_k, stride_c, stride_p;
  int64_t hw;
  const int32_t *chan_list; // device [n_ch] source channel of compact channel j; null: identity
  int32_t n_ch;
  int32_t max_bs;
  int32_t clamp;
  int32_t prune;                 // 1: skip the saturated tails (exact, see cdftab_count_kernel); 0: evaluate all of F
  int32_t tiles;                 // blocks per channel = ceil(hw / 256)
  int32_t pad_;
  uint32_t *hdr;                 // [n_ch*hw] 4-byte headers (written by the count pass, read by the fill pass and the host)
  void *hdr_out;                 // hdr_pack_kernel: the headers as the host gets them (uint32, or uint16 if hdr_compact)
  int32_t hdr_compact, pad3_;
  int32_t ch_begin, ch_end;      // compact channels the fill pass covers (a launch may fill an item piece by piece)
  uint8_t *pool;                 // rows, latent order (device memory, or pinned host memory written over PCIe)
  unsigned long long pool_cap;   // bytes
  unsigned long long *pool_used; // [0] bytes used, [1] overflow flag, [2 + k] byte offset of the first row of piece
                                 // k + 1 (k < n_piece - 1), piece k = compact channels [n_ch*k/n_piece, n_ch*(k+1)/n_piece),
                                 // [2 + kMaxPieces] some row is non-monotone (written when n_piece >= 1)
  int32_t n_piece;               // 0 or 1: no piece offsets wanted (at most FGMM_MAX_PIECES)
  int32_t pad2_;
  uint16_t *tmp;                 // null, or [n_ch*tiles][kTmpHdrRows + W][256]: per lane its evaluation window (j_lo, j_hi,
                                 // T_sat) and the edges the count pass evaluated (row t = each lane's t-th edge,
                                 // W = 2*max_bs+2): the fill pass formats rows from them, no second evaluation
  uint32_t *blk_sums;            // [n_ch*tiles] row bytes per block
  unsigned long long *blk_off;   // [n_ch*tiles] byte offset of each block's first row
};

// ---- decode-side table format v3 (documented in include/flashgmm_amd.h) -------------------------------------
//   hdr  (uint32): int16 a | cnt << 16 (15 bits) | nonmono << 31
//   hdr  (uint16, batched decode only, items with 2*max_bs+2 <= 254 and no non-monotone row): (a + max_bs) | cnt << 8
//   row i = F_i[a .. a+cnt), starting at the first non-zero edge (F_i[v < a] = 0, F_i[v >= a+cnt] = last entry);
//   rows in latent order, each 4-byte aligned, no stored offset:
//     raw (cnt < 64 or nonmono): uint16[round2(cnt)], padded with the last value
//     EF  (cnt >= 64, monotone): uint8 lows[round8(cnt)] ; uint64 upper[U], U = ceil((cnt + 256) / 64),
//                                bit ((E_j >> 8) + j) set for every entry j
constexpr int kMaxPieces = 8; // FGMM_MAX_PIECES
constexpr int kTmpHdrRows = 4; // see DecDesc::tmp
#ifndef FGMM_EF_MIN
#define FGMM_EF_MIN 64
#endif
constexpr uint32_t kTabEfMin = FGMM_EF_MIN; // rows with at least this many entries are Elias-Fano coded
FGMM_HD static inline uint32_t tab_hdr_pack(int32_t a, uint32_t cnt, uint32_t nonmono) {
  return (uint32_t)(uint16_t)(int16_t)a | ((cnt & 0x7FFFu) << 16) | (nonmono << 31);
}
FGMM_HD static inline int32_t tab_hdr_a(uint32_t h) { return (int32_t)(int16_t)(uint16_t)(h & 0xFFFFu); }
FGMM_HD static inline uint32_t tab_hdr_cnt(uint32_t h) { return (h >> 16) & 0x7FFFu; }
FGMM_HD static inline uint32_t tab_hdr_nonmono(uint32_t h) { return h >> 31; }
FGMM_HD static inline bool tab_row_is_ef(uint32_t cnt, uint32_t nonmono) { return cnt >= kTabEfMin && !nonmono; }
FGMM_HD static inline uint32_t tab_row_bytes(uint32_t cnt, uint32_t nonmono) {
  return tab_row_is_ef(cnt, nonmono) ? ((cnt + 7u) & ~7u) + 8u * ((cnt + 256u + 63u) >> 6) : 2u * ((cnt + 1u) & ~1u);
}

// ---- kernel launchers (fgmm_kernels.hip); stream is a hipStream_t; all return hipError_t as int ----------
int launch_quant_stats(const EncDesc *d_descs, int count, int M_max, void *stream);
// M_max, hw_max, n_max: the largest M, hw and M * hw of the batch.  linear: every hw is a multiple of 64 * vec, waves
// take consecutive coded symbols across channels (all waves full); else one block per (tile, channel).
int launch_symtab(const EncDesc *d_descs, int count, int M_max, int64_t hw_max, int64_t n_max, bool linear, int mode, int vec,
                  bool clamped, bool f16, void *stream);
int launch_cdf_pair(const int32_t *v, const float *scales, const float *means, const float *weights, int64_t n,
                    int64_t stride_n, int64_t stride_k, int mode, float *c1, float *c2, void *stream);
// count + scan (sizes and offsets), then fill (rows); launch_cdftab = both, back to back on one stream
int launch_cdftab_count(const DecDesc *d_descs, int count, int n_ch_max, int64_t hw_max, int mode, bool clamped, bool f16,
                        void *stream);
int launch_cdftab_fill(const DecDesc *d_descs, int count, int n_ch_max, int64_t hw_max, int mode, bool clamped, bool f16,
                       void *stream);
int launch_cdftab(const DecDesc *d_descs, int count, int n_ch_max, int64_t hw_max, int mode, bool clamped, bool f16,
                  void *stream);
// y_hat[c, p] = rank[c] < 0 ? 0 : (float)sym[rank[c] * hw + p]; sym is int16 (wide = 0) or int32 and may live in pinned
// host memory (read over PCIe)   (entropy_models.py:903-908)
// headers into the staging range (DecDesc::hdr_out), 2-byte form where DecDesc::hdr_compact; n_max = largest n_ch*hw
int launch_hdr_pack(const DecDesc *d_descs, int count, int64_t n_max, void *stream);
int launch_yhat_scatter(const void *sym, int wide, const int32_t *rank, float *y_hat, int M, int64_t hw, void *stream);
// checkerboard split (embed = false: [planes,h,w] -> [2,planes,h,w/2]) / merge (embed = true); w even, elem_bytes 2 or 4
int launch_ckbd(const void *src, void *dst, int64_t planes, int64_t h, int64_t w, int elem_bytes, int anchor_odd, bool embed,
                void *stream);
int launch_fastmath_selftest(int which, unsigned long long n, unsigned long long seed, unsigned long long *n_bad, void *stream);
// exhaustive check of the saturation lemmas behind the pruning; *n_bad (device) receives the number of violations
int launch_saturation_selftest(int mode, unsigned long long *n_bad, void *stream);

// ---- host rANS (fgmm_rans.cpp), integer only --------------------------------------------------------------
// returns 0 or an fgmm_status; *out malloc'ed
int rans_encode_symtab(const uint32_t *packed, const int32_t *symbols_or_null, int64_t n, int64_t n_bypass_hint,
                       uint8_t **out, size_t *out_len);
// two streams by one thread, interleaved (each output identical to rans_encode_symtab's)
int rans_encode_symtab2(const uint32_t *const packed[2], const int32_t *const symbols[2], const int64_t n[2],
                        const int64_t n_bypass_hint[2], uint8_t **out[2], size_t *out_len[2]);
// Tables that reach the host in pieces: after piece k has landed, the headers and rows of the first end[k] latents
// are valid (end[nseg-1] = n), the rows of piece k starting at base[k].  Piece 0 has landed before the decoder is called; wait(arg, k) blocks until piece k
// (k >= 1) has.  The decoder walks the latents in order, so it only ever waits for the next piece.
struct Landing {
  int nseg;
  const uint64_t *end;
  const uint8_t *const *base; // rows of piece k start at base[k] (the pieces need not be adjacent in memory)
  void *arg;
  int (*wait)(void *arg, int k); // FGMM_OK or an error status
};
// hdr16: null, or the headers in their 2-byte form (then `hdr` is ignored)
int rans_decode_cdftab(const uint8_t *enc, size_t enc_len, const uint32_t *hdr, const uint8_t *pool, int64_t n,
                       int32_t max_bs, int32_t *out, const Landing *land = nullptr, const uint16_t *hdr16 = nullptr);
FGMM_HD static inline bool tab_hdr_fits16(int32_t max_bs) { return 2 * max_bs + 2 <= 254; }

} // namespace fgmm

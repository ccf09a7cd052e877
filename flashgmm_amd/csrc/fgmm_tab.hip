// fgmm_tab.hip — decode-side edge tables on CDNA4 / gfx950 (MI355X): what the reference's per-symbol float bisection
// (RansDecoder::decode_with_indexes_gmm, compressai/cpp_exts/rans/rans_interface.cpp:826-862) can ever look at,
//     F_i[v] = (uint16)(cdf_i(v - 0.5) * 65535),   v in [-max_bs, max_bs + 1],
// evaluated once on the GPU, trimmed losslessly to the window outside which it is constant, and laid out for the host
// rANS decoder (format v5: fgmm_internal.h, include/flashgmm_amd.h).
//
//   tab_kernel            single pass, the production path (items whose half-width fits: tab_tl() > 0).
//                         Block = `tl` consecutive latents:
//                           0  lane = latent: twelve parameters -> LDS (64 B per latent, sigma clamped, refined reciprocal),
//                              the evaluation window [j_lo, j_hi) between the provably saturated tails
//                           1  block scan of the window lengths (in PAIRS of edges)
//                           2  FLATTENED over (latent, pair of consecutive edges): every lane of every wave evaluates two
//                              edges per step whatever the window lengths of its neighbours are (lane = latent left a
//                              third of the lanes idle), parameters from LDS — each is reused by every edge of its latent,
//                              ~51 times — both edges in packed fp32 (they share all twelve); edges -> LDS as uint16
//                           3  lane = latent: first non-zero edge, start of the trailing constant run, monotonicity ->
//                              header (2 / 4 / 8 bytes, straight into the host's layout) and row size
//                           4  block scan of the row sizes; ONE atomic add on the launch's cursor places the block's rows
//                           5  FLATTENED over the 4-byte words of the block's rows: formatted from LDS (uint16 rows, or
//                              Elias-Fano: low bytes + unary high parts), coalesced stores
//                         No temporary buffer in HBM, no second evaluation, no host round trip before the rows exist.
//   cdftab_count/scan/fill  generic two-pass path, lane = latent, any half-width (8-byte headers past 16382), rows
//                         sequential in latent order: the building-block API and items too wide for the LDS kernel.
// Both produce the same virtual table; tests compare them with each other and with the oracle.
#include <algorithm>

#include "fgmm_dev.h"

namespace fgmm {

// ---------------------------------------------------------------------------------------------------------
// one latent's parameters in registers (generic path; phase 0 of tab_kernel)
// ---------------------------------------------------------------------------------------------------------
template <int MODE, bool CLAMPED, typename PT> struct TabLatent {
  float mu[4], sg[4], pi[4], rs[4];
  int max_bs;

  __device__ __forceinline__ void load(const DecDesc &d, int c, int64_t p) {
    const int64_t base = (int64_t)c * d.stride_c + p * d.stride_p;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float s = ld1<PT>(d.scales, base + k * d.stride_k);
      sg[k] = CLAMPED ? clamp_scale(s) : s;
      rs[k] = CLAMPED ? rcp_refined(sg[k]) : 0.0f; // one refined reciprocal per component for the whole row
      mu[k] = ld1<PT>(d.means, base + k * d.stride_k);
      pi[k] = ld1<PT>(d.weights, base + k * d.stride_k);
    }
    if (d.logits) softmax4(pi);
    max_bs = d.max_bs;
  }
  __device__ __forceinline__ uint32_t edge(int j) const { // F[v = j - max_bs]
    const float x = (float)(j - max_bs) - 0.5f;
    if constexpr (CLAMPED) {
      bool ok = true;
      float c = mix4_clamped<MODE>(x, mu, sg, rs, pi, ok);
      if (__builtin_expect(!ok, 0))
        c = mix4_slow<MODE>(x, mu[0], mu[1], mu[2], mu[3], sg[0], sg[1], sg[2], sg[3], pi[0], pi[1], pi[2], pi[3]);
      return quant16(c);
    } else {
      return quant16(mix4<MODE>(x, mu, sg, pi));
    }
  }
};

// ---- evaluation window: skip the part of [-max_bs, max_bs+1] where every component is saturated -------------
// Saturation lemmas (fgmm_math.h Sat<MODE>, proved by exhaustive scan: fgmm_selftest_saturation):
//   all z_k <= -ZL  =>  F[v] == 0          all z_k >= +ZR  =>  F[v] == quant16((pi0+pi1)+(pi2+pi3))
// z_k(v) = ((float)v - 0.5f - mu_k) / sg_k is non-decreasing in v for finite mu and 0 < sg < inf (every IEEE
// operation is monotone), so it is enough to VERIFY the condition, with the kernel's own arithmetic, at one v:
// it then holds for every v beyond it.  Any latent whose parameters fall outside the lemmas' domain is
// evaluated over the full range instead.  Indices < j_lo are all zero, indices >= j_hi are all T_sat.
template <int MODE>
__device__ __forceinline__ void tab_window(const float (&mu)[4], const float (&sg)[4], const float (&pi)[4], int prune, int max_bs, int W,
                                           int &j_lo, int &j_hi, uint32_t &T_sat) {
  j_lo = 0;
  j_hi = W;
  T_sat = 0;
  if (prune) {
    bool ok = true;
    float tl = INFINITY, tr = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ok = ok && (sg[k] > 0.0f) && (sg[k] < INFINITY) && (fabsf(mu[k]) < INFINITY) && Sat<MODE>::weight_ok(pi[k]);
      tl = fminf(tl, __builtin_fmaf(-Sat<MODE>::ZL, sg[k], mu[k]));
      tr = fmaxf(tr, __builtin_fmaf(Sat<MODE>::ZR, sg[k], mu[k]));
    }
    if (ok) {
      const float lim = (float)max_bs + 4.0f;
      // left: largest candidate v with v - 0.5 <= tl, minus one for the rounding of tl itself
      const int vL = (int)fminf(fmaxf(floorf(tl + 0.5f) - 1.0f, -lim), lim);
      const int vR = (int)fminf(fmaxf(ceilf(tr + 0.5f) + 1.0f, -lim), lim);
      bool okL = true, okR = true;
      const float xl = (float)vL - 0.5f, xr = (float)vR - 0.5f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        okL = okL && ((xl - mu[k]) / sg[k] <= -Sat<MODE>::ZL);
        okR = okR && ((xr - mu[k]) / sg[k] >= Sat<MODE>::ZR);
      }
      const int64_t lo = (int64_t)vL + max_bs + 1, hi = (int64_t)vR + max_bs;
      if (okL) j_lo = (int)std::min<int64_t>(std::max<int64_t>(lo, 0), W); // indices < j_lo are v <= vL: all zero
      if (okR) j_hi = (int)std::min<int64_t>(std::max<int64_t>(hi, j_lo), W); // indices >= j_hi are v >= vR: all T_sat
      T_sat = quant16((pi[0] + pi[1]) + (pi[2] + pi[3]));
    }
  }
}

// trimming state of one row, fed edge by edge in index order (both paths run exactly this)
struct TrimState {
  int lead, run_start;
  bool allzero, nonmono;
  uint32_t prev;
  __device__ __forceinline__ void init(int j_lo) {
    lead = j_lo - 1;
    run_start = 0;
    allzero = true;
    nonmono = false;
    prev = 0;
  }
  __device__ __forceinline__ void step(int j, uint32_t E) {
    lead = (allzero && E == 0) ? j : lead; // index of the last leading zero
    allzero = allzero && E == 0;
    run_start = (j == 0 || E != prev) ? j : run_start;
    nonmono |= (j > 0) && (E < prev);
    prev = E;
  }
  // after the evaluated window: the saturated right part as one virtual step (F[j_hi .. W-1] == T_sat)
  __device__ __forceinline__ void finish(int j_hi, int W, uint32_t T_sat, int &a_idx, uint32_t &cnt) {
    if (j_hi < W) {
      if (allzero) {
        if (T_sat == 0) lead = W - 1; else allzero = false;
      }
      if (j_hi == 0 || T_sat != prev) run_start = j_hi;
      nonmono |= (j_hi > 0) && (T_sat < prev);
    }
    // the row starts at the first non-zero edge: F[v < a] = 0 is implied by the format, and the host takes "cf below the
    // first entry" as the interval [0, first entry) of the symbol before it
    a_idx = lead + 1;
    if (a_idx > run_start) a_idx = run_start;
    cnt = (uint32_t)(run_start - a_idx + 1);
  }
};

constexpr int kGenericMaxWindow = 1 << 20; // generic path: longest evaluation window of one latent (a lane walks it alone)

// ---------------------------------------------------------------------------------------------------------
// generic path, pass 1: window, every edge once -> header, row size -> block sums
// ---------------------------------------------------------------------------------------------------------
template <int MODE, bool CLAMPED, typename PT>
__global__ __launch_bounds__(kBlock) void cdftab_count_kernel(const DecDesc *__restrict__ descs) {
  const DecDesc &d = descs[blockIdx.z];
  const int cj = blockIdx.y;
  if (cj >= d.n_ch) return;
  const int64_t hw = d.hw;
  if ((int64_t)blockIdx.x * kBlock >= hw) return;
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool active = p < hw;
  const int c = d.chan_list ? d.chan_list[cj] : cj;
  TabLatent<MODE, CLAMPED, PT> L;
  L.load(d, c, active ? p : 0);
  const int max_bs = d.max_bs;
  const int W = 2 * max_bs + 2;

  int j_lo, j_hi;
  uint32_t T_sat;
  tab_window<MODE>(L.mu, L.sg, L.pi, d.prune, max_bs, W, j_lo, j_hi, T_sat);
  bool too_long = false;
  if (j_hi - j_lo > kGenericMaxWindow) { // refused (the host sees the flag): do not walk it
    too_long = active;
    j_hi = j_lo;
  }
  TrimState ts;
  ts.init(j_lo);
  for (int j = j_lo; j < j_hi; ++j) ts.step(j, L.edge(j));
  int a_idx;
  uint32_t cnt;
  ts.finish(j_hi, W, T_sat, a_idx, cnt);
  const uint32_t nm = ts.nonmono ? 1u : 0u;
  if (active) {
    const int64_t i = (int64_t)cj * hw + p;
    if (d.hdr_form == 8) static_cast<unsigned long long *>(d.hdr)[i] = tab_hdr8_pack(a_idx - max_bs, cnt, nm);
    else static_cast<uint32_t *>(d.hdr)[i] = tab_hdr_pack(a_idx - max_bs, cnt, nm);
  }
  __shared__ uint32_t s_tmp[kBlock / 64];
  // a block holds < 2^31 bytes of rows: 256 rows of at most 2 * (2^20 + 2) bytes
  const uint32_t total = block_reduce_add(active ? (uint32_t)tab_row_bytes(cnt, nm, d.ef_min) : 0u, s_tmp);
  const int any_nonmono = __syncthreads_or(active && ts.nonmono);
  const int any_long = __syncthreads_or(too_long);
  if (threadIdx.x == 0) {
    d.blk_sums[(int64_t)cj * d.tiles + blockIdx.x] = total | (any_nonmono ? 0x80000000u : 0u);
    if (any_long) d.pool_used[3] = 1ull;
  }
}

// one block per item: blk_off[b] = sum of blk_sums[0..b), pool_used[0] = total bytes, [1] = overflow flag,
// [2] = some row of the item is non-monotone  ([3] = some window was refused, set by the count pass)
__global__ __launch_bounds__(kBlock) void cdftab_scan_kernel(const DecDesc *__restrict__ descs) {
  const DecDesc &d = descs[blockIdx.x];
  const int64_t nb = (int64_t)d.n_ch * d.tiles;
  __shared__ unsigned long long s_tmp[kBlock / 64];
  __shared__ unsigned long long s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  uint32_t flagged = 0;
  for (int64_t b0 = 0; b0 < nb; b0 += kBlock) {
    const int64_t b = b0 + threadIdx.x;
    const uint32_t raw = b < nb ? d.blk_sums[b] : 0u;
    flagged |= raw >> 31;
    const uint32_t v = raw & 0x7FFFFFFFu;
    // 64-bit prefix: one block holds up to 2^29 bytes of rows (256 rows of a 2^20-edge window), 256 of them exceed 2^32
    const unsigned long long ex = block_scan_excl64(v, s_tmp);
    const unsigned long long carry = s_carry;
    if (b < nb) d.blk_off[b] = carry + ex;
    __syncthreads();
    if (threadIdx.x == kBlock - 1) s_carry = carry + ex + v;
    __syncthreads();
  }
  const int any_nonmono = __syncthreads_or((int)flagged);
  if (threadIdx.x == 0) {
    d.pool_used[0] = s_carry;
    d.pool_used[1] = s_carry > d.pool_cap ? 1ull : 0ull;
    d.pool_used[2] = (unsigned long long)(any_nonmono != 0);
  }
}

// ---------------------------------------------------------------------------------------------------------
// generic path, pass 2: re-evaluate the trimmed window and store the row at its offset (latent order)
// ---------------------------------------------------------------------------------------------------------
template <int MODE, bool CLAMPED, typename PT>
__global__ __launch_bounds__(kBlock) void cdftab_fill_kernel(const DecDesc *__restrict__ descs) {
  const DecDesc &d = descs[blockIdx.z];
  const int cj = (int)blockIdx.y;
  if (cj >= d.n_ch) return;
  const int64_t hw = d.hw;
  if ((int64_t)blockIdx.x * kBlock >= hw) return;
  if (d.pool_used[1] || d.pool_used[3]) return; // pool too small / a window refused: the host sees the flags
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool active = p < hw;
  const int c = d.chan_list ? d.chan_list[cj] : cj;
  TabLatent<MODE, CLAMPED, PT> L;
  L.load(d, c, active ? p : 0);

  int a_idx = 0;
  uint32_t cnt = 0, nonmono = 0;
  if (active) {
    const int64_t i = (int64_t)cj * hw + p;
    if (d.hdr_form == 8) {
      const unsigned long long h = static_cast<const unsigned long long *>(d.hdr)[i];
      a_idx = (int)((int64_t)(int32_t)(uint32_t)h + d.max_bs);
      cnt = (uint32_t)(h >> 32) & 0x7FFFFFFFu;
      nonmono = (uint32_t)(h >> 63);
    } else {
      const uint32_t h = static_cast<const uint32_t *>(d.hdr)[i];
      a_idx = tab_hdr_a(h) + d.max_bs;
      cnt = tab_hdr_cnt(h);
      nonmono = tab_hdr_nonmono(h);
    }
  }
  const uint32_t bytes = active ? (uint32_t)tab_row_bytes(cnt, nonmono, d.ef_min) : 0u;
  __shared__ uint32_t s_tmp[kBlock / 64];
  const uint32_t ex = block_scan_excl(bytes, s_tmp);
  if (!active) return;
  uint16_t *__restrict__ row = reinterpret_cast<uint16_t *>(d.pool + d.blk_off[(int64_t)cj * d.tiles + blockIdx.x] + ex); // 2-byte aligned

  if (!tab_row_is_ef(cnt, nonmono, d.ef_min)) {
    for (uint32_t j = 0; j < cnt; ++j) row[j] = (uint16_t)L.edge(a_idx + (int)j); // raw: uint16 entries
  } else {
    // Elias-Fano: one bit string written 16 bits at a time: the unary high parts, then the low parts (the lane walks its
    // row twice; this path serves the rare items too wide for the single-pass kernel)
    const uint32_t l = tab_ef_l(cnt);
    unsigned long long acc = 0; // bits not yet stored, from bit 0 up
    uint32_t nb = 0, at = 0;    // valid bits in acc | 16-bit units stored
    auto put = [&](uint32_t v, uint32_t bits) { // bits <= 16
      acc |= (unsigned long long)v << nb;
      nb += bits;
      while (nb >= 16) {
        row[at++] = (uint16_t)acc;
        acc >>= 16;
        nb -= 16;
      }
    };
    uint32_t pos_next = 0; // first bit of the high part not yet emitted
    for (uint32_t j = 0; j < cnt; ++j) {
      const uint32_t pos = (L.edge(a_idx + (int)j) >> l) + j; // strictly increasing: the row is monotone
      for (uint32_t gap = pos - pos_next; gap; gap -= gap < 16 ? gap : 16) put(0, gap < 16 ? gap : 16);
      put(1, 1);
      pos_next = pos + 1;
    }
    for (uint32_t gap = tab_ef_lb(cnt, l) - pos_next; gap; gap -= gap < 16 ? gap : 16) put(0, gap < 16 ? gap : 16);
    for (uint32_t j = 0; j < cnt; ++j) put(L.edge(a_idx + (int)j) & ((1u << l) - 1u), l);
    if (nb) put(0, 16 - nb);
  }
}

// =========================================================================================================
// tab_kernel — the single-pass path
// =========================================================================================================
// largest l in [0, n) with pref[l] <= t   (pref non-decreasing, pref[0] <= t < pref[n])
__device__ __forceinline__ int find_owner(const uint32_t *pref, int n, uint32_t t) {
  int lo = 0, hi = n; // pref[lo] <= t < pref[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (pref[mid] <= t) lo = mid; else hi = mid;
  }
  return lo;
}

struct TabSmem { // carve-up of the dynamic LDS of one block (every offset a multiple of 16)
  float4_t *P;       // [tl][4]: mu, sigma (clamped), pi, refined 1/sigma
  uint32_t *offP;    // [tl + 1] pairs of edges before latent l
  uint32_t *win;     // [tl] j_lo | (j_hi - j_lo) << 16
  uint32_t *meta;    // [tl] a_idx | cnt << 16   (tab_tl: W <= cap_e <= 32768, so both fit)
  uint32_t *rowoff;  // [tl + 1] byte offset of row l within the block's rows (rows are 2-byte aligned)
  uint16_t *tsat;    // [tl]
  uint8_t *flags;    // [tl] bit 0: parameters tame (fast evaluation allowed), bit 1: row non-monotone
  uint32_t *scratch; // [16]
  uint32_t *efoff;   // [tl + 1] Elias-Fano rows before latent l: entries (low 16 bits) | words of unary high parts (high 16 bits)
  uint32_t *bitmap;  // [cap_e / 32 + 9 * tl] the unary high parts of the block's Elias-Fano rows
  uint32_t *E32;     // [cap_e / 2] evaluated edges, two uint16 per word: entry k of latent l is uint16 2 * offP[l] + k
  __device__ __forceinline__ TabSmem(unsigned char *base, int tl, int cap_e) {
    size_t o = 0;
    auto take = [&](size_t bytes) {
      unsigned char *p = base + o;
      o += (bytes + 15) & ~(size_t)15;
      return p;
    };
    P = reinterpret_cast<float4_t *>(take(sizeof(float4_t) * 4 * (size_t)tl));
    offP = reinterpret_cast<uint32_t *>(take(4 * ((size_t)tl + 1)));
    win = reinterpret_cast<uint32_t *>(take(4 * (size_t)tl));
    meta = reinterpret_cast<uint32_t *>(take(4 * (size_t)tl));
    rowoff = reinterpret_cast<uint32_t *>(take(4 * ((size_t)tl + 1)));
    tsat = reinterpret_cast<uint16_t *>(take(2 * (size_t)tl));
    flags = reinterpret_cast<uint8_t *>(take((size_t)tl));
    scratch = reinterpret_cast<uint32_t *>(take(64));
    efoff = reinterpret_cast<uint32_t *>(take(4 * ((size_t)tl + 1)));
    bitmap = reinterpret_cast<uint32_t *>(take(4 * ((size_t)cap_e / 32 + 9 * (size_t)tl)));
    E32 = reinterpret_cast<uint32_t *>(take(2 * (size_t)cap_e));
  }
};
static size_t tab_smem_bytes(int tl, int cap_e) {
  auto r = [](size_t b) { return (b + 15) & ~(size_t)15; };
  return r(64 * (size_t)tl) + r(4 * ((size_t)tl + 1)) * 3 + r(4 * (size_t)tl) * 2 + r(2 * (size_t)tl) + r((size_t)tl) + 64 +
         r(4 * ((size_t)cap_e / 32 + 9 * (size_t)tl)) + r(2 * (size_t)cap_e);
}

// block-wide exclusive scan for up to kBlock values held by the first threads (others pass 0); total in *total
__device__ __forceinline__ uint32_t tab_scan(uint32_t v, uint32_t *scratch, uint32_t *total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) scratch[w] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kBlock / 64; ++i) {
    const uint32_t s = scratch[i];
    base += (i < w) ? s : 0u;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

#ifndef FGMM_TAB_WAVES
#define FGMM_TAB_WAVES 4
#endif
#ifdef FGMM_TAB_PROF // dev aid: cycles per phase of tab_kernel, summed over the blocks of every launch (fgmm_debug_tabprof)
__device__ unsigned long long g_tabprof[8];
#define TAB_T(i) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long t_ = clock64(); atomicAdd(&g_tabprof[i], t_ - tab_t0); tab_t0 = t_; } } while (0)
extern "C" int fgmm_debug_tabprof(unsigned long long *out, int reset) {
  int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tabprof), sizeof(g_tabprof));
  if (reset) {
    unsigned long long z[8] = {};
    rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tabprof), z, sizeof(z));
  }
  return rc;
}
#else
#define TAB_T(i)
#endif
template <int MODE, bool CLAMPED, typename PT>
__global__ __launch_bounds__(kBlock, FGMM_TAB_WAVES) void tab_kernel(const DecDesc *__restrict__ descs, int tl_max, int cap_e) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const DecDesc &d = descs[blockIdx.y];
  const int b = d.blk_begin + (int)blockIdx.x;
  if (b >= d.blk_end) return;
  const int tl = d.tl;
  const int64_t i0 = (int64_t)b * tl;
  const int nl = (int)std::min<int64_t>(tl, d.n - i0); // latents of this block (>= 1)
  TabSmem S(smem_raw, tl_max, cap_e);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int max_bs = d.max_bs;
  const int W = 2 * max_bs + 2;

#ifdef FGMM_TAB_PROF
  unsigned long long tab_t0 = clock64();
#endif
  // ---- phase 0: parameters -> LDS, evaluation window ------------------------------------------------------------
  uint32_t pairs = 0;
  if (tid < nl) {
    const int64_t i = i0 + tid;
    const int64_t cj = i / d.hw, p = i - cj * d.hw;
    const int c = d.chan_list ? d.chan_list[cj] : (int)cj;
    const int64_t base = (int64_t)c * d.stride_c + p * d.stride_p;
    float mu[4], sg[4], pi[4], rs[4];
    bool tame = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sg[k] = ld1<PT>(d.scales, base + k * d.stride_k);
      mu[k] = ld1<PT>(d.means, base + k * d.stride_k);
      pi[k] = ld1<PT>(d.weights, base + k * d.stride_k);
    }
    if (d.logits) softmax4(pi);
    if constexpr (CLAMPED) {
      Sigma4 S4;
      S4.set(sg[0], sg[1], sg[2], sg[3]);
      tame = S4.tame;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sg[k] = tame ? S4.sg[k] : clamp_scale(sg[k]); // a NaN sigma stays NaN (the IEEE path is taken for this latent)
        rs[k] = S4.rs[k];
      }
    } else {
      tame = false; // un-clamped sigma: IEEE division, scalar evaluation
#pragma unroll
      for (int k = 0; k < 4; ++k) rs[k] = 0.0f;
    }
    int j_lo, j_hi;
    uint32_t T_sat;
    tab_window<MODE>(mu, sg, pi, d.prune, max_bs, W, j_lo, j_hi, T_sat);
    S.P[4 * tid + 0] = (float4_t){mu[0], mu[1], mu[2], mu[3]};
    S.P[4 * tid + 1] = (float4_t){sg[0], sg[1], sg[2], sg[3]};
    S.P[4 * tid + 2] = (float4_t){pi[0], pi[1], pi[2], pi[3]};
    S.P[4 * tid + 3] = (float4_t){rs[0], rs[1], rs[2], rs[3]};
    S.win[tid] = (uint32_t)j_lo | ((uint32_t)(j_hi - j_lo) << 16);
    S.tsat[tid] = (uint16_t)T_sat;
    S.flags[tid] = tame ? 1 : 0;
    pairs = (uint32_t)(j_hi - j_lo + 1) >> 1;
  }
  // ---- phase 1: pairs before each latent -------------------------------------------------------------------------
  uint32_t NP;
  const uint32_t exP = tab_scan(pairs, S.scratch, &NP);
  if (tid < nl) S.offP[tid] = exP;
  if (tid == 0) S.offP[nl] = NP;
  __syncthreads();

  TAB_T(0); // phases 0 + 1
  // ---- phase 2: flattened evaluation, two consecutive edges per lane and step ------------------------------------
  {
    // each wave takes a contiguous quarter of the pairs (in steps of 64), so that from one step to the next a lane
    // moves on by 64 pairs: two or three latents further, found by walking the prefix array forward
    const uint32_t Q = (((NP + 3) >> 2) + 63u) & ~63u;
    const uint32_t t_end = std::min(NP, (uint32_t)(wave + 1) * Q);
    uint32_t t = (uint32_t)wave * Q + (uint32_t)lane;
    int l = 0;
    uint32_t l_beg = 0, l_end = 0;
    if (t < t_end) {
      l = find_owner(S.offP, nl, t);
      l_beg = S.offP[l];
      l_end = S.offP[l + 1];
    }
    for (; t < t_end; t += 64) {
      while (t >= l_end) { // next latent with a non-empty window
        ++l;
        l_beg = l_end;
        l_end = S.offP[l + 1];
      }
      const float4_t m4 = S.P[4 * l + 0], s4 = S.P[4 * l + 1], p4 = S.P[4 * l + 2], r4 = S.P[4 * l + 3];
      const float mu[4] = {m4[0], m4[1], m4[2], m4[3]}, pi[4] = {p4[0], p4[1], p4[2], p4[3]};
      const int j = (int)(S.win[l] & 0xFFFFu) + 2 * (int)(t - l_beg);
      const float x0 = (float)(j - max_bs) - 0.5f, x1 = (float)(j + 1 - max_bs) - 0.5f;
      float c0 = 0.0f, c1 = 0.0f;
      bool fast = false;
      if constexpr (CLAMPED) {
        Sigma4 S4;
        S4.tame = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          S4.sg[k] = s4[k];
          S4.rs[k] = r4[k];
        }
        bool ok = (S.flags[l] & 1) != 0;
        const f2 cc = mix4_clamped2<MODE>((f2){x0, x1}, mu, S4, pi, ok);
        c0 = cc.x;
        c1 = cc.y;
        fast = ok;
      }
      if (__builtin_expect(!fast, 0)) { // un-clamped sigma, NaN sigma, far-off or non-finite mean: IEEE evaluation
        c0 = mix4_slow<MODE>(x0, mu[0], mu[1], mu[2], mu[3], s4[0], s4[1], s4[2], s4[3], pi[0], pi[1], pi[2], pi[3]);
        c1 = mix4_slow<MODE>(x1, mu[0], mu[1], mu[2], mu[3], s4[0], s4[1], s4[2], s4[3], pi[0], pi[1], pi[2], pi[3]);
      }
      S.E32[t] = quant16(c0) | (quant16(c1) << 16);
    }
  }
  __syncthreads();
  const uint16_t *E16 = reinterpret_cast<const uint16_t *>(S.E32);

  TAB_T(1); // phase 2
  // ---- phase 3: trim every row, header, row size ------------------------------------------------------------------
  uint32_t bytes = 0;
  if (tid < nl) {
    const uint32_t w = S.win[tid];
    const int j_lo = (int)(w & 0xFFFFu), len = (int)(w >> 16), j_hi = j_lo + len;
    const uint16_t *e = E16 + 2 * (size_t)S.offP[tid];
    TrimState ts;
    ts.init(j_lo);
    // entries two at a time (one 32-bit LDS read), eight reads in flight: the loop is bound by LDS latency otherwise
    const uint32_t *e32 = S.E32 + S.offP[tid];
    int k = 0;
#pragma unroll 1
    for (; k + 16 <= len; k += 16) {
      uint32_t v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = e32[(k >> 1) + q];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        ts.step(j_lo + k + 2 * q, v[q] & 0xFFFFu);
        ts.step(j_lo + k + 2 * q + 1, v[q] >> 16);
      }
    }
    for (; k < len; ++k) ts.step(j_lo + k, e[k]);
    int a_idx;
    uint32_t cnt;
    ts.finish(j_hi, W, S.tsat[tid], a_idx, cnt);
    const uint32_t nm = ts.nonmono ? 1u : 0u;
    S.meta[tid] = (uint32_t)a_idx | (cnt << 16);
    if (nm) S.flags[tid] |= 2;
    bytes = (uint32_t)tab_row_bytes(cnt, nm, d.ef_min);
    const int64_t li = (int64_t)(b - d.blk_begin) * tl + tid; // latent index within this launch's header array
    if (d.hdr_form == 2) {
      const uint32_t c8 = nm ? kHdr2Escape : cnt; // W <= 254 here, so cnt <= 254
      static_cast<uint16_t *>(d.hdr_out)[li] = (uint16_t)((uint32_t)a_idx | (c8 << 8));
      if (nm) bytes += 4; // the escaped row carries its own 4-byte header
    } else if (d.hdr_form == 4) {
      static_cast<uint32_t *>(d.hdr_out)[li] = tab_hdr_pack(a_idx - max_bs, cnt, nm);
    } else {
      static_cast<unsigned long long *>(d.hdr_out)[li] = tab_hdr8_pack(a_idx - max_bs, cnt, nm);
    }
  }
  TAB_T(2); // phase 3
  // ---- phase 4: place the block's rows; Elias-Fano rows: entries and upper words before each ------------------------
  uint32_t B, EFT;
  const uint32_t exB = tab_scan(bytes, S.scratch, &B);
  const uint32_t B4 = (B + 3u) & ~3u; // blocks start 4-byte aligned (blk_off counts 4-byte units)
  uint32_t ef_pack = 0;
  if (tid < nl) {
    const uint32_t cnt = S.meta[tid] >> 16, nm = (S.flags[tid] >> 1) & 1u;
    if (tab_row_is_ef(cnt, nm, d.ef_min)) ef_pack = cnt | (((tab_ef_hb(cnt, tab_ef_l(cnt)) + 31u) >> 5) << 16); // both sums stay below 2^16 (cap_e <= 32768)
  }
  const uint32_t exEF = tab_scan(ef_pack, S.scratch, &EFT);
  if (tid < nl) {
    S.rowoff[tid] = exB;
    S.efoff[tid] = exEF;
  }
  // The block's place in the launch's row area: ONE returning atomic add per block on the launch's cursor (blocks lie in arrival
  // order; the chip does ~70 of those per microsecond: the kernel's bound only below 48 latents per block.  Round 4 built and
  // measured decoupled look-back instead - tables that are the same bytes on every run, a third slower, since a block cannot be
  // placed before every block ahead of it has been evaluated: profiles/r04_tab_place_sweep.txt, git history)
  if (tid == 0) {
    S.rowoff[nl] = B;
    S.efoff[nl] = EFT;
    const unsigned long long base = atomicAdd(&d.counters[0], (unsigned long long)B4);
    const bool fits = base + B4 <= d.rows_cap;
    if (!fits) atomicOr(&d.counters[1], 1ull);
    d.blkoff_out[b - d.blk_begin] = (uint32_t)(base >> 2);
    S.scratch[8] = (uint32_t)base;
    S.scratch[9] = (uint32_t)(base >> 32);
    S.scratch[10] = fits ? 1u : 0u;
    if (d.count_edges) atomicAdd(&d.counters[4 + (b & (kTabEdgeSlots - 1))], 2ull * NP);
  }
  for (uint32_t q = tid; q < (EFT >> 16); q += kBlock) S.bitmap[q] = 0;
  {
    const int any_nm = __syncthreads_or(tid < nl && (S.flags[tid] & 2)); // also publishes rowoff / efoff / scratch / bitmap
    if (tid == 0 && any_nm) atomicAdd(&d.counters[3], 1ull);
  }
  if (!S.scratch[10]) return; // the launch's row area is too small: the host re-runs it with what the cursor asks for

  // what a lane keeps of the row it is working on (reloaded from LDS only when it moves on to another latent)
  struct RowRef {
    const uint16_t *e; // first evaluated edge of the latent
    int j_lo, j_hi, a_idx;
    uint32_t cnt, nm, T_sat, aux; // aux: first entry / first word of the unary part of the row among the block's Elias-Fano rows
    uint32_t efl, HB, LB, M;      // Elias-Fano row: low bits, bits of the unary part, first bit of the low parts, 2^20 / efl rounded up; efl = 0: raw row
    __device__ __forceinline__ void load(const TabSmem &S, const uint16_t *E16, int l, uint32_t ef_min) {
      const uint32_t w = S.win[l], mt = S.meta[l];
      j_lo = (int)(w & 0xFFFFu);
      j_hi = j_lo + (int)(w >> 16);
      a_idx = (int)(mt & 0xFFFFu);
      cnt = mt >> 16;
      nm = (S.flags[l] >> 1) & 1u;
      T_sat = S.tsat[l];
      e = E16 + 2 * (size_t)S.offP[l] - j_lo; // e[idx] for j_lo <= idx < j_hi
      aux = S.efoff[l];
      efl = tab_row_is_ef(cnt, nm, ef_min) ? tab_ef_l(cnt) : 0u;
      HB = efl ? tab_ef_hb(cnt, efl) : 0u;
      LB = (HB + 7u) & ~7u;
      M = efl == 8u ? (1u << 20) / 8u + 1u : (efl == 12u ? (1u << 20) / 12u + 1u : (efl ? (1u << 20) / efl + 1u : 0u)); // no division for the two widths in use
    }
    // entry k of the row: F[a_idx + k] — an evaluated edge, or one of the two constants outside the evaluation window
    __device__ __forceinline__ uint32_t entry(uint32_t k) const {
      const int idx = a_idx + (int)k;
      return idx < j_lo ? 0u : (idx >= j_hi ? T_sat : (uint32_t)e[idx]);
    }
  };

  TAB_T(3); // phase 4
  // ---- phase 5a: unary high parts of the Elias-Fano rows, FLATTENED over their entries: bit ((E_j >> l) + j) ---------
  // Every lane takes a run of CONSECUTIVE entries (not every 64th): it stays inside one row for nearly all of them - the
  // row's description is loaded once, not once per entry - and the bits of entries that fall into one bitmap word are
  // gathered in a register and set with one LDS atomic (consecutive entries of a row are a few bits apart).
  {
    const uint32_t NE = EFT & 0xFFFFu;
    const uint32_t per = (NE + kBlock - 1u) / kBlock;
    uint32_t t = (uint32_t)tid * per;
    const uint32_t t_end = std::min(NE, t + per);
    if (t < t_end) {
      int lo = 0, hi = nl; // largest l with entries-before(l) <= t
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((S.efoff[mid] & 0xFFFFu) <= t) lo = mid; else hi = mid;
      }
      int l = lo;
      RowRef R;
      R.load(S, E16, l, d.ef_min);
      uint32_t l_beg = R.aux & 0xFFFFu, l_end = S.efoff[l + 1] & 0xFFFFu;
      uint32_t acc_w = ~0u, acc = 0; // bitmap word the gathered bits belong to | the bits
      for (; t < t_end; ++t) {
        if (t >= l_end) {
          do {
            ++l;
            l_beg = l_end;
            l_end = S.efoff[l + 1] & 0xFFFFu;
          } while (t >= l_end);
          R.load(S, E16, l, d.ef_min);
        }
        const uint32_t k = t - l_beg, pos = (R.entry(k) >> R.efl) + k;
        const uint32_t wi = (R.aux >> 16) + (pos >> 5);
        if (wi != acc_w) {
          if (acc) atomicOr(&S.bitmap[acc_w], acc);
          acc_w = wi;
          acc = 0;
        }
        acc |= 1u << (pos & 31u);
      }
      if (acc) atomicOr(&S.bitmap[acc_w], acc);
    }
  }
  __syncthreads(); // the unary parts are complete before phase 5b reads them
  uint8_t *__restrict__ out = d.rows + (((unsigned long long)S.scratch[9] << 32) | S.scratch[8]);

  TAB_T(4); // phase 5a
  // ---- phase 5b: FLATTENED over the 4-byte words of the block's rows; a word is two 16-bit units of one row, or - rows
  // are 2-byte aligned - the last unit of a row and the first of the next.  Every lane formats a run of CONSECUTIVE words
  // (a wave used to take 64 consecutive words per step, every lane landing in another row at every step: a search of the
  // row offsets and seven LDS reads of row description per word).  Now a lane meets one or two rows in all; its stores
  // are strided across the wave (each touches the lines its neighbours touch a step later: the L2 merges them).
  {
    const uint32_t NW = B4 >> 2;
    const uint32_t per = (NW + kBlock - 1u) / kBlock;
    uint32_t q = (uint32_t)tid * per;
    const uint32_t q_end = std::min(NW, q + per);
    int l = 0;
    uint32_t r_beg = 0, r_end = 0;
    RowRef R;
    if (q < q_end) {
      l = find_owner(S.rowoff, nl, 4 * q);
      r_beg = S.rowoff[l];
      r_end = S.rowoff[l + 1];
      R.load(S, E16, l, d.ef_min);
    }
    // 16-bit unit h of the row R describes
    auto unit = [&](uint32_t h) -> uint32_t {
      const bool escaped = d.hdr_form == 2 && R.nm;
      if (escaped && h < 2) return (tab_hdr_pack(R.a_idx - max_bs, R.cnt, 1u) >> (16 * h)) & 0xFFFFu;
      if (!R.efl) return R.entry(escaped ? h - 2 : h);
      const uint32_t bit0 = 16 * h;
      uint32_t v = 0;
      if (bit0 < R.HB) v = (S.bitmap[(R.aux >> 16) + (h >> 1)] >> (16 * (h & 1u))) & 0xFFFFu; // zero from bit HB on
      if (bit0 + 16 > R.LB) { // low parts that overlap bits [bit0, bit0 + 16)
        const int rel = (int)bit0 - (int)R.LB; // of the low area
        uint32_t j = rel > 0 ? ((uint32_t)rel * R.M) >> 20 : 0u; // rel / efl  (rel < 2^15: exact)
        const uint32_t mask = (1u << R.efl) - 1u;
        for (; j < R.cnt; ++j) {
          const int sh = (int)(j * R.efl) - rel;
          if (sh >= 16) break;
          const uint32_t lowbits = R.entry(j) & mask;
          v |= sh >= 0 ? lowbits << sh : lowbits >> -sh;
        }
        v &= 0xFFFFu;
      }
      return v;
    };
    for (; q < q_end; ++q) {
      uint32_t val = 0;
      if (4 * q < B) { // (else: the block's padding to 4 bytes... cannot be a whole word, kept for safety)
        if (4 * q >= r_end) {
          do {
            ++l;
            r_beg = r_end;
            r_end = S.rowoff[l + 1];
          } while (4 * q >= r_end);
          R.load(S, E16, l, d.ef_min);
        }
        const uint32_t h = (4 * q - r_beg) >> 1;
        val = unit(h);
        if (4 * q + 2 < r_end) {
          val |= unit(h + 1) << 16;
        } else if (4 * q + 2 < B) { // the upper half belongs to the next row (rows have at least one unit: no skipping)
          ++l;
          r_beg = r_end;
          r_end = S.rowoff[l + 1];
          R.load(S, E16, l, d.ef_min);
          val |= unit(0) << 16;
        }
      }
      stg<uint32_t>(out + 4 * (size_t)q, val);
    }
  }
  TAB_T(5); // phase 5b
}

// =========================================================================================================
// segdec_kernel — rANS decode of CHECKPOINTED bitstreams on the GPU: one WORKGROUP per segment, no tables at all
// =========================================================================================================
// A checkpointed bitstream (include/flashgmm_amd.h: fgmm_ckpt) falls into segments that can be decoded independently, each
// from its note of the coder state.  A Kodak batch call has thousands of them - enough to give every SIMD of the chip
// several waves - so the per-symbol chain  cf = x & 0xFFFF -> which symbol's interval holds cf -> advance the state  can run
// ON the GPU, next to the parameters, and the decode-side tables (57.6 B/latent across PCIe, the bound of the table path)
// need not exist.  Per segment a PRODUCER wave (two in small launches) evaluates the edges F[v] of the next batch of latents
// into LDS - the window between the saturated tails, the same arithmetic as tab_kernel - while the CONSUMER wave decodes the
// current batch: lane = one edge of the symbol's latent, one compare + popcount counts the edges <= cf - the wavefront form
// of the reference's bisection (rans_interface.cpp:826-862) - and the coder state is advanced on the scalar unit.
// The kernel only accepts what it can decide exactly as the reference does: a MONOTONE window with one interval around cf.
// Anything else (a non-monotone row, cf beyond every edge) flags the segment "hard"; a segment that does not end in the next
// checkpoint's (state, position) flags "mismatch" - the host then decodes that bitstream through the table path, so the
// result is the sequential decoder's in every case and a wrong note costs time, never a symbol.
#ifdef FGMM_SEG_PROF // dev aid: cycles per phase, summed over the waves of every launch (read with fgmm_debug_segprof)
__device__ unsigned long long g_segprof[8];
__device__ unsigned long long g_segtimes[2 * 16384]; // begin, end of the first 16384 segments of the last launch
#define SEG_T(v) const unsigned long long v = clock64()
#define SEG_ACC(i, v) seg_acc[i] += (unsigned long long)(v)
#else
#define SEG_T(v)
#define SEG_ACC(i, v)
#endif
constexpr int kSegCapE = 2048; // edges (uint16) of one batch of latents in LDS
// One batch as the producer wave hands it to the consumer wave (two of them: one being filled while the other is decoded)
constexpr int kSegBufE = 0;                       // uint16[kSegCapE] edges, latent after latent (each from an even index on)
constexpr int kSegBufPa = 2 * kSegCapE;           // uint32[64] first edge (12 bits) | edges (7) | j_lo - 1 - max_bs (13, signed)
constexpr int kSegBufPb = kSegBufPa + 256;        // uint32[64] F beyond the window (16; 0: nothing up there) | slow | decreases | has a tail
constexpr int kSegBufPc = kSegBufPb + 256;        // uint32[64] j_lo | edges << 16 (the slow path's: windows beyond 63 edges)
constexpr int kSegBufOut = kSegBufPc + 256;       // int64[64] where the latent's value goes in y_hat
constexpr int kSegBufNk = kSegBufOut + 512;       // uint32 latents in the batch
constexpr int kSegBufBytes = kSegBufNk + 16;
constexpr int kSegLdsP = 0;                       // producer's own: parameters of 64 latents | pair offsets | windows, tails, flags
constexpr int kSegLdsOff = 64 * 64;
constexpr int kSegLdsCtrl = kSegLdsOff + 4 * 68 + 3 * 4 * 64;
constexpr int kSegLdsBuf = kSegLdsCtrl + 32;
constexpr int kSegLds = kSegLdsBuf + 2 * kSegBufBytes; // 15.9 KB per segment: ten segments per CU
// Two or three waves per segment (launch_segdec).  The PRODUCER (wave 1; wave 2 takes a share of B) does what needs no coder
// state, a batch of up to 64 latents ahead:
//   A lane = latent: twelve parameters, clamp + reciprocals, the window between the saturated tails (tab_kernel's phase 0);
//     as many latents as fit kSegCapE edges form the batch
//   B flattened over (latent, pair of consecutive edges): tab_kernel's phase 2, the same packed arithmetic, edges -> LDS; the
//     pair in the lane below tells whether the row decreases there
//   per latent: everything of the search that does not depend on cf, packed into two words
// The CONSUMER (wave 0) walks the batch symbol by symbol on the scalar unit: lane = edge, one compare + popcount is the
// reference's bisection in a monotone row, two v_readlane pick F[J] and F[J + 1], the 64-bit state advances in SGPRs.  While it
// decodes batch n the producer evaluates batch n + 1 (on another SIMD of the CU): the sequential chain never waits for edges.
#ifdef FGMM_SEG_WAVES // (experiments: cap the registers for this many waves per SIMD)
#define FGMM_SEG_OCC __attribute__((amdgpu_waves_per_eu(FGMM_SEG_WAVES, FGMM_SEG_WAVES)))
#else
#define FGMM_SEG_OCC
#endif
template <int MODE, bool CLAMPED, typename PT>
__global__ __launch_bounds__(192) FGMM_SEG_OCC void segdec_kernel(const SegDesc *__restrict__ descs, const SegRef *__restrict__ segs, int64_t n_segs) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[kSegLds];
  float4_t *const P = reinterpret_cast<float4_t *>(lds + kSegLdsP);         // [64][4]: mu, sigma (clamped), pi, refined 1/sigma
  uint32_t *const offP = reinterpret_cast<uint32_t *>(lds + kSegLdsOff);    // [65] pairs before latent l of the batch
  uint32_t *const winL = offP + 68, *const tsfL = winL + 64; // [64] j_lo | len << 16;  T_sat | tame << 16  (read by OTHER lanes in
                                                             // phase B, some of which have left the loop: not a cross-lane register read)
  uint32_t *const nmL = tsfL + 64;                           // [64] the latent's row decreases somewhere
  // [0..1] the consumer has given up (per batch parity) | [2] producer 0 has laid out batch seq | [3] producer 1 has evaluated its share
  // of batch seq | [4] latents, [5] pairs of that batch | [6] a producer waited in vain
  volatile uint32_t *const ctrl = reinterpret_cast<volatile uint32_t *>(lds + kSegLdsCtrl);
  const int64_t wid = blockIdx.x;
  if (wid >= n_segs) return;
  const uint32_t lane = threadIdx.x & 63u;
  const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // 0: the consumer, 1 / 2: producers
  const bool producer = role != 0;
  const bool two_producers = blockDim.x > 128; // (the launch's choice: see launch_segdec)
  const SegRef ref = segs[wid];
  const SegDesc &d = descs[ref.item];
  const int64_t sg = ref.seg;
  const int64_t lo = sg * d.stride, hi = sg == d.n_ckpt ? d.n : lo + d.stride;
  const int32_t max_bs = d.max_bs;
  const int W = 2 * max_bs + 2; // <= kSegCapE (the host sends wider items through the table path)
  auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
  auto bcast = [](uint32_t v, uint32_t k) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)k); };
  auto buf = [&](int n) { return lds + kSegLdsBuf + (n & 1) * kSegBufBytes; };
#ifdef FGMM_SEG_PROF
  unsigned long long seg_acc[6] = {};
#endif
  SEG_T(t_begin);
#ifdef FGMM_SEG_PROF
  const unsigned long long w_begin = wall_clock64(); // (100 MHz, one counter for the chip: clock64 is per XCD)
#endif

  // =============================== producer: batch [base, base + nk) -> B ===============================
  // (producers meet at LDS flags: a partial barrier does not exist, and the consumer must not wait for them inside a batch)
  auto publish = [&](int slot, uint32_t v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) ctrl[slot] = v;
  };
  auto await = [&](int slot, uint32_t v) {
    for (int spins = 0; uni(ctrl[slot]) != v; ++spins) {
      __builtin_amdgcn_s_sleep(2);
      if (spins > (1 << 20)) { // (never seen; a segment that gives up is decoded by the table path)
        if (lane == 0) ctrl[6] = 1;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  // B. flattened over (latent, pair of consecutive edges), pairs [t0 + lane, t_end) in steps of 64: the tab_kernel's evaluation
  auto eval_pairs = [&](uint32_t t0, uint32_t t_end, int nk, uint32_t NP, uint32_t *E32) {
    int l = 0;
    uint32_t l_beg = 0, l_end = 0;
    uint32_t t = t0 + lane;
    if (t < t_end) {
      l = find_owner(offP, nk, t);
      l_beg = offP[l];
      l_end = l + 1 < nk ? offP[l + 1] : NP;
    }
    uint32_t carry = 0; // the pair before lane 0's: lane 63's of the step before (none before the first step: see the seam below)
    for (; t < t_end; t += 64) {
      while (t >= l_end) { // next latent with a non-empty window
        ++l;
        l_beg = l_end;
        l_end = l + 1 < nk ? offP[l + 1] : NP;
      }
      const float4_t m4 = P[4 * l + 0], s4 = P[4 * l + 1], p4 = P[4 * l + 2], r4 = P[4 * l + 3];
      const float mu_[4] = {m4[0], m4[1], m4[2], m4[3]}, pi_[4] = {p4[0], p4[1], p4[2], p4[3]};
      const uint32_t wl = winL[l], tl_ = tsfL[l];
      const int j = (int)(wl & 0xFFFFu) + 2 * (int)(t - l_beg);
      const float x0 = (float)(j - max_bs) - 0.5f, x1 = (float)(j + 1 - max_bs) - 0.5f;
      float c0 = 0.0f, c1 = 0.0f;
      bool fast = false;
      if constexpr (CLAMPED) {
        Sigma4 S4;
        S4.tame = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          S4.sg[k] = s4[k];
          S4.rs[k] = r4[k];
        }
        bool ok = (tl_ >> 16) != 0;
        const f2 cc = mix4_clamped2<MODE>((f2){x0, x1}, mu_, S4, pi_, ok);
        c0 = cc.x;
        c1 = cc.y;
        fast = ok;
      }
      if (__builtin_expect(!fast, 0)) { // un-clamped sigma, NaN sigma, far-off or non-finite mean: IEEE evaluation
        c0 = mix4_slow<MODE>(x0, mu_[0], mu_[1], mu_[2], mu_[3], s4[0], s4[1], s4[2], s4[3], pi_[0], pi_[1], pi_[2], pi_[3]);
        c1 = mix4_slow<MODE>(x1, mu_[0], mu_[1], mu_[2], mu_[3], s4[0], s4[1], s4[2], s4[3], pi_[0], pi_[1], pi_[2], pi_[3]);
      }
      const uint32_t q0 = quant16(c0), q1 = quant16(c1), pk = q0 | (q1 << 16);
      E32[t] = pk;
      // ... and which rows DECREASE somewhere.  A count of the edges <= cf is the symbol's interval only in a monotone row; the
      // reference's bisection may answer differently when the row decreases anywhere (rans_interface.cpp:833-854), so such a
      // latent is left to the table path.  The pair before this one sits in the lane below (wave_shr:1; lane 0: `carry`); an
      // odd window's last pair holds F[j_hi] as well: a real edge, checked like the others.
      const uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)pk, 0x138, 0xF, 0xF, false);
      const uint32_t prev = t > l_beg ? before >> 16 : 0u; // F below the window is 0
      if (prev > q0 || q0 > q1) nmL[l] = 1;
      carry = bcast(pk, 63u);
    }
  };
  // producer 0 evaluates the pairs below the seam (it also does A and the per-latent pass), producer 1 those from the seam on
  auto seam = [](uint32_t NP) { return std::min(NP, ((NP * 7u) / 16u + 63u) & ~63u); };
  auto produce1 = [&](unsigned char *B, uint32_t seq) {
    await(2, seq);
    const int nk = (int)uni(ctrl[4]);
    const uint32_t NP = uni(ctrl[5]);
    eval_pairs(seam(NP), NP, nk, NP, reinterpret_cast<uint32_t *>(B + kSegBufE));
    __builtin_amdgcn_s_waitcnt(0xc07f);
    publish(3, seq);
  };
  auto produce = [&](int64_t base, unsigned char *B, uint32_t seq) {
    SEG_T(t_a);
    uint32_t *const E32 = reinterpret_cast<uint32_t *>(B + kSegBufE);
    const uint16_t *const E16 = reinterpret_cast<const uint16_t *>(E32);
    // ---- A. lane = latent base + lane: parameters -> LDS, evaluation window; how many latents fit the edge budget
    const int64_t i = std::min(base + lane, hi - 1);
    const int64_t cj = i / d.hw, p = i - cj * d.hw;
    const int c = d.chan_list ? ldg<int32_t>(d.chan_list + cj) : (int)cj;
    const int64_t pbase = (int64_t)c * d.stride_c + p * d.stride_p;
    float mu[4], sgm[4], pi[4], rs[4];
    bool tame = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sgm[k] = ld1<PT>(d.scales, pbase + k * d.stride_k);
      mu[k] = ld1<PT>(d.means, pbase + k * d.stride_k);
      pi[k] = ld1<PT>(d.weights, pbase + k * d.stride_k);
    }
    if (d.logits) softmax4(pi);
    if constexpr (CLAMPED) {
      Sigma4 S4;
      S4.set(sgm[0], sgm[1], sgm[2], sgm[3]);
      tame = S4.tame;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sgm[k] = tame ? S4.sg[k] : clamp_scale(sgm[k]);
        rs[k] = S4.rs[k];
      }
    } else {
      tame = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) rs[k] = 0.0f;
    }
    int j_lo, j_hi;
    uint32_t T_sat;
    tab_window<MODE>(mu, sgm, pi, 1, max_bs, W, j_lo, j_hi, T_sat);
    P[4 * lane + 0] = (float4_t){mu[0], mu[1], mu[2], mu[3]};
    P[4 * lane + 1] = (float4_t){sgm[0], sgm[1], sgm[2], sgm[3]};
    P[4 * lane + 2] = (float4_t){pi[0], pi[1], pi[2], pi[3]};
    P[4 * lane + 3] = (float4_t){rs[0], rs[1], rs[2], rs[3]};
    const uint32_t win = (uint32_t)j_lo | ((uint32_t)(j_hi - j_lo) << 16);
    const int nk_max = (int)std::min<int64_t>(64, hi - base);
    const uint32_t pairs = (int)lane < nk_max ? (uint32_t)(j_hi - j_lo + 1) >> 1 : 0u;
    uint32_t incl = pairs;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
      if (lane >= (uint32_t)o) incl += t;
    }
    // latents 0 .. nk - 1 of the batch fit kSegCapE edges (one latent always does: W <= kSegCapE)
    const int nk = std::max(1, (int)__popcll(__ballot((int)lane < nk_max && 2 * incl <= (uint32_t)kSegCapE)));
    const uint32_t excl = incl - pairs;
    offP[lane] = excl;
    winL[lane] = win;
    tsfL[lane] = T_sat | (tame ? 0x10000u : 0u);
    nmL[lane] = 0;
    const uint32_t NP = bcast(incl, (uint32_t)(nk - 1));
    if (lane == 0) {
      offP[64] = NP; // (offP[nk] is what a walk past the last latent reads: lanes' own entries hold it for nk < 64)
      ctrl[4] = (uint32_t)nk;
      ctrl[5] = NP;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): this wave's LDS writes are done
    __builtin_amdgcn_wave_barrier();
    if (two_producers) publish(2, seq); // producer 1 may start
    SEG_T(t_b);
    // ---- B. this producer's share of the pairs
    const uint32_t H = two_producers ? seam(NP) : NP;
    eval_pairs(0u, H, nk, NP, E32);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (two_producers) await(3, seq); // producer 1's share is in LDS
    if (lane == 0 && H < NP) { // the seam: producer 1's first pair against the one before it
      const int l = find_owner(offP, nk, H);
      if (H > offP[l] && (E32[H - 1] >> 16) > (E32[H] & 0xFFFFu)) nmL[l] = 1;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // ---- per latent, lane = latent: everything of the search that does not depend on cf.  A latent is "plain" when its window is
    // monotone - into the saturated tail too -, fits one pass of the wave with a lane to spare, and cannot put cf below F[0]
    const uint32_t len_l = win >> 16, jl_l = win & 0xFFFFu;
    const bool tail_l = (int)(jl_l + len_l) < W;
    const bool mine = (int)lane < nk && len_l != 0; // (lanes past the batch would read past its edges)
    const uint32_t first_l = mine ? (uint32_t)E16[2u * excl] : 1u, last_l = mine ? (uint32_t)E16[2u * excl + len_l - 1u] : 0u;
    const uint32_t nm_l = nmL[lane];
    const bool slow_l = nm_l != 0 || len_l > 63u || (tail_l && T_sat < last_l) || (jl_l == 0 && first_l != 0);
    uint32_t *const Bw = reinterpret_cast<uint32_t *>(B);
    Bw[kSegBufPa / 4 + lane] = (2u * excl) | ((len_l & 0x7Fu) << 12) | ((uint32_t)((int)jl_l - 1 - max_bs) << 19);
    Bw[kSegBufPb / 4 + lane] = (tail_l ? T_sat : 0u) | (slow_l ? 0x10000u : 0u) | (nm_l ? 0x20000u : 0u) | (tail_l ? 0x40000u : 0u);
    Bw[kSegBufPc / 4 + lane] = win;
    reinterpret_cast<int64_t *>(B + kSegBufOut)[lane] = (int64_t)c * d.hw + p;
    if (lane == 0) Bw[kSegBufNk / 4] = (uint32_t)nk;
    SEG_T(t_e);
    SEG_ACC(0, t_b - t_a);
    SEG_ACC(1, t_e - t_b);
  };

  // =============================== consumer: the coder ===============================
  // the coder at symbol lo: the stream's own head (segment 0) or the checkpoint before this segment.  (The descriptor's pointers
  // are generic: a load through one counts as divergent and everything computed from it leaves the scalar unit - hence global
  // loads and an explicit "this is uniform" on what the coder starts from.)
  const uint32_t *__restrict__ w = d.words + 2;
  const int64_t nw = d.n_words - 2;
  uint32_t x_lo = 0, x_hi = 0, err = kSegOk, wv = 0, wp = 0, wleft = 0;
  int64_t wbase = 0;
  auto uni64 = [&](uint64_t v) { return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v); };
  if (!producer) {
    if (sg == 0) {
      x_lo = uni(ldg<uint32_t>(d.words));
      x_hi = uni(ldg<uint32_t>(d.words + 1));
    } else {
      const uint64_t cx = ldg<uint64_t>(&d.ckpt[sg - 1].x), cp = ldg<uint64_t>(&d.ckpt[sg - 1].pos);
      x_lo = uni((uint32_t)cx);
      x_hi = uni((uint32_t)(cx >> 32));
      wbase = (int64_t)uni64(cp);
    }
    if (wbase < 0 || wbase > nw) {
      err = kSegMismatch;
      wbase = 0;
    }
    // the next 64 words of the bitstream across the lanes; `wp` of them are consumed, `wleft` are left in the stream
    wv = wbase + lane < nw ? ldg<uint32_t>(w + wbase + lane) : 0u;
    wleft = (uint32_t)std::min<int64_t>(nw - wbase, 0x7FFFFFFF);
  }
  uint32_t sout = 0; // the stream ran out
  auto next_word = [&]() -> uint32_t { // wave-uniform
    if (wp == 64) {
      wbase += 64;
      wv = wbase + lane < nw ? ldg<uint32_t>(w + wbase + lane) : 0u;
      wp = 0;
    }
    if (wleft == 0) sout = 1; else --wleft;
    const uint32_t r = bcast(wv, wp);
    ++wp;
    return r;
  };
  // x = freq * (x >> 16) + bias, 64 bits, spelled out for the scalar unit (the compiler multiplies 64-bit values on the VALU)
  auto advance = [&](uint32_t freq, uint32_t bias) {
    const uint64_t xs = (((uint64_t)x_hi << 32) | x_lo) >> 16;
    const uint32_t s_lo = (uint32_t)xs, s_hi = (uint32_t)(xs >> 32);
    uint32_t lo_, hi_, t;
    asm("s_mul_hi_u32 %1, %3, %4\n\ts_mul_i32 %2, %3, %5\n\ts_mul_i32 %0, %3, %4\n\ts_add_u32 %1, %1, %2\n\ts_add_u32 %0, %0, %6\n\ts_addc_u32 %1, %1, 0"
        : "=&s"(lo_), "=&s"(hi_), "=&s"(t)
        : "s"(freq), "s"(s_lo), "s"(s_hi), "s"(bias)
        : "scc");
    if (hi_ == 0 && lo_ < 0x80000000u) { // x < 2^31: one more word (Rans64DecRenorm, rans64.h:136-142)
      hi_ = lo_;
      lo_ = next_word();
    }
    x_lo = lo_;
    x_hi = hi_;
  };
  auto consume = [&](const unsigned char *B, int nk) {
    SEG_T(t_c);
    const uint16_t *const E16 = reinterpret_cast<const uint16_t *>(B + kSegBufE);
    const uint32_t *const Bw = reinterpret_cast<const uint32_t *>(B);
    const uint32_t pa = Bw[kSegBufPa / 4 + lane], pb = Bw[kSegBufPb / 4 + lane], pc = Bw[kSegBufPc / 4 + lane];
    int32_t myval = 0;
    // myval of lane k = v (both wave-uniform; two SGPR operands exceed the constant bus: the lane goes through m0, which the
    // compiler sets up itself before each of its own uses - the clobber keeps the statement out of such a pair)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    auto put = [&](int32_t v, int k) { asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(myval) : "s"(v), "s"(k) : "m0"); };
#pragma clang diagnostic pop
    uint32_t bad_acc = 0; // hard cases of the straight path, looked at once per batch (nothing below can leave the stream or the LDS)
    // (the edges of symbol k + 1 do not depend on the coder: they are read from LDS while symbol k is searched and advanced)
    uint32_t a_n = bcast(pa, 0u), b_n = bcast(pb, 0u);
    uint32_t E_n = (uint32_t)E16[(a_n & 0xFFFu) + lane]; // (lanes past the window read their neighbours' edges: masked out below)
    for (int k = 0; k < nk; ++k) {
      // ---- a run of plain symbols: one exit, nothing but the straight path inside
      uint32_t a, b, E, cf;
      for (;;) {
        a = a_n, b = b_n, E = E_n;
        const uint32_t k1 = (uint32_t)std::min(k + 1, nk - 1);
        a_n = bcast(pa, k1);
        b_n = bcast(pb, k1);
        E_n = (uint32_t)E16[(a_n & 0xFFFu) + lane];
        cf = x_lo & 0xFFFFu;
        if (__builtin_expect((((cf + 1u) | b) & 0x10000u) != 0, 0)) break; // a bypass escape (cf = 0xFFFF) or a latent that is not plain
        // count the edges <= cf: the wavefront form of the reference's bisection (rans_interface.cpp:826-862) in a monotone row.
        // Lanes past the window hold 0x10000 for the count and for "the edge before the first" (& 0xFFFF: F below the window is 0;
        // lane 63 is never inside a plain window), and F beyond the window for "the edge after the last".
        const bool inside = lane < ((a >> 12) & 0x7Fu);
        const uint32_t Ec = inside ? E : 0x10000u, Ey = inside ? E : (b & 0xFFFFu);
        const uint32_t n = (uint32_t)__popcll(__ballot(Ec <= cf));
        const uint32_t start = bcast(Ec, n - 1u) & 0xFFFFu, next = bcast(Ey, n);
        // F[J] <= cf < F[J + 1], J = j_lo + n - 1 in [0, W - 2] (rans_interface.cpp:826-833): J >= 0 by the producer's choice of
        // plain latents; J = W - 1 has next = 0 and so a freq <= 0
        const uint32_t freq = next - start;
        // ... and the interval must HOLD cf: with every window edge <= cf the walk ends at the saturated tail, next = T_sat, and
        // a cf in [T_sat, 0xFFFF) - weights that sum to just under 1; only a corrupt stream asks for it - lies beyond it: the
        // reference's bisection ends elsewhere (rans_interface.cpp:865-875), so the segment goes back to the table path
        bad_acc |= ((freq - 1u) >> 16) | (uint32_t)(cf - start >= freq);
        advance(freq, cf - start); // Rans64DecAdvance, rans64.h:124-134
        put(((int32_t)a >> 19) + (int32_t)n, k);
        if (++k >= nk) break;
      }
      if (k >= nk) break;
      if (bad_acc | sout) break; // (the paths below loop over the stream: not with a state that is already known to be wrong)
      int32_t value;
      if (cf == 0xFFFFu) { // bypass escape: Rans64DecAdvance(65535, 1), then the nibbles (rans_interface.cpp:808-824)
        uint64_t xx = ((((uint64_t)x_hi << 32) | x_lo) >> 16) + cf - 0xFFFFu;
        auto renorm = [&]() {
          if (xx < (1ull << 31)) xx = (xx << 32) | next_word();
        };
        renorm();
        auto get_bits = [&]() {
          const uint32_t v = (uint32_t)xx & 15u;
          xx >>= 4;
          renorm();
          return v;
        };
        int32_t val = (int32_t)get_bits(), nn = val;
        while (val == 15 && !sout) {
          val = (int32_t)get_bits();
          nn += val;
        }
        uint32_t raw = 0;
        for (int j = 0; j < nn && !sout; ++j) raw |= get_bits() << ((j * 4) & 31);
        value = (int32_t)raw;
        x_lo = uni((uint32_t)xx);
        x_hi = uni((uint32_t)(xx >> 32));
      } else { // a window of 64 edges and more, or a row the reference's bisection must see itself (left to the table path)
        const uint32_t wk = bcast(pc, (uint32_t)k), e0 = a & 0xFFFu;
        const int jl = (int)(wk & 0xFFFFu), len = (int)(wk >> 16);
        const bool tail = (b & 0x40000u) != 0;
        const uint32_t tsat = b & 0xFFFFu;
        uint32_t below = 0, last = 0, start = 0, next = 0; // `last`: the edge before this pass (F below the window is 0)
        bool found = false, bad = (b & 0x20000u) != 0;
        for (int off = 0; off < len; off += 64) {
          const int q = off + (int)lane;
          const bool valid = q < len;
          const uint32_t Eq = valid ? (uint32_t)E16[e0 + (uint32_t)q] : 0x10000u; // lanes past the window never count as <= cf
          const uint32_t n_valid = (uint32_t)std::min(64, len - off);
          if (!found) {
            const uint32_t n_le = (uint32_t)__popcll(__ballot(valid && Eq <= cf));
            if (n_le < n_valid) { // the first edge > cf lies in this pass: F[J] = the edge before it, F[J + 1] = that edge
              next = bcast(Eq, n_le);
              start = n_le ? bcast(Eq, n_le - 1) : last;
              found = true;
            }
            below += n_le;
          }
          last = bcast(Eq, n_valid - 1);
          if (found) { // (the rest of a long window is only needed for its last edge)
            if (off + 64 < len) last = (uint32_t)E16[e0 + (uint32_t)len - 1u];
            break;
          }
        }
        bad = bad || (tail && tsat < last); // ... monotone into the saturated tail too
        if (!found) { // every edge of the window is <= cf: the interval ends at the saturated tail, if there is one above cf
          start = last;
          next = tsat;
          bad = bad || !(tail && tsat > cf);
        }
        // F[J] <= cf < F[J + 1] with J = jl + below - 1 >= 0  (J = jl - 1: the zeros below the window; the reference's range is
        // J in [0, W - 2]: rans_interface.cpp:826-833)
        const int J = jl + (int)below - 1;
        const uint32_t freq = next - start;
        if (__builtin_expect(bad || J < 0 || J > W - 2 || freq == 0 || freq > 0xFFFFu, 0)) {
          bad_acc = 1;
          break;
        }
        value = J - max_bs;
        advance(freq, cf - start);
        x_lo = uni(x_lo);
        x_hi = uni(x_hi);
      }
      if (lane == (uint32_t)k) myval = value;
    }
    if (sout) err = kSegStream;
    else if (bad_acc) err = kSegHard;
    if (!err && (int)lane < nk) stg<float>(d.y_hat + reinterpret_cast<const int64_t *>(B + kSegBufOut)[lane], (float)myval);
    SEG_T(t_e);
    SEG_ACC(3, t_e - t_c);
    SEG_ACC(4, nk);
    SEG_ACC(5, 1);
  };

  // =============================== the two in step: batch n is decoded while batch n + 1 is evaluated ===============================
  if (threadIdx.x < 8) ctrl[threadIdx.x] = 0;
  __syncthreads();
  if (role == 1) produce(lo, buf(0), 1u);
  else if (role == 2) produce1(buf(0), 1u);
  else if (lane == 0) ctrl[0] = err;
  __syncthreads();
  int64_t base = lo;
  for (int n = 0;; ++n) {
    unsigned char *const B = buf(n);
    if (ctrl[n & 1]) break; // (batch 0: a note that points outside the stream)
    const int nk = (int)uni(reinterpret_cast<const uint32_t *>(B)[kSegBufNk / 4]);
    const int64_t next = base + nk;
    if (producer) {
      if (next < hi) {
        if (role == 1) produce(next, buf(n + 1), (uint32_t)n + 2u);
        else produce1(buf(n + 1), (uint32_t)n + 2u);
      }
    } else {
      consume(B, nk);
      if (lane == 0) ctrl[(n + 1) & 1] = err;
    }
    SEG_T(t_w);
    __syncthreads();
    SEG_ACC(2, clock64() - t_w);
    if (next >= hi) break;
    base = next;
  }
  if (!producer) {
    // ---- the segment must end exactly where the next checkpoint says the coder stands
    if (!err && ctrl[6]) err = kSegHard;
    if (!err && sg < d.n_ckpt) {
      const uint64_t cx = ldg<uint64_t>(&d.ckpt[sg].x), cp = ldg<uint64_t>(&d.ckpt[sg].pos);
      const uint64_t x = ((uint64_t)x_hi << 32) | x_lo;
      if (x != cx || (uint64_t)(wbase + wp) != cp) err = kSegMismatch;
    }
    if (lane == 0) stg<uint32_t>(d.status + sg, err);
  }
#ifdef FGMM_SEG_PROF
  if (lane == 0) {
    for (int q = 0; q < 6; ++q) atomicAdd(&g_segprof[q], seg_acc[q]);
    if (!producer) {
      atomicAdd(&g_segprof[6], (unsigned long long)(clock64() - t_begin));
      atomicAdd(&g_segprof[7], 1ull);
      if (wid < 16384) g_segtimes[2 * wid] = w_begin, g_segtimes[2 * wid + 1] = wall_clock64();
    }
  }
#endif
}
#ifdef FGMM_SEG_PROF
extern "C" int fgmm_debug_segprof(unsigned long long *out, int reset) {
  int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_segprof), sizeof(g_segprof));
  if (reset) {
    unsigned long long z[8] = {};
    rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_segprof), z, sizeof(z));
  }
  return rc;
}
extern "C" int fgmm_debug_segtimes(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_segtimes), sizeof(g_segtimes)); }
#endif

// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------
static inline int launch_err() { return (int)hipGetLastError(); }

// Two or three waves per segment: a second producer shortens a segment by a fifth (its widest rows by more) and costs the chip
// a tenth of its throughput (the producers' handshake, fewer segments per CU).  A launch that fills the chip less than twice over
// ends when its slowest segment does - three waves; a larger one is bound by the work - two.
template <bool CLAMPED, typename PT>
static int launch_segdec_c(const SegDesc *d, const SegRef *segs, int64_t n_segs, int mode, hipStream_t s) {
  const dim3 grid((unsigned)n_segs), block(n_segs <= 4096 ? 192 : 128); // one segment per workgroup
  switch (mode) {
  case MODE_AS: hipLaunchKernelGGL((segdec_kernel<MODE_AS, CLAMPED, PT>), grid, block, 0, s, d, segs, n_segs); break;
  case MODE_LOGISTIC: hipLaunchKernelGGL((segdec_kernel<MODE_LOGISTIC, CLAMPED, PT>), grid, block, 0, s, d, segs, n_segs); break;
  default: hipLaunchKernelGGL((segdec_kernel<MODE_POLYA, CLAMPED, PT>), grid, block, 0, s, d, segs, n_segs); break;
  }
  return launch_err();
}
// channels without a coded symbol are zero in y_hat (entropy_models.py:903-908): one launch for all items of a call
__global__ __launch_bounds__(kBlock) void segzero_kernel(const SegDesc *__restrict__ descs) {
  const SegDesc &d = descs[blockIdx.y];
  if ((int64_t)blockIdx.x >= d.n_dead) return;
  float *out = d.y_hat + (int64_t)ldg<int32_t>(d.dead_list + blockIdx.x) * d.hw;
  for (int64_t i = threadIdx.x; i < d.hw; i += kBlock) stg<float>(out + i, 0.0f);
}
int launch_segzero(const SegDesc *d_descs, int count, int64_t max_dead, void *stream) {
  if (count <= 0 || max_dead <= 0) return 0;
  if (max_dead > 0x7FFFFFFFll) return (int)hipErrorInvalidValue;
  for (int c0 = 0; c0 < count; c0 += 65535) { // grid.y holds 65535 items: a larger call is zeroed in several launches
    hipLaunchKernelGGL(segzero_kernel, dim3((unsigned)max_dead, (unsigned)std::min(count - c0, 65535)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), d_descs + c0);
    if (const int e = launch_err()) return e;
  }
  return 0;
}
int launch_segdec(const SegDesc *d_descs, const SegRef *d_segs, int64_t n_segs, int mode, bool clamped, bool f16, void *stream) {
  if (n_segs <= 0) return 0;
  if (n_segs > 0x7FFFFFFFll) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  if (f16) return clamped ? launch_segdec_c<true, _Float16>(d_descs, d_segs, n_segs, mode, s) : launch_segdec_c<false, _Float16>(d_descs, d_segs, n_segs, mode, s);
  return clamped ? launch_segdec_c<true, float>(d_descs, d_segs, n_segs, mode, s) : launch_segdec_c<false, float>(d_descs, d_segs, n_segs, mode, s);
}

template <bool CLAMPED, typename PT>
static int launch_cdftab_c(const DecDesc *d_descs, int count, int n_ch_max, int64_t hw_max, int mode, int pass, hipStream_t s) {
  dim3 grid((unsigned)((hw_max + kBlock - 1) / kBlock), (unsigned)n_ch_max, (unsigned)count);
#define FGMM_TAB_LAUNCH(M)                                                                                          \
  if (pass & 1) {                                                                                                   \
    hipLaunchKernelGGL((cdftab_count_kernel<M, CLAMPED, PT>), grid, dim3(kBlock), 0, s, d_descs);                   \
    hipLaunchKernelGGL(cdftab_scan_kernel, dim3((unsigned)count), dim3(kBlock), 0, s, d_descs);                     \
  }                                                                                                                 \
  if (pass & 2) hipLaunchKernelGGL((cdftab_fill_kernel<M, CLAMPED, PT>), grid, dim3(kBlock), 0, s, d_descs);
  switch (mode) {
  case MODE_AS: FGMM_TAB_LAUNCH(MODE_AS) break;
  case MODE_LOGISTIC: FGMM_TAB_LAUNCH(MODE_LOGISTIC) break;
  default: FGMM_TAB_LAUNCH(MODE_POLYA) break;
  }
#undef FGMM_TAB_LAUNCH
  return launch_err();
}

static int launch_cdftab_pass(const DecDesc *d_descs, int count, int n_ch_max, int64_t hw_max, int mode, bool clamped, bool f16,
                              int pass, void *stream) {
  if (count <= 0 || n_ch_max <= 0 || hw_max <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (f16) return clamped ? launch_cdftab_c<true, _Float16>(d_descs, count, n_ch_max, hw_max, mode, pass, s)
                          : launch_cdftab_c<false, _Float16>(d_descs, count, n_ch_max, hw_max, mode, pass, s);
  return clamped ? launch_cdftab_c<true, float>(d_descs, count, n_ch_max, hw_max, mode, pass, s)
                 : launch_cdftab_c<false, float>(d_descs, count, n_ch_max, hw_max, mode, pass, s);
}
int launch_cdftab_count(const DecDesc *d_descs, int count, int n_ch_max, int64_t hw_max, int mode, bool clamped, bool f16,
                        void *stream) {
  return launch_cdftab_pass(d_descs, count, n_ch_max, hw_max, mode, clamped, f16, 1, stream);
}
int launch_cdftab_fill(const DecDesc *d_descs, int count, int n_ch_max, int64_t hw_max, int mode, bool clamped, bool f16,
                       void *stream) {
  return launch_cdftab_pass(d_descs, count, n_ch_max, hw_max, mode, clamped, f16, 2, stream);
}
int launch_cdftab(const DecDesc *d_descs, int count, int n_ch_max, int64_t hw_max, int mode, bool clamped, bool f16,
                  void *stream) {
  return launch_cdftab_pass(d_descs, count, n_ch_max, hw_max, mode, clamped, f16, 3, stream);
}

template <bool CLAMPED, typename PT>
static int launch_tab_c(const DecDesc *d, int count, int blocks_max, int tl_max, int cap_e, int mode, hipStream_t s) {
  const dim3 grid((unsigned)blocks_max, (unsigned)count);
  const size_t lds = tab_smem_bytes(tl_max, cap_e);
  switch (mode) {
  case MODE_AS: hipLaunchKernelGGL((tab_kernel<MODE_AS, CLAMPED, PT>), grid, dim3(kBlock), lds, s, d, tl_max, cap_e); break;
  case MODE_LOGISTIC: hipLaunchKernelGGL((tab_kernel<MODE_LOGISTIC, CLAMPED, PT>), grid, dim3(kBlock), lds, s, d, tl_max, cap_e); break;
  default: hipLaunchKernelGGL((tab_kernel<MODE_POLYA, CLAMPED, PT>), grid, dim3(kBlock), lds, s, d, tl_max, cap_e); break;
  }
  return launch_err();
}
int launch_tab(const DecDesc *d_descs, int count, int blocks_max, int tl_max, int cap_e, int mode, bool clamped, bool f16,
               void *stream) {
  if (count <= 0 || blocks_max <= 0) return 0;
  if (tl_max < 1 || tl_max > kTabMaxTl || cap_e < 32 || cap_e > 32768 || (cap_e & 31) || tab_smem_bytes(tl_max, cap_e) > 160 * 1024) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  if (f16) return clamped ? launch_tab_c<true, _Float16>(d_descs, count, blocks_max, tl_max, cap_e, mode, s)
                          : launch_tab_c<false, _Float16>(d_descs, count, blocks_max, tl_max, cap_e, mode, s);
  return clamped ? launch_tab_c<true, float>(d_descs, count, blocks_max, tl_max, cap_e, mode, s)
                 : launch_tab_c<false, float>(d_descs, count, blocks_max, tl_max, cap_e, mode, s);
}

} // namespace fgmm

// fgmm_kernels.hip — hand-written HIP kernels of the GMM entropy-coding path for CDNA4 / gfx950 (MI355X).
//
//   quant_stats_kernel   y -> round(y), per-channel {min, max, any-nonzero}            (entropy_models.py:834-842)
//   symtab_kernel        (y | symbols, sigma, mu, pi) -> packed start|range<<16         (rans_interface.cpp:487-517)
//   cdf_pair_kernel      float CDF pair probe                                           (rans_interface.cpp:250-292)
//   cdftab_{count,scan,fill}  (sigma, mu, pi, max_bs) -> trimmed per-latent edge tables (rans_interface.cpp:826-862)
//
// All kernels are batched over `count` independent bitstreams (blockIdx.z = item) through a device array of
// descriptors, because one Kodak-sized half (<= 147 456 latents) is far too small to fill 256 CUs on its own.
// They are HBM-streaming / transcendental-VALU kernels: no MFMA, no data reuse, so no LDS tiling — mixture
// parameters are read exactly once, straight to VGPRs, as 16-byte-per-lane coalesced loads of the planar
// (k, c, p) layout.  Wave = 64 lanes throughout.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "fgmm_dev.h"

namespace fgmm {

// ---------------------------------------------------------------------------------------------------------
// quant_stats_kernel: one block per (channel, item).  8 B/latent of traffic (4 in, 4 out).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void quant_stats_kernel(const EncDesc *__restrict__ descs) {
  const EncDesc &d = descs[blockIdx.z];
  const int c = blockIdx.x;
  if (c >= d.M) return;
  const float *__restrict__ y = d.y + (int64_t)c * d.hw;
  float *__restrict__ yq = d.yq ? d.yq + (int64_t)c * d.hw : nullptr;
  float mn = INFINITY, mx = -INFINITY;
  int nz = 0;
  const int64_t hw = d.hw;
  const bool vec = ((hw & 3) == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0) &&
                   (!yq || (reinterpret_cast<uintptr_t>(yq) & 15) == 0);
  if (vec) {
    for (int64_t p = (int64_t)threadIdx.x * 4; p < hw; p += (int64_t)kBlock * 4) {
      const float4 v = *reinterpret_cast<const float4 *>(y + p);
      float4 q;
      q.x = __builtin_rintf(v.x); q.y = __builtin_rintf(v.y); q.z = __builtin_rintf(v.z); q.w = __builtin_rintf(v.w);
      mn = fminf(fminf(mn, v.x), fminf(fminf(v.y, v.z), v.w));
      mx = fmaxf(fmaxf(mx, v.x), fmaxf(fmaxf(v.y, v.z), v.w));
      nz |= (q.x != 0.0f) | (q.y != 0.0f) | (q.z != 0.0f) | (q.w != 0.0f);
      if (yq) *reinterpret_cast<float4 *>(yq + p) = q;
    }
  } else {
    for (int64_t p = threadIdx.x; p < hw; p += kBlock) {
      const float v = y[p];
      const float q = __builtin_rintf(v); // round-half-even == torch.round
      mn = fminf(mn, v);
      mx = fmaxf(mx, v);
      nz |= (q != 0.0f);
      if (yq) yq[p] = q;
    }
  }
  mn = wave_min(mn);
  mx = wave_max(mx);
  nz = __any(nz) ? 1 : 0;
  __shared__ float s_mn[kBlock / 64], s_mx[kBlock / 64];
  __shared__ int s_nz[kBlock / 64];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_mn[w] = mn; s_mx[w] = mx; s_nz[w] = nz;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 1; i < kBlock / 64; ++i) {
      mn = fminf(mn, s_mn[i]); mx = fmaxf(mx, s_mx[i]); nz |= s_nz[i];
    }
    d.chan_min[c] = mn;
    d.chan_max[c] = mx;
    d.chan_nz[c] = nz;
  }
}

// ---------------------------------------------------------------------------------------------------------
// chan_compact_kernel: one block per item; chan_nz[M] -> chan_list[j] = j-th non-zero channel, chan_list[M] = count.
// (entropy_models.py:844 `nonzero`.)  Lets the CDF kernel find its channel with ONE wave-uniform scalar load
// instead of a dependent global-load + block reduction in front of its parameter loads.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void chan_compact_kernel(const EncDesc *__restrict__ descs) {
  const EncDesc &d = descs[blockIdx.x];
  if (!d.chan_nz) return;
  __shared__ int s_base;
  __shared__ int s_wave[kBlock / 64];
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int c0 = 0; c0 < d.M; c0 += kBlock) {
    const int c = c0 + threadIdx.x;
    const bool nz = c < d.M && d.chan_nz[c] != 0;
    const unsigned long long m = __ballot(nz);
    if (lane == 0) s_wave[w] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int i = 0; i < w; ++i) off += s_wave[i];
    if (nz) d.chan_list[off + __popcll(m & ((1ull << lane) - 1ull))] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int i = 0; i < kBlock / 64; ++i) t += s_wave[i];
      s_base += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) d.chan_list[d.M] = s_base;
}

// ---------------------------------------------------------------------------------------------------------
// symtab_kernel — THE encode-side CDF kernel.  Algorithmic traffic 56 B/latent at K = 4:
//   4 (y or symbol) + 3*4*4 (sigma, mu, pi planes) in, 4 out (start | range << 16).
// grid = (tiles over hw, compact channel j, item): block j codes the j-th NON-ZERO channel (chan_list, built on the
// device by chan_compact_kernel — entropy_models.py:844-845 channel compaction with no host round trip) and writes
// row j of the table; blocks with j >= count leave after one scalar load.
// VEC = 4: each lane owns 4 consecutive positions, every plane read is one 16-B load (1 KiB per wave-instr).
// ---------------------------------------------------------------------------------------------------------
#ifndef FGMM_SYMTAB_WAVES
#define FGMM_SYMTAB_WAVES 5 // min waves per SIMD the register allocator must leave room for (<= 96 VGPRs)
#endif
template <int MODE, int VEC, bool CLAMPED, typename PT, bool LINEAR>
__global__ __launch_bounds__(kBlock, VEC == 8 ? 4 : FGMM_SYMTAB_WAVES) void symtab_kernel(const EncDesc *__restrict__ descs) { // (VEC = 8: 128 VGPRs, else it spills)
  const EncDesc &d = descs[blockIdx.z];
  const int64_t hw = d.hw;
  const int n_nz = d.chan_list ? d.chan_list[d.M] : d.M; // wave-uniform scalar load
  int rank;     // compact (coded) channel of this wave: wave-uniform in both forms, so all addressing stays scalar
  int64_t p0;   // position of the lane's first symbol within the channel
  int64_t slot; // where this wave leaves its bypass count
  bool active;  // lanes past the end stay for the wave reduction below
  if constexpr (LINEAR) {
    // Every hw of the batch is a multiple of 64 * VEC (checked by the host): the coded symbols of an item are one
    // linear range [0, n_nz * hw) and each WAVE takes 64 * VEC consecutive ones, never straddling a channel.  All
    // waves are full whatever hw is (a 768-symbol Kodak plane fills only 3 of the 4 waves of a per-channel block).
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t w0 = ((int64_t)blockIdx.x * kBlock + wave * 64) * VEC;
    if (w0 >= (int64_t)n_nz * hw) return;
    rank = __builtin_amdgcn_readfirstlane((int)(w0 / hw));
    p0 = (w0 - (int64_t)rank * hw) + (int64_t)(threadIdx.x & 63) * VEC;
    slot = (int64_t)blockIdx.x * (kBlock / 64) + wave;
    active = true;
  } else {
    // one block per (tile of kBlock * VEC positions, compact channel)
    rank = blockIdx.y;
    if (rank >= n_nz) return;
    if ((int64_t)blockIdx.x * kBlock * VEC >= hw) return;
    p0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * VEC;
    const int64_t tiles = (hw + (int64_t)kBlock * VEC - 1) / ((int64_t)kBlock * VEC);
    slot = ((int64_t)rank * tiles + blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6);
    active = p0 < hw;
  }
  const int c = d.chan_list ? d.chan_list[rank] : rank;
  // where the compact channel's entries go: the table is one piece, or up to four segments of compact channels (wave-uniform)
  const int seg = (rank >= d.seg_b[0]) + (rank >= d.seg_b[1]) + (rank >= d.seg_b[2]);
  uint32_t *const row_out = (seg ? d.packed_seg[seg] : d.packed) + (int64_t)(rank - seg * d.cps) * hw;
  int nbypass = 0;
  if (!active) {
  } else if constexpr (VEC == 8) {
    // fp16 planes, 8 positions per lane: every plane read is ONE 16-byte load (the VEC = 4 form reads 8 bytes per lane and plane);
    // the halves stay packed in registers (48 VGPRs for the twelve planes) and are widened as each position is evaluated.
    // Measured against VEC = 4 on ELIC-4K batches (profiles/r05_symtab_fp16_vec8_ab.txt): 127.6 / 133.1 against 123.0 / 126.8 G
    // symbols per second (2 / 4 images) - the kernel is bound by VALU issue (0.86 of the issue roof, bench.py's valu_frac) and
    // this form issues fewer load and address instructions.  The default for aligned fp16 planes; option "enc_vec" = 4 is the A/B.
    typedef PT pvec_t __attribute__((ext_vector_type(8)));
    typedef float f4_t __attribute__((ext_vector_type(4)));
    typedef int i4_t __attribute__((ext_vector_type(4)));
    typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
    const int64_t base = (int64_t)c * d.stride_c + p0;
    float vq[8];
    int vi[8];
    if (d.sym) {
      const i4_t t0 = ldg<i4_t>(d.sym + (int64_t)c * hw + p0), t1 = ldg<i4_t>(d.sym + (int64_t)c * hw + p0 + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        vi[e] = e < 4 ? t0[e & 3] : t1[e & 3];
        vq[e] = (float)vi[e];
      }
    } else {
      const f4_t t0 = ldg<f4_t>(d.y + (int64_t)c * hw + p0), t1 = ldg<f4_t>(d.y + (int64_t)c * hw + p0 + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        vq[e] = __builtin_rintf(e < 4 ? t0[e & 3] : t1[e & 3]);
        vi[e] = (int)vq[e];
      }
    }
    pvec_t rS[4], rM[4], rP[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      rS[k] = ldg<pvec_t>(static_cast<const PT *>(d.scales) + base + k * d.stride_k);
      rM[k] = ldg<pvec_t>(static_cast<const PT *>(d.means) + base + k * d.stride_k);
      rP[k] = ldg<pvec_t>(static_cast<const PT *>(d.weights) + base + k * d.stride_k);
    }
    u4_t out[2];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float mu[4], sg[4], pi[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sg[k] = (float)rS[k][e];
        mu[k] = (float)rM[k][e];
        pi[k] = (float)rP[k][e];
      }
      if (d.logits) softmax4(pi);
      int bp;
      out[e >> 2][e & 3] = sym_entry<MODE, CLAMPED>(vq[e], vi[e], mu, sg, pi, bp);
      nbypass += __popcll(__ballot(bp));
    }
    stg<u4_t>(row_out + p0, out[0]);
    stg<u4_t>(row_out + p0 + 4, out[1]);
  } else if constexpr (VEC > 1) {
    // planar, aligned (checked by the host): one VEC-wide load per plane per lane (16 B fp32 / 8 B fp16 at VEC = 4)
    typedef float fvec_t __attribute__((ext_vector_type(VEC)));
    typedef int ivec_t __attribute__((ext_vector_type(VEC)));
    typedef uint32_t uvec_t __attribute__((ext_vector_type(VEC)));
    const int64_t base = (int64_t)c * d.stride_c + p0;
    float vq[VEC];
    int vi[VEC];
    if (d.sym) {
      const ivec_t t = ldg<ivec_t>(d.sym + (int64_t)c * hw + p0);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        vi[e] = t[e];
        vq[e] = (float)vi[e];
      }
    } else {
      const fvec_t t = ldg<fvec_t>(d.y + (int64_t)c * hw + p0);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        vq[e] = __builtin_rintf(t[e]);
        vi[e] = (int)vq[e];
      }
    }
    float S[4][VEC], Mu[4][VEC], Pi[4][VEC];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ldv<PT, VEC>(d.scales, base + k * d.stride_k, S[k]);
      ldv<PT, VEC>(d.means, base + k * d.stride_k, Mu[k]);
      ldv<PT, VEC>(d.weights, base + k * d.stride_k, Pi[k]);
    }
    uvec_t out;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float mu[4], sg[4], pi[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sg[k] = S[k][e];
        mu[k] = Mu[k][e];
        pi[k] = Pi[k][e];
      }
      if (d.logits) softmax4(pi);
      int bp;
      out[e] = sym_entry<MODE, CLAMPED>(vq[e], vi[e], mu, sg, pi, bp);
      nbypass += __popcll(__ballot(bp)); // the wave's count on the scalar unit: no lane-wise sum, no cross-lane reduction
    }
    stg<uvec_t>(row_out + p0, out);
  } else {
    const int64_t base = (int64_t)c * d.stride_c + p0 * d.stride_p;
    float vq;
    int vi;
    if (d.sym) {
      vi = ldg<int32_t>(d.sym + (int64_t)c * hw + p0);
      vq = (float)vi;
    } else {
      vq = __builtin_rintf(ldg<float>(d.y + (int64_t)c * hw + p0));
      vi = (int)vq;
    }
    float mu[4], sg[4], pi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sg[k] = ld1<PT>(d.scales, base + k * d.stride_k);
      mu[k] = ld1<PT>(d.means, base + k * d.stride_k);
      pi[k] = ld1<PT>(d.weights, base + k * d.stride_k);
    }
    if (d.logits) softmax4(pi);
    int bp;
    stg<uint32_t>(row_out + p0, sym_entry<MODE, CLAMPED>(vq, vi, mu, sg, pi, bp));
    nbypass = __popcll(__ballot(bp));
  }
  // bypass census (the host sizes its output buffer from it): one plain store per wave that saw any, no atomics.  nbypass is
  // the same on every active lane (ballots); lanes past the end of a channel are the wave's last ones, so lane 0 is active
  // whenever any lane is
  if ((threadIdx.x & 63) == 0 && nbypass)
    d.meta[slot] = (uint32_t)nbypass;
}

// ---------------------------------------------------------------------------------------------------------
// cdf_pair_kernel: float probe of the mixture CDF at both edges of v (parity tests: 1e-5 bar, in fact bit-exact)
// ---------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void cdf_pair_kernel(const int32_t *__restrict__ v, const float *__restrict__ scales,
                                                          const float *__restrict__ means, const float *__restrict__ weights,
                                                          int64_t n, int64_t sn, int64_t sk, float *__restrict__ c1,
                                                          float *__restrict__ c2) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  float mu[4], sg[4], pi[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    sg[k] = scales[i * sn + k * sk];
    mu[k] = means[i * sn + k * sk];
    pi[k] = weights[i * sn + k * sk];
  }
  const float vq = (float)v[i];
  c1[i] = mix4<MODE>(vq - 0.5f, mu, sg, pi);
  c2[i] = mix4<MODE>(vq - 0.5f + 1.0f, mu, sg, pi);
}

// ---------------------------------------------------------------------------------------------------------
// yhat_scatter_kernel: the decoded symbols of the coded channels back into the full [M, hw] latent, as floats, zero
// channels restored (entropy_models.py:903-908).  `sym` is read where the host decoder left it (pinned host memory).
// ---------------------------------------------------------------------------------------------------------
template <typename ST>
__global__ __launch_bounds__(kBlock) void yhat_scatter_kernel(const ST *__restrict__ sym, const int32_t *__restrict__ rank,
                                                             float *__restrict__ y_hat, int64_t hw) {
  const int c = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= hw) return;
  const int r = rank[c];
  y_hat[(int64_t)c * hw + p] = r < 0 ? 0.0f : (float)sym[(int64_t)r * hw + p];
}

// The same by ROUNDS over a batch of items (ScatDesc): round r = the compact symbols [bound[r], bound[r + 1]) of every item; the
// channels without a coded symbol are zeroed once, by yhat_zero_dead_kernel.
__global__ __launch_bounds__(kBlock) void yhat_scatter_round_kernel(const ScatDesc *__restrict__ descs, int round) {
  const ScatDesc &d = descs[blockIdx.y];
  if (!d.y_hat) return;
  const int64_t lo = d.bound[round], hi = d.bound[round + 1], hw = d.hw;
  const bool small = hi <= 0x7FFFFFFFll;
  for (int64_t i = lo + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < hi; i += (int64_t)gridDim.x * kBlock) {
    const int64_t r = small ? (int64_t)((uint32_t)i / (uint32_t)hw) : i / hw;
    d.y_hat[(int64_t)d.chan_list[r] * hw + (i - r * hw)] = (float)d.sym[i];
  }
}
__global__ __launch_bounds__(kBlock) void yhat_zero_dead_kernel(const ScatDesc *__restrict__ descs) {
  const ScatDesc &d = descs[blockIdx.z];
  const int c = blockIdx.y;
  if (!d.y_hat || c >= d.M || d.rank[c] >= 0) return;
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p < d.hw) d.y_hat[(int64_t)c * d.hw + p] = 0.0f;
}

// softmax_probe_kernel: the kernels' mixture-weight sequence on (n, 4) rows of logits (tests: against torch.softmax)
__global__ __launch_bounds__(kBlock) void softmax_probe_kernel(const float *__restrict__ logits, float *__restrict__ pi, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  float w[4] = {logits[4 * i], logits[4 * i + 1], logits[4 * i + 2], logits[4 * i + 3]};
  softmax4(w);
#pragma unroll
  for (int k = 0; k < 4; ++k) pi[4 * i + k] = w[k];
}

// ---------------------------------------------------------------------------------------------------------
// Checkerboard split / merge (latent_codecs/checkerboard.py:333-377), pure data movement, HBM bound:
//   unembed: full [planes, h, w] -> halves [2, planes, h, w/2]   (half 0 = anchors, half 1 = non-anchors)
//   embed  : the inverse
// Lane = one horizontal pair (columns 2j, 2j+1): the pair is one 2*sizeof(T) access of the full tensor, its two
// elements one coalesced access each of the two halves.  In row i the anchor is column 2j + ((i & 1) ^ anchor_odd).
// ---------------------------------------------------------------------------------------------------------
template <typename T, bool EMBED, int V>
__global__ __launch_bounds__(kBlock) void ckbd_kernel(const T *__restrict__ src, T *__restrict__ dst, int64_t rows, int64_t h,
                                                     int64_t w2, int anchor_odd) {
  // V consecutive pairs per lane (w2 % V == 0, checked by the launcher: they share a row): 16-byte accesses of the
  // full tensor for V = 2 (4-byte elements) / V = 4 (2-byte elements)
  struct alignas(2 * sizeof(T) * V) Pairs { T v[2 * V]; };
  struct alignas(sizeof(T) * V) Halves { T v[V]; };
  const int64_t half = rows * w2; // pairs in all = elements of one half
  const int64_t groups = half / V;
  const bool h_even = (h & 1) == 0, small = half <= 0xFFFFFFFFll;
  // pairs are taken in linear order, whatever the row length (a Kodak half-row is 24 pairs: one row per block would
  // leave 9 lanes in 10 idle)
  for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < groups; g += (int64_t)gridDim.x * kBlock) {
    const int64_t e = g * V;
    const int64_t row = small ? (int64_t)((uint32_t)e / (uint32_t)w2) : e / w2;
    const int i_par = h_even ? (int)(row & 1) : (int)((small ? (int64_t)((uint32_t)row % (uint32_t)h) : row % h) & 1);
    const int b = i_par ^ anchor_odd;
    if constexpr (EMBED) {
      const Halves anchor = reinterpret_cast<const Halves *>(src)[g], other = reinterpret_cast<const Halves *>(src + half)[g];
      Pairs p;
#pragma unroll
      for (int t = 0; t < V; ++t) {
        p.v[2 * t] = b ? other.v[t] : anchor.v[t];
        p.v[2 * t + 1] = b ? anchor.v[t] : other.v[t];
      }
      reinterpret_cast<Pairs *>(dst)[g] = p;
    } else {
      const Pairs p = reinterpret_cast<const Pairs *>(src)[g];
      Halves anchor, other;
#pragma unroll
      for (int t = 0; t < V; ++t) {
        anchor.v[t] = b ? p.v[2 * t + 1] : p.v[2 * t];
        other.v[t] = b ? p.v[2 * t] : p.v[2 * t + 1];
      }
      reinterpret_cast<Halves *>(dst)[g] = anchor;
      reinterpret_cast<Halves *>(dst + half)[g] = other;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// saturation_selftest_kernel: exhaustive proof-by-enumeration of the lemmas in fgmm_math.h (Sat<MODE>):
// every binary32 z >= ZR (up to and including +inf) must give phi(z) == 1, every z <= -ZL must give
// 0 <= phi(z) <= LEFT_MAX.  ~2.1e9 evaluations per mode; a few milliseconds.
// ---------------------------------------------------------------------------------------------------------
template <int MODE> __global__ __launch_bounds__(kBlock) void saturation_selftest_kernel(unsigned long long *n_bad) {
  const uint32_t inf_bits = 0x7F800000u;
  const uint32_t r0 = f2bits(Sat<MODE>::ZR), l0 = f2bits(Sat<MODE>::ZL);
  const uint64_t nr = (uint64_t)inf_bits - r0 + 1, nl = (uint64_t)inf_bits - l0 + 1;
  unsigned long long bad = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < nr + nl; i += (uint64_t)gridDim.x * kBlock) {
    if (i < nr) {
      const float z = bits2f(r0 + (uint32_t)i);
      bad += !(phi<MODE>(z) == 1.0f);
    } else {
      const float z = -bits2f(l0 + (uint32_t)(i - nr));
      const float v = phi<MODE>(z);
      bad += !(v >= 0.0f && v <= Sat<MODE>::LEFT_MAX);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(n_bad, bad);
}

// ---------------------------------------------------------------------------------------------------------
// fastmath_selftest_kernel: the hand-expanded cores of fgmm_math.h against the compiler's IEEE '/' and sqrtf.
//   which = 0: sqrt_core(x) == sqrtf(x) for EVERY binary32 x in {0} U [2^-96, 2] (plus -1, NaN)  (exhaustive;
//              the kernels only ever take sqrt(1 - e) with e in [0,1]: 0 or >= 2^-24)
//   which = 1: div_clamped(a, s, rcp_refined(s)) == a / s   for n hashed pairs; a = 0, or any magnitude >= 2^-60
//              incl. inf and NaN (a = x - mu with |x| >= 0.5 is +0 or >= 2^-26 by Sterbenz, never -0: the core
//              would return +0 where IEEE returns -0), s over [0.11, 256]
//              with the end points, powers of two and all-ones mantissas over-represented
//   which = 2: rcp_ge1(d) == 1 / d for EVERY binary32 d in [1, +inf] and NaN                   (exhaustive)
//   which = 3, 4, 5: Phi<MODE, true>(z) == Phi<MODE, false>(z) (MODE = which - 3) for EVERY binary32 z with
//              |z| < 2^48, both signs, zeros and denormals included — the domain the kernels' guard admits (exhaustive)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hash32(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (uint32_t)x;
}
__device__ __forceinline__ bool same_f32(float a, float b) { return f2bits(a) == f2bits(b) || (a != a && b != b); }
template <int MODE> __device__ __forceinline__ unsigned long long phi_fast_vs_plain(uint64_t t0, uint64_t stride) {
  const uint32_t top = f2bits(0x1p48f), top15 = f2bits(0x1p15f); // magnitudes [0, 2^48)
  unsigned long long bad = 0;
  for (uint64_t i = t0; i < 2ull * top; i += stride) {
    const float z = bits2f(i < top ? (uint32_t)i : (0x80000000u | (uint32_t)(i - top)));
    const float want = Phi<MODE, false>::eval(z);
    bad += !same_f32(Phi<MODE, true>::eval(z), want);
    // the packed form (domain |z| < 2^15), z in either half beside an unrelated value in the other (hashed)
    if (__builtin_fabsf(z) < 0x1p15f) {
      const float other = bits2f((hash32(i) % top15) | (hash32(i + 1) & 0x80000000u));
      const f2 a = Phi2<MODE>::eval((f2){z, other}), b = Phi2<MODE>::eval((f2){other, z});
      bad += !same_f32(a.x, want) + !same_f32(b.y, want);
    }
  }
  return bad;
}


__global__ __launch_bounds__(kBlock) void fastmath_selftest_kernel(int which, unsigned long long n, unsigned long long seed,
                                                                   unsigned long long *n_bad) {
  unsigned long long bad = 0;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  const uint64_t t0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (which == 0) {
    const uint32_t bot = f2bits(0x1p-96f), top = f2bits(2.0f);
    for (uint64_t i = t0; i <= (uint64_t)(top - bot) + 3; i += stride) {
      const uint64_t k = i + bot;
      const float x = k <= top ? bits2f((uint32_t)k) : (k == (uint64_t)top + 1 ? -1.0f : (k == (uint64_t)top + 2 ? 0.0f : bits2f(0x7FC00000u)));
      bad += !same_f32(sqrt_core(x), __builtin_sqrtf(x));
    }
  } else if (which == 6) { // sqrt_unit over {0} U [2^-24, 1]
    const uint32_t bot = f2bits(0x1p-24f), top = f2bits(1.0f);
    for (uint64_t i = t0; i <= (uint64_t)(top - bot) + 1; i += stride) {
      const float x = i <= (uint64_t)(top - bot) ? bits2f(bot + (uint32_t)i) : 0.0f;
      bad += !same_f32(sqrt_unit(x), __builtin_sqrtf(x));
      const f2 sq = sqrt_unit2((f2){x, bits2f(bot + (hash32(i) % (top - bot + 1)))});
      bad += !same_f32(sq.x, __builtin_sqrtf(x));
      const f2 sq2 = sqrt_unit2((f2){0.0f, x});
      bad += !same_f32(sq2.y, __builtin_sqrtf(x)) + !same_f32(sq2.x, 0.0f);
    }
  } else if (which == 3) {
    bad = phi_fast_vs_plain<MODE_POLYA>(t0, stride);
  } else if (which == 4) {
    bad = phi_fast_vs_plain<MODE_AS>(t0, stride);
  } else if (which == 5) {
    bad = phi_fast_vs_plain<MODE_LOGISTIC>(t0, stride);
  } else if (which == 2) {
    const uint32_t lo = f2bits(1.0f), hi = 0x7F800000u;
    for (uint64_t i = t0; i <= (uint64_t)(hi - lo) + 1; i += stride) {
      const float d = i <= (uint64_t)(hi - lo) ? bits2f(lo + (uint32_t)i) : bits2f(0x7FC00000u);
      bad += !same_f32(rcp_ge1(d), 1.0f / d);
    }
  } else {
    for (uint64_t i = t0; i < n; i += stride) {
      const uint32_t h1 = hash32(i * 2 + seed), h2 = hash32(i * 2 + 1 + seed * 0x9E3779B97F4A7C15ULL);
      float a = bits2f(h1); // both signs, NaN/inf included
      if (__builtin_fabsf(a) < 0x1p-60f) a = (h1 & 1u) ? 0.0f : a * 0x1p80f; // domain: +0 or >= 2^-60
      if (a == 0.0f) a = 0.0f; // a = x - mu is never -0 (x != 0; x - x = +0 in round-to-nearest)
      if ((h2 & 7u) == 0) a = (float)(int)(h1 >> 20) * 0.5f - bits2f((h1 & 0x007FFFFFu) | 0x3F000000u); // x - mu like
      float s;
      switch ((h2 >> 3) & 7u) {
      case 0: s = 0.11f; break;
      case 1: s = 256.0f; break;
      case 2: s = bits2f(((h2 >> 8) % 12u + 124u) << 23); break;                 // powers of two 2^-3 .. 2^8
      case 3: s = bits2f((((h2 >> 8) % 12u + 124u) << 23) | 0x007FFFFFu); break; // all-ones mantissas
      default: s = bits2f((((h2 >> 8) % 12u + 123u) << 23) | (hash32(h2 + i) & 0x007FFFFFu)); break;
      }
      s = clamp_scale(s);
      const bool ok = same_f32(div_clamped(a, s, rcp_refined(s)), a / s);
      if (!ok) { n_bad[1] = f2bits(a); n_bad[2] = f2bits(s); } // a witness for the report
      bad += !ok;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(n_bad, bad);
}

// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------
static inline int launch_err() { return (int)hipGetLastError(); }

int launch_quant_stats(const EncDesc *d_descs, int count, int M_max, void *stream) {
  if (count <= 0 || M_max <= 0) return 0;
  dim3 grid((unsigned)M_max, 1, (unsigned)count);
  hipLaunchKernelGGL(quant_stats_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, d_descs);
  int e = launch_err();
  if (e) return e;
  hipLaunchKernelGGL(chan_compact_kernel, dim3((unsigned)count), dim3(kBlock), 0, (hipStream_t)stream, d_descs);
  return launch_err();
}

template <int VEC, bool CLAMPED, typename PT, bool LINEAR>
static int launch_symtab_v(const EncDesc *d, int count, int M_max, int64_t hw_max, int64_t n_max, int mode, hipStream_t s) {
  const int64_t per_block = (int64_t)kBlock * VEC;
  const dim3 grid = LINEAR ? dim3((unsigned)((n_max + per_block - 1) / per_block), 1u, (unsigned)count)
                           : dim3((unsigned)((hw_max + per_block - 1) / per_block), (unsigned)M_max, (unsigned)count);
  switch (mode) {
  case MODE_AS: hipLaunchKernelGGL((symtab_kernel<MODE_AS, VEC, CLAMPED, PT, LINEAR>), grid, dim3(kBlock), 0, s, d); break;
  case MODE_LOGISTIC: hipLaunchKernelGGL((symtab_kernel<MODE_LOGISTIC, VEC, CLAMPED, PT, LINEAR>), grid, dim3(kBlock), 0, s, d); break;
  default: hipLaunchKernelGGL((symtab_kernel<MODE_POLYA, VEC, CLAMPED, PT, LINEAR>), grid, dim3(kBlock), 0, s, d); break;
  }
  return launch_err();
}
template <typename PT, bool LINEAR>
static int launch_symtab_t(const EncDesc *d, int count, int M_max, int64_t hw_max, int64_t n_max, int mode, int vec, bool clamped,
                           hipStream_t s) {
  if constexpr (sizeof(PT) == 2) // (fp16 planes only: 16-byte loads per plane)
    if (vec == 8) return clamped ? launch_symtab_v<8, true, PT, LINEAR>(d, count, M_max, hw_max, n_max, mode, s)
                                 : launch_symtab_v<8, false, PT, LINEAR>(d, count, M_max, hw_max, n_max, mode, s);
  if (vec >= 4) return clamped ? launch_symtab_v<4, true, PT, LINEAR>(d, count, M_max, hw_max, n_max, mode, s)
                               : launch_symtab_v<4, false, PT, LINEAR>(d, count, M_max, hw_max, n_max, mode, s);
  if (vec == 2) return clamped ? launch_symtab_v<2, true, PT, LINEAR>(d, count, M_max, hw_max, n_max, mode, s)
                               : launch_symtab_v<2, false, PT, LINEAR>(d, count, M_max, hw_max, n_max, mode, s);
  return clamped ? launch_symtab_v<1, true, PT, LINEAR>(d, count, M_max, hw_max, n_max, mode, s)
                 : launch_symtab_v<1, false, PT, LINEAR>(d, count, M_max, hw_max, n_max, mode, s);
}

int launch_symtab(const EncDesc *d_descs, int count, int M_max, int64_t hw_max, int64_t n_max, bool linear, int mode, int vec,
                  bool clamped, bool f16, void *stream) {
  if (count <= 0 || M_max <= 0 || hw_max <= 0) return 0;
  if (linear && (n_max + kBlock - 1) / kBlock > 0x7FFFFFFFll) linear = false; // grid.x
  hipStream_t s = (hipStream_t)stream;
  if (linear)
    return f16 ? launch_symtab_t<_Float16, true>(d_descs, count, M_max, hw_max, n_max, mode, vec, clamped, s)
               : launch_symtab_t<float, true>(d_descs, count, M_max, hw_max, n_max, mode, vec, clamped, s);
  return f16 ? launch_symtab_t<_Float16, false>(d_descs, count, M_max, hw_max, n_max, mode, vec, clamped, s)
             : launch_symtab_t<float, false>(d_descs, count, M_max, hw_max, n_max, mode, vec, clamped, s);
}

int launch_cdf_pair(const int32_t *v, const float *scales, const float *means, const float *weights, int64_t n,
                    int64_t stride_n, int64_t stride_k, int mode, float *c1, float *c2, void *stream) {
  if (n <= 0) return 0;
  dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
  hipStream_t s = (hipStream_t)stream;
  switch (mode) {
  case MODE_AS: hipLaunchKernelGGL((cdf_pair_kernel<MODE_AS>), grid, dim3(kBlock), 0, s, v, scales, means, weights, n, stride_n, stride_k, c1, c2); break;
  case MODE_LOGISTIC: hipLaunchKernelGGL((cdf_pair_kernel<MODE_LOGISTIC>), grid, dim3(kBlock), 0, s, v, scales, means, weights, n, stride_n, stride_k, c1, c2); break;
  default: hipLaunchKernelGGL((cdf_pair_kernel<MODE_POLYA>), grid, dim3(kBlock), 0, s, v, scales, means, weights, n, stride_n, stride_k, c1, c2); break;
  }
  return launch_err();
}

int launch_softmax_probe(const float *logits, float *pi, int64_t n, void *stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(softmax_probe_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, logits, pi, n);
  return launch_err();
}

int launch_yhat_scatter(const void *sym, int wide, const int32_t *rank, float *y_hat, int M, int64_t hw, void *stream) {
  if (M <= 0 || hw <= 0) return 0;
  dim3 grid((unsigned)((hw + kBlock - 1) / kBlock), (unsigned)M);
  if (wide) hipLaunchKernelGGL(yhat_scatter_kernel<int32_t>, grid, dim3(kBlock), 0, (hipStream_t)stream, (const int32_t *)sym, rank, y_hat, hw);
  else hipLaunchKernelGGL(yhat_scatter_kernel<int16_t>, grid, dim3(kBlock), 0, (hipStream_t)stream, (const int16_t *)sym, rank, y_hat, hw);
  return launch_err();
}

int launch_yhat_scatter_round(const ScatDesc *d_descs, int count, int round, int64_t max_range, void *stream) {
  if (count <= 0 || max_range <= 0) return 0;
  if (count > 65535 || round < 0 || round >= kMaxPieces) return (int)hipErrorInvalidValue;
  const dim3 grid((unsigned)std::min<int64_t>((max_range + kBlock - 1) / kBlock, 1024), (unsigned)count);
  hipLaunchKernelGGL(yhat_scatter_round_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, d_descs, round);
  return launch_err();
}
int launch_yhat_zero_dead(const ScatDesc *d_descs, int count, int M_max, int64_t hw_max, void *stream) {
  if (count <= 0 || M_max <= 0 || hw_max <= 0) return 0;
  if (count > 65535 || M_max > 65535) return (int)hipErrorInvalidValue;
  const dim3 grid((unsigned)((hw_max + kBlock - 1) / kBlock), (unsigned)M_max, (unsigned)count);
  hipLaunchKernelGGL(yhat_zero_dead_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, d_descs);
  return launch_err();
}

template <typename T, int V>
static void launch_ckbd_t(const void *src, void *dst, int64_t rows, int64_t h, int64_t w2, int anchor_odd, bool embed, hipStream_t s) {
  dim3 grid((unsigned)std::min<int64_t>((rows * w2 / V + kBlock - 1) / kBlock, 1 << 20));
  if (embed) hipLaunchKernelGGL((ckbd_kernel<T, true, V>), grid, dim3(kBlock), 0, s, (const T *)src, (T *)dst, rows, h, w2, anchor_odd);
  else hipLaunchKernelGGL((ckbd_kernel<T, false, V>), grid, dim3(kBlock), 0, s, (const T *)src, (T *)dst, rows, h, w2, anchor_odd);
}
int launch_ckbd(const void *src, void *dst, int64_t planes, int64_t h, int64_t w, int elem_bytes, int anchor_odd, bool embed,
                void *stream) {
  const int64_t rows = planes * h, w2 = w / 2;
  if (rows <= 0 || w2 <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  // wide form: V pairs per lane, when rows hold a multiple of V pairs and both tensors are 16-byte aligned (the second
  // half starts rows * w2 elements into the halves tensor)
  const int V = elem_bytes == 4 ? 2 : 4;
  const bool wide = w2 % V == 0 && (reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) % 16 == 0 &&
                    (rows * w2 * elem_bytes) % 16 == 0;
  if (elem_bytes == 4) {
    if (wide) launch_ckbd_t<uint32_t, 2>(src, dst, rows, h, w2, anchor_odd, embed, s);
    else launch_ckbd_t<uint32_t, 1>(src, dst, rows, h, w2, anchor_odd, embed, s);
  } else {
    if (wide) launch_ckbd_t<uint16_t, 4>(src, dst, rows, h, w2, anchor_odd, embed, s);
    else launch_ckbd_t<uint16_t, 1>(src, dst, rows, h, w2, anchor_odd, embed, s);
  }
  return launch_err();
}

int launch_fastmath_selftest(int which, unsigned long long n, unsigned long long seed, unsigned long long *n_bad, void *stream) {
  hipLaunchKernelGGL(fastmath_selftest_kernel, dim3(256 * 16), dim3(kBlock), 0, (hipStream_t)stream, which, n, seed, n_bad);
  return launch_err();
}

int launch_saturation_selftest(int mode, unsigned long long *n_bad, void *stream) {
  dim3 grid(256 * 16);
  hipStream_t s = (hipStream_t)stream;
  switch (mode) {
  case MODE_AS: hipLaunchKernelGGL((saturation_selftest_kernel<MODE_AS>), grid, dim3(kBlock), 0, s, n_bad); break;
  case MODE_LOGISTIC: hipLaunchKernelGGL((saturation_selftest_kernel<MODE_LOGISTIC>), grid, dim3(kBlock), 0, s, n_bad); break;
  default: hipLaunchKernelGGL((saturation_selftest_kernel<MODE_POLYA>), grid, dim3(kBlock), 0, s, n_bad); break;
  }
  return launch_err();
}

} // namespace fgmm

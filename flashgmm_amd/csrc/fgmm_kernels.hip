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

#include "fgmm_internal.h"
#include <algorithm>
#include "fgmm_math.h"

namespace fgmm {

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------------------------------------
// wave / block helpers (wave64)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// parameter planes are float32, or float16 converted on load (BASELINE configs[4]: "fp16 (mu,sigma,pi) with fp32 CDF
// accumulate"): every value is widened exactly, then the fp32 path runs unchanged
// Descriptor pointers are generic (they come out of a struct in memory): cast to the global address space so that the
// accesses are global_load / global_store (a flat_load also takes a slot of the LDS queue and is waited for out of order).
#define FGMM_GLOBAL __attribute__((address_space(1)))
#ifndef FGMM_NT_LOADS
#define FGMM_NT_LOADS 1 // stream the inputs with the non-temporal hint (measured +3-4 % on the symtab kernel; 0: A/B)
#endif
template <typename T> __device__ __forceinline__ T ldg(const void *p) {
  const FGMM_GLOBAL T *g = (const FGMM_GLOBAL T *)p;
#if FGMM_NT_LOADS
  return __builtin_nontemporal_load(g);
#else
  return *g;
#endif
}
template <typename T> __device__ __forceinline__ void stg(void *p, T v) { *(FGMM_GLOBAL T *)p = v; }
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));
template <typename PT> __device__ __forceinline__ float ld1(const void *base, int64_t idx) {
  return (float)ldg<PT>(static_cast<const PT *>(base) + idx);
}
template <typename PT> __device__ __forceinline__ void ld4(const void *base, int64_t idx, float (&out)[4]);
template <> __device__ __forceinline__ void ld4<float>(const void *base, int64_t idx, float (&out)[4]) {
  const float4_t v = ldg<float4_t>(static_cast<const float *>(base) + idx); // 16 B / lane
  out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
}
template <> __device__ __forceinline__ void ld4<_Float16>(const void *base, int64_t idx, float (&out)[4]) {
  const half4_t v = ldg<half4_t>(static_cast<const _Float16 *>(base) + idx); // 8 B / lane
  out[0] = (float)v[0]; out[1] = (float)v[1]; out[2] = (float)v[2]; out[3] = (float)v[3];
}

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
template <int MODE, bool CLAMPED>
__device__ __forceinline__ uint32_t sym_entry(float vq, int vi, const float (&mu)[4], const float (&sg)[4], const float (&pi)[4],
                                              int &bypass) {
  const float x1 = vq - 0.5f;          // static_cast<float>(value) - offset             (:499)
  const float x2 = vq - 0.5f + 1.0f;   // static_cast<float>(value) - offset + 1.0f
  uint32_t lo, hi;
  if constexpr (CLAMPED) {
    Sigma4 S;
    S.set(sg[0], sg[1], sg[2], sg[3]); // clamp + refined reciprocals, shared by both edges
    bool ok = S.tame;
    const f2 cc = mix4_clamped2<MODE>((f2){x1, x2}, mu, S, pi, ok); // both edges share every parameter: packed fp32
    float c1 = cc.x, c2 = cc.y;
    if (__builtin_expect(!ok, 0)) { // far-off / non-finite mean, NaN sigma: one rare out-of-line IEEE evaluation
      const float s0 = clamp_scale(sg[0]), s1 = clamp_scale(sg[1]), s2 = clamp_scale(sg[2]), s3 = clamp_scale(sg[3]);
      c1 = mix4_slow<MODE>(x1, mu[0], mu[1], mu[2], mu[3], s0, s1, s2, s3, pi[0], pi[1], pi[2], pi[3]);
      c2 = mix4_slow<MODE>(x2, mu[0], mu[1], mu[2], mu[3], s0, s1, s2, s3, pi[0], pi[1], pi[2], pi[3]);
    }
    lo = quant16(c1);
    hi = quant16(c2);
  } else {
    lo = quant16(mix4<MODE>(x1, mu, sg, pi));
    hi = quant16(mix4<MODE>(x2, mu, sg, pi));
  }
  const uint32_t pmf = (hi - lo) & 0xFFFFu; // uint16_t pmf = next - value                (:512)
  bypass = (pmf == 0);
  return pmf ? (lo | (pmf << 16)) : ((uint32_t)vi & 0xFFFFu); // bypass: low 16 bits of the int32 symbol
}

#ifndef FGMM_SYMTAB_WAVES
#define FGMM_SYMTAB_WAVES 5 // min waves per SIMD the register allocator must leave room for (<= 96 VGPRs)
#endif
template <int MODE, int VEC, bool CLAMPED, typename PT, bool LINEAR>
__global__ __launch_bounds__(kBlock, FGMM_SYMTAB_WAVES) void symtab_kernel(const EncDesc *__restrict__ descs) {
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
  int nbypass = 0;
  if (!active) {
  } else if constexpr (VEC == 4) {
    // planar, aligned (checked by the host): one 4-wide load per plane per lane (16 B fp32 / 8 B fp16)
    const int64_t base = (int64_t)c * d.stride_c + p0;
    float vq[4];
    int vi[4];
    if (d.sym) {
      typedef int int4_t __attribute__((ext_vector_type(4)));
      const int4_t t = ldg<int4_t>(d.sym + (int64_t)c * hw + p0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        vi[e] = t[e];
        vq[e] = (float)vi[e];
      }
    } else {
      const float4_t t = ldg<float4_t>(d.y + (int64_t)c * hw + p0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        vq[e] = __builtin_rintf(t[e]);
        vi[e] = (int)vq[e];
      }
    }
    float S[4][4], Mu[4][4], Pi[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ld4<PT>(d.scales, base + k * d.stride_k, S[k]);
      ld4<PT>(d.means, base + k * d.stride_k, Mu[k]);
      ld4<PT>(d.weights, base + k * d.stride_k, Pi[k]);
    }
    typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));
    uint4_t out;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float mu[4], sg[4], pi[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sg[k] = S[k][e];
        mu[k] = Mu[k][e];
        pi[k] = Pi[k][e];
      }
      int bp;
      out[e] = sym_entry<MODE, CLAMPED>(vq[e], vi[e], mu, sg, pi, bp);
      nbypass += bp;
    }
    stg<uint4_t>(d.packed + (int64_t)rank * hw + p0, out);
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
    int bp;
    stg<uint32_t>(d.packed + (int64_t)rank * hw + p0, sym_entry<MODE, CLAMPED>(vq, vi, mu, sg, pi, bp));
    nbypass = bp;
  }
  // bypass census (the host sizes its output buffer from it): one plain store per wave that saw any, no atomics
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nbypass += __shfl_xor(nbypass, o, 64);
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
// Decode-side edge tables (format v3, include/flashgmm_amd.h).  Lane = latent.  Three launches per group of items:
//   cdftab_count_kernel  find, exactly, the window outside which F_i is constant (leading zeros, trailing constant
//                        run, non-monotone flag) -> 4-byte header; row byte length -> per-block sums
//   cdftab_scan_kernel   exclusive scan of the block sums (one block per item) -> block offsets, total bytes
//   cdftab_fill_kernel   re-evaluate only the window and store the row at its offset: rows lie in LATENT ORDER with
//                        no per-row offset (the host walks them sequentially), narrow / non-monotone rows as
//                        uint16 entries, wide monotone rows Elias-Fano coded (low bytes + unary high parts)
// F[v] = quant16(cdf(v - 0.5)) over v in [-max_bs, max_bs+1] is all the reference's bisection can probe
// (rans_interface.cpp:826-862).  Transcendental-VALU bound (about 150 VALU ops per edge), not HBM bound.
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

__device__ __forceinline__ uint32_t block_reduce_add(uint32_t v, uint32_t *s_tmp) { // kBlock threads, result in all
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = v;
  __syncthreads();
  uint32_t t = 0;
#pragma unroll
  for (int i = 0; i < kBlock / 64; ++i) t += s_tmp[i];
  __syncthreads();
  return t;
}
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t *s_tmp) { // exclusive prefix over the block
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_tmp[w] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (int i = 0; i < w; ++i) base += s_tmp[i];
  __syncthreads();
  return base + incl - v;
}

// ---- evaluation window: skip the part of [-max_bs, max_bs+1] where every component is saturated -------------
// Saturation lemmas (fgmm_math.h Sat<MODE>, proved by exhaustive scan: fgmm_selftest_saturation):
//   all z_k <= -ZL  =>  F[v] == 0          all z_k >= +ZR  =>  F[v] == quant16((pi0+pi1)+(pi2+pi3))
// z_k(v) = ((float)v - 0.5f - mu_k) / sg_k is non-decreasing in v for finite mu and 0 < sg < inf (every IEEE
// operation is monotone), so it is enough to VERIFY the condition, with the kernel's own arithmetic, at one v:
// it then holds for every v beyond it.  Any latent whose parameters fall outside the lemmas' domain is
// evaluated over the full range instead.  Indices < j_lo are all zero, indices >= j_hi are all T_sat.
template <int MODE, bool CLAMPED, typename PT>
__device__ __forceinline__ void tab_window(const TabLatent<MODE, CLAMPED, PT> &L, int prune, int max_bs, int W, int &j_lo, int &j_hi,
                                           uint32_t &T_sat) {
  j_lo = 0;
  j_hi = W;
  T_sat = 0;
  if (prune) {
    bool ok = true;
    float tl = INFINITY, tr = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ok = ok && (L.sg[k] > 0.0f) && (L.sg[k] < INFINITY) && (fabsf(L.mu[k]) < INFINITY) && Sat<MODE>::weight_ok(L.pi[k]);
      tl = fminf(tl, __builtin_fmaf(-Sat<MODE>::ZL, L.sg[k], L.mu[k]));
      tr = fmaxf(tr, __builtin_fmaf(Sat<MODE>::ZR, L.sg[k], L.mu[k]));
    }
    if (ok) {
      const float lim = (float)max_bs + 4.0f;
      // left: largest candidate v with v - 0.5 <= tl, minus one for the rounding of tl itself
      int vL = (int)fminf(fmaxf(floorf(tl + 0.5f) - 1.0f, -lim), lim);
      int vR = (int)fminf(fmaxf(ceilf(tr + 0.5f) + 1.0f, -lim), lim);
      bool okL = true, okR = true;
      const float xl = (float)vL - 0.5f, xr = (float)vR - 0.5f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        okL = okL && ((xl - L.mu[k]) / L.sg[k] <= -Sat<MODE>::ZL);
        okR = okR && ((xr - L.mu[k]) / L.sg[k] >= Sat<MODE>::ZR);
      }
      if (okL) j_lo = min(max(vL + max_bs + 1, 0), W); // indices < j_lo are v <= vL: all zero
      if (okR) j_hi = min(max(vR + max_bs, j_lo), W);  // indices >= j_hi are v >= vR: all T_sat
      T_sat = quant16((L.pi[0] + L.pi[1]) + (L.pi[2] + L.pi[3]));
    }
  }
}

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
  tab_window(L, d.prune, max_bs, W, j_lo, j_hi, T_sat);
  // temp buffer of this block: 4 header rows (j_lo, j_hi, T_sat of each lane, one spare), then W rows of edges
  uint16_t *__restrict__ tmp = d.tmp ? d.tmp + ((int64_t)cj * d.tiles + blockIdx.x) * ((int64_t)(W + kTmpHdrRows) * kBlock) + threadIdx.x : nullptr;
  if (tmp) {
    tmp[0] = (uint16_t)j_lo; // W <= 2 * FGMM_MAX_BS + 2 < 65536
    tmp[kBlock] = (uint16_t)j_hi;
    tmp[2 * kBlock] = (uint16_t)T_sat;
    tmp += kTmpHdrRows * kBlock;
  }

  int lead = j_lo - 1, run_start = 0;
  bool allzero = true, nonmono = false;
  uint32_t prev = 0;
  for (int j = j_lo; j < j_hi; ++j) {
    const uint32_t E = L.edge(j);
    if (tmp) tmp[(int64_t)(j - j_lo) * kBlock] = (uint16_t)E; // lanes of a block side by side: coalesced
    if (allzero) {
      if (E == 0) lead = j; else allzero = false;
    }
    if (j == 0 || E != prev) run_start = j;
    nonmono |= (j > 0) && (E < prev);
    prev = E;
  }
  if (j_hi < W) { // the saturated right part, one virtual step: F[j_hi .. W-1] == T_sat
    if (allzero) {
      if (T_sat == 0) lead = W - 1; else allzero = false;
    }
    if (j_hi == 0 || T_sat != prev) run_start = j_hi;
    nonmono |= (j_hi > 0) && (T_sat < prev);
  }
  // the row starts at the first non-zero edge: F[v < a] = 0 is implied by the format, and the host takes "cf below the
  // first entry" as the interval [0, first entry) of the symbol before it (2 bytes less per row than storing that zero)
  int a_idx = lead + 1;
  if (a_idx > run_start) a_idx = run_start;
  const uint32_t cnt = (uint32_t)(run_start - a_idx + 1);

  if (active) d.hdr[(int64_t)cj * hw + p] = tab_hdr_pack(a_idx - max_bs, cnt, nonmono ? 1u : 0u);
  __shared__ uint32_t s_tmp[kBlock / 64];
  const uint32_t total = block_reduce_add(active ? tab_row_bytes(cnt, nonmono ? 1u : 0u) : 0u, s_tmp);
  const int any_nonmono = __syncthreads_or(active && nonmono);
  // bit 31: some row of the block is non-monotone (a block holds < 2^31 bytes of rows: 256 rows of < 64 KiB)
  if (threadIdx.x == 0) d.blk_sums[(int64_t)cj * d.tiles + blockIdx.x] = total | (any_nonmono ? 0x80000000u : 0u);
}

// one block per item: blk_off[b] = sum of blk_sums[0..b), pool_used[0] = total bytes, [1] = overflow flag,
// [2 .. 2+n_piece-1) piece offsets, [2 + kMaxPieces] = some row of the item is non-monotone
__global__ __launch_bounds__(kBlock) void cdftab_scan_kernel(const DecDesc *__restrict__ descs) {
  const DecDesc &d = descs[blockIdx.x];
  const int64_t nb = (int64_t)d.n_ch * d.tiles;
  __shared__ uint32_t s_tmp[kBlock / 64];
  __shared__ unsigned long long s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  uint32_t flagged = 0;
  for (int64_t b0 = 0; b0 < nb; b0 += kBlock) {
    const int64_t b = b0 + threadIdx.x;
    const uint32_t raw = b < nb ? d.blk_sums[b] : 0u;
    flagged |= raw >> 31;
    const uint32_t v = raw & 0x7FFFFFFFu;
    const uint32_t ex = block_scan_excl(v, s_tmp);
    const unsigned long long carry = s_carry;
    if (b < nb) d.blk_off[b] = carry + ex;
    __syncthreads();
    if (threadIdx.x == kBlock - 1) s_carry = carry + ex + v;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    d.pool_used[0] = s_carry;
    d.pool_used[1] = s_carry > d.pool_cap ? 1ull : 0ull;
  }
  const int any_nonmono = __syncthreads_or((int)flagged);
  if (d.n_piece >= 1 && threadIdx.x == 0) d.pool_used[2 + kMaxPieces] = (unsigned long long)(any_nonmono != 0);
  if (d.n_piece > 1 && (int)threadIdx.x < d.n_piece - 1) { // where the pieces of the item begin (blk_off is this block's own)
    const int64_t ch = (int64_t)d.n_ch * ((int)threadIdx.x + 1) / d.n_piece;
    d.pool_used[2 + threadIdx.x] = ch * d.tiles < nb ? d.blk_off[ch * d.tiles] : s_carry;
  }
}

// Headers for the host, next to the rows in the staging range: the 4-byte form, or — for items whose half-width fits
// (2*max_bs+2 <= 254) and that have no non-monotone row — 2 bytes: (a + max_bs) | cnt << 8.
__global__ __launch_bounds__(kBlock) void hdr_pack_kernel(const DecDesc *__restrict__ descs) {
  const DecDesc &d = descs[blockIdx.y];
  const int64_t n = (int64_t)d.n_ch * d.hw;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const uint32_t h = d.hdr[i];
    if (d.hdr_compact) reinterpret_cast<uint16_t *>(d.hdr_out)[i] = (uint16_t)(((uint32_t)(tab_hdr_a(h) + d.max_bs) & 0xFFu) | (tab_hdr_cnt(h) << 8));
    else reinterpret_cast<uint32_t *>(d.hdr_out)[i] = h;
  }
}

template <int MODE, bool CLAMPED, typename PT>
__global__ __launch_bounds__(kBlock) void cdftab_fill_kernel(const DecDesc *__restrict__ descs) {
  const DecDesc &d = descs[blockIdx.z];
  const int cj = (int)blockIdx.y + d.ch_begin;
  if (cj >= d.ch_end) return;
  const int64_t hw = d.hw;
  if ((int64_t)blockIdx.x * kBlock >= hw) return;
  if (d.pool_used[1]) return; // pool too small: the host sees the flag
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool active = p < hw;
  const int c = d.chan_list ? d.chan_list[cj] : cj;
  TabLatent<MODE, CLAMPED, PT> L;
  if (!d.tmp) L.load(d, c, active ? p : 0); // with a temp buffer the fill pass never looks at the parameters

  const uint32_t h = active ? d.hdr[(int64_t)cj * hw + p] : 0u;
  const int a_idx = tab_hdr_a(h) + d.max_bs;
  const uint32_t cnt = tab_hdr_cnt(h), nonmono = tab_hdr_nonmono(h);
  const uint32_t bytes = active ? tab_row_bytes(cnt, nonmono) : 0u;
  __shared__ uint32_t s_tmp[kBlock / 64];
  const uint32_t ex = block_scan_excl(bytes, s_tmp);
  if (!active) return;
  uint8_t *__restrict__ row = d.pool + d.blk_off[(int64_t)cj * d.tiles + blockIdx.x] + ex; // 4-byte aligned
  // the row's entries: what the count pass left in the temp buffer (outside its evaluation window the entries are
  // the saturated constants), or a second evaluation when there is no temp buffer
  const int W = 2 * d.max_bs + 2;
  int j_lo = 0, j_hi = W;
  uint32_t T_sat = 0;
  const uint16_t *__restrict__ tmp = nullptr;
  if (d.tmp) {
    tmp = d.tmp + ((int64_t)cj * d.tiles + blockIdx.x) * ((int64_t)(W + kTmpHdrRows) * kBlock) + threadIdx.x;
    j_lo = tmp[0];
    j_hi = tmp[kBlock];
    T_sat = tmp[2 * kBlock];
    tmp += kTmpHdrRows * kBlock;
  }
  auto edge_at = [&](int idx) -> uint32_t {
    if (!tmp) return L.edge(idx);
    if (idx < j_lo) return 0u;
    if (idx >= j_hi) return T_sat;
    return tmp[(int64_t)(idx - j_lo) * kBlock];
  };

  // Entries are fetched 8 at a time before any is used: a lane's loop is a chain of dependent steps, and with one
  // temp-buffer load per step the kernel would be bound by memory latency times the longest row of the wave.
  constexpr int CH = 8;
  if (!tab_row_is_ef(cnt, nonmono)) {
    // raw: uint16 entries, padded to an even count with the last value (rows are 4-byte aligned)
    const uint32_t len2 = (cnt + 1u) & ~1u;
    uint32_t last = 0;
    for (uint32_t j0 = 0; j0 < len2; j0 += CH) {
      uint32_t e[CH];
#pragma unroll
      for (int t = 0; t < CH; ++t) e[t] = (j0 + t < cnt) ? edge_at(a_idx + (int)(j0 + t)) : 0u;
#pragma unroll
      for (int t = 0; t < CH; ++t) {
        if (j0 + t < cnt) last = e[t];
        e[t] = last;
      }
#pragma unroll
      for (int t = 0; t < CH; t += 2)
        if (j0 + t < len2) *reinterpret_cast<uint32_t *>(row + 2 * (j0 + t)) = e[t] | (e[t + 1] << 16);
    }
  } else {
    // Elias-Fano, 8 low bits: lows[cnt] (padded to 8), then U 64-bit words with bit ((E_j >> 8) + j) set
    const uint32_t lows_bytes = (cnt + 7u) & ~7u, U = (cnt + 256u + 63u) >> 6;
    uint32_t *__restrict__ up32 = reinterpret_cast<uint32_t *>(row + lows_bytes); // the 64-bit words, as two halves each
    auto put_word = [&](uint32_t wi, unsigned long long w) {
      up32[2 * wi] = (uint32_t)w;
      up32[2 * wi + 1] = (uint32_t)(w >> 32);
    };
    unsigned long long wcur = 0;
    uint32_t widx = 0;
    for (uint32_t j0 = 0; j0 < lows_bytes; j0 += CH) { // lows_bytes is a multiple of 8 = CH
      uint32_t e[CH];
#pragma unroll
      for (int t = 0; t < CH; ++t) e[t] = (j0 + t < cnt) ? edge_at(a_idx + (int)(j0 + t)) : 0u;
      uint32_t lo0 = 0, lo1 = 0; // the low bytes of the 8 entries (zero padding past the row)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        lo0 |= (e[t] & 0xFFu) << (8 * t);
        lo1 |= (e[t + 4] & 0xFFu) << (8 * t);
      }
      reinterpret_cast<uint32_t *>(row + j0)[0] = lo0; // rows are only 4-byte aligned
      reinterpret_cast<uint32_t *>(row + j0)[1] = lo1;
#pragma unroll
      for (int t = 0; t < CH; ++t) {
        if (j0 + t < cnt) {
          const uint32_t pos = (e[t] >> 8) + j0 + t; // strictly increasing: the row is monotone
          const uint32_t wi = pos >> 6;
          while (widx < wi) {
            put_word(widx++, wcur);
            wcur = 0;
          }
          wcur |= 1ull << (pos & 63u);
        }
      }
    }
    while (widx < U) {
      put_word(widx++, wcur);
      wcur = 0;
    }
  }
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
  if (vec == 4) return clamped ? launch_symtab_v<4, true, PT, LINEAR>(d, count, M_max, hw_max, n_max, mode, s)
                               : launch_symtab_v<4, false, PT, LINEAR>(d, count, M_max, hw_max, n_max, mode, s);
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

int launch_hdr_pack(const DecDesc *d_descs, int count, int64_t n_max, void *stream) {
  if (count <= 0 || n_max <= 0) return 0;
  dim3 grid((unsigned)std::min<int64_t>((n_max + kBlock - 1) / kBlock, 1024), (unsigned)count);
  hipLaunchKernelGGL(hdr_pack_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, d_descs);
  return launch_err();
}

int launch_yhat_scatter(const void *sym, int wide, const int32_t *rank, float *y_hat, int M, int64_t hw, void *stream) {
  if (M <= 0 || hw <= 0) return 0;
  dim3 grid((unsigned)((hw + kBlock - 1) / kBlock), (unsigned)M);
  if (wide) hipLaunchKernelGGL(yhat_scatter_kernel<int32_t>, grid, dim3(kBlock), 0, (hipStream_t)stream, (const int32_t *)sym, rank, y_hat, hw);
  else hipLaunchKernelGGL(yhat_scatter_kernel<int16_t>, grid, dim3(kBlock), 0, (hipStream_t)stream, (const int16_t *)sym, rank, y_hat, hw);
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

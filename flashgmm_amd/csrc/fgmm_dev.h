// fgmm_dev.h — device helpers shared by fgmm_kernels.hip (encode side, misc) and fgmm_tab.hip (decode-side tables).
#pragma once
#include <hip/hip_runtime.h>

#include "fgmm_internal.h"
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

// N consecutive parameters (N = 2, 4), widened to float
template <typename PT, int N> __device__ __forceinline__ void ldv(const void *base, int64_t idx, float (&out)[N]) {
  typedef PT pvec_t __attribute__((ext_vector_type(N)));
  const pvec_t v = ldg<pvec_t>(static_cast<const PT *>(base) + idx);
#pragma unroll
  for (int e = 0; e < N; ++e) out[e] = (float)v[e];
}

// One entry of the encode-side table: both CDF edges of a symbol as one packed-fp32 pair, 16-bit quantisation, pmf == 0 -> the
// reference's bypass escape (rans_interface.cpp:498-517).  Shared by symtab_kernel (parameters from planes) and the fused parameter
// head (parameters straight from the MFMA accumulators, fgmm_head.hip).
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

// the same with 64-bit sums (values that may add up past 2^32 within one block)
__device__ __forceinline__ unsigned long long block_scan_excl64(unsigned long long v, unsigned long long *s_tmp) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned long long incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_tmp[w] = incl;
  __syncthreads();
  unsigned long long base = 0;
  for (int i = 0; i < w; ++i) base += s_tmp[i];
  __syncthreads();
  return base + incl - v;
}

} // namespace fgmm

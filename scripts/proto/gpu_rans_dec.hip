// PROTOTYPE (not part of the product): rANS decode of GPU-built tables ON the GPU, one wave per bitstream.
// Question it answers: how many ns per symbol does a wave need for the dependent chain
//   cf = x & 0xFFFF -> search the row -> (start, freq) -> x = freq * (x >> 16) + cf - start -> renormalise
// when the row's entries sit across the lanes (one compare + ballot finds the symbol) and the state lives in SGPRs?
// Tables: 4-byte headers, raw uint16 rows, sequential (fgmm_build_cdftab_hip with FGMM_TAB_RAW_ROWS); monotone rows of at
// most 192 entries; everything else raises the stream's error flag.  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC.
#include <hip/hip_runtime.h>
#include <stdint.h>

struct ProtoStream {
  const uint32_t *words; // the bitstream (device copy)
  int64_t n_words;
  const uint32_t *hdr;   // [n]
  const uint16_t *rows;  // raw rows, sequential
  int64_t n;
  int32_t max_bs, pad;
  int32_t *out;          // [n] symbols
  int32_t *status;       // 0 ok, 1 unsupported row, 2 underrun
};

constexpr int kLdsBytes = 64 * 2 * 192 + 64; // rows of 64 latents

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t lane_val(uint32_t v, uint32_t k) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)k); }

extern "C" __global__ __launch_bounds__(64) void proto_rans_dec(const ProtoStream *__restrict__ streams) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[kLdsBytes];
  const ProtoStream s = streams[blockIdx.x];
  const uint32_t lane = threadIdx.x;
  const uint32_t *__restrict__ w = s.words;
  if (s.n_words < 2) {
    if (lane == 0) *s.status = 2;
    return;
  }
  // the coder state and everything derived from it are wave-uniform: kept in SGPRs (readfirstlane pins them there, the
  // arithmetic on them is then selected for the scalar unit)
  uint32_t x_lo = uni(w[0]), x_hi = uni(w[1]);
  // the next 64 words of the bitstream across the lanes; word q is lane q - wbase
  int64_t wbase = 2;
  uint32_t wv = wbase + lane < s.n_words ? w[wbase + lane] : 0u;
  uint32_t wp = 0; // words consumed from wv
  int err = 0;
  uint64_t row_base = 0; // bytes
  const uint16_t *lds16 = reinterpret_cast<const uint16_t *>(lds);
  auto next_word = [&]() -> uint32_t { // uniform
    if (wp == 64) {
      wbase += 64;
      wv = wbase + lane < s.n_words ? w[wbase + lane] : 0u;
      wp = 0;
    }
    if (wbase + wp >= s.n_words) err = 2;
    const uint32_t r = lane_val(wv, wp);
    ++wp;
    return r;
  };
  for (int64_t base = 0; base < s.n && !err; base += 64) {
    const int64_t i = base + lane;
    const uint32_t h = i < s.n ? s.hdr[i] : 0u;
    const int32_t a_k = (int32_t)(int16_t)(uint16_t)(h & 0xFFFFu);
    const uint32_t cnt_k = (h >> 16) & 0x7FFFu, nm_k = h >> 31;
    uint32_t incl = 2 * cnt_k;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o, 64);
      if (lane >= (uint32_t)o) incl += t;
    }
    const uint32_t off_k = incl - 2 * cnt_k;
    const uint32_t total = lane_val(incl, 63);
    if (__ballot(cnt_k > 192u || nm_k) || total + 4 > (uint32_t)kLdsBytes) {
      err = 1;
      break;
    }
    const uint32_t skew = (uint32_t)(row_base & 2u);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(s.rows) + (row_base - skew));
    for (uint32_t o = lane; 4 * o < total + skew; o += 64) reinterpret_cast<uint32_t *>(lds)[o] = src[o];
    __syncthreads();
    row_base += total;
    int32_t myval = 0;
    const int nk = (int)(s.n - base < 64 ? s.n - base : 64);
    for (int k = 0; k < nk; ++k) {
      const int32_t a = (int32_t)lane_val((uint32_t)a_k, (uint32_t)k);
      const uint32_t cnt = lane_val(cnt_k, (uint32_t)k);
      const uint32_t e_off = (lane_val(off_k, (uint32_t)k) + skew) >> 1; // in uint16 units
      const uint32_t e0 = lane < cnt ? (uint32_t)lds16[e_off + lane] : 0x10000u;
      const uint32_t cf = x_lo & 0xFFFFu;
      int32_t value;
      if (__builtin_expect(cf == 0xFFFFu, 0)) { // bypass: Rans64DecAdvance(65535, 1) then nibbles
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
        while (val == 15 && !err) {
          val = (int32_t)get_bits();
          nn += val;
        }
        uint32_t raw = 0;
        for (int j = 0; j < nn && !err; ++j) raw |= get_bits() << ((j * 4) & 31);
        value = (int32_t)raw;
        x_lo = uni((uint32_t)xx);
        x_hi = uni((uint32_t)(xx >> 32));
      } else {
        uint32_t j, start, e_j;
        if (__builtin_expect(cnt <= 64, 1)) {
          const uint64_t m0 = __ballot(e0 > cf);
          j = m0 ? (uint32_t)__builtin_ctzll(m0) : cnt;
          start = lane_val(e0, (j - 1u) & 63u);
          e_j = lane_val(e0, j & 63u);
        } else { // up to 192 entries: two more registers
          const uint32_t e1 = lane + 64 < cnt ? (uint32_t)lds16[e_off + 64 + lane] : 0x10000u;
          const uint32_t e2 = lane + 128 < cnt ? (uint32_t)lds16[e_off + 128 + lane] : 0x10000u;
          const uint64_t m0 = __ballot(e0 > cf), m1 = __ballot(e1 > cf), m2 = __ballot(e2 > cf);
          j = m0 ? (uint32_t)__builtin_ctzll(m0) : (m1 ? 64u + (uint32_t)__builtin_ctzll(m1) : (m2 ? 128u + (uint32_t)__builtin_ctzll(m2) : cnt));
          auto entry = [&](uint32_t q) -> uint32_t {
            return q < 64 ? lane_val(e0, q & 63u) : (q < 128 ? lane_val(e1, q & 63u) : lane_val(e2, q & 63u));
          };
          start = entry(j - 1u);
          e_j = entry(j);
        }
        if (j == 0) start = 0; // the implied zero edge before the row
        if (__builtin_expect(j >= cnt, 0)) { // cf at or above the last entry: the reference's bisection replay (prototype: flag)
          err = 1;
          e_j = start + 1;
        }
        uint32_t freq = (e_j - start) & 0xFFFFu;
        if (__builtin_expect(!freq, 0)) {
          err = 1;
          freq = 1;
        }
        value = a + (int32_t)j - 1;
        // x = freq * (x >> 16) + (cf - start), 64 bits, on the scalar unit
        const uint32_t s_lo = (x_lo >> 16) | (x_hi << 16), s_hi = x_hi >> 16;
        const uint64_t p = (uint64_t)freq * s_lo + (((uint64_t)(freq * s_hi)) << 32) + (uint64_t)(cf - start);
        uint32_t n_lo = (uint32_t)p, n_hi = (uint32_t)(p >> 32);
        if (n_hi == 0 && n_lo < 0x80000000u) { // x < 2^31: one more word
          n_hi = n_lo;
          n_lo = next_word();
        }
        x_lo = uni(n_lo);
        x_hi = uni(n_hi);
      }
      if (lane == (uint32_t)k) myval = value;
      if (err) break;
    }
    if (i < s.n) s.out[i] = myval;
    __syncthreads();
  }
  if (lane == 0) *s.status = err;
}

extern "C" int proto_launch(const ProtoStream *d_streams, int count, void *stream) {
  hipLaunchKernelGGL(proto_rans_dec, dim3((unsigned)count), dim3(64), 0, (hipStream_t)stream, d_streams);
  return (int)hipGetLastError();
}

// fgmm_head16.hip — the parameter head's last layer on the BF16 matrix cores with binary32 accuracy ("bf16x6"), an OPTION beside the exact
// form of fgmm_head.hip (FGMM_HEAD_BF16X6 at fgmm_head_create_ex).
//
// Every weight and every feature is split into three bfloat16 parts, v = v1 + v2 + v3 with v1 = bf16(v), v2 = bf16(v - v1),
// v3 = bf16(v - v1 - v2) (the subtractions are exact in binary32; the three parts carry 24 bits of v), and a product w·x is the sum of
// the six part products of weight 2^0 .. 2^-16 — (w1,x3) (w3,x1) (w2,x2) (w1,x2) (w2,x1) (w1,x1), smallest first — each ONE
// v_mfma_f32_32x32x16_bf16 over 16 input channels, accumulated in binary32.  The three dropped products are below 2^-24 of w·x:
// the result is within ~2e-7·Σ|w·x| of the exact sum (tests: 1e-5, the bar stated for the head), at 6 × 32 instead of 8 × 64 matrix-pipe
// cycles per 16 input channels.  What it is NOT: a sequence a CPU can restate bit for bit (the matrix instruction's internal summation
// order is not documented).  Deterministic on gfx950 — the same weights and features give the same parameters on every launch, fused or
// not, encoder or decoder — so streams coded with it decode with it on MI355X; the exact form stays the default and the one pinned
// against the oracle.
//
// Tiling: 256 threads = 4 waves, one block per CU (a wave holds 192 accumulator registers): block = 16 latent channels (192 rows of W) x
// 256 positions, wave = 192 rows x 64 positions = 6 x 2 accumulator tiles; K tiles of 32 input channels through LDS as bf16 parts:
// W tile 3 x 192 x 32 from a PRE-SPLIT packed copy (36 KB contiguous per block and tile); the x tile from a pre-split copy of the
// features as well - head16_split_kernel writes it once per call, [K tile][part][position][32 channels], 1.5 x the features' bytes: the
// twelve channel-group blocks of a position tile would otherwise each split the same tile (the first cut did: 0.87 ms per Kodak batch,
// a third of it that arithmetic) - so that staging is sixteen-byte copies only and a lane's eight k values are one 16-byte read.  Rows
// of 40 bf16 (80 bytes) in LDS: the 16-byte reads and writes of 16 consecutive lanes fall on 64 different banks.  Epilogue and block
// placement as in fgmm_head.hip.
#include "fgmm_dev.h"

namespace fgmm {
namespace {

constexpr int kCG = kHeadCG, kRows = 12 * kCG, kBK = kHeadBK, kPB = 256, kTiles = kRows / 32;
constexpr int kPitch = 40;                                           // bf16 per LDS row
constexpr int kA16 = 3 * kRows * kPitch, kB16 = 3 * kPB * kPitch;    // bf16 elements of the W tile / x tile in LDS
constexpr size_t kLds16 = sizeof(uint16_t) * (size_t)(kA16 + kB16);  // 107 520 bytes

typedef float f16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u4_t __attribute__((ext_vector_type(4)));

struct Split3 {
  uint16_t p[3];
};
__device__ __forceinline__ uint16_t bf_bits(__bf16 b) { return __builtin_bit_cast(uint16_t, b); }
__device__ __forceinline__ Split3 split3(float v) {
  const __bf16 a = (__bf16)v; // round to nearest even (v_cvt_pk_bf16_f32)
  const float r1 = v - (float)a;
  const __bf16 b = (__bf16)r1;
  const float r2 = r1 - (float)b;
  const __bf16 c = (__bf16)r2;
  return Split3{{bf_bits(a), bf_bits(b), bf_bits(c)}};
}

// packed weights: bf16 [channel group][K tile][part 3][192 rows][32 input channels in natural order]; bias as in fgmm_head.hip
__global__ __launch_bounds__(256) void head16_pack_kernel(const float *__restrict__ w, const float *__restrict__ bias, int M, int c_in, int n_cg, int n_kt,
                                                          uint16_t *__restrict__ wp, float *__restrict__ bp) {
  const int64_t total = (int64_t)n_cg * n_kt * kRows * kBK;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total + (int64_t)n_cg * kRows; i += (int64_t)gridDim.x * 256) {
    const bool is_bias = i >= total;
    const int64_t e = is_bias ? i - total : i;
    const int q = is_bias ? 0 : (int)(e % kBK);
    const int r = (int)((is_bias ? e : e / kBK) % kRows);
    const int kt = is_bias ? 0 : (int)((e / (kBK * kRows)) % n_kt);
    const int cg = (int)(is_bias ? e / kRows : e / ((int64_t)kBK * kRows * n_kt));
    const int tile = r >> 5, ri = r & 31;
    const int g = tile / 3, t = tile % 3, cl = ri >> 2, k = ri & 3;
    const int c = cg * kCG + g * 8 + cl;
    const int64_t o = (int64_t)t * 4 * M + (int64_t)k * M + c;
    if (is_bias) {
      bp[e] = (c < M && bias) ? bias[o] : 0.0f;
    } else {
      const int kin = kt * kBK + q;
      const Split3 s = split3((c < M && kin < c_in) ? w[o * c_in + kin] : 0.0f);
      const int64_t base = ((int64_t)cg * n_kt + kt) * 3 * (kRows * kBK) + (int64_t)r * kBK + q;
#pragma unroll
      for (int part = 0; part < 3; ++part) wp[base + (int64_t)part * (kRows * kBK)] = s.p[part];
    }
  }
}

// features [c_in, hw] float32 -> three bf16 parts, [K tile][part][position][32 input channels] (channels past c_in: zero).  A thread =
// (position, group of 8 channels): eight strided loads (64-byte segments per wave), three 16-byte stores (contiguous per wave).
__global__ __launch_bounds__(256) void head16_split_kernel(const float *__restrict__ x0, uint16_t *__restrict__ xs0, int64_t hw, int c_in, int n_kt, int64_t x_stride,
                                                           int64_t xs_stride) {
  const float *x = x0 + (int64_t)blockIdx.z * x_stride;
  uint16_t *xs = xs0 + (int64_t)blockIdx.z * xs_stride;
  const int kt = blockIdx.y, g8 = threadIdx.x & 3;
  const int64_t pos = (int64_t)blockIdx.x * 64 + (threadIdx.x >> 2);
  if (pos >= hw) return;
  uint16_t parts[3][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int kin = kt * kBK + 8 * g8 + j;
    const Split3 s = split3(kin < c_in ? ldg<float>(x + (int64_t)kin * hw + pos) : 0.0f);
    parts[0][j] = s.p[0], parts[1][j] = s.p[1], parts[2][j] = s.p[2];
  }
#pragma unroll
  for (int part = 0; part < 3; ++part) {
    u4_t v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (uint32_t)parts[part][2 * e] | ((uint32_t)parts[part][2 * e + 1] << 16);
    stg<u4_t>(xs + (((int64_t)kt * 3 + part) * hw + pos) * kBK + 8 * g8, v);
  }
}

template <int MODE, bool CLAMPED, bool FUSED>
__global__ __launch_bounds__(256, 1) void head16_kernel(const EncDesc *__restrict__ edescs, const HeadDesc *__restrict__ hdescs, HeadW hw_, int pt_max,
                                                         int cg_max, int total) {
  extern __shared__ __attribute__((aligned(16))) uint16_t s16[];
  uint16_t *const sA = s16, *const sB = s16 + kA16;
  __shared__ int s_rank[kCG];
  const int per_xcd = (int)gridDim.x >> 3;
  const int L = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  if (L >= total) return;
  const int item = L / (pt_max * cg_max);
  const int rem = L - item * (pt_max * cg_max);
  const int pt = rem / cg_max, cg = rem - pt * cg_max;
  const uint16_t *xs; // the item's features, split by head16_split_kernel (the descriptor's x points at that copy)
  int64_t hw;
  int M;
  if constexpr (FUSED) {
    xs = reinterpret_cast<const uint16_t *>(edescs[item].x), hw = edescs[item].hw, M = edescs[item].M;
  } else {
    xs = reinterpret_cast<const uint16_t *>(hdescs[item].x), hw = hdescs[item].hw, M = hw_.M;
  }
  const int64_t P0 = (int64_t)pt * kPB;
  if (P0 >= hw || cg * kCG >= M) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
  if constexpr (FUSED) { // the compact channel of each of the block's 16 channels (fgmm_head.hip)
    const EncDesc &d = edescs[item];
    __shared__ unsigned long long s_nzmask[4];
    __shared__ int s_below[4];
    int below = 0;
    const int c_first = cg * kCG;
    for (int base = 0; base <= c_first; base += 256) {
      const int c = base + tid;
      const unsigned long long m = __ballot(c < M && d.chan_nz[c] != 0);
      if (base + 256 <= c_first) {
        below += __popcll(m);
        continue;
      }
      if (lane == 0) s_nzmask[wave] = m;
    }
    if (lane == 0) s_below[wave] = below;
    __syncthreads();
    if (tid < kCG) {
      const int c = c_first + tid, off = c & 255;
      int r = -1;
      if (c < M && ((s_nzmask[off >> 6] >> (off & 63)) & 1ull)) {
        r = s_below[0] + s_below[1] + s_below[2] + s_below[3];
        for (int wv = 0; wv < (off >> 6); ++wv) r += __popcll(s_nzmask[wv]);
        r += __popcll(s_nzmask[off >> 6] & ((1ull << (off & 63)) - 1ull));
      }
      s_rank[tid] = r;
    }
    __syncthreads();
    bool any = false;
#pragma unroll
    for (int i = 0; i < kCG; ++i) any = any || s_rank[i] >= 0;
    if (!any) return;
  }
  // the lane's sixteen latents (2 channel halves x 4 channels x 2 position tiles), fetched under the K loop
  float yv[16];
  if constexpr (FUSED) {
    const EncDesc &d = edescs[item];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int cl = ((q >> 3) & 1) * 8 + 2 * ((q >> 1) & 3) + h, np = q & 1;
      const int64_t p = P0 + wave * 64 + np * 32 + col;
      yv[q] = (s_rank[cl] >= 0 && p < hw) ? ldg<float>(d.y + (int64_t)(cg * kCG + cl) * hw + p) : 0.0f;
    }
  }
  f16_t acc[kTiles][2];
  {
    const float *bp = hw_.bp + (int64_t)cg * kRows;
#pragma unroll
    for (int tl = 0; tl < kTiles; ++tl)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float b = bp[tl * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
        acc[tl][0][r] = b, acc[tl][1][r] = b;
      }
  }
  const int n_kt = hw_.n_kt;
  const uint16_t *wp = static_cast<const uint16_t *>(hw_.wp) + (int64_t)cg * n_kt * 3 * (kRows * kBK);
  // ---- staging: sixteen-byte copies only - nine chunks of the W parts, twelve of the x parts per thread and tile
  u4_t ra[9], rb[12];
  unsigned rb_ok = 0; // bit j: chunk j lies inside the features (applied when the tile is written: a select right after the load waits for it)
  auto load_tile = [&](int kt) {
    rb_ok = 0;
    const u4_t *src = reinterpret_cast<const u4_t *>(wp + (int64_t)kt * 3 * (kRows * kBK));
#pragma unroll
    for (int j = 0; j < 9; ++j) ra[j] = src[tid + 256 * j];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int f = tid + 256 * j, part = f >> 10, r = f & 1023; // 1024 chunks per part: 256 positions x 4 groups of 8 channels
      const int64_t pos = P0 + (r >> 2);
      const uint16_t *g = xs + (((int64_t)kt * 3 + part) * hw + (pos < hw ? pos : 0)) * kBK + 8 * (r & 3);
      rb[j] = ldg<u4_t>(g);
      rb_ok |= (pos < hw ? 1u : 0u) << j;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int f = tid + 256 * j, part = f / 768, r768 = f - part * 768;
      *reinterpret_cast<u4_t *>(&sA[(part * kRows + (r768 >> 2)) * kPitch + (r768 & 3) * 8]) = ra[j];
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int f = tid + 256 * j, part = f >> 10, r = f & 1023;
      *reinterpret_cast<u4_t *>(&sB[(part * kPB + (r >> 2)) * kPitch + 8 * (r & 3)]) = (rb_ok >> j) & 1u ? rb[j] : (u4_t){0u, 0u, 0u, 0u};
    }
  };
  // the fragments of one step of 16 input channels: a lane's eight k values of a row (W) or a position (x) are one 16-byte read
  auto frag = [&](const uint16_t *base) { return __builtin_bit_cast(bf8_t, *reinterpret_cast<const u4_t *>(base)); };
#define FGMM_FENCE() __builtin_amdgcn_sched_barrier(0)
  load_tile(0);
  store_tile();
  __syncthreads();
  for (int kt = 0; kt < n_kt; ++kt) {
    FGMM_FENCE();
    load_tile(kt + 1 < n_kt ? kt + 1 : kt); // in flight under this tile's products (the last tile loads itself again: harmless)
    FGMM_FENCE();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf8_t a[kTiles][3], b[2][3];
      const int ko = 16 * s + 8 * h;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int tl = 0; tl < kTiles; ++tl) a[tl][q] = frag(&sA[(q * kRows + tl * 32 + col) * kPitch + ko]);
#pragma unroll
        for (int np = 0; np < 2; ++np) b[np][q] = frag(&sB[(q * kPB + wave * 64 + np * 32 + col) * kPitch + ko]);
      }
      FGMM_FENCE();
      // the six part products, smallest first
      constexpr int qa[6] = {0, 2, 1, 0, 1, 0}, qb[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int tl = 0; tl < kTiles; ++tl)
#pragma unroll
          for (int np = 0; np < 2; ++np) acc[tl][np] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tl][qa[t]], b[np][qb[t]], acc[tl][np], 0, 0, 0);
      FGMM_FENCE();
    }
    __syncthreads(); // every wave has read the tile
    store_tile();
    __syncthreads();
  }
#undef FGMM_FENCE
  // ---- epilogue (as fgmm_head.hip, two position tiles per lane)
  if constexpr (FUSED) {
    const EncDesc &d = edescs[item];
    int nbypass = 0;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cl = g * 8 + 2 * j + h;
        const int rank = s_rank[cl];
        uint32_t *row_out = nullptr;
        if (rank >= 0) {
          const int seg = (rank >= d.seg_b[0]) + (rank >= d.seg_b[1]) + (rank >= d.seg_b[2]);
          row_out = (seg ? d.packed_seg[seg] : d.packed) + (int64_t)(rank - seg * d.cps) * hw;
        }
#pragma unroll
        for (int np = 0; np < 2; ++np) {
          const int64_t p = P0 + wave * 64 + np * 32 + col;
          int bp = 0;
          if (rank >= 0 && p < hw) {
            float sg[4], mu[4], pi[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              sg[k] = acc[g * 3 + 0][np][4 * j + k];
              mu[k] = acc[g * 3 + 1][np][4 * j + k];
              pi[k] = acc[g * 3 + 2][np][4 * j + k];
            }
            softmax4(pi);
            const float vq = __builtin_rintf(yv[g * 8 + j * 2 + np]);
            stg<uint32_t>(row_out + p, sym_entry<MODE, CLAMPED>(vq, (int)vq, mu, sg, pi, bp));
          }
          nbypass += __popcll(__ballot(bp));
        }
      }
    if (lane == 0 && nbypass) atomicAdd(d.meta + ((L * 4 + wave) % (int)d.meta_slots), (uint32_t)nbypass);
  } else {
    float *out = hdescs[item].out;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = cg * kCG + g * 8 + 2 * j + h;
#pragma unroll
        for (int np = 0; np < 2; ++np) {
          const int64_t p = P0 + wave * 64 + np * 32 + col;
          if (c < M && p < hw) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
              for (int k = 0; k < 4; ++k) out[((int64_t)t * 4 * M + (int64_t)k * M + c) * hw + p] = acc[g * 3 + t][np][4 * j + k];
          }
        }
      }
  }
}

template <typename K> int lds16(K kernel) {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds16);
}

template <int MODE, bool CLAMPED> int launch_fused16(const EncDesc *descs, const HeadW &w, int count, int M_max, int64_t hw_max, hipStream_t st) {
  const int pt_max = (int)((hw_max + kPB - 1) / kPB), cg_max = (M_max + kCG - 1) / kCG;
  const int64_t total = (int64_t)count * pt_max * cg_max;
  if (total <= 0) return 0;
  if (total > (1ll << 30)) return (int)hipErrorInvalidValue;
  auto kernel = head16_kernel<MODE, CLAMPED, true>;
  if (int e = lds16(kernel)) return e;
  hipLaunchKernelGGL(kernel, dim3((unsigned)((total + 7) / 8 * 8)), dim3(256), kLds16, st, descs, (const HeadDesc *)nullptr, w, pt_max, cg_max, (int)total);
  return (int)hipGetLastError();
}

} // namespace

size_t head16_packed_bytes(int M, int c_in) {
  const size_t n_cg = ((size_t)M + kCG - 1) / kCG, n_kt = ((size_t)c_in + kBK - 1) / kBK;
  return n_cg * n_kt * 3 * kRows * kBK * sizeof(uint16_t) + n_cg * kRows * sizeof(float);
}

int launch_head16_pack(const float *w, const float *bias, int M, int c_in, void *packed, void *stream) {
  const int n_cg = (M + kCG - 1) / kCG, n_kt = (c_in + kBK - 1) / kBK;
  const int64_t total = (int64_t)n_cg * n_kt * kRows * kBK + (int64_t)n_cg * kRows;
  uint16_t *wp = static_cast<uint16_t *>(packed);
  float *bp = reinterpret_cast<float *>(wp + (size_t)n_cg * n_kt * 3 * kRows * kBK);
  hipLaunchKernelGGL(head16_pack_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, w, bias, M, c_in, n_cg, n_kt,
                     wp, bp);
  return (int)hipGetLastError();
}

size_t head16_split_elems(int c_in, int64_t hw) { return (size_t)((c_in + kBK - 1) / kBK) * 3 * (size_t)hw * kBK; }

// `count` items of one size whose features (x0 + i * x_stride floats) and split copies (xs0 + i * xs_stride bf16) are evenly spaced
int launch_head16_split(const float *x0, void *xs0, int64_t hw, int c_in, int count, int64_t x_stride, int64_t xs_stride, void *stream) {
  if (count <= 0 || hw <= 0) return 0;
  const int n_kt = (c_in + kBK - 1) / kBK;
  hipLaunchKernelGGL(head16_split_kernel, dim3((unsigned)((hw + 63) / 64), (unsigned)n_kt, (unsigned)count), dim3(256), 0, (hipStream_t)stream, x0,
                     static_cast<uint16_t *>(xs0), hw, c_in, n_kt, x_stride, xs_stride);
  return (int)hipGetLastError();
}

int launch_head16_params(const HeadDesc *d_descs, const HeadW &w, int count, int64_t hw_max, void *stream) {
  const int pt_max = (int)((hw_max + kPB - 1) / kPB), cg_max = w.n_cg;
  const int64_t total = (int64_t)count * pt_max * cg_max;
  if (total <= 0) return 0;
  if (total > (1ll << 30)) return (int)hipErrorInvalidValue;
  auto kernel = head16_kernel<0, true, false>;
  if (int e = lds16(kernel)) return e;
  hipLaunchKernelGGL(kernel, dim3((unsigned)((total + 7) / 8 * 8)), dim3(256), kLds16, (hipStream_t)stream, (const EncDesc *)nullptr, d_descs, w, pt_max, cg_max,
                     (int)total);
  return (int)hipGetLastError();
}

int launch_head16_symtab(const EncDesc *d_descs, const HeadW &w, int count, int M_max, int64_t hw_max, int mode, bool clamped, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (mode * 2 + (clamped ? 1 : 0)) {
  case 0: return launch_fused16<0, false>(d_descs, w, count, M_max, hw_max, st);
  case 1: return launch_fused16<0, true>(d_descs, w, count, M_max, hw_max, st);
  case 2: return launch_fused16<1, false>(d_descs, w, count, M_max, hw_max, st);
  case 3: return launch_fused16<1, true>(d_descs, w, count, M_max, hw_max, st);
  case 4: return launch_fused16<2, false>(d_descs, w, count, M_max, hw_max, st);
  case 5: return launch_fused16<2, true>(d_descs, w, count, M_max, hw_max, st);
  }
  return (int)hipErrorInvalidValue;
}

} // namespace fgmm

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
// Tiling: 512 threads = 8 waves, one block per CU, two waves per SIMD: block = 16 latent channels (192 rows of W) x 256 positions, wave =
// 96 rows x 64 positions = 3 x 2 accumulator tiles; K tiles of 16 input channels (one matrix step) through two LDS buffers as bf16 parts:
// W tile 3 x 192 x 16 from a PRE-SPLIT packed copy (18 KB contiguous per block and tile); the x tile from a pre-split copy of the
// features as well - head16_split_kernel writes it once per call, [K tile][part][position][16 channels], 1.5 x the features' bytes: the
// twelve channel-group blocks of a position tile would otherwise each split the same tile (the first cut did: 0.87 ms per Kodak batch,
// a third of it that arithmetic) - so that staging is sixteen-byte copies only and a lane's eight k values are one 16-byte read.  Rows
// of 24 bf16 (48 bytes) in LDS: the 16-byte reads and writes of 16 consecutive lanes fall on 64 different banks.  Epilogue and block
// placement as in fgmm_head.hip.
#include "fgmm_dev.h"

namespace fgmm {
namespace {

constexpr int kCG = kHeadCG, kRows = 12 * kCG, kBK = 16, kPB = 256; // (K tiles of SIXTEEN input channels here: one matrix step)
constexpr int kPitch = 24;                                           // bf16 per LDS row (48 bytes)
constexpr int kA16 = 3 * kRows * kPitch, kB16 = 3 * kPB * kPitch;    // bf16 elements of the W tile / x tile in LDS
constexpr int kBuf16 = kA16 + kB16;                                  // one staged (W tile, x tile) pair: 64 512 bytes
constexpr size_t kLds16 = 2 * sizeof(uint16_t) * (size_t)kBuf16;     // two of them: 129 024 bytes

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

// packed weights: bf16 [channel group][K tile of 16][part 3][192 rows][16 input channels in natural order]; bias as in fgmm_head.hip
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

// features [c_in, hw] float32 -> three bf16 parts, [K tile of 16][part][position][16 input channels] (channels past c_in: zero).  A thread
// = (position, group of 8 channels): eight strided loads (128-byte segments per wave), three 16-byte stores (contiguous per wave).
__global__ __launch_bounds__(256) void head16_split_kernel(const float *__restrict__ x0, uint16_t *__restrict__ xs0, int64_t hw, int c_in, int n_kt, int64_t x_stride,
                                                           int64_t xs_stride) {
  const float *x = x0 + (int64_t)blockIdx.z * x_stride;
  uint16_t *xs = xs0 + (int64_t)blockIdx.z * xs_stride;
  const int kt = blockIdx.y, g8 = threadIdx.x & 1;
  const int64_t pos = (int64_t)blockIdx.x * 128 + (threadIdx.x >> 1);
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

// 512 threads = 8 waves, two per SIMD (each within 256 registers): wave = (row half rh, position quarter pq) = 96 rows (the three
// parameter tiles of channels 8 rh .. 8 rh + 7) x 64 positions = 3 x 2 accumulator tiles.  One wave's LDS reads and barrier waits fall
// under its SIMD partner's products.
constexpr int kThreads = 512, kWaves = kThreads / 64;
template <int MODE, bool CLAMPED, bool FUSED>
__global__ __launch_bounds__(kThreads) void head16_kernel(const EncDesc *__restrict__ edescs, const HeadDesc *__restrict__ hdescs, HeadW hw_, int pt_max,
                                                           int cg_max, int total) {
  extern __shared__ __attribute__((aligned(16))) uint16_t s16[];
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
  const int rh = wave & 1, pq = wave >> 1;
  if constexpr (FUSED) { // the compact channel of each of the block's 16 channels (fgmm_head.hip; chunks of 512 channels here)
    const EncDesc &d = edescs[item];
    __shared__ unsigned long long s_nzmask[kWaves];
    __shared__ int s_below[kWaves];
    int below = 0;
    const int c_first = cg * kCG;
    for (int base = 0; base <= c_first; base += kThreads) {
      const int c = base + tid;
      const unsigned long long m = __ballot(c < M && d.chan_nz[c] != 0);
      if (base + kThreads <= c_first) {
        below += __popcll(m);
        continue;
      }
      if (lane == 0) s_nzmask[wave] = m;
    }
    if (lane == 0) s_below[wave] = below;
    __syncthreads();
    if (tid < kCG) {
      const int c = c_first + tid, off = c & (kThreads - 1);
      int r = -1;
      if (c < M && ((s_nzmask[off >> 6] >> (off & 63)) & 1ull)) {
        r = 0;
        for (int wv = 0; wv < kWaves; ++wv) r += s_below[wv];
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
  // the lane's eight latents (4 channels x 2 position tiles), fetched under the K loop
  float yv[8];
  if constexpr (FUSED) {
    const EncDesc &d = edescs[item];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int cl = rh * 8 + 2 * (q >> 1) + h, np = q & 1;
      const int64_t p = P0 + pq * 64 + np * 32 + col;
      yv[q] = (s_rank[cl] >= 0 && p < hw) ? ldg<float>(d.y + (int64_t)(cg * kCG + cl) * hw + p) : 0.0f;
    }
  }
  f16_t acc[3][2];
  {
    const float *bp = hw_.bp + (int64_t)cg * kRows;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float b = bp[(rh * 3 + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
        acc[t][0][r] = b, acc[t][1][r] = b;
      }
  }
  const int n_kt = (hw_.c_in + kBK - 1) / kBK;
  const uint16_t *wp = static_cast<const uint16_t *>(hw_.wp) + (int64_t)cg * n_kt * 3 * (kRows * kBK);
  // ---- staging: sixteen-byte copies only - per thread and tile two or three chunks of the W parts (1152 in all), three of the x parts;
  // TWO register sets: the loads of tile t + 2 are issued in step t and written to LDS at the end of step t + 1 (one step of products
  // is about an L2 round trip under load: with one set the write waited for its loads)
  struct Stage {
    u4_t ra[3], rb[3];
    unsigned ok = 0; // bit j: x chunk j lies inside the features (applied when the tile is written: a select right after the load waits for it)
  };
  auto load_tile = [&](Stage &g, int kt) {
    g.ok = 0;
    const u4_t *src = reinterpret_cast<const u4_t *>(wp + (int64_t)kt * 3 * (kRows * kBK));
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int f = tid + kThreads * j;
      g.ra[j] = src[f < 1152 ? f : 0];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int f = tid + kThreads * j, part = f >> 9, r = f & 511; // 512 chunks per part: 256 positions x 2 groups of 8 channels
      const int64_t pos = P0 + (r >> 1);
      g.rb[j] = ldg<u4_t>(xs + (((int64_t)kt * 3 + part) * hw + (pos < hw ? pos : 0)) * kBK + 8 * (r & 1));
      g.ok |= (pos < hw ? 1u : 0u) << j;
    }
  };
  auto store_tile = [&](const Stage &g, uint16_t *bA, uint16_t *bB) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int f = tid + kThreads * j, part = f / 384, r384 = f - part * 384;
      if (f < 1152) *reinterpret_cast<u4_t *>(&bA[(part * kRows + (r384 >> 1)) * kPitch + (r384 & 1) * 8]) = g.ra[j];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int f = tid + kThreads * j, part = f >> 9, r = f & 511;
      *reinterpret_cast<u4_t *>(&bB[(part * kPB + (r >> 1)) * kPitch + 8 * (r & 1)]) = (g.ok >> j) & 1u ? g.rb[j] : (u4_t){0u, 0u, 0u, 0u};
    }
  };
  // Two LDS buffers, one barrier per tile of 16 input channels: tile t is multiplied out of buffer t & 1 (15 sixteen-byte fragment reads,
  // 36 products per wave) while the global loads of tiles t + 1 and t + 2 are in flight; tile t + 1 is written into the other buffer -
  // last read a tile ago, a barrier behind - at the end of the step.  Tiles past the last load the last one again: written, never multiplied.
#define FGMM_FENCE() __builtin_amdgcn_sched_barrier(0)
  uint16_t *const buf0 = s16, *const buf1 = s16 + kBuf16;
  Stage g0, g1;
  load_tile(g0, 0);
  load_tile(g1, n_kt > 1 ? 1 : 0);
  store_tile(g0, buf0, buf0 + kA16);
  __syncthreads();
  auto step = [&](int kt, Stage &issue, const Stage &land) { // `issue` takes tile kt + 2, `land` holds tile kt + 1
    const uint16_t *bA = (kt & 1) ? buf1 : buf0, *bB = bA + kA16;
    uint16_t *nA = (kt & 1) ? buf0 : buf1;
    FGMM_FENCE();
    load_tile(issue, kt + 2 < n_kt ? kt + 2 : n_kt - 1);
    bf8_t a[3][3], b[2][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
      for (int t = 0; t < 3; ++t) a[t][q] = __builtin_bit_cast(bf8_t, *reinterpret_cast<const u4_t *>(&bA[(q * kRows + (rh * 3 + t) * 32 + col) * kPitch + 8 * h]));
#pragma unroll
      for (int np = 0; np < 2; ++np) b[np][q] = __builtin_bit_cast(bf8_t, *reinterpret_cast<const u4_t *>(&bB[(q * kPB + pq * 64 + np * 32 + col) * kPitch + 8 * h]));
    }
    FGMM_FENCE();
    constexpr int qa[6] = {0, 2, 1, 0, 1, 0}, qb[6] = {2, 0, 1, 1, 0, 0}; // the six part products, smallest first
#pragma unroll
    for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int np = 0; np < 2; ++np) acc[t][np] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][qa[t6]], b[np][qb[t6]], acc[t][np], 0, 0, 0);
    FGMM_FENCE();
    store_tile(land, nA, nA + kA16);
    FGMM_FENCE();
    __syncthreads();
  };
  for (int kt = 0; kt < n_kt; kt += 2) {
    step(kt, g0, g1);
    if (kt + 1 < n_kt) step(kt + 1, g1, g0);
  }
#undef FGMM_FENCE
  // ---- epilogue (as fgmm_head.hip): the lane's eight latents, all twelve parameters in registers
  if constexpr (FUSED) {
    const EncDesc &d = edescs[item];
    int nbypass = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cl = rh * 8 + 2 * j + h;
      const int rank = s_rank[cl];
      uint32_t *row_out = nullptr;
      if (rank >= 0) {
        const int seg = (rank >= d.seg_b[0]) + (rank >= d.seg_b[1]) + (rank >= d.seg_b[2]);
        row_out = (seg ? d.packed_seg[seg] : d.packed) + (int64_t)(rank - seg * d.cps) * hw;
      }
#pragma unroll
      for (int np = 0; np < 2; ++np) {
        const int64_t p = P0 + pq * 64 + np * 32 + col;
        int bp = 0;
        if (rank >= 0 && p < hw) {
          float sg[4], mu[4], pi[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            sg[k] = acc[0][np][4 * j + k];
            mu[k] = acc[1][np][4 * j + k];
            pi[k] = acc[2][np][4 * j + k];
          }
          softmax4(pi);
          const float vq = __builtin_rintf(yv[j * 2 + np]);
          stg<uint32_t>(row_out + p, sym_entry<MODE, CLAMPED>(vq, (int)vq, mu, sg, pi, bp));
        }
        nbypass += __popcll(__ballot(bp));
      }
    }
    if (lane == 0 && nbypass) atomicAdd(d.meta + ((L * kWaves + wave) % (int)d.meta_slots), (uint32_t)nbypass);
  } else {
    float *out = hdescs[item].out;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = cg * kCG + rh * 8 + 2 * j + h;
#pragma unroll
      for (int np = 0; np < 2; ++np) {
        const int64_t p = P0 + pq * 64 + np * 32 + col;
        if (c < M && p < hw) {
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) out[((int64_t)t * 4 * M + (int64_t)k * M + c) * hw + p] = acc[t][np][4 * j + k];
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
  hipLaunchKernelGGL(kernel, dim3((unsigned)((total + 7) / 8 * 8)), dim3(kThreads), kLds16, st, descs, (const HeadDesc *)nullptr, w, pt_max, cg_max, (int)total);
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
  hipLaunchKernelGGL(head16_split_kernel, dim3((unsigned)((hw + 127) / 128), (unsigned)n_kt, (unsigned)count), dim3(256), 0, (hipStream_t)stream, x0,
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
  hipLaunchKernelGGL(kernel, dim3((unsigned)((total + 7) / 8 * 8)), dim3(kThreads), kLds16, (hipStream_t)stream, (const EncDesc *)nullptr, d_descs, w, pt_max, cg_max,
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

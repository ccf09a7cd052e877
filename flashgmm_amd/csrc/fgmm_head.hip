// fgmm_head.hip — the parameter head's last layer on the matrix cores, fused with the encode-side CDF kernel.
//
// What it replaces: the final 1x1 convolution of `entropy_parameters`, nn.Conv2d(N*10//3, 3*K*N, 1) (compressai/models/ckbd_gmm.py:115-121),
// the chunk(3, 1) into scales | means | weights and the softmax over K (compressai/latent_codecs/gaussian_mixture_conditional.py:183-202),
// the sigma clamp (compressai/entropy_models/entropy_models.py:817) - and, in the FUSED form, symtab_kernel: the 3*K parameters of a
// latent never exist in HBM, they go from the MFMA accumulators straight into sym_entry() and only the packed 4-byte table entry is
// written (56 -> 8 B of HBM traffic per symbol beside the features the convolution reads anyway).
//
// Arithmetic: out[o][p] = bias[o] + sum_k W[o][k] * x[k][p] on v_mfma_f32_32x32x2_f32 - exact binary32, and bit for bit the chain
//     acc = bias[o];  for k = 0 .. c_in - 1:  acc = fmaf(W[o][k], x[k][p], acc)
// (one rounding per product, k ascending: lane half 0 of the instruction holds the even k of a pair, half 1 the odd one).  The order is
// the library's, not a BLAS's: encoder and decoder get the same parameters from the same weights on any ROCm / torch / MIOpen version,
// which the reference silently relies on.  The CPU checker under tests restates the chain with fmaf: the GPU tests compare
// bit for bit.
//
// Tiling (wave64, 256 threads = 4 waves, TWO blocks per CU - a wave holds 96 accumulator registers and at most 256 in all):
//   block  = 16 latent channels (their 3*K = 12 parameters each: 192 rows of W) x 128 positions
//   wave   = all 192 rows x 32 positions = 6 row tiles of 32x32 accumulators
//   K loop = tiles of 32 input channels staged through LDS (W tile 192 x 32 from a PRE-PACKED copy of the weights: one contiguous 24 KB
//            read per block and tile; x tile 32 x 128): the global loads of tile i + 1 are in flight while tile i is multiplied, the
//            operand fragments are read from LDS one group of four k-pairs ahead of the products that use them.  One LDS buffer,
//            two barriers per tile: the CU's OTHER block multiplies meanwhile (two blocks drift out of phase by themselves - a block's
//            prologue, barriers and epilogue fall under the other's products; one block per CU with twice the tile: 0.67 of the roof)
//   rows   : row tile (g, t), g = which 8 of the 16 channels, t = scales | means | logits; row 4 * cl + k within it = parameter
//            (t, k) of channel 8 g + cl.  The 32x32 accumulator map (row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), column = lane & 31)
//            then gives every lane, for ITS position, registers 4 j + k = parameter (t, k) of channel 8 g + 2 j + (lane >> 5): a lane
//            owns all twelve parameters of its latents - the epilogue needs no lane movement and no LDS.
//   blocks that share an x tile (the channel groups of one position tile) get consecutive slots on ONE XCD: its L2 holds the tile.
#include "fgmm_dev.h"

#ifndef FGMM_HEAD_EXP
#define FGMM_HEAD_EXP 0 // experiments (wrong results): bit 0 = no global loads in the K loop, bit 1 = no LDS writes, bit 2 = no barrier
#endif

namespace fgmm {
namespace {

constexpr int kCG = kHeadCG;          // latent channels per block
constexpr int kRows = 12 * kCG;       // rows of W per block (192)
constexpr int kBK = kHeadBK;          // input channels per LDS tile
constexpr int kPB = 128;              // positions per block (4 waves x 32)
constexpr int kALd = kBK + 4;         // LDS row pitch of the W tile (floats): 16-byte reads of 64 lanes hit 64 different banks
constexpr int kBLd = kPB + 32;        // ... of the x tile: the two lane halves read rows k, k + 1 -> banks 32 apart
constexpr int kTiles = kRows / 32;    // 6 row tiles

typedef float f16_t __attribute__((ext_vector_type(16)));

// packed weights: [channel group][K tile][192 rows][32 columns]; column h * 16 + s of a tile = input channel tile * 32 + 2 s + h
__global__ __launch_bounds__(256) void head_pack_kernel(const float *__restrict__ w, const float *__restrict__ bias, int M, int c_in, int n_cg, int n_kt,
                                                        float *__restrict__ wp, float *__restrict__ bp) {
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
    const int64_t o = (int64_t)t * 4 * M + (int64_t)k * M + c; // output channel of the convolution: chunk t, component k, latent channel c
    if (is_bias) {
      bp[e] = (c < M && bias) ? bias[o] : 0.0f;
    } else {
      const int kin = kt * kBK + 2 * (q & 15) + (q >> 4);
      wp[e] = (c < M && kin < c_in) ? w[o * c_in + kin] : 0.0f;
    }
  }
}

// FUSED: the epilogue evaluates the table entry (EncDesc: y, channel census of quant_stats_kernel, the table's place); else it writes the
// three parameter tensors as planes [3 * 4 * M, hw] (scales | means | logits, channel k * M + c).
// VEC: every item's positions are a multiple of 4 and its features 16-byte aligned (checked by the host): the x tile is staged with
// 16-byte loads; else element by element (any shape, slowly).
template <int MODE, bool CLAMPED, bool FUSED, bool VEC>
__global__ __launch_bounds__(256, 2) void head_kernel(const EncDesc *__restrict__ edescs, const HeadDesc *__restrict__ hdescs, HeadW hw_, int pt_max,
                                                       int cg_max, int total) {
  __shared__ __attribute__((aligned(16))) float sA[kRows * kALd];
  __shared__ __attribute__((aligned(16))) float sB[kBK * kBLd];
  __shared__ int s_rank[kCG];
  // ---- which (item, position tile, channel group): consecutive logical slots on one XCD (blocks are dealt to the 8 XCDs round robin)
  const int per_xcd = (int)gridDim.x >> 3;
  const int L = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  if (L >= total) return;
  const int item = L / (pt_max * cg_max);
  const int rem = L - item * (pt_max * cg_max);
  const int pt = rem / cg_max, cg = rem - pt * cg_max;
  const float *x;
  int64_t hw;
  int M;
  if constexpr (FUSED) {
    x = edescs[item].x, hw = edescs[item].hw, M = edescs[item].M;
  } else {
    x = hdescs[item].x, hw = hdescs[item].hw, M = hw_.M;
  }
  const int64_t P0 = (int64_t)pt * kPB;
  if (P0 >= hw || cg * kCG >= M) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
  if constexpr (FUSED) {
    // compact (coded) channel of each of the block's 16 channels, -1 = no coded symbol: a prefix sum over quant_stats' census
    const EncDesc &d = edescs[item];
    // (every thread takes channels tid, tid + 256 ...: one ballot per wave and chunk, the sixteen ranks are popcounts below a channel)
    __shared__ unsigned long long s_nzmask[4];
    int below = 0; // coded channels in the chunks before the one that holds this block's channels
    const int c_first = cg * kCG;
    for (int base = 0; base <= c_first; base += 256) {
      const int c = base + tid;
      const unsigned long long m = __ballot(c < M && d.chan_nz[c] != 0);
      if (base + 256 <= c_first) {
        below += __popcll(m); // (per wave; summed over the waves below)
        continue;
      }
      if (lane == 0) s_nzmask[wave] = m;
    }
    __shared__ int s_below[4];
    if (lane == 0) s_below[wave] = below;
    __syncthreads();
    if (tid < kCG) {
      const int c = c_first + tid, off = c & 255; // (the block's channels lie in one chunk: 256 is a multiple of 16)
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
    if (!any) return; // sixteen channels that round to zero everywhere: nothing to code, nothing to multiply
  }
  // the lane's eight latents, fetched now: their latency lies under the K loop instead of in front of the epilogue
  float yv[8];
  if constexpr (FUSED) {
    const EncDesc &d = edescs[item];
    const int64_t p = P0 + wave * 32 + col;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = cg * kCG + (q >> 2) * 8 + 2 * (q & 3) + h;
      yv[q] = (s_rank[(q >> 2) * 8 + 2 * (q & 3) + h] >= 0 && p < hw) ? ldg<float>(d.y + (int64_t)c * hw + p) : 0.0f;
    }
  }
  // ---- accumulators start at the bias
  f16_t acc[kTiles];
  {
    const float *bp = hw_.bp + (int64_t)cg * kRows;
#pragma unroll
    for (int tl = 0; tl < kTiles; ++tl)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tl][r] = bp[tl * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
  }
  // ---- K loop
  const int n_kt = hw_.n_kt, c_in = hw_.c_in;
  const float *wp = static_cast<const float *>(hw_.wp) + (int64_t)cg * n_kt * (kRows * kBK);
  // Staging, branch-free (the K loop is ONE basic block, so that the instruction order below is the order issued): a load that would
  // fall outside the features (input channels past c_in in the last tile, positions past hw in the last position tile) reads a valid
  // address instead and is replaced by zero; the packed weights are zero there as well.
  float4_t ra[6], rb[4];
  unsigned rb_ok = 0; // bit 4 j + e: element e of rb[j] lies inside the features (applied when the tile is WRITTEN: a select right after
                      // the load would wait for it, and the loads are there to be in flight under a tile's products)
  auto load_tile = [&](int kt) {
    rb_ok = 0;
    const float4_t *src = reinterpret_cast<const float4_t *>(wp + (int64_t)kt * (kRows * kBK));
#pragma unroll
    for (int j = 0; j < 6; ++j) ra[j] = src[tid + 256 * j];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = tid + 256 * j, kk = f >> 5, p4 = f & 31;
      const int kin = kt * kBK + kk;
      const int64_t p = P0 + 4 * p4;
      const bool k_ok = kin < c_in;
      const float *g = x + (int64_t)(k_ok ? kin : c_in - 1) * hw;
      float4_t v;
      if constexpr (VEC) {
        v = ldg<float4_t>(g + (p < hw ? p : 0));
        rb_ok |= (k_ok && p < hw ? 0xFu : 0u) << (4 * j); // (hw is a multiple of 4: the four positions are inside or outside together)
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = g[p + e < hw ? p + e : 0];
          rb_ok |= (k_ok && p + e < hw ? 1u : 0u) << (4 * j + e);
        }
      }
      rb[j] = v;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int f = tid + 256 * j;
      *reinterpret_cast<float4_t *>(&sA[(f >> 3) * kALd + (f & 7) * 4]) = ra[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = tid + 256 * j;
      float4_t v = rb[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (rb_ok >> (4 * j + e)) & 1u ? v[e] : 0.0f;
      *reinterpret_cast<float4_t *>(&sB[(f >> 5) * kBLd + (f & 31) * 4]) = v;
    }
  };
  // the fragments of four k-pairs (one 16-byte read per row tile, one 4-byte read per k-pair): read one group AHEAD of the products
  // that use them
  struct Frag {
    float4_t a[kTiles];
    float b[4];
  };
  auto read_frag = [&](Frag &f, int s4) {
    const float *a_base = sA + col * kALd + h * 16 + s4 * 4;
    const float *b_base = sB + h * kBLd + wave * 32 + col + 8 * s4 * kBLd;
#pragma unroll
    for (int tl = 0; tl < kTiles; ++tl) f.a[tl] = *reinterpret_cast<const float4_t *>(a_base + tl * 32 * kALd);
#pragma unroll
    for (int e = 0; e < 4; ++e) f.b[e] = b_base[2 * e * kBLd];
  };
  auto multiply = [&](const Frag &f) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int tl = 0; tl < kTiles; ++tl) acc[tl] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[tl][e], f.b[e], acc[tl], 0, 0, 0);
  };
  // __builtin_amdgcn_sched_barrier(0): nothing is scheduled across it.  Left to itself the scheduler sinks every LDS read to just before
  // its first use and the global loads to just before the LDS writes - each then waited for with the matrix pipe idle.
#define FGMM_FENCE() __builtin_amdgcn_sched_barrier(0)
  load_tile(0);
  store_tile();
  __syncthreads();
  Frag f0, f1;
  read_frag(f0, 0);
  for (int kt = 0; kt < n_kt; ++kt) {
    // 96 products per wave and tile in four groups of 24; before a group, what a LATER step needs is issued: the global loads of the
    // next tile (the last tile loads itself again: harmless) and the fragments of the next group
    FGMM_FENCE();
#if !(FGMM_HEAD_EXP & 1)
    load_tile(kt + 1 < n_kt ? kt + 1 : kt);
#endif
    read_frag(f1, 1);
    FGMM_FENCE();
    multiply(f0);
    FGMM_FENCE();
    read_frag(f0, 2);
    FGMM_FENCE();
    multiply(f1);
    FGMM_FENCE();
    read_frag(f1, 3);
    FGMM_FENCE();
    multiply(f0);
    multiply(f1);
    FGMM_FENCE();
#if !(FGMM_HEAD_EXP & 4)
    __syncthreads(); // every wave has read the tile
#endif
#if !(FGMM_HEAD_EXP & 2)
    store_tile();
#endif
#if !(FGMM_HEAD_EXP & 4)
    __syncthreads();
#endif
    read_frag(f0, 0); // (after the last tile: a read that nobody uses)
  }
#undef FGMM_FENCE
  // ---- epilogue: the lane's 8 latents (2 channel halves x 4 channels at its position), all twelve parameters in registers
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
        const int64_t p = P0 + wave * 32 + col;
        int bp = 0;
        if (rank >= 0 && p < hw) {
          float sg[4], mu[4], pi[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            sg[k] = acc[g * 3 + 0][4 * j + k];
            mu[k] = acc[g * 3 + 1][4 * j + k];
            pi[k] = acc[g * 3 + 2][4 * j + k];
          }
          softmax4(pi);
          const float vq = __builtin_rintf(yv[g * 4 + j]);
          stg<uint32_t>(row_out + p, sym_entry<MODE, CLAMPED>(vq, (int)vq, mu, sg, pi, bp));
        }
        nbypass += __popcll(__ballot(bp));
      }
    // bypass census: the host sums the item's slots; one atomic per wave that saw any (rare: 0.2 % of the symbols)
    if (lane == 0 && nbypass) atomicAdd(d.meta + ((L * 4 + wave) % (int)d.meta_slots), (uint32_t)nbypass);
  } else {
    float *out = hdescs[item].out;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = cg * kCG + g * 8 + 2 * j + h;
        const int64_t p = P0 + wave * 32 + col;
        if (c < M && p < hw) {
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) out[((int64_t)t * 4 * M + (int64_t)k * M + c) * hw + p] = acc[g * 3 + t][4 * j + k];
        }
      }
  }
}

template <int MODE, bool CLAMPED, bool VEC>
int launch_fused(const EncDesc *descs, const HeadW &w, int count, int M_max, int64_t hw_max, hipStream_t st) {
  const int pt_max = (int)((hw_max + kPB - 1) / kPB), cg_max = (M_max + kCG - 1) / kCG;
  const int64_t total = (int64_t)count * pt_max * cg_max;
  if (total <= 0) return 0;
  if (total > (1ll << 30)) return (int)hipErrorInvalidValue;
  const unsigned grid = (unsigned)((total + 7) / 8 * 8);
  auto kernel = head_kernel<MODE, CLAMPED, true, VEC>;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, st, descs, (const HeadDesc *)nullptr, w, pt_max, cg_max, (int)total);
  return (int)hipGetLastError();
}

} // namespace

int launch_head_pack(const float *w, const float *bias, int M, int c_in, float *wp, float *bp, void *stream) {
  const int n_cg = (M + kCG - 1) / kCG, n_kt = (c_in + kBK - 1) / kBK;
  const int64_t total = (int64_t)n_cg * n_kt * kRows * kBK + (int64_t)n_cg * kRows;
  const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(head_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, bias, M, c_in, n_cg, n_kt, wp, bp);
  return (int)hipGetLastError();
}

int launch_head_params(const HeadDesc *d_descs, const HeadW &w, int count, int64_t hw_max, bool vec, void *stream) {
  const int pt_max = (int)((hw_max + kPB - 1) / kPB), cg_max = w.n_cg;
  const int64_t total = (int64_t)count * pt_max * cg_max;
  if (total <= 0) return 0;
  if (total > (1ll << 30)) return (int)hipErrorInvalidValue;
  const unsigned grid = (unsigned)((total + 7) / 8 * 8);
  auto kernel = vec ? head_kernel<0, true, false, true> : head_kernel<0, true, false, false>;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const EncDesc *)nullptr, d_descs, w, pt_max, cg_max, (int)total);
  return (int)hipGetLastError();
}

int launch_head_symtab(const EncDesc *d_descs, const HeadW &w, int count, int M_max, int64_t hw_max, int mode, bool clamped, bool vec, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  switch ((mode * 2 + (clamped ? 1 : 0)) * 2 + (vec ? 1 : 0)) {
  case 0: return launch_fused<0, false, false>(d_descs, w, count, M_max, hw_max, st);
  case 1: return launch_fused<0, false, true>(d_descs, w, count, M_max, hw_max, st);
  case 2: return launch_fused<0, true, false>(d_descs, w, count, M_max, hw_max, st);
  case 3: return launch_fused<0, true, true>(d_descs, w, count, M_max, hw_max, st);
  case 4: return launch_fused<1, false, false>(d_descs, w, count, M_max, hw_max, st);
  case 5: return launch_fused<1, false, true>(d_descs, w, count, M_max, hw_max, st);
  case 6: return launch_fused<1, true, false>(d_descs, w, count, M_max, hw_max, st);
  case 7: return launch_fused<1, true, true>(d_descs, w, count, M_max, hw_max, st);
  case 8: return launch_fused<2, false, false>(d_descs, w, count, M_max, hw_max, st);
  case 9: return launch_fused<2, false, true>(d_descs, w, count, M_max, hw_max, st);
  case 10: return launch_fused<2, true, false>(d_descs, w, count, M_max, hw_max, st);
  case 11: return launch_fused<2, true, true>(d_descs, w, count, M_max, hw_max, st);
  }
  return (int)hipErrorInvalidValue;
}

} // namespace fgmm

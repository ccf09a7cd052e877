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
// which the reference silently relies on.  oracle/fgmm_oracle.c restates the chain with fmaf (fgo_head_params): the GPU tests compare
// bit for bit.
//
// Tiling (wave64, 256 threads = 4 waves, one block per CU - the kernel holds 192 accumulator registers per lane):
//   block  = 16 latent channels (their 3*K = 12 parameters each: 192 rows of W) x 256 positions
//   wave   = all 192 rows x 64 positions = 6 row tiles x 2 position tiles of 32x32 accumulators
//   K loop = tiles of 32 input channels staged through LDS, two buffers (W tile 192 x 32 from a PRE-PACKED copy of the weights: one
//            contiguous 24 KB read per block and tile; x tile 32 x 256): the global loads of tile i + 1 are in flight and its LDS writes
//            issue while tile i is multiplied, one barrier per tile; the operand fragments are read from LDS one group of four k-pairs
//            ahead of the products that use them
//   rows   : row tile (g, t), g = which 8 of the 16 channels, t = scales | means | logits; row 4 * cl + k within it = parameter
//            (t, k) of channel 8 g + cl.  The 32x32 accumulator map (row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), column = lane & 31)
//            then gives every lane, for ITS position, registers 4 j + k = parameter (t, k) of channel 8 g + 2 j + (lane >> 5): a lane
//            owns all twelve parameters of its latents - the epilogue needs no lane movement and no LDS.
//   blocks that share an x tile (the channel groups of one position tile) get consecutive slots on ONE XCD: its L2 holds the tile.
#include "fgmm_dev.h"

namespace fgmm {
namespace {

constexpr int kCG = kHeadCG;          // latent channels per block
constexpr int kRows = 12 * kCG;       // rows of W per block (192)
constexpr int kBK = kHeadBK;          // input channels per LDS tile
constexpr int kPB = 256;              // positions per block
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
constexpr int kStage = kRows * kALd + kBK * kBLd;                   // floats of one staged (W tile, x tile) pair
constexpr size_t kHeadLds = 2 * sizeof(float) * (size_t)kStage;     // two of them: 129 024 bytes of the CU's 160 KB
template <int MODE, bool CLAMPED, bool FUSED, bool VEC>
__global__ __launch_bounds__(256, 1) void head_kernel(const EncDesc *__restrict__ edescs, const HeadDesc *__restrict__ hdescs, HeadW hw_, int pt_max,
                                                       int cg_max, int total) {
  extern __shared__ __attribute__((aligned(16))) float s_stage[];
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
    if (tid < kCG) {
      const int c = cg * kCG + tid;
      int r = -1;
      if (c < M && d.chan_nz[c]) {
        r = 0;
        for (int cc = 0; cc < c; ++cc) r += d.chan_nz[cc] != 0;
      }
      s_rank[tid] = r;
    }
    __syncthreads();
    bool any = false;
#pragma unroll
    for (int i = 0; i < kCG; ++i) any = any || s_rank[i] >= 0;
    if (!any) return; // sixteen channels that round to zero everywhere: nothing to code, nothing to multiply
  }
  // ---- accumulators start at the bias
  f16_t acc[kTiles][2];
  {
    const float *bp = hw_.bp + (int64_t)cg * kRows;
#pragma unroll
    for (int tl = 0; tl < kTiles; ++tl)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float b = bp[tl * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
        acc[tl][0][r] = b;
        acc[tl][1][r] = b;
      }
  }
  // ---- K loop
  const int n_kt = hw_.n_kt, c_in = hw_.c_in;
  const float *wp = hw_.wp + (int64_t)cg * n_kt * (kRows * kBK);
  // Staging, branch-free (the K loop is ONE basic block, so that the instruction order below is the order issued): a load that would
  // fall outside the features (input channels past c_in in the last tile, positions past hw in the last position tile) reads a valid
  // address instead and is replaced by zero; the packed weights are zero there as well.
  float4_t ra[6], rb[8];
  unsigned rb_ok = 0; // bit 4 j + e: element e of rb[j] lies inside the features (applied when the tile is WRITTEN: a select right after
                      // the load would wait for it, and the loads are there to be in flight under a tile's products)
  auto load_tile = [&](int kt) {
    rb_ok = 0;
    const float4_t *src = reinterpret_cast<const float4_t *>(wp + (int64_t)kt * (kRows * kBK));
#pragma unroll
    for (int j = 0; j < 6; ++j) ra[j] = src[tid + 256 * j];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = tid + 256 * j, kk = f >> 6, p4 = f & 63;
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
  auto store_tile = [&](float *sA, float *sB) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int f = tid + 256 * j;
      *reinterpret_cast<float4_t *>(&sA[(f >> 3) * kALd + (f & 7) * 4]) = ra[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = tid + 256 * j;
      float4_t v = rb[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (rb_ok >> (4 * j + e)) & 1u ? v[e] : 0.0f;
      *reinterpret_cast<float4_t *>(&sB[(f >> 6) * kBLd + (f & 63) * 4]) = v;
    }
  };
  // the fragments of four k-pairs (one 16-byte read per row tile, the two position tiles' values of a k-pair in one ds_read2): read
  // one group AHEAD of the products that use them, so that the matrix pipe never waits for the LDS (one wave per SIMD: nobody else
  // would fill the gap)
  struct Frag {
    float4_t a[kTiles];
    float b[4][2];
  };
  auto read_frag = [&](Frag &f, const float *sA, const float *sB, int s4) {
    const float *a_base = sA + col * kALd + h * 16 + s4 * 4;
    const float *b_base = sB + h * kBLd + wave * 64 + col + 8 * s4 * kBLd;
#pragma unroll
    for (int tl = 0; tl < kTiles; ++tl) f.a[tl] = *reinterpret_cast<const float4_t *>(a_base + tl * 32 * kALd);
#pragma unroll
    for (int e = 0; e < 4; ++e) f.b[e][0] = b_base[2 * e * kBLd], f.b[e][1] = b_base[2 * e * kBLd + 32];
  };
  auto multiply = [&](const Frag &f) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int tl = 0; tl < kTiles; ++tl) {
        acc[tl][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[tl][e], f.b[e][0], acc[tl][0], 0, 0, 0);
        acc[tl][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[tl][e], f.b[e][1], acc[tl][1], 0, 0, 0);
      }
  };
  // __builtin_amdgcn_sched_barrier(0): nothing is scheduled across it.  Left to itself the scheduler sinks every LDS read to just before
  // its first use and the global loads to just before the LDS writes - each then waited for with the matrix pipe idle (measured: 0.62
  // of the MFMA roof; the bursts below cost a few issue slots per 48 products instead).
#define FGMM_FENCE() __builtin_amdgcn_sched_barrier(0)
  load_tile(0);
  store_tile(s_stage, s_stage + kRows * kALd);
  __syncthreads();
  Frag f0, f1;
  read_frag(f0, s_stage, s_stage + kRows * kALd, 0);
  for (int kt = 0; kt < n_kt; ++kt) {
    const float *sA = s_stage + (kt & 1) * kStage, *sB = sA + kRows * kALd;
    float *nA = s_stage + ((kt + 1) & 1) * kStage, *nB = nA + kRows * kALd;
    // 192 products per wave and tile in four groups of 48; before each group, what a LATER step needs is issued:
    //   group 0: the global loads of the next tile (the last tile loads itself again: harmless) and the fragments of group 1
    //   group 1, 2: the fragments of groups 2, 3
    //   group 3: the next tile's LDS writes - into the OTHER buffer, which was last read a tile ago, a barrier behind us
    FGMM_FENCE();
    load_tile(kt + 1 < n_kt ? kt + 1 : kt);
    read_frag(f1, sA, sB, 1);
    FGMM_FENCE();
    multiply(f0);
    FGMM_FENCE();
    read_frag(f0, sA, sB, 2);
    FGMM_FENCE();
    multiply(f1);
    FGMM_FENCE();
    read_frag(f1, sA, sB, 3);
    FGMM_FENCE();
    multiply(f0);
    FGMM_FENCE();
    store_tile(nA, nB);
    FGMM_FENCE();
    multiply(f1);
    FGMM_FENCE();
    __syncthreads();
    read_frag(f0, nA, nB, 0); // (after the last tile: a read of the other buffer that nobody uses)
  }
#undef FGMM_FENCE
  // ---- epilogue: the lane's 16 latents (2 channel halves x 4 channels x 2 position tiles), all twelve parameters in registers
  if constexpr (FUSED) {
    const EncDesc &d = edescs[item];
    int nbypass = 0;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cl = g * 8 + 2 * j + h;
        const int rank = s_rank[cl];
        const int c = cg * kCG + cl;
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
            const float vq = __builtin_rintf(ldg<float>(d.y + (int64_t)c * hw + p));
            stg<uint32_t>(row_out + p, sym_entry<MODE, CLAMPED>(vq, (int)vq, mu, sg, pi, bp));
          }
          nbypass += __popcll(__ballot(bp));
        }
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

template <typename K>
int head_lds(K kernel) { // (more than the 64 KB a kernel gets without asking)
  return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kHeadLds);
}

template <int MODE, bool CLAMPED, bool VEC>
int launch_fused(const EncDesc *descs, const HeadW &w, int count, int M_max, int64_t hw_max, hipStream_t st) {
  const int pt_max = (int)((hw_max + kPB - 1) / kPB), cg_max = (M_max + kCG - 1) / kCG;
  const int64_t total = (int64_t)count * pt_max * cg_max;
  if (total <= 0) return 0;
  if (total > (1ll << 30)) return (int)hipErrorInvalidValue;
  const unsigned grid = (unsigned)((total + 7) / 8 * 8);
  auto kernel = head_kernel<MODE, CLAMPED, true, VEC>;
  if (int e = head_lds(kernel)) return e;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), kHeadLds, st, descs, (const HeadDesc *)nullptr, w, pt_max, cg_max, (int)total);
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
  if (int e = head_lds(kernel)) return e;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), kHeadLds, (hipStream_t)stream, (const EncDesc *)nullptr, d_descs, w, pt_max, cg_max, (int)total);
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

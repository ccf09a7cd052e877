// mfma_f32_rate.hip — what v_mfma_f32_32x32x2_f32 sustains in the SHAPE the parameter-head kernel uses it (256 threads = one wave per SIMD,
// twelve 32x32 accumulators per wave, 192 products per K tile), with nothing else in the loop, and with the head kernel's LDS reads.
//   hipcc -O3 --offload-arch=gfx950 scripts/mfma_f32_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16_t __attribute__((ext_vector_type(16)));
typedef float f4_t __attribute__((ext_vector_type(4)));

template <int LDS>
__global__ __launch_bounds__(256, 1) void k(float *out, int iters) {
  __shared__ __attribute__((aligned(16))) float s[192 * 36 + 32 * 288];
  for (int i = threadIdx.x; i < 192 * 36 + 32 * 288; i += 256) s[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, col = lane & 31;
  f16_t acc[6][2];
  for (int t = 0; t < 6; ++t)
    for (int r = 0; r < 16; ++r) acc[t][0][r] = acc[t][1][r] = 0.0f;
  float a = 1.0f + lane * 1e-3f, b = 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      f4_t av[6];
      float bv[4][2];
      if (LDS) {
#pragma unroll
        for (int t = 0; t < 6; ++t) av[t] = *reinterpret_cast<const f4_t *>(&s[(t * 32 + col) * 36 + h * 16 + s4 * 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e][0] = s[192 * 36 + (8 * s4 + 2 * e + h) * 288 + wave * 64 + col], bv[e][1] = s[192 * 36 + (8 * s4 + 2 * e + h) * 288 + wave * 64 + col + 32];
      } else {
#pragma unroll
        for (int t = 0; t < 6; ++t) av[t] = (f4_t){a, a, a, a};
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e][0] = bv[e][1] = b;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < 6; ++t) {
          acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][e], bv[e][0], acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][e], bv[e][1], acc[t][1], 0, 0, 0);
        }
    }
  }
  float sum = 0;
  for (int t = 0; t < 6; ++t)
    for (int r = 0; r < 16; ++r) sum += acc[t][0][r] + acc[t][1][r];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}

int main() {
  float *out;
  hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int lds = 0; lds < 2; ++lds)
    for (int blocks : {256, 1728, 2048}) {
      const int iters = 200; // K tiles of 32 (the head: 20)
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (lds) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * 32 * 32 * 2 * 192.0 * iters * 4 * blocks;
        if (rep == 2) printf("lds_reads=%d blocks=%d iters=%d: %.3f ms  %.1f TFLOP/s (%.3f of 157.3)\n", lds, blocks, iters, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
      }
    }
  return 0;
}

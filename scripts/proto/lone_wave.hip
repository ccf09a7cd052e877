// Dev aid: what one wave alone on a SIMD pays per instruction (s_memtime ticks and wall time), for chains like segdec's phase C.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lone_wave scripts/proto/lone_wave.hip && /tmp/lone_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin(unsigned long long ticks, unsigned long long *out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long t;
  do { t = __builtin_amdgcn_s_memtime(); } while (t - t0 < ticks);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t - t0;
}
template <int KIND> __global__ void chain(int n, unsigned long long *out, unsigned *sink) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)(i * 7);
  __syncthreads();
  unsigned x = __builtin_amdgcn_readfirstlane(n * 2654435761u + blockIdx.x), acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
    if (KIND == 0) { // 16 dependent SALU
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("s_add_u32 %0, %0, 0x1234567\n\ts_xor_b32 %0, %0, 0x3333" : "+s"(x) : : "scc");
    } else if (KIND == 1) { // 32 dependent VALU
      unsigned v = x + threadIdx.x;
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("v_add_u32 %0, 0x1234567, %0\n\tv_xor_b32 %0, 0x3333, %0" : "+v"(v));
      acc += v;
    } else if (KIND == 2) { // LDS read -> ballot -> popcount -> readlane -> scalar dependent (the search)
      const unsigned cf = x & 0xFFFFu;
      const unsigned E = lds[(x >> 4 & 0xFFF) % 4000 + threadIdx.x > 4095 ? 0 : (x >> 4 & 0xFFF) % 4000 + threadIdx.x];
      const unsigned nle = __popcll(__ballot(E <= cf));
      const unsigned s = __builtin_amdgcn_readlane(E, nle & 63);
      x = x * 1664525u + s + 1013904223u;
    } else if (KIND == 3) { // taken branches: 8 per iteration
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("s_cmp_eq_u32 %0, %0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:\n\ts_add_u32 %0, %0, 1" : "+s"(x) : : "scc");
    } else if (KIND == 4) { // readlane -> salu -> v_mov -> readlane ... (VALU <-> SALU ping-pong), 8 round trips
      unsigned v = x + threadIdx.x;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        unsigned s = __builtin_amdgcn_readlane(v, 5);
        asm volatile("s_add_u32 %0, %0, 77" : "+s"(s) : : "scc");
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(v) : "s"(s));
      }
      acc += v;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; sink[blockIdx.x] = x + acc; }
}
int main() {
  unsigned long long *d_out; unsigned *d_sink;
  hipMalloc(&d_out, 8 * 65536); hipMalloc(&d_sink, 4 * 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 100000000ull, d_out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; hipMemcpy(&t, d_out, 8, hipMemcpyDeviceToHost);
    printf("spin: %llu ticks in %.3f ms -> s_memtime at %.1f MHz\n", t, ms, t / ms / 1e3);
  }
  const char *names[] = {"32 dependent SALU", "32 dependent VALU", "lds->ballot->bcnt->readlane->salu (search chain, ~12 instr)", "8 x (cmp, taken branch, add)", "8 x (readlane, salu, valu)"};
  const int per_iter[] = {32, 32, 1, 8, 8};
  for (int blocks : {1, 256, 1024, 4096, 16384}) {
    printf("blocks (waves) %d:\n", blocks);
    for (int kind = 0; kind < 5; ++kind) {
      const int n = 20000;
      float best = 1e9; unsigned long long t = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        switch (kind) {
        case 0: hipLaunchKernelGGL(chain<0>, dim3(blocks), dim3(64), 0, 0, n, d_out, d_sink); break;
        case 1: hipLaunchKernelGGL(chain<1>, dim3(blocks), dim3(64), 0, 0, n, d_out, d_sink); break;
        case 2: hipLaunchKernelGGL(chain<2>, dim3(blocks), dim3(64), 0, 0, n, d_out, d_sink); break;
        case 3: hipLaunchKernelGGL(chain<3>, dim3(blocks), dim3(64), 0, 0, n, d_out, d_sink); break;
        default: hipLaunchKernelGGL(chain<4>, dim3(blocks), dim3(64), 0, 0, n, d_out, d_sink); break;
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; hipMemcpy(&t, d_out, 8, hipMemcpyDeviceToHost); }
      }
      printf("  %-62s %8.3f ms  %7.1f ticks / %6.1f ns per unit (wave 0)\n", names[kind], best, (double)t / n / per_iter[kind], best * 1e6 / n / per_iter[kind]);
    }
  }
  return 0;
}

// Dev aid: VALU issue rate of gfx950 per instruction kind and waves per SIMD (is a wave64 v_fma_f32 2 or 4 cycles?).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/bin/valu_peak scripts/valu_peak.hip && scripts/bin/valu_peak
// Each lane runs CH independent dependency chains of one instruction kind for ITER iterations; the grid is
// 256 CUs x (waves per SIMD) x 4 SIMDs waves, so every SIMD holds exactly `wps` waves.  Reported: cycles per wave-
// instruction per SIMD at the clock the run sustained (s_memtime / wall), and lane-ops/s chip-wide.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int CH = 8;
constexpr int ITER = 4096;

template <int KIND> __device__ __forceinline__ float op(float a, float b, float c) {
  if constexpr (KIND == 0) return __builtin_fmaf(a, b, c);                 // v_fma_f32
  else if constexpr (KIND == 1) return a * b;                              // v_mul_f32
  else if constexpr (KIND == 2) return __builtin_amdgcn_rcpf(a);           // v_rcp_f32
  else if constexpr (KIND == 3) return __builtin_amdgcn_rsqf(a);           // v_rsq_f32
  else if constexpr (KIND == 4) return __builtin_floorf(a);                // v_floor_f32
  else if constexpr (KIND == 5) return __builtin_ldexpf(a, (int)b);        // v_cvt + v_ldexp
  else if constexpr (KIND == 6) return (float)(int)a;                      // v_cvt_i32_f32 + v_cvt_f32_i32
  else if constexpr (KIND == 7) return (a < b) ? a : c;                    // v_cmp + v_cndmask
  else if constexpr (KIND == 8) return __uint_as_float((__float_as_uint(a) & 0x80000000u) | (__float_as_uint(b) & 0x7fffffffu)); // v_bfi
  else return a + b;                                                       // v_add_f32
}

template <int KIND> __global__ __launch_bounds__(256) void k(float *out, float s0, float s1, unsigned long long *clk) {
  float v[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) v[i] = s0 + (float)(threadIdx.x + i) * 1e-3f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) v[i] = op<KIND>(v[i], s1, s0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0;
#pragma unroll
  for (int i = 0; i < CH; ++i) acc += v[i];
  if (acc == 12345.678f) out[0] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int KIND> void run(const char *name, int instr_per_op) {
  float *out;
  unsigned long long *clk;
  CK(hipMalloc(&out, 64));
  CK(hipMalloc(&clk, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%-22s", name);
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = 256 * wps; // 256-thread blocks = 4 waves = one per SIMD of a CU
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.999f, clk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.999f, clk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c;
    CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    const double winstr_per_simd = (double)wps * CH * ITER * instr_per_op;
    // s_memtime ticks at 100 MHz-derived constant rate on gfx9?  report wall-based numbers at an assumed 2.4 GHz too
    printf("  wps%d: %6.3f ms  %5.2f cyc/instr@2.4GHz  (%5.1f T lane-op/s)", wps, ms, ms * 1e-3 * 2.4e9 / winstr_per_simd,
           256.0 * 4 * winstr_per_simd * 64 / (ms * 1e-3) / 1e12);
    (void)c;
  }
  printf("\n");
  CK(hipFree(out));
  CK(hipFree(clk));
}

int main() {
  run<0>("v_fma_f32", 1);
  run<1>("v_mul_f32", 1);
  run<9>("v_add_f32", 1);
  run<2>("v_rcp_f32", 1);
  run<3>("v_rsq_f32", 1);
  run<4>("v_floor_f32", 1);
  run<5>("cvt_i32+ldexp", 1);
  run<6>("cvt_i32+cvt_f32", 2);
  run<7>("cmp+cndmask", 2);
  run<8>("v_bfi/and_or", 1);
  return 0;
}

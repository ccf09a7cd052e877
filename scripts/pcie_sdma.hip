// Dev aid: does hipMemcpyAsync D2H keep using the SDMA engine when the copies are chained to kernels of another
// stream through events (wait before, record after)?  A spin kernel keeps every CU busy: a copy that runs as a blit
// kernel cannot start until it ends, an SDMA copy is unaffected.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin_kernel(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; ++i) a = a * b + 0.5f;
  if (a == 123.f) out[0] = a;
}
__global__ void tiny_kernel(float* out) { if (out[1] == 123.f) out[2] = 1.f; }
__global__ void copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t bytes = (size_t)400 << 20;
  void *d, *h; float* dout;
  CK(hipMalloc(&d, bytes)); CK(hipMalloc(&dout, 64)); CK(hipMemset(dout, 0, 64));
  CK(hipHostMalloc(&h, bytes, hipHostMallocDefault)); memset(h, 0, bytes);
  hipStream_t sc, sk, sb; CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  const int P = 9;
  hipEvent_t evk[P], evc[P];
  for (int i = 0; i < P; ++i) { CK(hipEventCreateWithFlags(&evk[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&evc[i], hipEventDisableTiming)); }
  for (int variant = 0; variant < 4; ++variant)
    for (int busy = 0; busy < 2; ++busy)
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        if (busy) spin_kernel<<<4096, 256, 0, sb>>>(dout, 3000000);
        double t0 = now();
        size_t per = bytes / P;
        for (int p = 0; p < P; ++p) {
          if (variant >= 1) { tiny_kernel<<<1, 64, 0, sk>>>(dout); CK(hipEventRecord(evk[p], sk)); CK(hipStreamWaitEvent(sc, evk[p], 0)); }
          if (variant == 3) copy_kernel<<<64, 256, 0, sc>>>((uint4*)((char*)h + p * per), (const uint4*)((char*)d + p * per), per / 16);
          else CK(hipMemcpyAsync((char*)h + p * per, (char*)d + p * per, per, hipMemcpyDeviceToHost, sc));
          if (variant >= 2) CK(hipEventRecord(evc[p], sc));
        }
        CK(hipStreamSynchronize(sc));
        double t1 = now();
        if (rep == 1) printf("variant %d (%s) busy=%d: %.2f ms  %.1f GB/s\n", variant,
               variant == 0 ? "plain copies" : variant == 1 ? "wait-event before each copy" : variant == 2 ? "wait before + record after" : "own 64-block copy kernel, wait + record", busy, 1e3 * (t1 - t0), bytes / (t1 - t0) * 1e-9);
        CK(hipDeviceSynchronize());
      }
  return 0;
}

// Dev aid: how fast do SHADER STORES reach pinned host memory, by store width and by how much compute sits between them?
//   hipcc --offload-arch=gfx950 -O3 -o scripts/bin/pcie_store scripts/pcie_store.hip && scripts/bin/pcie_store
// The decode-side table kernel could store its rows straight across PCIe ("tab_direct") instead of into a device staging
// area that a copy fetches; its stores are 4 bytes per lane.  Reference point: hipMemcpyAsync device -> pinned.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// every block stores `chunk` contiguous bytes at a cursor-like scattered position (block b -> slot perm(b)), W bytes per lane
template <int W> __global__ __launch_bounds__(256) void store_k(uint8_t *dst, size_t chunk, unsigned nblk, int spin, float *sink) {
  const unsigned b = blockIdx.x;
  const unsigned slot = (unsigned)(((unsigned long long)b * 2654435761ull) % nblk); // scattered like the cursor's placement
  uint8_t *out = dst + (size_t)slot * chunk;
  float acc = (float)threadIdx.x;
  for (int i = 0; i < spin; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 0.5f); // compute between a block's start and its stores
  const uint32_t v = __float_as_uint(acc) | b;
  if constexpr (W == 2) {
    for (size_t o = (size_t)threadIdx.x * 2; o < chunk; o += 256 * 2) *reinterpret_cast<uint16_t *>(out + o) = (uint16_t)v;
  } else if constexpr (W == 4) {
    for (size_t o = (size_t)threadIdx.x * 4; o < chunk; o += 256 * 4) *reinterpret_cast<uint32_t *>(out + o) = v;
  } else {
    for (size_t o = (size_t)threadIdx.x * 16; o < chunk; o += 256 * 16) *reinterpret_cast<u4 *>(out + o) = (u4){v, v, v, v};
  }
  if (acc == 12345.678f) *sink = acc;
}

int main() {
  const size_t total = 179ull << 20, chunk = 6144; // one decode call's tables; a block's rows
  const unsigned nblk = (unsigned)(total / chunk);
  uint8_t *h = nullptr, *d = nullptr;
  float *sink = nullptr;
  CK(hipHostMalloc((void **)&h, total, hipHostMallocDefault));
  CK(hipMalloc((void **)&d, total));
  CK(hipMalloc((void **)&sink, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timeit = [&](const char *what, auto &&fn) {
    fn();
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(e0, nullptr));
      fn();
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    printf("%-64s %7.3f ms  %5.1f GB/s\n", what, best, (double)total / best / 1e6);
  };
  timeit("hipMemcpyAsync device -> pinned", [&] { CK(hipMemcpyAsync(h, d, total, hipMemcpyDeviceToHost, nullptr)); });
  for (int spin : {0, 2000}) {
    char name[128];
    snprintf(name, sizeof name, "kernel -> pinned,  4 B/lane, %5d FMAs before the stores", spin);
    timeit(name, [&] { hipLaunchKernelGGL(store_k<4>, dim3(nblk), dim3(256), 0, nullptr, h, chunk, nblk, spin, sink); });
    snprintf(name, sizeof name, "kernel -> pinned, 16 B/lane, %5d FMAs before the stores", spin);
    timeit(name, [&] { hipLaunchKernelGGL(store_k<16>, dim3(nblk), dim3(256), 0, nullptr, h, chunk, nblk, spin, sink); });
    snprintf(name, sizeof name, "kernel -> pinned,  2 B/lane, %5d FMAs before the stores", spin);
    timeit(name, [&] { hipLaunchKernelGGL(store_k<2>, dim3(nblk), dim3(256), 0, nullptr, h, chunk, nblk, spin, sink); });
    snprintf(name, sizeof name, "kernel -> HBM,     4 B/lane, %5d FMAs before the stores", spin);
    timeit(name, [&] { hipLaunchKernelGGL(store_k<4>, dim3(nblk), dim3(256), 0, nullptr, d, chunk, nblk, spin, sink); });
  }
  return 0;
}

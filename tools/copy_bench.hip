// Micro-benchmark used to choose the shape of the local reduce-copy loop (threads per
// workgroup, packs per thread, grid, nontemporal or plain accesses) on MI355X.
//   hipcc --offload-arch=gfx950 -O3 -o tools/copy_bench tools/copy_bench.hip && ./tools/copy_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT, bool LOADS_FIRST>
__global__ void copyk(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t npacks) {
  const size_t tile = (size_t)blockDim.x * U;
  for (size_t base = (size_t)blockIdx.x * tile; base + tile <= npacks; base += (size_t)gridDim.x * tile) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const u32x4* p = s + base + threadIdx.x + (size_t)u * blockDim.x;
      v[u] = NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      u32x4* p = d + base + threadIdx.x + (size_t)u * blockDim.x;
      if (NT) __builtin_nontemporal_store(v[u], p); else *p = v[u];
    }
  }
}
// contiguous slab per workgroup instead of grid-stride interleave
template <int U, bool NT>
__global__ void copyslab(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t npacks) {
  const size_t per = npacks / gridDim.x;
  const size_t lo = per * blockIdx.x, hi = lo + per;
  const size_t tile = (size_t)blockDim.x * U;
  for (size_t base = lo; base + tile <= hi; base += tile) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const u32x4* p = s + base + threadIdx.x + (size_t)u * blockDim.x;
      v[u] = NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      u32x4* p = d + base + threadIdx.x + (size_t)u * blockDim.x;
      if (NT) __builtin_nontemporal_store(v[u], p); else *p = v[u];
    }
  }
}

template <class F> float timeit(F f, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; i++) f();
  hipEventRecord(a);
  for (int i = 0; i < iters; i++) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / iters;
}

int main() {
  const size_t bytes = 1ull << 30, npacks = bytes / 16;
  u32x4 *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes);
  hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
  const int grids[] = {256, 512, 1024, 2048, 4096, 8192, 16384};
  const int blocks[] = {256, 512, 1024};
#define RUN(NAME, K)                                                                        \
  for (int b : blocks) for (int g : grids) {                                                \
    float ms = timeit([&]() { hipLaunchKernelGGL(K, dim3(g), dim3(b), 0, 0, s, d, npacks); }, 10); \
    printf("%-28s block %4d grid %5d  %.4f ms  %.0f GB/s\n", NAME, b, g, ms, 2.0 * bytes / ms / 1e6); }
  RUN("stride U1 nt", (copyk<1, true, true>))
  RUN("stride U2 nt", (copyk<2, true, true>))
  RUN("stride U4 nt", (copyk<4, true, true>))
  RUN("stride U8 nt", (copyk<8, true, true>))
  RUN("stride U4 plain", (copyk<4, false, true>))
  RUN("stride U8 plain", (copyk<8, false, true>))
  RUN("slab U4 nt", (copyslab<4, true>))
  RUN("slab U8 nt", (copyslab<8, true>))
  RUN("slab U4 plain", (copyslab<4, false>))
  float ms = timeit([&]() { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); }, 10);
  printf("%-28s %.4f ms  %.0f GB/s\n", "hipMemcpyAsync D2D", ms, 2.0 * bytes / ms / 1e6);
  return 0;
}

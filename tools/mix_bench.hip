// Micro-benchmark: what the step structure of the collective kernels costs by itself, without any flags or peers.
// Persistent 1024-thread workgroups each copy steps of STEP bytes (8 packs per thread and tile, like the reduce-copy
// loops); between steps either nothing, or what Ctx::step does (every wave waits for its stores, barrier).  Stores
// plain streaming (nt) or system-scope write-through (sc0 sc1 nt) into ordinary or uncached memory.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mix_bench tools/mix_bench.hip && ./tools/mix_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const u32x4 gc_u32x4;
typedef __attribute__((address_space(1))) u32x4 g_u32x4;

template <bool WT, bool DRAIN>
__global__ __launch_bounds__(1024) void stepcopy(const char* s, char* d, size_t bytes, size_t step) {
  const size_t nsteps = bytes / step;
  const uint32_t lane16 = threadIdx.x * 16u, row = blockDim.x * 16u;
  const size_t tile = (size_t)row * 8;
  for (size_t k = blockIdx.x; k < nsteps; k += gridDim.x) {
    const uint64_t sb = (uint64_t)s + k * step, db = (uint64_t)d + k * step;
    for (size_t b = 0; b + tile <= step; b += tile) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = __builtin_nontemporal_load((gc_u32x4*)(sb + b + lane16 + u * row));
#pragma unroll
      for (int u = 0; u < 8; u++) {
        if (WT) __builtin_amdgcn_raw_buffer_store_b128(v[u], __builtin_amdgcn_make_buffer_rsrc((void*)(db + b), 0, -1, 0x00020000), (int)(lane16 + u * row), 0, 19);
        else __builtin_nontemporal_store(v[u], (g_u32x4*)(db + b + lane16 + u * row));
      }
    }
    if (DRAIN) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
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
  const size_t bytes = 3ull << 30;
  char *s, *d, *du;
  if (hipMalloc(&s, bytes) != hipSuccess || hipMalloc(&d, bytes) != hipSuccess ||
      hipExtMallocWithFlags((void**)&du, bytes, hipDeviceMallocUncached) != hipSuccess) { printf("allocation failed\n"); return 1; }
  hipMemset(s, 1, bytes); hipMemset(d, 0, bytes); hipMemset(du, 0, bytes);
  const int grids[] = {64, 128, 192, 240, 256, 512};
  const size_t steps[] = {768u << 10, 256u << 10};
#define RUN(NAME, K, DST)                                                                                         \
  for (size_t st : steps) for (int g : grids) {                                                                   \
    float ms = timeit([&]() { hipLaunchKernelGGL(K, dim3(g), dim3(1024), 0, 0, s, DST, bytes, st); }, 5);        \
    printf("%-44s step %4zu KiB grid %4d  %.4f ms  %.0f GB/s\n", NAME, st >> 10, g, ms, 2.0 * bytes / ms / 1e6); }
  RUN("nt stores, no drain, ordinary memory", (stepcopy<false, false>), d)
  RUN("nt stores, drain per step, ordinary memory", (stepcopy<false, true>), d)
  RUN("write-through, no drain, ordinary memory", (stepcopy<true, false>), d)
  RUN("write-through, drain per step, ordinary", (stepcopy<true, true>), d)
  RUN("write-through, no drain, uncached memory", (stepcopy<true, false>), du)
  RUN("write-through, drain per step, uncached", (stepcopy<true, true>), du)
  return 0;
}

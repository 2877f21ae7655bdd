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

// The reducer's shape: NS sources folded (fp32 adds) into ND destinations per step, U packs per source and thread and tile,
// the first destination written with streaming stores, the others write-through; drain + barrier after every step.
template <int NS, int ND, int U>
__global__ __launch_bounds__(1024) void stepfold(const char* s, char* d, size_t region, size_t step) {
  const size_t nsteps = region / step;
  const uint32_t lane16 = threadIdx.x * 16u, row = blockDim.x * 16u;
  const size_t tile = (size_t)row * U;
  for (size_t k = blockIdx.x; k < nsteps; k += gridDim.x) {
    for (size_t b = 0; b + tile <= step; b += tile) {
      u32x4 v[NS][U];
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int j = 0; j < NS; j++)
          v[j][u] = __builtin_nontemporal_load((gc_u32x4*)((uint64_t)s + j * region + k * step + b + lane16 + u * row));
#pragma unroll
      for (int u = 0; u < U; u++) {
        float4 acc = __builtin_bit_cast(float4, v[0][u]);
#pragma unroll
        for (int j = 1; j < NS; j++) { const float4 x = __builtin_bit_cast(float4, v[j][u]); acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w; }
        v[0][u] = __builtin_bit_cast(u32x4, acc);
      }
#pragma unroll
      for (int q = 0; q < ND; q++)
#pragma unroll
        for (int u = 0; u < U; u++) {
          const uint64_t db = (uint64_t)d + q * region + k * step + b;
          if (q) __builtin_amdgcn_raw_buffer_store_b128(v[0][u], __builtin_amdgcn_make_buffer_rsrc((void*)db, 0, -1, 0x00020000), (int)(lane16 + u * row), 0, 19);
          else __builtin_nontemporal_store(v[0][u], (g_u32x4*)(db + lane16 + u * row));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

// The same fold, one source after the other: a tile is U packs per thread of ONE source (U x 16 KiB contiguous), added into
// an accumulator tile while the next source's tile is in flight; stores when all sources are in.  Same association order.
template <int NS, int ND, int U>
__global__ __launch_bounds__(1024) void stepfoldSerial(const char* s, char* d, size_t region, size_t step) {
  const size_t nsteps = region / step;
  const uint32_t lane16 = threadIdx.x * 16u, row = blockDim.x * 16u;
  const size_t tile = (size_t)row * U;
  for (size_t k = blockIdx.x; k < nsteps; k += gridDim.x) {
    for (size_t b = 0; b + tile <= step; b += tile) {
      u32x4 a[U], x[U], y[U];
      auto load = [&](u32x4 (&v)[U], int j) {
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load((gc_u32x4*)((uint64_t)s + j * region + k * step + b + lane16 + u * row));
      };
      auto add = [&](u32x4 (&v)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
          float4 p = __builtin_bit_cast(float4, a[u]); const float4 q = __builtin_bit_cast(float4, v[u]);
          p.x += q.x; p.y += q.y; p.z += q.z; p.w += q.w; a[u] = __builtin_bit_cast(u32x4, p);
        }
      };
      load(a, 0);
      if (NS > 1) load(x, 1);
#pragma unroll
      for (int j = 1; j < NS; j += 2) {
        if (j + 1 < NS) load(y, j + 1);
        add(x);
        if (j + 1 < NS) { if (j + 2 < NS) load(x, j + 2); add(y); }
      }
#pragma unroll
      for (int q = 0; q < ND; q++)
#pragma unroll
        for (int u = 0; u < U; u++) {
          const uint64_t db = (uint64_t)d + q * region + k * step + b;
          if (q) __builtin_amdgcn_raw_buffer_store_b128(a[u], __builtin_amdgcn_make_buffer_rsrc((void*)db, 0, -1, 0x00020000), (int)(lane16 + u * row), 0, 19);
          else __builtin_nontemporal_store(a[u], (g_u32x4*)(db + lane16 + u * row));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
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
  // reducer shapes at 2, 4 and 8 ranks: sources x 384 MiB read, destinations x 384 MiB written (s and d hold 3 GiB each)
  const size_t region = 384ull << 20;
#define RUNF(NAME, K, NS, ND)                                                                                    \
  for (int g : {128, 192, 240, 256}) {                                                                           \
    float ms = timeit([&]() { hipLaunchKernelGGL(K, dim3(g), dim3(1024), 0, 0, s, du, region, (size_t)(768u << 10)); }, 5); \
    printf("%-44s step  768 KiB grid %4d  %.4f ms  %.0f GB/s\n", NAME, g, ms, (double)(NS + ND) * region / ms / 1e6); }
  RUNF("fold 2 sources -> 2 destinations, U=4", (stepfold<2, 2, 4>), 2, 2)
  RUNF("fold 4 sources -> 4 destinations, U=2", (stepfold<4, 4, 2>), 4, 4)
  RUNF("fold 8 sources -> 8 destinations, U=1", (stepfold<8, 8, 1>), 8, 8)
  RUNF("fold 8 sources -> 1 destination, U=1", (stepfold<8, 1, 1>), 8, 1)
  RUNF("copy 1 source -> 8 destinations, U=8", (stepfold<1, 8, 8>), 1, 8)
  RUNF("serial fold 8 -> 8, tiles of 8 packs", (stepfoldSerial<8, 8, 8>), 8, 8)
  RUNF("serial fold 8 -> 8, tiles of 4 packs", (stepfoldSerial<8, 8, 4>), 8, 8)
  RUNF("serial fold 8 -> 1, tiles of 8 packs", (stepfoldSerial<8, 1, 8>), 8, 1)
  RUNF("serial fold 4 -> 4, tiles of 8 packs", (stepfoldSerial<4, 4, 8>), 4, 4)
  return 0;
}

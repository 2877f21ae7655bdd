// One translation unit per element type: compiled twelve times with -DRA_DTYPE=0..11 (the
// ncclDataType_t value) so that the builds run in parallel.  Holds the collective kernel
// and the local reduce-copy kernel for every reduction op of that type.
#include "kernels_impl.h"

#ifndef RA_DTYPE
#error "compile with -DRA_DTYPE=<ncclDataType_t value 0..11>"
#endif

#if RA_DTYPE == 0
typedef int8_t elem_t;
#elif RA_DTYPE == 1
typedef uint8_t elem_t;
#elif RA_DTYPE == 2
typedef int32_t elem_t;
#elif RA_DTYPE == 3
typedef uint32_t elem_t;
#elif RA_DTYPE == 4
typedef int64_t elem_t;
#elif RA_DTYPE == 5
typedef uint64_t elem_t;
#elif RA_DTYPE == 6
typedef f16_t elem_t;
#elif RA_DTYPE == 7
typedef float elem_t;
#elif RA_DTYPE == 8
typedef double elem_t;
#elif RA_DTYPE == 9
typedef bf16_t elem_t;
#elif RA_DTYPE == 10
typedef f8e4m3_t elem_t;
#elif RA_DTYPE == 11
typedef f8e5m2_t elem_t;
#endif
#define RA_IS_INT (RA_DTYPE <= 5)

#define RA_CAT2(a, b) a##b
#define RA_CAT(a, b) RA_CAT2(a, b)

template <int OP>
static hipError_t collOp(const RaDevComm* comm, const RaWork& w, int grid, int block, hipStream_t s) {
  if (w.oneShot == 2) hipLaunchKernelGGL((raTwoExchangeKernel<elem_t, OP>), dim3(grid), dim3(1024), 0, s, comm, w);
  else if (w.oneShot) hipLaunchKernelGGL((raOneShotKernel<elem_t, OP>), dim3(grid), dim3(1024), 0, s, comm, w);
  else if (w.algo == RA_ALGO_DIRECT && w.coll == RA_COLL_ALLREDUCE)
    hipLaunchKernelGGL((raCollKernel<elem_t, OP, RA_K_DIRECT_ALLREDUCE>), dim3(grid), dim3(block), 0, s, comm, w);
  else if (w.algo == RA_ALGO_DIRECT && w.coll == RA_COLL_REDUCESCATTER)
    hipLaunchKernelGGL((raCollKernel<elem_t, OP, RA_K_DIRECT_REDUCESCATTER>), dim3(grid), dim3(block), 0, s, comm, w);
  else if (w.algo == RA_ALGO_DIRECT && w.coll == RA_COLL_REDUCE)
    hipLaunchKernelGGL((raCollKernel<elem_t, OP, RA_K_DIRECT_REDUCE>), dim3(grid), dim3(block), 0, s, comm, w);
  else if (w.algo == RA_ALGO_RING && w.coll == RA_COLL_ALLREDUCE)
    hipLaunchKernelGGL((raCollKernel<elem_t, OP, RA_K_RING_ALLREDUCE>), dim3(grid), dim3(block), 0, s, comm, w);
  else if (w.algo == RA_ALGO_RING && w.coll == RA_COLL_REDUCESCATTER)
    hipLaunchKernelGGL((raCollKernel<elem_t, OP, RA_K_RING_REDUCESCATTER>), dim3(grid), dim3(block), 0, s, comm, w);
  else if (w.algo == RA_ALGO_TREE && w.coll == RA_COLL_ALLREDUCE)
    hipLaunchKernelGGL((raCollKernel<elem_t, OP, RA_K_CHAIN_ALLREDUCE>), dim3(grid), dim3(block), 0, s, comm, w);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
template <int OP>
static hipError_t llOp(const RaDevComm* comm, const RaWork& w, hipStream_t s) {
  if (w.algo == RA_ALGO_RING) hipLaunchKernelGGL((raLLRingKernel<elem_t, OP>), dim3(w.nChannels), dim3(RA_LL_THREADS), 0, s, comm, w);
  else if (w.algo == RA_ALGO_TREE) hipLaunchKernelGGL((raLLChainKernel<elem_t, OP>), dim3(w.nChannels), dim3(RA_LL_THREADS), 0, s, comm, w);
  else hipLaunchKernelGGL((raLLAllReduceKernel<elem_t, OP>), dim3(w.nChannels), dim3(RA_LL_THREADS), 0, s, comm, w);
  return hipGetLastError();
}
template <int OP>
static hipError_t noAllocOp(const RaNaArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((raNoAllocAllReduceKernel<elem_t, OP>), dim3(RA_NA_GRID), dim3(RA_LL_THREADS), 0, s, a);
  return hipGetLastError();
}
template <int OP>
static hipError_t localOp(const RcArgs& a, size_t nelem, const void* opArgPtr, int grid, hipStream_t s) {
  hipLaunchKernelGGL((raLocalKernel<elem_t, OP>), dim3(grid), dim3(RA_LOCAL_THREADS), 0, s, a, nelem, opArgPtr);
  return hipGetLastError();
}

hipError_t RA_CAT(raLaunchCollDt, RA_DTYPE)(const RaDevComm* comm, const RaWork& w, int op, int grid, int block,
                                            hipStream_t s) {
  switch (op) {
    case RA_OP_SUM: return collOp<RA_OP_SUM>(comm, w, grid, block, s);
    case RA_OP_PROD: return collOp<RA_OP_PROD>(comm, w, grid, block, s);
    case RA_OP_MAX: return collOp<RA_OP_MAX>(comm, w, grid, block, s);
    case RA_OP_MIN: return collOp<RA_OP_MIN>(comm, w, grid, block, s);
    case RA_OP_PREMULSUM: return collOp<RA_OP_PREMULSUM>(comm, w, grid, block, s);
#if RA_IS_INT
    case RA_OP_SUMPOSTDIV: return collOp<RA_OP_SUMPOSTDIV>(comm, w, grid, block, s);
#endif
    default: return hipErrorInvalidValue;
  }
}

hipError_t RA_CAT(raLaunchLLDt, RA_DTYPE)(const RaDevComm* comm, const RaWork& w, int op, hipStream_t s) {
  switch (op) {
    case RA_OP_SUM: return llOp<RA_OP_SUM>(comm, w, s);
    case RA_OP_PROD: return llOp<RA_OP_PROD>(comm, w, s);
    case RA_OP_MAX: return llOp<RA_OP_MAX>(comm, w, s);
    case RA_OP_MIN: return llOp<RA_OP_MIN>(comm, w, s);
    case RA_OP_PREMULSUM: return llOp<RA_OP_PREMULSUM>(comm, w, s);
#if RA_IS_INT
    case RA_OP_SUMPOSTDIV: return llOp<RA_OP_SUMPOSTDIV>(comm, w, s);
#endif
    default: return hipErrorInvalidValue;
  }
}

hipError_t RA_CAT(raLaunchLocalDt, RA_DTYPE)(const RcArgs& a, size_t nelem, const void* opArgPtr, int op, int grid, hipStream_t s) {
  switch (op) {
    case RA_OP_SUM: return localOp<RA_OP_SUM>(a, nelem, opArgPtr, grid, s);
    case RA_OP_PROD: return localOp<RA_OP_PROD>(a, nelem, opArgPtr, grid, s);
    case RA_OP_MAX: return localOp<RA_OP_MAX>(a, nelem, opArgPtr, grid, s);
    case RA_OP_MIN: return localOp<RA_OP_MIN>(a, nelem, opArgPtr, grid, s);
    case RA_OP_PREMULSUM: return localOp<RA_OP_PREMULSUM>(a, nelem, opArgPtr, grid, s);
#if RA_IS_INT
    case RA_OP_SUMPOSTDIV: return localOp<RA_OP_SUMPOSTDIV>(a, nelem, opArgPtr, grid, s);
#endif
    default: return hipErrorInvalidValue;
  }
}

hipError_t RA_CAT(raLaunchNoAllocDt, RA_DTYPE)(const RaNaArgs& a, int op, hipStream_t s) {
  switch (op) {
    case RA_OP_SUM: return noAllocOp<RA_OP_SUM>(a, s);
    case RA_OP_PROD: return noAllocOp<RA_OP_PROD>(a, s);
    case RA_OP_MAX: return noAllocOp<RA_OP_MAX>(a, s);
    case RA_OP_MIN: return noAllocOp<RA_OP_MIN>(a, s);
    case RA_OP_PREMULSUM: return noAllocOp<RA_OP_PREMULSUM>(a, s);
#if RA_IS_INT
    case RA_OP_SUMPOSTDIV: return noAllocOp<RA_OP_SUMPOSTDIV>(a, s);
#endif
    default: return hipErrorInvalidValue;
  }
}

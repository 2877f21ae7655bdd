// Type-independent kernels (all-gather moves bytes) and the dtype dispatch over the
// per-type translation units built from kernels_dtype.hip.
#include "kernels_impl.h"

__global__ __launch_bounds__(1024) void raGatherKernel(const RaDevComm* __restrict__ comm, RaWork w) {
  __shared__ ChanShared sh;
  Ctx cx;
  const int role = blockIdx.x / w.nChannels;
  cx.init(comm, &sh, blockIdx.x % w.nChannels, w.stepBytes);
  if (w.coll == RA_COLL_BROADCAST) directBroadcast(cx, w, role);
  else if (w.algo == RA_ALGO_DIRECT) directAllGather(cx, w, role);
  else ringAllGather(cx, w);
  cx.finish();
}

#define RA_DECL(N)                                                                                             \
  hipError_t raLaunchCollDt##N(const RaDevComm*, const RaWork&, int op, int grid, int block, hipStream_t s);  \
  hipError_t raLaunchLocalDt##N(const RcArgs&, size_t nelem, const void* opArgPtr, int op, int grid, hipStream_t s);         \
  hipError_t raLaunchLLDt##N(const RaDevComm*, const RaWork&, int op, hipStream_t s);
RA_DECL(0) RA_DECL(1) RA_DECL(2) RA_DECL(3) RA_DECL(4) RA_DECL(5) RA_DECL(6) RA_DECL(7) RA_DECL(8) RA_DECL(9) RA_DECL(10) RA_DECL(11)
#undef RA_DECL

hipError_t raLaunchColl(const RaDevComm* comm, const RaWork& w, int dtype, int op, int grid, int block, hipStream_t s) {
  if (w.coll == RA_COLL_ALLGATHER || w.coll == RA_COLL_BROADCAST) {
    hipLaunchKernelGGL(raGatherKernel, dim3(grid), dim3(block), 0, s, comm, w);
    return hipGetLastError();
  }
  switch (dtype) {
#define RA_CASE(N) case N: return raLaunchCollDt##N(comm, w, op, grid, block, s);
    RA_CASE(0) RA_CASE(1) RA_CASE(2) RA_CASE(3) RA_CASE(4) RA_CASE(5) RA_CASE(6) RA_CASE(7) RA_CASE(8) RA_CASE(9) RA_CASE(10) RA_CASE(11)
#undef RA_CASE
    default: return hipErrorInvalidValue;
  }
}

hipError_t raLaunchLL(const RaDevComm* comm, const RaWork& w, int dtype, int op, hipStream_t s) {
  switch (dtype) {
#define RA_CASE(N) case N: return raLaunchLLDt##N(comm, w, op, s);
    RA_CASE(0) RA_CASE(1) RA_CASE(2) RA_CASE(3) RA_CASE(4) RA_CASE(5) RA_CASE(6) RA_CASE(7) RA_CASE(8) RA_CASE(9) RA_CASE(10) RA_CASE(11)
#undef RA_CASE
    default: return hipErrorInvalidValue;
  }
}

hipError_t raLaunchLocal(void* dst, const void* const* srcs, int nsrc, int preIdx, bool post, uint64_t opArg,
                         const void* opArgPtr, size_t nelem, int dtype, int op, int grid, hipStream_t s) {
  if (nsrc < 1 || nsrc > RA_MAX_SRCS) return hipErrorInvalidValue;
  RcArgs a;
  for (int j = 0; j < nsrc; j++) a.src[j] = (const char*)srcs[j];
  a.dst[0] = (char*)dst;
  a.ns = nsrc; a.nd = 1; a.nLocal = 1; a.preIdx = preIdx; a.post = post; a.opArg = opArg;
  switch (dtype) {
#define RA_CASE(N) case N: return raLaunchLocalDt##N(a, nelem, opArgPtr, op, grid, s);
    RA_CASE(0) RA_CASE(1) RA_CASE(2) RA_CASE(3) RA_CASE(4) RA_CASE(5) RA_CASE(6) RA_CASE(7) RA_CASE(8) RA_CASE(9) RA_CASE(10) RA_CASE(11)
#undef RA_CASE
    default: return hipErrorInvalidValue;
  }
}

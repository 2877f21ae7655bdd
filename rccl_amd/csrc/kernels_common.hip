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
  hipError_t raLaunchLLDt##N(const RaDevComm*, const RaWork&, int op, hipStream_t s);                            \
  hipError_t raLaunchNoAllocDt##N(const RaNaArgs&, int op, hipStream_t s);
RA_DECL(0) RA_DECL(1) RA_DECL(2) RA_DECL(3) RA_DECL(4) RA_DECL(5) RA_DECL(6) RA_DECL(7) RA_DECL(8) RA_DECL(9) RA_DECL(10) RA_DECL(11)
#undef RA_DECL

hipError_t raLaunchColl(const RaDevComm* comm, const RaWork& w, int dtype, int op, int grid, int block, hipStream_t s) {
  if ((w.coll == RA_COLL_ALLGATHER && !w.oneShot) || w.coll == RA_COLL_BROADCAST) {
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

// ------------------------------------------------------------------ the all-reduce that allocates nothing
// Its line buffers: one array per GPU, part of this library's code object, so it exists on every GPU the library
// runs on from the moment the module is loaded until the process ends, zero-filled at first (no line carries a
// valid epoch).  `scratch` names the array for the compiler only; its address on a device comes from
// hipGetSymbolAddress with that device current, once, and only that address is ever handed to a kernel.
static __device__ __align__(256) unsigned char scratch[RA_NA_SCRATCH_BYTES];

#include <atomic>
#include <mutex>

static std::mutex gNaMutex;
static char* gNaLines[64];             // per device: address of its `scratch`, null until set up
static bool gNaPeer[64][64];           // [a][b]: kernels on device a may store into device b's memory
static std::atomic<uint32_t> gNaEpoch{0};

// The address of `dev`'s array (set-up on the first call for a device; the current device is left as found).
static hipError_t naLines(int dev, char** out) {
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> g(gNaMutex);
  if (!gNaLines[dev]) {
    int cur = 0;
    hipError_t e = hipGetDevice(&cur);
    if (e != hipSuccess) return e;
    if ((e = hipSetDevice(dev)) != hipSuccess) return e;
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    hipError_t back = hipSetDevice(cur);
    if (e != hipSuccess) return e;
    if (back != hipSuccess) return back;
    if (!p) return hipErrorInvalidSymbol;
    gNaLines[dev] = (char*)p;
  }
  *out = gNaLines[dev];
  return hipSuccess;
}

hipError_t raNoAllocAllReduce(const void* const* send, void* const* recv, const int* devices, int nRanks, size_t count,
                              int dtype, int op, uint64_t opArg, uint64_t timeoutTicks, const hipStream_t* streams) {
  if (nRanks < 1 || nRanks > RA_NA_MAX_RANKS) return hipErrorInvalidValue;
  RaNaArgs a;
  for (int p = 0; p < RA_NA_MAX_RANKS; p++) a.lines[p] = nullptr;
  for (int p = 0; p < nRanks; p++) {
    hipError_t e = naLines(devices[p], &a.lines[p]);
    if (e != hipSuccess) return e;
  }
  int cur = 0;
  hipError_t e = hipGetDevice(&cur);
  if (e != hipSuccess) return e;
  {
    // once per ordered pair of distinct devices: the peer mapping the line stores need
    std::lock_guard<std::mutex> g(gNaMutex);
    for (int x = 0; x < nRanks; x++) {
      for (int y = 0; y < nRanks; y++) {
        const int dx = devices[x], dy = devices[y];
        if (dx == dy || gNaPeer[dx][dy]) continue;
        if ((e = hipSetDevice(dx)) != hipSuccess) return e;
        e = hipDeviceEnablePeerAccess(dy, 0);
        if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
        if (e != hipSuccess) { (void)hipSetDevice(cur); return e; }
        gNaPeer[dx][dy] = true;
      }
    }
  }
  uint32_t epoch = gNaEpoch.fetch_add(1) + 1;
  if (epoch == 0) epoch = gNaEpoch.fetch_add(1) + 1;
  a.count = count; a.opArg = opArg; a.timeoutTicks = timeoutTicks; a.epoch = epoch; a.nRanks = nRanks;
  for (int r = 0; r < nRanks && e == hipSuccess; r++) {
    a.rank = r; a.send = send[r]; a.recv = recv[r];
    if ((e = hipSetDevice(devices[r])) != hipSuccess) break;
    switch (dtype) {
#define RA_CASE(N) case N: e = raLaunchNoAllocDt##N(a, op, streams[r]); break;
      RA_CASE(0) RA_CASE(1) RA_CASE(2) RA_CASE(3) RA_CASE(4) RA_CASE(5) RA_CASE(6) RA_CASE(7) RA_CASE(8) RA_CASE(9) RA_CASE(10) RA_CASE(11)
#undef RA_CASE
      default: e = hipErrorInvalidValue;
    }
  }
  hipError_t back = hipSetDevice(cur);
  return e != hipSuccess ? e : back;
}

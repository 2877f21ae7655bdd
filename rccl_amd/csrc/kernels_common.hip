// Type-independent kernels (all-gather moves bytes) and the dtype dispatch over the
// per-type translation units built from kernels_dtype.hip.
#include "kernels_impl.h"

enum { RA_G_BROADCAST, RA_G_DIRECT_ALLGATHER, RA_G_RING_ALLGATHER };
template <int K>
__global__ __launch_bounds__(1024) void raGatherKernel(const RaDevComm* __restrict__ comm, RaWork w) {
  __shared__ ChanShared shA;
  Ctx cx;
  const int role = blockIdx.x / w.nChannels;
  cx.init(comm, RA_LDS_CHAN(shA), blockIdx.x % w.nChannels, w.stepBytes);
  if constexpr (K == RA_G_BROADCAST) directBroadcast(cx, w, role);
  else if constexpr (K == RA_G_DIRECT_ALLGATHER) directAllGather(cx, w, role);
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
    if (w.coll == RA_COLL_BROADCAST) hipLaunchKernelGGL(raGatherKernel<RA_G_BROADCAST>, dim3(grid), dim3(block), 0, s, comm, w);
    else if (w.algo == RA_ALGO_DIRECT) hipLaunchKernelGGL(raGatherKernel<RA_G_DIRECT_ALLGATHER>, dim3(grid), dim3(block), 0, s, comm, w);
    else hipLaunchKernelGGL(raGatherKernel<RA_G_RING_ALLGATHER>, dim3(grid), dim3(block), 0, s, comm, w);
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
// `naWords` (the GPU's status words, RA_NA_WORD_*) is handled the same way.
static __device__ __align__(256) unsigned char scratch[RA_NA_SCRATCH_BYTES];
static __device__ __align__(256) uint32_t naWords[RA_NA_WORDS];

#include <atomic>
#include <mutex>

static std::mutex gNaMutex;
static char* gNaLines[64];             // per device: address of its `scratch`, null until set up
static uint32_t* gNaWords[64];         // per device: address of its `naWords`
static std::atomic<uint32_t> gNaEpoch{0};

// The addresses of `dev`'s arrays (set-up on the first call for a device; the current device is left as found).
static hipError_t naSetup(int dev, char** lines, uint32_t** words) {
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> g(gNaMutex);
  if (!gNaLines[dev] || !gNaWords[dev]) {
    int cur = 0;
    hipError_t e = hipGetDevice(&cur);
    if (e != hipSuccess) return e;
    if ((e = hipSetDevice(dev)) != hipSuccess) return e;
    void* p = nullptr;
    void* q = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e == hipSuccess) e = hipGetSymbolAddress(&q, HIP_SYMBOL(naWords));
    hipError_t back = hipSetDevice(cur);
    if (e != hipSuccess) return e;
    if (back != hipSuccess) return back;
    if (!p || !q) return hipErrorInvalidSymbol;
    gNaLines[dev] = (char*)p;
    gNaWords[dev] = (uint32_t*)q;
  }
  *lines = gNaLines[dev];
  *words = gNaWords[dev];
  return hipSuccess;
}

// All ranks on ONE device (the caller has checked that).  A launch that fails after earlier ranks' kernels are already
// waiting cancels the call: the call's number goes into the GPU's cancel word, which every waiting kernel polls.
hipError_t raNoAllocAllReduce(const void* const* send, void* const* recv, int device, int nRanks, size_t count,
                              int dtype, int op, uint64_t opArg, uint64_t timeoutTicks, const hipStream_t* streams) {
  if (nRanks < 1 || nRanks > RA_NA_MAX_RANKS) return hipErrorInvalidValue;
  RaNaArgs a;
  char* lines = nullptr;
  hipError_t e = naSetup(device, &lines, &a.words);
  if (e != hipSuccess) return e;
  for (int p = 0; p < RA_NA_MAX_RANKS; p++) a.lines[p] = p < nRanks ? lines : nullptr;
  int cur = 0;
  if ((e = hipGetDevice(&cur)) != hipSuccess) return e;
  if (cur != device && (e = hipSetDevice(device)) != hipSuccess) return e;
  uint32_t epoch = gNaEpoch.fetch_add(1) + 1;
  if (epoch == 0) epoch = gNaEpoch.fetch_add(1) + 1;
  a.count = count; a.opArg = opArg; a.timeoutTicks = timeoutTicks; a.epoch = epoch; a.nRanks = nRanks;
  int launched = 0;
  for (int r = 0; r < nRanks && e == hipSuccess; r++) {
    a.rank = r; a.send = send[r]; a.recv = recv[r];
    switch (dtype) {
#define RA_CASE(N) case N: e = raLaunchNoAllocDt##N(a, op, streams[r]); break;
      RA_CASE(0) RA_CASE(1) RA_CASE(2) RA_CASE(3) RA_CASE(4) RA_CASE(5) RA_CASE(6) RA_CASE(7) RA_CASE(8) RA_CASE(9) RA_CASE(10) RA_CASE(11)
#undef RA_CASE
      default: e = hipErrorInvalidValue;
    }
    if (e == hipSuccess) launched++;
  }
  if (e != hipSuccess && launched > 0)      // the kernels already launched would wait for lines that never come
    (void)hipMemcpy(a.words + RA_NA_WORD_CANCEL, &epoch, sizeof epoch, hipMemcpyHostToDevice);
  hipError_t back = cur != device ? hipSetDevice(cur) : hipSuccess;
  return e != hipSuccess ? e : back;
}

// The call number of the first call that gave up on `device` (0: none).  A 4-byte copy from the device: no allocation.
hipError_t raNoAllocFailedCall(int device, uint32_t* call) {
  char* lines = nullptr;
  uint32_t* words = nullptr;
  hipError_t e = naSetup(device, &lines, &words);
  if (e != hipSuccess) return e;
  int cur = 0;
  if ((e = hipGetDevice(&cur)) != hipSuccess) return e;
  if (cur != device && (e = hipSetDevice(device)) != hipSuccess) return e;
  e = hipMemcpy(call, words + RA_NA_WORD_FAILED, sizeof *call, hipMemcpyDeviceToHost);
  hipError_t back = cur != device ? hipSetDevice(cur) : hipSuccess;
  return e != hipSuccess ? e : back;
}

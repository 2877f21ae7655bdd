// Host-callable launchers implemented in kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include "device_types.h"

// grid = roles * w.nChannels workgroups of `block` threads on stream s.
hipError_t raLaunchColl(const RaDevComm* comm, const RaWork& w, int dtype, int op, int grid, int block, hipStream_t s);

// One-shot low-latency collectives (payload <= RA_LL_MAX_BYTES): w.nChannels (<= RA_LL_MAX_GRID) x RA_LL_THREADS threads.
hipError_t raLaunchLL(const RaDevComm* comm, const RaWork& w, int dtype, int op, hipStream_t s);

// Local kernels run RA_LOCAL_THREADS threads per workgroup; one workgroup moves RA_LOCAL_TILE_BYTES of
// output per pass (1024 threads x 128 B in flight each: the shape that measured fastest for a 1 GiB copy,
// 6.28 TB/s, with one tile per workgroup).
#define RA_LOCAL_THREADS 1024
#define RA_LOCAL_TILE_BYTES (RA_LOCAL_THREADS * 128)

// dst[i] = fold of srcs[0..nsrc-1][i] on the current device, `grid` workgroups of RA_LOCAL_THREADS threads.
// opArgPtr (may be null): device address of the pre-multiply scalar, read by the kernel instead of opArg.
hipError_t raLaunchLocal(void* dst, const void* const* srcs, int nsrc, int preIdx, bool post, uint64_t opArg,
                         const void* opArgPtr, size_t nelem, int dtype, int op, int grid, hipStream_t s);

// All-reduce over nRanks buffers on ONE device, rank r on streams[r], that allocates nothing: its line buffers are a
// static array the GPU holds (payload per rank at most RA_NA_MAX_BYTES, at most RA_NA_MAX_RANKS ranks).
hipError_t raNoAllocAllReduce(const void* const* send, void* const* recv, int device, int nRanks, size_t count,
                              int dtype, int op, uint64_t opArg, uint64_t timeoutTicks, const hipStream_t* streams);
// Number of the first such call one of whose kernels gave up waiting on `device` (0: none).
hipError_t raNoAllocFailedCall(int device, uint32_t* call);

// Grid for a local kernel producing `outBytes` from `nsrc` sources: one tile per workgroup (a tile is
// RA_LOCAL_THREADS x U packs with nsrc*U = 8, see reduceCopy), grid-stride beyond 64 Ki tiles.
static inline int raLocalGrid(size_t outBytes, int nsrc) {
  const int u = nsrc <= 1 ? 8 : nsrc == 2 ? 4 : nsrc == 8 ? 1 : 2;
  const size_t tile = (size_t)RA_LOCAL_THREADS * 16 * u;
  size_t g = (outBytes + tile - 1) / tile;
  if (g < 1) g = 1;
  if (g > 65536) g = 65536;
  return (int)g;
}

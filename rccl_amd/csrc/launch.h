// Host-callable launchers implemented in kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include "device_types.h"

// grid = roles * w.nChannels workgroups of `block` threads on stream s.
hipError_t raLaunchColl(const RaDevComm* comm, const RaWork& w, int dtype, int op, int grid, int block, hipStream_t s);

// dst[i] = fold of srcs[0..nsrc-1][i] on the current device, `grid` workgroups of 256 threads.
hipError_t raLaunchLocal(void* dst, const void* const* srcs, int nsrc, int preIdx, bool post, uint64_t opArg,
                         size_t nelem, int dtype, int op, int grid, hipStream_t s);

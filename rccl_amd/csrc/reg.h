// User-buffer registration (ncclCommRegister / ncclCommDeregister) and the zero-copy gather it enables.
//
// The reference keeps a per-communicator cache of registered page ranges (src/register.cc:88-185) and
// uses it for network, NVLS and CollNet buffers; its ring kernels never write into a peer's user
// buffer (Direct = 0, src/device/all_reduce.h:62).  Here a registered receive buffer lets the
// all-pairs all-reduce write finished pieces straight into the peers' receive buffers, which removes
// the broadcast-phase FIFO write, its read-back and the gather copy.
//
// Every rank publishes its registrations (allocation base, size, HIP IPC handle) on a small board in
// POSIX shared memory.  Nothing is exchanged at registration time: before an all-reduce launch a rank
// looks at its peers' boards (one generation counter each) and maps what is new.  Which registration a
// rank's receive buffer belongs to travels to the peers inside the collective itself (one flag word
// with the first scatter step of every channel), so a peer that has not mapped it yet simply keeps
// using the FIFO for that launch.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "rccl_amd.h"

struct ncclComm;

ncclResult_t raRegInit(ncclComm* c, uint64_t key);
void raRegShutdown(ncclComm* c);
ncclResult_t raRegRegister(ncclComm* c, void* buff, size_t size, void** handle);
ncclResult_t raRegDeregister(ncclComm* c, void* handle);
// Before an all-reduce launch on `stream`: *word = what this rank announces for [recv, recv+bytes)
// (0 when the range is not registered); *havePeerRegs = some peer registration is mapped here.
ncclResult_t raRegPrepare(ncclComm* c, const void* recv, size_t bytes, hipStream_t stream, uint64_t* word,
                          int* havePeerRegs);
// A fresh key for the boards of a communicator whose ranks all live in this process.
uint64_t raRegLocalKey();

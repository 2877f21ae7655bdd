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
//
// Memory-type rule.  A peer may write only into memory that is not cached on the owner's side: fine-grained /
// uncached device memory, the one type HSA defines for concurrent access by several agents (and the only type the
// reference lets a peer write, src/transport/p2p.cc:231-235, src/include/alloc.h:34-36).  Ordinary (coarse-grained)
// hipMalloc memory is coherent only at the owner's own kernel boundaries: lines the owner's earlier kernels left in
// its caches can hide or later overwrite what a peer stored, so a registration of such memory stays local (it is
// cached as the reference caches it, and its collectives keep using the FIFOs).  ncclMemAlloc returns memory of
// the peer-writable type; buffers from hipExtMallocWithFlags(hipDeviceMallocUncached / Finegrained) qualify too.
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
// ncclMemAlloc / ncclMemFree: peer-writable device memory, remembered per process so that raRegRegister knows it.
ncclResult_t raMemAlloc(void** ptr, size_t size);
ncclResult_t raMemFree(void* ptr);
// True when [base, base+size) is device memory a peer may store into (see the memory-type rule above).
bool raMemPeerWritable(const void* base, size_t size);
// A fresh key for the boards of a communicator whose ranks all live in this process.
uint64_t raRegLocalKey();

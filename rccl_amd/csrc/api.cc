// The exported C ABI (include/rccl_amd.h).  Every symbol forwards to an internal function
// so that calls made inside this library never bind to another RCCL loaded in the process.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include "comm.h"
#include "net.h"
#include "debug.h"
#include "launch.h"
#include "reg.h"

#define RA_API extern "C" __attribute__((visibility("default")))

RA_API ncclResult_t ncclGetVersion(int* version) {
  if (!version) return ncclInvalidArgument;
  *version = RCCL_AMD_NCCL_VERSION_CODE;
  return ncclSuccess;
}

RA_API const char* ncclGetErrorString(ncclResult_t r) {
  // texts of src/init.cc ncclGetErrorString
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled cuda error (run with NCCL_DEBUG=INFO for details)";
    case ncclSystemError: return "unhandled system error (run with NCCL_DEBUG=INFO for details)";
    case ncclInternalError: return "internal error - please report this issue to the NCCL developers";
    case ncclInvalidArgument: return "invalid argument (run with NCCL_DEBUG=WARN for details)";
    case ncclInvalidUsage: return "invalid usage (run with NCCL_DEBUG=WARN for details)";
    case ncclRemoteError: return "remote process exited or there was a network error";
    case ncclInProgress: return "NCCL operation in progress";
    default: return "unknown result code";
  }
}

RA_API const char* ncclGetLastError(ncclComm_t) { return raLastError(); }

RA_API ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t* asyncError) {
  RA_CHECK(raCommCheck(comm, "ncclCommGetAsyncError"));
  if (!asyncError) return ncclInvalidArgument;
  *asyncError = raCommPollError(comm);
  return ncclSuccess;
}

RA_API ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) { RA_WARN("ncclGetUniqueId: uniqueId is NULL"); return ncclInvalidArgument; }
  return raBootstrapCreate(id);
}

RA_API ncclResult_t ncclCommInitRankConfig(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank, ncclConfig_t* config) {
  if (raGroupDepth() > 0) {
    int dev = 0;
    RA_HIP(hipGetDevice(&dev));
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    raGroupAddInit(comm, nranks, id, rank, config, dev);
    return ncclSuccess;
  }
  return raCommInitRank(comm, nranks, id, rank, config);
}
RA_API ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  return ncclCommInitRankConfig(comm, nranks, id, rank, nullptr);
}
RA_API ncclResult_t ncclCommInitAll(ncclComm_t* comm, int ndev, const int* devlist) { return raCommInitAll(comm, ndev, devlist); }

RA_API ncclResult_t ncclCommFinalize(ncclComm_t comm) {
  RA_CHECK(raCommEnsureReady(comm, "ncclCommFinalize"));
  if (comm->haveLast) {
    int old = 0;
    (void)hipGetDevice(&old);
    (void)hipSetDevice(comm->device);
    hipError_t e = hipStreamSynchronize(comm->lastStream);
    (void)hipSetDevice(old);
    if (e != hipSuccess) return ncclUnhandledCudaError;
  }
  comm->finalized = true;
  return raCommPollError(comm);
}
RA_API ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  if (comm == nullptr) return ncclSuccess;   // src/init.cc: destroying NULL is a no-op
  return raCommDestroy(comm, false);
}
RA_API ncclResult_t ncclCommAbort(ncclComm_t comm) {
  if (comm == nullptr) return ncclSuccess;
  return raCommDestroy(comm, true);
}
RA_API ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
  RA_CHECK(raCommCheck(comm, "ncclCommCount"));
  if (!count) return ncclInvalidArgument;
  *count = comm->nRanks;
  return ncclSuccess;
}
RA_API ncclResult_t ncclCommCuDevice(const ncclComm_t comm, int* device) {
  RA_CHECK(raCommCheck(comm, "ncclCommCuDevice"));
  if (!device) return ncclInvalidArgument;
  *device = comm->device;
  return ncclSuccess;
}
RA_API ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) {
  RA_CHECK(raCommCheck(comm, "ncclCommUserRank"));
  if (!rank) return ncclInvalidArgument;
  *rank = comm->rank;
  return ncclSuccess;
}

// The reference's ncclMemAlloc is a plain device allocation on this platform (src/init.cc:2891-2966: the cuMem path
// needs multicast support, else cudaMalloc).  Here it returns fine-grained (uncached) device memory: the type a
// registered buffer must have for peers to write into it (reg.h, memory-type rule).
RA_API ncclResult_t ncclMemAlloc(void** ptr, size_t size) { return raMemAlloc(ptr, size); }
RA_API ncclResult_t ncclMemFree(void* ptr) { return raMemFree(ptr); }

RA_API ncclResult_t ncclCommRegister(const ncclComm_t comm, void* buff, size_t size, void** handle) {
  RA_CHECK(raCommEnsureReady(comm, "ncclCommRegister"));       // src/register.cc:154-160
  return raRegRegister(comm, buff, size, handle);
}
RA_API ncclResult_t ncclCommDeregister(const ncclComm_t comm, void* handle) {
  RA_CHECK(raCommEnsureReady(comm, "ncclCommDeregister"));
  return raRegDeregister(comm, handle);
}
RA_API ncclResult_t rcclAmdCommStats(ncclComm_t comm, uint64_t* directPieces, uint64_t* fifoPieces) {
  RA_CHECK(raCommEnsureReady(comm, "rcclAmdCommStats"));
  uint64_t v[2] = {0, 0};
  if (comm->stats) {
    int old = 0;
    (void)hipGetDevice(&old);
    (void)hipSetDevice(comm->device);
    hipError_t e = hipMemcpy(v, comm->stats, sizeof v, hipMemcpyDeviceToHost);
    (void)hipSetDevice(old);
    if (e != hipSuccess) return ncclUnhandledCudaError;
  }
  if (directPieces) *directPieces = v[0];
  if (fifoPieces) *fifoPieces = v[1];
  return ncclSuccess;
}

RA_API ncclResult_t rcclAmdCommNetStats(ncclComm_t comm, uint64_t* out) {
  RA_CHECK(raCommEnsureReady(comm, "rcclAmdCommNetStats"));
  if (!out) return ncclInvalidArgument;
  raNetStats(comm, out);
  return ncclSuccess;
}

RA_API ncclResult_t rcclAmdCommStepTimers(ncclComm_t comm, uint64_t* ticks) {
  RA_CHECK(raCommEnsureReady(comm, "rcclAmdCommStepTimers"));
  if (!ticks) return ncclInvalidArgument;
  memset(ticks, 0, 15 * sizeof(uint64_t));
  if (comm->stats) {
    int old = 0;
    (void)hipGetDevice(&old);
    (void)hipSetDevice(comm->device);
    hipError_t e = hipMemcpy(ticks, comm->stats + 8, 15 * sizeof(uint64_t), hipMemcpyDeviceToHost);
    (void)hipSetDevice(old);
    if (e != hipSuccess) return ncclUnhandledCudaError;
  }
  return ncclSuccess;
}

RA_API ncclResult_t ncclRedOpCreatePreMulSum(ncclRedOp_t* op, void* scalar, ncclDataType_t datatype,
                                             ncclScalarResidence_t residence, ncclComm_t comm) {
  // src/enqueue.cc:2219-2270
  RA_CHECK(raCommEnsureReady(comm, "ncclRedOpCreatePreMulSum"));
  if (!op || !scalar) return ncclInvalidArgument;
  const size_t es = raTypeSize(datatype);
  if (es == 0) { RA_WARN("ncclRedOpCreatePreMulSum: unsupported datatype %d", datatype); return ncclInvalidArgument; }
  for (int i = 0; i < RA_MAX_USER_OPS; i++) {
    if (comm->userOps[i].used) continue;
    RaUserOp& u = comm->userOps[i];
    u.used = true; u.dtype = datatype; u.arg = 0; u.argPtr = nullptr;
    if (residence == ncclScalarHostImmediate) memcpy(&u.arg, scalar, es);
    else u.argPtr = scalar;
    *op = (ncclRedOp_t)((int)ncclNumOps + i);
    return ncclSuccess;
  }
  RA_WARN("ncclRedOpCreatePreMulSum: too many live custom operators");
  return ncclInvalidUsage;
}
RA_API ncclResult_t ncclRedOpDestroy(ncclRedOp_t op, ncclComm_t comm) {
  RA_CHECK(raCommEnsureReady(comm, "ncclRedOpDestroy"));
  const int ix = (int)op - (int)ncclNumOps;
  if (ix < 0 || ix >= RA_MAX_USER_OPS || !comm->userOps[ix].used) { RA_WARN("ncclRedOpDestroy: operator %d is not a live custom operator", (int)op); return ncclInvalidArgument; }
  comm->userOps[ix].used = false;
  return ncclSuccess;
}

RA_API ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype,
                                  ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
  RaCollArgs a{RA_COLL_ALLREDUCE, sendbuff, recvbuff, count, (int)datatype, (int)op, comm, stream};
  return raEnqueue(a);
}
RA_API ncclResult_t ncclReduceScatter(const void* sendbuff, void* recvbuff, size_t recvcount, ncclDataType_t datatype,
                                      ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
  RaCollArgs a{RA_COLL_REDUCESCATTER, sendbuff, recvbuff, recvcount, (int)datatype, (int)op, comm, stream};
  return raEnqueue(a);
}
RA_API ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype,
                                  ncclComm_t comm, hipStream_t stream) {
  RaCollArgs a{RA_COLL_ALLGATHER, sendbuff, recvbuff, sendcount, (int)datatype, (int)ncclSum, comm, stream};
  return raEnqueue(a);
}
RA_API ncclResult_t ncclGroupStart(void) { return raGroupStart(); }
RA_API ncclResult_t ncclGroupEnd(void) { return raGroupEnd(); }

static ncclResult_t outOfScope(const char* name) {
  RA_WARN("%s is outside this library's path (all-reduce, reduce-scatter, all-gather)", name);
  return ncclInvalidUsage;
}
RA_API ncclResult_t ncclReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                               int root, ncclComm_t comm, hipStream_t stream) {
  // src/collectives.cc ncclReduce_impl
  RaCollArgs a{RA_COLL_REDUCE, sendbuff, recvbuff, count, (int)datatype, (int)op, comm, stream, root};
  return raEnqueue(a);
}
RA_API ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root,
                                  ncclComm_t comm, hipStream_t stream) {
  // src/collectives.cc ncclBroadcast_impl
  RaCollArgs a{RA_COLL_BROADCAST, sendbuff, recvbuff, count, (int)datatype, (int)ncclSum, comm, stream, root};
  return raEnqueue(a);
}
RA_API ncclResult_t ncclBcast(void* buff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream) {
  return ncclBroadcast(buff, buff, count, datatype, root, comm, stream);   // the deprecated in-place form
}
RA_API ncclResult_t ncclSend(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) { return outOfScope("ncclSend"); }
RA_API ncclResult_t ncclRecv(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) { return outOfScope("ncclRecv"); }
RA_API ncclResult_t ncclCommSplit(ncclComm_t comm, int color, int key, ncclComm_t* newcomm, ncclConfig_t* config) {
  // src/init.cc:2753-2820
  RA_CHECK(raCommCheck(comm, "ncclCommSplit"));
  if (!newcomm) { RA_WARN("ncclCommSplit: newcomm is NULL"); return ncclInvalidArgument; }
  if (color < 0 && color != NCCL_SPLIT_NOCOLOR) { RA_WARN("ncclCommSplit: invalid colour %d", color); return ncclInvalidArgument; }
  if (config && (config->magic != 0xcafebeef || config->size < sizeof(size_t) + 2 * sizeof(unsigned))) {
    RA_WARN("ncclCommSplit: config was not initialised with NCCL_CONFIG_INITIALIZER");
    return ncclInvalidArgument;
  }
  const ncclResult_t ready = raCommPollError(comm);
  if (ready != ncclSuccess) {
    RA_WARN("ncclCommSplit: the parent communicator is not ready (%s)", ncclGetErrorString(ready));
    return ready == ncclInProgress ? ncclInvalidArgument : ready;
  }
  *newcomm = NCCL_COMM_NULL;               // stays null until the split has ended
  if (raGroupDepth() > 0) { raGroupAddSplit(comm, color, key, newcomm, config); return ncclSuccess; }
  return raCommSplit(comm, color, key, newcomm, config);
}
RA_API ncclResult_t ncclGather(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) { return outOfScope("ncclGather"); }
RA_API ncclResult_t ncclScatter(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) { return outOfScope("ncclScatter"); }
RA_API ncclResult_t ncclAllToAll(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) { return outOfScope("ncclAllToAll"); }
RA_API ncclResult_t ncclAllToAllv(const void*, const size_t[], const size_t[], void*, const size_t[], const size_t[], ncclDataType_t,
                                  ncclComm_t, hipStream_t) { return outOfScope("ncclAllToAllv"); }
// The MSCCL entry points (src/msccl.cc): the reference routes small all-reduces through that interpreter on MI300X;
// here the planner has its own small-message path, so there is nothing to load or run.
RA_API ncclResult_t mscclLoadAlgo(const char*, mscclAlgoHandle_t*, int) { return outOfScope("mscclLoadAlgo"); }
RA_API ncclResult_t mscclRunAlgo(const void*, const size_t[], const size_t[], void*, const size_t[], const size_t[], size_t,
                                 ncclDataType_t, int, int, ncclRedOp_t, mscclAlgoHandle_t, ncclComm_t, hipStream_t) { return outOfScope("mscclRunAlgo"); }
RA_API ncclResult_t mscclUnloadAlgo(mscclAlgoHandle_t) { return outOfScope("mscclUnloadAlgo"); }

RA_API ncclResult_t rcclAmdReduceCopy(void* dst, const void* const* srcs, int nsrc, size_t count, ncclDataType_t datatype,
                                      ncclRedOp_t op, hipStream_t stream) {
  if (!dst || !srcs || nsrc < 1 || nsrc > RA_MAX_SRCS) { RA_WARN("rcclAmdReduceCopy: 1..%d sources", RA_MAX_SRCS); return ncclInvalidArgument; }
  if (datatype < 0 || datatype >= ncclNumTypes || op < 0 || op > ncclMin) { RA_WARN("rcclAmdReduceCopy: unsupported type/op"); return ncclInvalidArgument; }
  if (count == 0) return ncclSuccess;
  const size_t bytes = count * raTypeSize(datatype);
  const int grid = raLocalGrid(bytes, nsrc);
  static const int devOps[] = {RA_OP_SUM, RA_OP_PROD, RA_OP_MAX, RA_OP_MIN};
  hipError_t e = raLaunchLocal(dst, srcs, nsrc, -1, false, 0, nullptr, count, datatype, devOps[op], grid, stream);
  if (e != hipSuccess) { RA_WARN("rcclAmdReduceCopy: launch failed: %s", hipGetErrorString(e)); return ncclUnhandledCudaError; }
  return ncclSuccess;
}


// No hipMalloc, hipFree or other allocating call on this path: the line buffers are a static array on each GPU
// (kernels_common.hip), everything else travels in the kernel arguments.
RA_API ncclResult_t rcclAmdAllReduceNoAlloc(const void* const* sendbuffs, void* const* recvbuffs, const int* devices,
                                            int nranks, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                                            const hipStream_t* streams) {
  if (!sendbuffs || !recvbuffs || !devices || !streams) { RA_WARN("rcclAmdAllReduceNoAlloc: null argument"); return ncclInvalidArgument; }
  if (nranks < 1 || nranks > RA_NA_MAX_RANKS) {
    RA_WARN("rcclAmdAllReduceNoAlloc: %d ranks, 1..%d supported", nranks, RA_NA_MAX_RANKS);
    return ncclInvalidArgument;
  }
  if (datatype < 0 || datatype >= ncclNumTypes || op < 0 || op >= ncclNumOps) {
    RA_WARN("rcclAmdAllReduceNoAlloc: unsupported type %d / op %d (built-in operators only)", (int)datatype, (int)op);
    return ncclInvalidArgument;
  }
  const size_t bytes = count * raTypeSize(datatype);
  if (count > RA_NA_MAX_BYTES || bytes > RA_NA_MAX_BYTES) {
    RA_WARN("rcclAmdAllReduceNoAlloc: %zu elements of %zu bytes exceed the limit of %u bytes per rank; use ncclAllReduce",
            count, raTypeSize(datatype), RA_NA_MAX_BYTES);
    return ncclInvalidArgument;
  }
  int nDev = 0;
  if (hipGetDeviceCount(&nDev) != hipSuccess) return ncclUnhandledCudaError;
  for (int r = 0; r < nranks; r++) {
    if (devices[r] < 0 || devices[r] >= nDev || devices[r] >= 64) { RA_WARN("rcclAmdAllReduceNoAlloc: rank %d names device %d", r, devices[r]); return ncclInvalidArgument; }
    if (count && (!sendbuffs[r] || !recvbuffs[r])) { RA_WARN("rcclAmdAllReduceNoAlloc: rank %d has a null buffer", r); return ncclInvalidArgument; }
    // The staging lines are part of the library's code object, which is ordinary (coarse-grained) device memory: a peer
    // GPU may not store into that (peers write only fine-grained or uncached memory, see reg.h), so every rank of a
    // call has to be on the same device.
    if (devices[r] != devices[0]) {
      RA_WARN("rcclAmdAllReduceNoAlloc: ranks on different devices (%d and %d): all ranks of a call must share one device; "
              "between devices use ncclAllReduce on a communicator", devices[0], devices[r]);
      return ncclInvalidUsage;
    }
    // The n kernels wait for one another: two ranks on one stream would run one after the other and never meet.
    for (int q = 0; q < r; q++)
      if (streams[q] == streams[r]) {
        RA_WARN("rcclAmdAllReduceNoAlloc: ranks %d and %d name the same stream; every rank needs a stream of its own", q, r);
        return ncclInvalidUsage;
      }
  }
  // Every rank's kernel waits for the others, so all must be running at once, each on a hardware queue of its own; a
  // process has few of those per device (GPU_MAX_HW_QUEUES, default 4, per stream priority; 8 ranks of one process on
  // one device measured: the kernels queue behind one another and run into their time limit).
  if (nranks > RA_NA_MAX_RANKS_PER_DEVICE) {
    RA_WARN("rcclAmdAllReduceNoAlloc: %d ranks on device %d; at most %d ranks may share a device (one hardware queue each)",
            nranks, devices[0], RA_NA_MAX_RANKS_PER_DEVICE);
    return ncclInvalidUsage;
  }
  if (count == 0) return ncclSuccess;
  int devOp = RA_OP_SUM;
  uint64_t arg = 0;
  static const int devOps[] = {RA_OP_SUM, RA_OP_PROD, RA_OP_MAX, RA_OP_MIN};
  if (op == ncclAvg) RA_CHECK(raAvgOp(nranks, datatype, &devOp, &arg));
  else devOp = devOps[op];
  long ms = 120000;                                     // the same knob and default as a communicator's waits
  if (const char* t = getenv("RCCL_AMD_SPIN_TIMEOUT_MS")) { if (*t) ms = strtol(t, nullptr, 0); }
  if (ms < 1) ms = 1;
  hipError_t e = raNoAllocAllReduce(sendbuffs, recvbuffs, devices[0], nranks, count, datatype, devOp, arg,
                                    (uint64_t)ms * 100000ull, streams);
  if (e != hipSuccess) { RA_WARN("rcclAmdAllReduceNoAlloc: %s", hipGetErrorString(e)); return ncclUnhandledCudaError; }
  return ncclSuccess;
}

RA_API ncclResult_t rcclAmdNoAllocAsyncError(int device, ncclResult_t* asyncError) {
  if (!asyncError) return ncclInvalidArgument;
  uint32_t call = 0;
  hipError_t e = raNoAllocFailedCall(device, &call);
  if (e != hipSuccess) { RA_WARN("rcclAmdNoAllocAsyncError: %s", hipGetErrorString(e)); return e == hipErrorInvalidDevice ? ncclInvalidArgument : ncclUnhandledCudaError; }
  if (call) RA_WARN("rcclAmdAllReduceNoAlloc: call %u on device %d gave up waiting for its peers; its receive buffers are incomplete", call, device);
  *asyncError = call ? ncclRemoteError : ncclSuccess;
  return ncclSuccess;
}

// Window a communicator of `nranks` ranks would allocate with `channels` channels and the default slot and exchange
// settings (environment included), after the fit into the size budget; host logic only.
RA_API ncclResult_t rcclAmdWindowBytesFor(int nranks, int channels, size_t* totalBytes, size_t* fifoBytes, size_t* exchangeBytes,
                                          int* channelsOut) {
  if (nranks < 2 || nranks > RA_MAX_RANKS || channels < 1 || channels > RA_MAX_CHANNELS) return ncclInvalidArgument;
  RaWindowLayout l;
  int ch = channels;
  raWindowLayoutFor(nranks, &ch, &l);
  if (totalBytes) *totalBytes = l.totalBytes;
  if (fifoBytes) *fifoBytes = raDataBytes(l);
  if (exchangeBytes) *exchangeBytes = l.totalBytes - l.osFlagOffset;
  if (channelsOut) *channelsOut = ch;
  return ncclSuccess;
}

RA_API ncclResult_t rcclAmdPlanInfo(ncclComm_t comm, int coll, size_t count, ncclDataType_t datatype, int* algo, int* proto,
                                    int* nChannels, size_t* chunkElems) {
  RA_CHECK(raCommEnsureReady(comm, "rcclAmdPlanInfo"));
  if (coll < 0 || coll >= RA_NUM_COLLS) return ncclInvalidArgument;
  RaPlan p;
  RA_CHECK(raPlan(comm, coll, count, datatype, &p));
  if (algo) *algo = p.algo;
  if (proto) *proto = p.proto;
  if (nChannels) *nChannels = p.nChannels;
  if (chunkElems) *chunkElems = p.chunkElems;
  return ncclSuccess;
}

RA_API ncclResult_t rcclAmdPlanForConfig(int nranks, int coll, size_t count, ncclDataType_t datatype, int algo, int maxChannels,
                                         size_t slotBytes, size_t minChunkBytes, int* algoOut, int* nChannels,
                                         size_t* chunkElems, size_t* lastChunkElems, int* grid) {
  if (coll < 0 || coll > RA_COLL_ALLGATHER) return ncclInvalidArgument;
  RaPlan p;
  RA_CHECK(raPlanFor(nranks, coll, count, datatype, algo, maxChannels, slotBytes, minChunkBytes, 1024, 0, 0, 0, &p));
  if (algoOut) *algoOut = p.algo;
  if (nChannels) *nChannels = p.nChannels;
  if (chunkElems) *chunkElems = p.chunkElems;
  if (lastChunkElems) *lastChunkElems = p.lastChunkElems;
  if (grid) *grid = p.grid;
  return ncclSuccess;
}

RA_API ncclResult_t rcclAmdPlanProtoForConfig(int nranks, int coll, size_t count, ncclDataType_t datatype, int algo, int allowLL,
                                              size_t oneShotMaxBytes, size_t twoExchangeMaxBytes, int* algoOut, int* proto) {
  if (coll < 0 || coll >= RA_NUM_COLLS || allowLL < 0 || allowLL > 2) return ncclInvalidArgument;
  RaPlan p;
  RA_CHECK(raPlanFor(nranks, coll, count, datatype, algo, 16, 512 << 10, 32 << 10, 1024, allowLL, oneShotMaxBytes, twoExchangeMaxBytes, &p));
  if (algoOut) *algoOut = p.algo;
  if (proto) *proto = p.proto;
  return ncclSuccess;
}

RA_API ncclResult_t rcclAmdBootstrapAllGather(const ncclUniqueId* id, int rank, int nranks, const void* mine, void* all,
                                              size_t bytes) {
  if (!id || !mine || !all || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  RaBootstrap b;
  ncclResult_t r = b.connect(*id, rank, nranks);
  if (r == ncclSuccess) r = b.allGather(mine, all, bytes);
  if (r == ncclSuccess) r = b.barrier();
  b.close();
  return r;
}

// "p"-prefixed twins of the public entry points (src/include/core.h:18-27): an interposer defines ncclX and
// calls pncclX.  Plain forwarding functions (this file is also seen by the device pass, which has no definitions
// an alias could point to).
RA_API ncclResult_t pncclGetVersion(int* version) { return ncclGetVersion(version); }
RA_API const char* pncclGetErrorString(ncclResult_t result) { return ncclGetErrorString(result); }
RA_API const char* pncclGetLastError(ncclComm_t comm) { return ncclGetLastError(comm); }
RA_API ncclResult_t pncclCommGetAsyncError(ncclComm_t comm, ncclResult_t* asyncError) { return ncclCommGetAsyncError(comm, asyncError); }
RA_API ncclResult_t pncclGetUniqueId(ncclUniqueId* uniqueId) { return ncclGetUniqueId(uniqueId); }
RA_API ncclResult_t pncclCommInitRankConfig(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank, ncclConfig_t* config) { return ncclCommInitRankConfig(comm, nranks, commId, rank, config); }
RA_API ncclResult_t pncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank) { return ncclCommInitRank(comm, nranks, commId, rank); }
RA_API ncclResult_t pncclCommInitAll(ncclComm_t* comm, int ndev, const int* devlist) { return ncclCommInitAll(comm, ndev, devlist); }
RA_API ncclResult_t pncclCommFinalize(ncclComm_t comm) { return ncclCommFinalize(comm); }
RA_API ncclResult_t pncclCommDestroy(ncclComm_t comm) { return ncclCommDestroy(comm); }
RA_API ncclResult_t pncclCommAbort(ncclComm_t comm) { return ncclCommAbort(comm); }
RA_API ncclResult_t pncclCommCount(const ncclComm_t comm, int* count) { return ncclCommCount(comm, count); }
RA_API ncclResult_t pncclCommCuDevice(const ncclComm_t comm, int* device) { return ncclCommCuDevice(comm, device); }
RA_API ncclResult_t pncclCommUserRank(const ncclComm_t comm, int* rank) { return ncclCommUserRank(comm, rank); }
RA_API ncclResult_t pncclMemAlloc(void** ptr, size_t size) { return ncclMemAlloc(ptr, size); }
RA_API ncclResult_t pncclMemFree(void* ptr) { return ncclMemFree(ptr); }
RA_API ncclResult_t pncclCommRegister(const ncclComm_t comm, void* buff, size_t size, void** handle) { return ncclCommRegister(comm, buff, size, handle); }
RA_API ncclResult_t pncclCommDeregister(const ncclComm_t comm, void* handle) { return ncclCommDeregister(comm, handle); }
RA_API ncclResult_t pncclRedOpCreatePreMulSum(ncclRedOp_t* op, void* scalar, ncclDataType_t datatype, ncclScalarResidence_t residence, ncclComm_t comm) { return ncclRedOpCreatePreMulSum(op, scalar, datatype, residence, comm); }
RA_API ncclResult_t pncclRedOpDestroy(ncclRedOp_t op, ncclComm_t comm) { return ncclRedOpDestroy(op, comm); }
RA_API ncclResult_t pncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) { return ncclAllReduce(sendbuff, recvbuff, count, datatype, op, comm, stream); }
RA_API ncclResult_t pncclReduceScatter(const void* sendbuff, void* recvbuff, size_t recvcount, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) { return ncclReduceScatter(sendbuff, recvbuff, recvcount, datatype, op, comm, stream); }
RA_API ncclResult_t pncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) { return ncclAllGather(sendbuff, recvbuff, sendcount, datatype, comm, stream); }
RA_API ncclResult_t pncclGroupStart(void) { return ncclGroupStart(); }
RA_API ncclResult_t pncclGroupEnd(void) { return ncclGroupEnd(); }
RA_API ncclResult_t pncclReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root, ncclComm_t comm, hipStream_t stream) { return ncclReduce(sendbuff, recvbuff, count, datatype, op, root, comm, stream); }
RA_API ncclResult_t pncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream) { return ncclBroadcast(sendbuff, recvbuff, count, datatype, root, comm, stream); }
RA_API ncclResult_t pncclBcast(void* buff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream) { return ncclBcast(buff, count, datatype, root, comm, stream); }
RA_API ncclResult_t pncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) { return ncclSend(sendbuff, count, datatype, peer, comm, stream); }
RA_API ncclResult_t pncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) { return ncclRecv(recvbuff, count, datatype, peer, comm, stream); }
RA_API ncclResult_t pncclCommSplit(ncclComm_t comm, int color, int key, ncclComm_t* newcomm, ncclConfig_t* config) { return ncclCommSplit(comm, color, key, newcomm, config); }
RA_API ncclResult_t pncclGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream) { return ncclGather(sendbuff, recvbuff, sendcount, datatype, root, comm, stream); }
RA_API ncclResult_t pncclScatter(const void* sendbuff, void* recvbuff, size_t recvcount, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream) { return ncclScatter(sendbuff, recvbuff, recvcount, datatype, root, comm, stream); }
RA_API ncclResult_t pncclAllToAll(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) { return ncclAllToAll(sendbuff, recvbuff, count, datatype, comm, stream); }
RA_API ncclResult_t pncclAllToAllv(const void* sendbuff, const size_t sendcounts[], const size_t sdispls[], void* recvbuff, const size_t recvcounts[], const size_t rdispls[], ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) { return ncclAllToAllv(sendbuff, sendcounts, sdispls, recvbuff, recvcounts, rdispls, datatype, comm, stream); }
RA_API ncclResult_t pmscclLoadAlgo(const char* mscclAlgoFilePath, mscclAlgoHandle_t* mscclAlgoHandle, int rank) { return mscclLoadAlgo(mscclAlgoFilePath, mscclAlgoHandle, rank); }
RA_API ncclResult_t pmscclRunAlgo(const void* sendBuff, const size_t sendCounts[], const size_t sDisPls[], void* recvBuff, const size_t recvCounts[], const size_t rDisPls[], size_t count, ncclDataType_t dataType, int root, int peer, ncclRedOp_t op, mscclAlgoHandle_t mscclAlgoHandle, ncclComm_t comm, hipStream_t stream) { return mscclRunAlgo(sendBuff, sendCounts, sDisPls, recvBuff, recvCounts, rDisPls, count, dataType, root, peer, op, mscclAlgoHandle, comm, stream); }
RA_API ncclResult_t pmscclUnloadAlgo(mscclAlgoHandle_t mscclAlgoHandle) { return mscclUnloadAlgo(mscclAlgoHandle); }

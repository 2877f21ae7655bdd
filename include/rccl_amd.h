/*
 * rccl_amd.h — C ABI of the MI355X-native collective engine.
 *
 * The entry points below carry the names, argument order, argument meaning and
 * error convention of the reference's public header (ROCm/rccl src/nccl.h.in), so
 * that code written against <rccl/rccl.h> for ncclAllReduce / ncclReduceScatter /
 * ncclAllGather and the communicator lifecycle links against librccl_amd.so
 * unchanged.  Each declaration cites the reference line it replaces.
 *
 * Plain C: pointers, sizes and enums only.  A hipStream_t is passed as an opaque
 * pointer, so this header needs no HIP include.
 */
#ifndef RCCL_AMD_H_
#define RCCL_AMD_H_

#include <limits.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The NCCL API level this library tracks (reference: RCCL 2.21.5; nccl.h.in:15-21). */
#define NCCL_MAJOR 2
#define NCCL_MINOR 21
#define NCCL_PATCH 5
#define NCCL_SUFFIX ""
#define NCCL_VERSION_CODE 22105
#define NCCL_VERSION(X,Y,Z) (((X) <= 2 && (Y) <= 8) ? (X) * 1000 + (Y) * 100 + (Z) : (X) * 10000 + (Y) * 100 + (Z))
#define RCCL_AMD_NCCL_VERSION_CODE NCCL_VERSION_CODE
/* feature macros of the reference header (nccl.h.in:23-26) */
#define RCCL_BFLOAT16 1
#define RCCL_FLOAT8 1
#define RCCL_GATHER_SCATTER 1
#define RCCL_ALLTOALLV 1

/* Same declaration as <hip/hip_runtime_api.h>; repeating an identical typedef is legal in C11 and C++. */
typedef struct ihipStream_t* hipStream_t;

/* nccl.h.in:36-42 */
typedef struct ncclComm* ncclComm_t;
#define NCCL_COMM_NULL NULL
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;

/* nccl.h.in:50-60 */
typedef enum {
  ncclSuccess = 0,
  ncclUnhandledCudaError = 1,
  ncclSystemError = 2,
  ncclInternalError = 3,
  ncclInvalidArgument = 4,
  ncclInvalidUsage = 5,
  ncclRemoteError = 6,
  ncclInProgress = 7,
  ncclNumResults = 8
} ncclResult_t;

/* nccl.h.in:63-99 */
#define NCCL_CONFIG_UNDEF_INT INT_MIN
#define NCCL_CONFIG_UNDEF_PTR NULL
#define NCCL_SPLIT_NOCOLOR -1
typedef struct ncclConfig_v21700 {
  size_t size;
  unsigned int magic;
  unsigned int version;
  int blocking;
  int cgaClusterSize;
  int minCTAs;
  int maxCTAs;
  const char* netName;
  int splitShare;
} ncclConfig_t;
#define NCCL_CONFIG_INITIALIZER { sizeof(ncclConfig_t), 0xcafebeef, RCCL_AMD_NCCL_VERSION_CODE, \
  NCCL_CONFIG_UNDEF_INT, NCCL_CONFIG_UNDEF_INT, NCCL_CONFIG_UNDEF_INT, NCCL_CONFIG_UNDEF_INT,   \
  NCCL_CONFIG_UNDEF_PTR, NCCL_CONFIG_UNDEF_INT }

/* nccl.h.in:341-360 */
typedef enum { ncclNumOps_dummy = 5 } ncclRedOp_dummy_t;
typedef enum {
  ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4,
  ncclNumOps = 5,
  ncclMaxRedOp = 0x7fffffff >> (32 - 8 * sizeof(ncclRedOp_dummy_t))
} ncclRedOp_t;

/* nccl.h.in:364-377.  The two fp8 types use the reference's encoding (src/include/rccl_float8.h:
 * exponent bias 2^(we-1), 0x80 is NaN, no infinities), not the OCP one. */
typedef enum {
  ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3,
  ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclHalf = 6, ncclFloat32 = 7, ncclFloat = 7,
  ncclFloat64 = 8, ncclDouble = 8, ncclBfloat16 = 9, ncclFp8E4M3 = 10, ncclFp8E5M2 = 11,
  ncclNumTypes = 12
} ncclDataType_t;

/* nccl.h.in:393-396 */
typedef enum { ncclScalarDevice = 0, ncclScalarHostImmediate = 1 } ncclScalarResidence_t;

/* ---- version / errors ------------------------------------------------------ */
ncclResult_t ncclGetVersion(int* version);                                   /* nccl.h.in:121 */
const char*  ncclGetErrorString(ncclResult_t result);                        /* nccl.h.in:262 */
const char*  ncclGetLastError(ncclComm_t comm);                              /* nccl.h.in:268 */
ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t* asyncError); /* nccl.h.in:279 */

/* ---- communicator lifecycle ------------------------------------------------ */
ncclResult_t ncclGetUniqueId(ncclUniqueId* uniqueId);                        /* nccl.h.in:144 */
ncclResult_t ncclCommInitRankConfig(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank,
                                    ncclConfig_t* config);                   /* nccl.h.in:161 */
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank); /* nccl.h.in:177 */
ncclResult_t ncclCommInitAll(ncclComm_t* comm, int ndev, const int* devlist); /* nccl.h.in:193 */
ncclResult_t ncclCommFinalize(ncclComm_t comm);                              /* nccl.h.in:207 */
ncclResult_t ncclCommDestroy(ncclComm_t comm);                               /* nccl.h.in:217 */
ncclResult_t ncclCommAbort(ncclComm_t comm);                                 /* nccl.h.in:228 */
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count);               /* nccl.h.in:295 */
ncclResult_t ncclCommCuDevice(const ncclComm_t comm, int* device);           /* nccl.h.in:306 */
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank);             /* nccl.h.in:317 */
/* Device memory of the type peers may write into (fine-grained, not cached by the owner's kernels). */
ncclResult_t ncclMemAlloc(void** ptr, size_t size);                          /* nccl.h.in:105 */
ncclResult_t ncclMemFree(void* ptr);                                         /* nccl.h.in:108 */
/* Registers a device buffer with the communicator; local call, reference-counted per allocation
 * (src/register.cc:106-160).  An all-reduce whose recvbuff lies in a registered allocation on a rank
 * lets the other ranks write their finished pieces straight into it instead of through the FIFO; so does an
 * all-gather of at least 4 MiB per rank.  Peers write only into fine-grained device memory (ncclMemAlloc, or
 * hipExtMallocWithFlags with the uncached / fine-grained flag): a registration of ordinary hipMalloc memory, and of
 * allocations of 2 GiB and more, stays local and its collectives keep using the FIFOs. */
ncclResult_t ncclCommRegister(const ncclComm_t comm, void* buff, size_t size, void** handle);   /* nccl.h.in:324 */
ncclResult_t ncclCommDeregister(const ncclComm_t comm, void* handle);                           /* nccl.h.in:330 */

/* ---- custom pre-multiply-sum operator --------------------------------------- */
ncclResult_t ncclRedOpCreatePreMulSum(ncclRedOp_t* op, void* scalar, ncclDataType_t datatype,
                                      ncclScalarResidence_t residence, ncclComm_t comm); /* nccl.h.in:412 */
ncclResult_t ncclRedOpDestroy(ncclRedOp_t op, ncclComm_t comm);              /* nccl.h.in:425 */

/* ---- the hot path ----------------------------------------------------------- */
/* Each call returns once the operation is enqueued on `stream` (nccl.h.in:435-438).
 * In place when sendbuff == recvbuff (all-reduce), recvbuff == sendbuff + rank*recvcount
 * (reduce-scatter) or sendbuff == recvbuff + rank*sendcount (all-gather). */
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype,
                           ncclRedOp_t op, ncclComm_t comm, hipStream_t stream);        /* nccl.h.in:519-520 */
ncclResult_t ncclReduceScatter(const void* sendbuff, void* recvbuff, size_t recvcount,
                               ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                               hipStream_t stream);                                      /* nccl.h.in:542-544 */
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount,
                           ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream); /* nccl.h.in:565-566 */
ncclResult_t ncclGroupStart(void);                                           /* nccl.h.in:808 */
ncclResult_t ncclGroupEnd(void);                                             /* nccl.h.in:818 */

/* ---- one-to-all / all-to-one, one hop over the all-pairs links ------------------------------ */
/* recvbuff may be NULL off the root of ncclReduce, sendbuff off the root of ncclBroadcast. */
ncclResult_t ncclReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype,
                        ncclRedOp_t op, int root, ncclComm_t comm, hipStream_t stream);  /* nccl.h.in:458 */
ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype,
                           int root, ncclComm_t comm, hipStream_t stream);               /* nccl.h.in:499 */
ncclResult_t ncclBcast(void* buff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm,
                       hipStream_t stream);                                              /* nccl.h.in:478 */

/* ---- communicator split --------------------------------------------------------------------- */
/* Ranks that pass the same colour form a new communicator, ordered by (key, rank in `comm`); NCCL_SPLIT_NOCOLOR
 * leaves *newcomm NULL.  Collective over `comm`: every rank must call it (one thread that owns several ranks: inside
 * ncclGroupStart/End).  config NULL: the new communicator is blocking or not as its parent is. */
ncclResult_t ncclCommSplit(ncclComm_t comm, int color, int key, ncclComm_t* newcomm,
                           ncclConfig_t* config);                                        /* nccl.h.in:247 */

/* ---- declared by the reference, outside this library's path ----------------- */
/* These exist so that a program using them still links; each returns
 * ncclInvalidUsage and records a message for ncclGetLastError. */
ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer,
                      ncclComm_t comm, hipStream_t stream);                              /* nccl.h.in:587 */
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm,
                      hipStream_t stream);                                               /* nccl.h.in:609 */
ncclResult_t ncclGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, int root,
                        ncclComm_t comm, hipStream_t stream);                            /* nccl.h.in:632 */
ncclResult_t ncclScatter(const void* sendbuff, void* recvbuff, size_t recvcount, ncclDataType_t datatype, int root,
                         ncclComm_t comm, hipStream_t stream);                           /* nccl.h.in:654 */
ncclResult_t ncclAllToAll(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype,
                          ncclComm_t comm, hipStream_t stream);                          /* nccl.h.in:677 */
ncclResult_t ncclAllToAllv(const void* sendbuff, const size_t sendcounts[], const size_t sdispls[], void* recvbuff,
                           const size_t recvcounts[], const size_t rdispls[], ncclDataType_t datatype,
                           ncclComm_t comm, hipStream_t stream);                         /* nccl.h.in:702 */
typedef int mscclAlgoHandle_t;                                                           /* nccl.h.in:717 */
ncclResult_t mscclLoadAlgo(const char* mscclAlgoFilePath, mscclAlgoHandle_t* mscclAlgoHandle, int rank);   /* nccl.h.in:729 */
ncclResult_t mscclRunAlgo(const void* sendBuff, const size_t sendCounts[], const size_t sDisPls[], void* recvBuff,
                          const size_t recvCounts[], const size_t rDisPls[], size_t count, ncclDataType_t dataType,
                          int root, int peer, ncclRedOp_t op, mscclAlgoHandle_t mscclAlgoHandle, ncclComm_t comm,
                          hipStream_t stream);                                           /* nccl.h.in:755 */
ncclResult_t mscclUnloadAlgo(mscclAlgoHandle_t mscclAlgoHandle);                         /* nccl.h.in:776 */


/* ---- profiling aliases ---------------------------------------------------------------------- */
/* The reference exports every public entry point a second time with a "p" prefix (src/include/core.h:18-27,
 * the pncclX declarations in nccl.h.in) so that an interposer can define ncclX and call through.  Same here. */
ncclResult_t pncclGetVersion(int* version);
const char* pncclGetErrorString(ncclResult_t result);
const char* pncclGetLastError(ncclComm_t comm);
ncclResult_t pncclCommGetAsyncError(ncclComm_t comm, ncclResult_t* asyncError);
ncclResult_t pncclGetUniqueId(ncclUniqueId* uniqueId);
ncclResult_t pncclCommInitRankConfig(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank, ncclConfig_t* config);
ncclResult_t pncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t pncclCommInitAll(ncclComm_t* comm, int ndev, const int* devlist);
ncclResult_t pncclCommFinalize(ncclComm_t comm);
ncclResult_t pncclCommDestroy(ncclComm_t comm);
ncclResult_t pncclCommAbort(ncclComm_t comm);
ncclResult_t pncclCommCount(const ncclComm_t comm, int* count);
ncclResult_t pncclCommCuDevice(const ncclComm_t comm, int* device);
ncclResult_t pncclCommUserRank(const ncclComm_t comm, int* rank);
ncclResult_t pncclMemAlloc(void** ptr, size_t size);
ncclResult_t pncclMemFree(void* ptr);
ncclResult_t pncclCommRegister(const ncclComm_t comm, void* buff, size_t size, void** handle);
ncclResult_t pncclCommDeregister(const ncclComm_t comm, void* handle);
ncclResult_t pncclRedOpCreatePreMulSum(ncclRedOp_t* op, void* scalar, ncclDataType_t datatype, ncclScalarResidence_t residence, ncclComm_t comm);
ncclResult_t pncclRedOpDestroy(ncclRedOp_t op, ncclComm_t comm);
ncclResult_t pncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream);
ncclResult_t pncclReduceScatter(const void* sendbuff, void* recvbuff, size_t recvcount, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream);
ncclResult_t pncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
ncclResult_t pncclGroupStart(void);
ncclResult_t pncclGroupEnd(void);
ncclResult_t pncclReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root, ncclComm_t comm, hipStream_t stream);
ncclResult_t pncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream);
ncclResult_t pncclBcast(void* buff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream);
ncclResult_t pncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t pncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t pncclCommSplit(ncclComm_t comm, int color, int key, ncclComm_t* newcomm, ncclConfig_t* config);
ncclResult_t pncclGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream);
ncclResult_t pncclScatter(const void* sendbuff, void* recvbuff, size_t recvcount, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream);
ncclResult_t pncclAllToAll(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
ncclResult_t pncclAllToAllv(const void* sendbuff, const size_t sendcounts[], const size_t sdispls[], void* recvbuff, const size_t recvcounts[], const size_t rdispls[], ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
ncclResult_t pmscclLoadAlgo(const char* mscclAlgoFilePath, mscclAlgoHandle_t* mscclAlgoHandle, int rank);
ncclResult_t pmscclRunAlgo(const void* sendBuff, const size_t sendCounts[], const size_t sDisPls[], void* recvBuff, const size_t recvCounts[], const size_t rDisPls[], size_t count, ncclDataType_t dataType, int root, int peer, ncclRedOp_t op, mscclAlgoHandle_t mscclAlgoHandle, ncclComm_t comm, hipStream_t stream);
ncclResult_t pmscclUnloadAlgo(mscclAlgoHandle_t mscclAlgoHandle);

/* ---- extensions (no counterpart in nccl.h.in) -------------------------------- */

/* Local N-source reduce-copy on the current device: dst[i] = src[0][i] (op) ... (op) src[nsrc-1][i],
 * folded left to right in element type `datatype`.  This is the kernel every reducing hop runs
 * (the role of reduceCopy, src/device/common_kernel.h:185-265); exported for parity tests and for
 * measuring it against the HBM roofline.  1 <= nsrc <= 8.  Enqueued on `stream`. */
ncclResult_t rcclAmdReduceCopy(void* dst, const void* const* srcs, int nsrc, size_t count,
                               ncclDataType_t datatype, ncclRedOp_t op, hipStream_t stream);

/* All-reduce that allocates nothing.  It has no counterpart in the reference: it is for callers that cannot
 * allocate (code under HIP graph capture rules that forbid hipMalloc, tight serving loops, harnesses that forbid
 * allocation) and therefore cannot create a communicator, whose window ncclCommInitAll / ncclCommInitRank
 * allocate.  Same arithmetic as ncclAllReduce on a communicator made by ncclCommInitAll(devices): rank r's input
 * is sendbuffs[r], its result goes to recvbuffs[r] (may equal sendbuffs[r]), contributions are folded in rank
 * order, so results are bit-identical to ncclAllReduce's.  The kernel of rank r is enqueued on streams[r].
 * The staging lines live in a static 8 MiB array that every GPU the library runs on holds until the process ends.
 * Limits:
 *  - all ranks in this process and ON ONE DEVICE: devices[r] must all be equal, anything else returns
 *    ncclInvalidUsage.  (The static array is ordinary device memory of the library's code object, which a peer GPU
 *    must not store into; between devices use ncclAllReduce on a communicator.)
 *  - 1 <= nranks <= 4 (more than 8 returns ncclInvalidArgument, 5 to 8 ncclInvalidUsage: that many kernels of one
 *    process do not run at the same time on one device, see the streams below).
 *  - count * sizeof(datatype) <= 256 KiB per rank (larger counts return ncclInvalidArgument: use ncclAllReduce);
 *    built-in operators only (ncclSum .. ncclAvg).
 *  - The n kernels wait for one another, so all of them must be running at once: every rank needs a stream of its
 *    own (two ranks naming one stream, NULL included, return ncclInvalidUsage), and the streams must sit on different
 *    hardware queues.  HIP gives a process GPU_MAX_HW_QUEUES (default 4) hardware queues per stream priority and
 *    deals streams onto them in turn, so up to 2 ranks any two streams will do; for 3 or 4, create the streams with
 *    hipStreamCreateWithPriority spread over the priority levels.  Ranks whose kernels end up behind one another on one queue wait for their time limit
 *    (RCCL_AMD_SPIN_TIMEOUT_MS, default 120 s) and leave their receive buffers unwritten.
 *  - Every call uses the device's one array: two calls whose kernels could run at the same time race, so a call
 *    must be ordered (stream order or a synchronisation) after the previous call has finished on all its streams.
 *    A call is not meant to be replayed from a captured graph (it carries a fresh call number in its arguments).
 * The call returns once the kernels are enqueued.  A kernel that gave up waiting reports through
 * rcclAmdNoAllocAsyncError; when the launch of one rank fails the ranks already launched are told to leave at once
 * and the call returns ncclUnhandledCudaError.  The first call on a device looks up that device's arrays
 * (hipGetSymbolAddress); later calls make no HIP calls besides hipSetDevice and the kernel launches. */
ncclResult_t rcclAmdAllReduceNoAlloc(const void* const* sendbuffs, void* const* recvbuffs, const int* devices,
                                     int nranks, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                                     const hipStream_t* streams);

/* Asynchronous result of the rcclAmdAllReduceNoAlloc calls that ran on `device` (the role ncclCommGetAsyncError has for
 * a communicator): *asyncError = ncclSuccess, or ncclRemoteError once a kernel of such a call gave up waiting for its
 * peers and left its receive buffer incomplete (sticky).  Synchronous (a 4-byte copy from the device); allocates
 * nothing. */
ncclResult_t rcclAmdNoAllocAsyncError(int device, ncclResult_t* asyncError);

/* Ring schedule synthesis for a single node whose GPUs are joined by direct xGMI links
 * (the job of the ring search + pre-baked models, src/graph/search.cc:919-986,
 * src/graph/rome_models.cc:549-560).  `adj` is an ngpus x ngpus 0/1 matrix of direct links.
 * Writes up to maxRings rings of ngpus ranks each into `rings` (row-major) and their number into
 * *nrings.  Every ring is a Hamiltonian cycle over links of `adj`; over the whole set each directed
 * link is used the same number of times when such a set exists. */
ncclResult_t rcclAmdTopoRings(int ngpus, const int* adj, int maxRings, int* rings, int* nrings);

/* Parses a topology XML in the reference's format (the .xml files under tools/topo_expl/models, written by
 * src/graph/xml.cc) and fills the ngpus x ngpus direct-xGMI adjacency matrix.  `adj` must hold
 * maxGpus*maxGpus ints; *ngpus receives the GPU count. */
ncclResult_t rcclAmdTopoParseXml(const char* xmlText, int maxGpus, int* adj, int* ngpus);

/* Link table -> adjacency (host logic only): linkType / hops are ngpus x ngpus as hipExtGetLinkTypeAndHopCount reports
 * them (type 4 = xGMI); adj[a][b] = 1 when both directions of the pair are one xGMI hop, the links the reference turns
 * into <xgmi> edges (src/graph/xml.cc:780-791); *allDirect = 1 when every pair is, which is when the all-pairs schedule
 * is the default (otherwise the rings over the links that exist are). */
ncclResult_t rcclAmdTopoFromLinks(int ngpus, const unsigned* linkType, const unsigned* hops, int* adj, int* allDirect);

/* Bytes of the window a communicator of `nranks` ranks allocates per rank for `channels` channels with the default slot
 * and exchange settings (environment included), after it has been fitted into the size budget
 * (RCCL_AMD_WINDOW_MAX_BYTES, default 1.5 GiB): total, FIFO storage, exchange region, channels kept.  Host logic only
 * (cf. the reference's per-connection buffers, src/transport/p2p.cc:447-450, src/init.cc:837-875). */
ncclResult_t rcclAmdWindowBytesFor(int nranks, int channels, size_t* totalBytes, size_t* fifoBytes, size_t* exchangeBytes,
                                   int* channelsOut);

/* What the planner would choose for a collective on `comm`: algorithm (0 direct all-pairs,
 * 1 ring, 2 chain tree), protocol (0 simple, 1 low-latency), channels, chunk elements.
 * coll: 0 all-reduce, 1 reduce-scatter, 2 all-gather. */
ncclResult_t rcclAmdPlanInfo(ncclComm_t comm, int coll,
                             size_t count, ncclDataType_t datatype, int* algo, int* proto,
                             int* nChannels, size_t* chunkElems);

/* The same plan computed from explicit settings, without a communicator or a GPU (host logic only):
 * algo < 0 = default choice; maxChannels / slotBytes / minChunkBytes as a communicator would hold
 * them (defaults 16, 524288, 32768).  grid = workgroups the launch would use. */
ncclResult_t rcclAmdPlanForConfig(int nranks, int coll, size_t count, ncclDataType_t datatype, int algo,
                                  int maxChannels, size_t slotBytes, size_t minChunkBytes, int* algoOut,
                                  int* nChannels, size_t* chunkElems, size_t* lastChunkElems, int* grid);

/* Which protocol and schedule a collective of `count` elements takes for explicit settings (host logic only):
 * allowLL 0 = NCCL_PROTO=Simple, 1 = default, 2 = NCCL_PROTO=LL named alone; oneShotMaxBytes / twoExchangeMaxBytes =
 * RCCL_AMD_ONESHOT_MAX_BYTES / RCCL_AMD_TWO_EXCHANGE_MAX_BYTES as a communicator would hold them.
 * proto: 0 = FIFO schedule, 1 = one-hop line protocol, 2 = one exchange, 3 = two exchanges. */
ncclResult_t rcclAmdPlanProtoForConfig(int nranks, int coll, size_t count, ncclDataType_t datatype, int algo, int allowLL,
                                       size_t oneShotMaxBytes, size_t twoExchangeMaxBytes, int* algoOut, int* proto);

/* Counters of the all-reduce gather phase on this rank since the communicator was created: pieces written
 * straight into a peer's registered receive buffer, and pieces sent through the FIFO.  Synchronous. */
ncclResult_t rcclAmdCommStats(ncclComm_t comm, uint64_t* directPieces, uint64_t* fifoPieces);

/* Measurement aid (RCCL_AMD_TUNE_FLAGS=32): 100 MHz ticks thread 0 of every collective workgroup spent, summed over
 * workgroups and launches, per role (3 roles x {waiting for peers, reduce-copy body, draining memory operations,
 * fence and posting, number of steps}); ticks must hold 15 values. */
ncclResult_t rcclAmdCommStepTimers(ncclComm_t comm, uint64_t* ticks);
/* Host-staged (TCP) transport only: where the two proxy threads of this rank have spent their time since the
 * communicator was made, 12 values: tx thread ns in send() for slot data, ns in send() for credits, ns in passes that
 * found nothing to ship, ns in total, data messages, payload bytes, channel scans, scans skipped because the channel's
 * stamp had not changed; rx thread ns in recv() for slot data, ns in poll(), ns in total, data messages.  All zero on a
 * communicator that does not use the transport.  (The reference has no such query; its proxy progress is
 * src/transport/net.cc:1123-1350 with the profiler hooks of src/proxy.cc.) */
ncclResult_t rcclAmdCommNetStats(ncclComm_t comm, uint64_t* out);

/* Rendezvous check without a GPU: joins the endpoint named by `id` as `rank` of `nranks`, all-gathers
 * `bytes` bytes per rank into `all`, and leaves.  It is the exchange ncclCommInitRank performs for the
 * window handles (role of bootstrapAllGather, src/bootstrap.cc), callable from CPU-only tests. */
ncclResult_t rcclAmdBootstrapAllGather(const ncclUniqueId* id, int rank, int nranks, const void* mine,
                                       void* all, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* RCCL_AMD_H_ */

// Structures shared by the host planner and the HIP kernels.
//
// Vocabulary
//   window   : one fine-grained device allocation per rank that every peer maps
//              (HIP IPC or direct pointer).  It holds that rank's receive FIFOs and all
//              flag words other ranks write.  Role of the reference's per-connection
//              ncclRecvMem/ncclSendMem + SIMPLE buffers (src/include/comm.h:44-68,
//              src/transport/p2p.cc:212-246,490-561), folded into one region.
//   channel  : an independent slice of the work driven by its own workgroups
//              (reference: one workgroup per channel, src/device/common.h:230-379).
//   conn     : FIFO (channel, sender, receiver, kind).  Slots live in the receiver's
//              window; `tail` (slots published) sits beside them and is written by the
//              sender; `head` (slots consumed) sits in the SENDER's window and is written
//              by the receiver.  Both ends therefore poll local memory only
//              (reference protocol: src/device/prims_simple.h:111-182).
//   kind     : 0 = reduce phase traffic (scatter / ring / tree-up),
//              1 = broadcast phase traffic (gather / tree-down).
#pragma once
#include <stddef.h>
#include <stdint.h>

#define RA_MAX_RANKS 16
#define RA_MAX_CHANNELS 128
#define RA_NKINDS 2
#define RA_FLAG_STRIDE 128  // bytes between two flag words: one writer per line
#define RA_MAX_SRCS 8   // sources / destinations of one reduce-copy

// collectives
enum { RA_COLL_ALLREDUCE = 0, RA_COLL_REDUCESCATTER = 1, RA_COLL_ALLGATHER = 2, RA_COLL_BROADCAST = 3, RA_COLL_REDUCE = 4,
       RA_NUM_COLLS = 5 };
// algorithms
enum { RA_ALGO_DIRECT = 0, RA_ALGO_RING = 1, RA_ALGO_TREE = 2 };
// device reduction ops (reference ncclDevRedOp_t, src/include/device.h; built by
// hostToDevRedOp, src/enqueue.cc:1962-2057)
enum { RA_OP_SUM = 0, RA_OP_PROD = 1, RA_OP_MAX = 2, RA_OP_MIN = 3, RA_OP_PREMULSUM = 4, RA_OP_SUMPOSTDIV = 5,
       RA_NUM_OPS = 6 };
// device error codes written to RaDevComm::errorFlag
enum { RA_ERR_NONE = 0, RA_ERR_TIMEOUT = 1, RA_ERR_ABORTED = 2 };

struct RaWindowLayout {
  int nRanks;
  int nChannels;      // channels the window has room for
  int nSlots;         // FIFO depth (reference NCCL_STEPS = 8, src/include/device.h:32)
  uint32_t slotBytes; // bytes per slot (reference step size 512 KiB, src/init.cc:839)
  size_t dataOffset;  // start of the slot storage inside the window
  size_t llOffset;    // start of the low-latency line buffers (2 buffers x nRanks senders)
  size_t llBytes;     // bytes of one sender's line buffer (16 B per 8 B of payload)
  // One-exchange all-reduce (mid-size payloads): plain 16-byte packs, 2 buffers x nRanks senders x osBytes, and one
  // flag line per (buffer, workgroup, sender) in front of them.  osBytes 0: not available.
  size_t osFlagOffset;
  size_t osDataOffset;
  size_t osBytes;     // payload one sender's region has room for
  size_t totalBytes;
};

// Flag word index -> byte offset.  which: 0 = tail (written by the sender `peer`),
// 1 = head (written by the receiver `peer`), 2 + s = byte count of slot s of the FIFO from `peer`
// (written by the sender; only the host-staged transport reads it, to know how much to ship).
__host__ __device__ inline size_t raFlagOffset(const RaWindowLayout& l, int which, int ch, int peer, int kind) {
  return ((((size_t)which * l.nChannels + ch) * l.nRanks + peer) * RA_NKINDS + kind) * RA_FLAG_STRIDE;
}
__host__ __device__ inline size_t raFlagBytes(const RaWindowLayout& l) {
  return (size_t)(2 + l.nSlots) * l.nChannels * l.nRanks * RA_NKINDS * RA_FLAG_STRIDE;
}
// Slot storage of the FIFO on which `sender` sends to the window's `owner`.  A window holds no FIFO from its owner to
// itself: nRanks - 1 senders per channel, the owner skipped (an eighth of the FIFO storage at 8 ranks, half at 2).
__host__ __device__ inline size_t raDataOffset(const RaWindowLayout& l, int ch, int owner, int sender, int kind) {
  const int ix = sender < owner ? sender : sender - 1;
  return l.dataOffset + ((((size_t)ch * (l.nRanks - 1) + ix) * RA_NKINDS + kind) * l.nSlots) * (size_t)l.slotBytes;
}
__host__ __device__ inline size_t raDataBytes(const RaWindowLayout& l) {
  return (size_t)l.nChannels * (l.nRanks > 1 ? l.nRanks - 1 : 1) * RA_NKINDS * l.nSlots * (size_t)l.slotBytes;
}

// Low-latency protocol (reference: src/device/prims_ll.h:137-260): 16-byte lines
// {data lo, flag, data hi, flag}; the flag is the launch epoch, so data and "ready" arrive in one
// store and the receiver needs no separate flag round trip.
#define RA_LL_MAX_BYTES (1u << 20)     // payload one sender's line buffer has room for
#define RA_LL_DEFAULT_BYTES (512u << 10)   // largest payload that takes the low-latency protocol unless RCCL_AMD_LL_MAX_BYTES says otherwise
#define RA_LL_GRID 16                  // workgroups of a low-latency launch when ranks share a GPU (8 ranks x 16 stay resident)
#define RA_LL_MAX_GRID 32              // ... and when every rank has its own GPU; fixed per communicator (each workgroup keeps its own epoch)
#define RA_LL_THREADS 1024

// One-exchange all-reduce: workgroups of a launch (fixed per communicator, every rank the same), 1024 threads each.
#define RA_OS_MAX_GRID 128
__host__ __device__ inline size_t raOsFlagOffset(const RaWindowLayout& l, int buf, int wg, int sender) {
  return l.osFlagOffset + (((size_t)buf * RA_OS_MAX_GRID + wg) * l.nRanks + sender) * RA_FLAG_STRIDE;
}
__host__ __device__ inline size_t raOsDataOffset(const RaWindowLayout& l, int buf, int sender) {
  return l.osDataOffset + ((size_t)buf * l.nRanks + sender) * l.osBytes;
}

// The all-reduce that allocates nothing (rcclAmdAllReduceNoAlloc): the one-shot line protocol above with its line
// buffers in a static array every GPU holds, instead of a communicator's window.
#define RA_NA_MAX_BYTES (256u << 10)   // largest payload per rank
#define RA_NA_MAX_RANKS 8
#define RA_NA_MAX_RANKS_PER_DEVICE 4     // ranks of one process whose kernels can run at the same time on one device
#define RA_NA_GRID 16
#define RA_NA_SENDER_BYTES ((size_t)RA_NA_MAX_BYTES * 2)                    // a 16-byte line per 8 payload bytes
#define RA_NA_SCRATCH_BYTES (2 * RA_NA_MAX_RANKS * RA_NA_SENDER_BYTES)      // two alternating buffers: 8 MiB per GPU
struct RaNaArgs {
  char* lines[RA_NA_MAX_RANKS];   // every rank's line array (its GPU's static array)
  const void* send;
  void* recv;
  size_t count;
  uint64_t opArg;                 // an average's divisor (integers) or 1/n in the element type (floating point)
  uint64_t timeoutTicks;
  uint32_t epoch;                 // the same number on every rank of one call, larger than any earlier call's
  int rank, nRanks;
  uint32_t* words;                // this GPU's status words (RA_NA_WORD_*): how a kernel reports that it gave up, and how the
                                  // host tells kernels already launched to leave when a later launch of the call failed
};
enum { RA_NA_WORD_FAILED = 0,     // call number of the first call one of whose kernels gave up on this GPU (0: none)
       RA_NA_WORD_CANCEL = 1,     // call number whose kernels must stop waiting at once
       RA_NA_WORDS = 8 };
// Broadcast and reduce put one presence line per workgroup in the tail of a sender's line buffer; their payloads stay at
// or below RA_LL_DEFAULT_BYTES, so data lines never reach it.
#define RA_LL_TOKEN_LINE ((size_t)RA_LL_MAX_BYTES / 8 - RA_LL_MAX_GRID)
__host__ __device__ inline size_t raLLOffset(const RaWindowLayout& l, int buf, int sender) {
  return l.llOffset + ((size_t)buf * l.nRanks + sender) * l.llBytes;
}

// Registered receive buffers (zero-copy gather of the all-pairs all-reduce).  A rank announces which of its
// registrations its receive buffer lies in with one word: valid bit, board entry, entry serial, byte offset.
#define RA_MAX_REGS 32
struct RaRegSlot { uint64_t base; uint64_t serial; };   // a peer's registered allocation as mapped HERE; base 0 = not mapped
#define RA_REG_VALID (1ull << 63)
__host__ __device__ inline uint64_t raRegWord(int idx, uint32_t serial, uint64_t off) {
  return RA_REG_VALID | ((uint64_t)(idx & 63) << 57) | ((uint64_t)(serial & 0xffffu) << 41) | (off & ((1ull << 41) - 1));
}
#define RA_REG_IDX(w) ((int)(((w) >> 57) & 63))
#define RA_REG_SERIAL(w) (((w) >> 41) & 0xffffu)
#define RA_REG_OFF(w) ((w) & ((1ull << 41) - 1))
#define RA_AUX_DIRECT 0xD1D1D1D1D1D1D1D1ull   // "this launch's pieces were written into your receive buffer"
// All-gathers of at least this many bytes per rank start with the exchange of registration words (one flag
// round trip per launch, under 3 % of the shortest such all-gather); smaller ones always go through the FIFOs.
#define RA_AG_ANNOUNCE_BYTES ((size_t)4 << 20)

struct RaDevComm {
  int rank;
  int nRanks;
  RaWindowLayout lay;
  char* win[RA_MAX_RANKS];    // win[rank] is local; the others are peer mappings
  uint64_t* sendStep;         // [nChannels][nRanks][RA_NKINDS] slots this rank has published
  uint64_t* recvStep;         // [nChannels][nRanks][RA_NKINDS] slots this rank has consumed
  uint32_t* llEpoch;          // [RA_LL_MAX_GRID] launches each low-latency workgroup has run
  uint64_t* osEpoch;          // [RA_OS_MAX_GRID] one-exchange launches each workgroup has run
  uint32_t* abortFlag;        // host-mapped; non-zero asks every spin to give up
  uint32_t* errorFlag;        // host-mapped; first device error code
  uint64_t timeoutTicks;      // per-wait limit in 100 MHz wall-clock ticks
  uint32_t netMode;           // 1: windows are host-pinned staging areas pumped over TCP by a proxy thread
  const RaRegSlot* regTable;  // [RA_MAX_RANKS][RA_MAX_REGS]
  uint64_t* stats;            // [0] pieces written straight into a peer's receive buffer, [1] pieces sent through the FIFO (all-reduce gather phase)
  uint32_t tuneFlags;         // measurement switches (RCCL_AMD_TUNE_FLAGS): 2 = no acquire fence (breaks results), 8 = agent-scope release fence before flag posts, 8 + 4 = system-scope release, 32 = step timers, 64 = steps that consume data dawdle before their body (slow-consumer test)
  // ring r uses order ringRanks[r][0..nRanks-1]; channel c runs ring c % nRings
  int nRings;
  int8_t ringRanks[RA_MAX_CHANNELS][RA_MAX_RANKS];
};

struct RaWork {
  int coll;
  int algo;
  int nChannels;
  int elemSize;
  const char* send;
  char* recv;
  size_t count;        // all-reduce: elements; reduce-scatter / all-gather: elements per rank
  size_t chunkElems;   // elements one owner handles per round (a multiple of 16 bytes)
  size_t lastChunkElems;  // same, for the final partial round
  uint64_t opArg;      // PreMulSum scalar bits / SumPostDiv divisor
  const void* opArgPtr;  // non-null: the PreMulSum scalar lives in device memory and is read when the kernel runs
                         // (ncclScalarDevice; the reference's scalarArgIsPtr, src/enqueue.cc:1962-2057)
  uint32_t stepBytes;  // bytes one FIFO slot carries in this launch (chunkElems * elemSize)
  int root;            // broadcast / reduce
  int pairRoles;       // all-pairs all-reduce: scatter and gather workgroups serve two channels each (see raCollKernel)
  int oneShot;         // all-reduce in one exchange: every rank sends its whole input to every peer and folds all n
                       // contributions itself (raOneShotKernel)
  uint64_t regWord;    // announce word for recv (0: not registered)
  int peerRegs;        // some peer registration is mapped on this rank: read the peers' announce words
};

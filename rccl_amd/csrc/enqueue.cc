// Argument checks, reduction-op translation, planning and launch.
//
// Replaces the reference's enqueue path for the three collectives (src/enqueue.cc:
// 2182-2217 ncclEnqueueCheck, 2075-2180 taskAppend, 1962-2057 hostToDevRedOp,
// 1574-1727 algorithm/channel choice, 1844-1913 chunk sizing, 1434-1446 launch) and the
// group bracket (src/group.cc:96-102,384-448).  One kernel launch per collective per
// rank; the work descriptor travels as a kernel argument instead of through a work FIFO.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <thread>
#include <vector>

#include "comm.h"
#include "debug.h"
#include "launch.h"
#include "reg.h"

size_t raTypeSize(int dt) {
  switch (dt) {
    case ncclInt8: case ncclUint8: case ncclFp8E4M3: case ncclFp8E5M2: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

namespace {

size_t alignUp(size_t x, size_t a) { return (x + a - 1) / a * a; }
size_t divUp(size_t x, size_t a) { return (x + a - 1) / a; }

struct DeviceGuard {
  int old = -1;
  bool changed = false;
  void set(int dev) {
    if (hipGetDevice(&old) == hipSuccess && old != dev) { (void)hipSetDevice(dev); changed = true; }
  }
  ~DeviceGuard() { if (changed) (void)hipSetDevice(old); }
};

uint16_t f32ToF16Bits(float f) {   // IEEE binary16, round to nearest even
  uint32_t x; memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u, mant = x & 0x7fffffu;
  int32_t exp = (int32_t)((x >> 23) & 0xffu);
  if (exp == 0xff) return (uint16_t)(sign | 0x7c00u | (mant ? 0x200u | (mant >> 13) : 0u));
  exp = exp - 127 + 15;
  if (exp >= 0x1f) return (uint16_t)(sign | 0x7c00u);
  if (exp <= 0) {
    if (exp < -10) return (uint16_t)sign;
    mant |= 0x800000u;
    uint32_t shift = (uint32_t)(14 - exp), h = mant >> shift, rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (h & 1u))) h++;
    return (uint16_t)(sign | h);
  }
  uint32_t h = ((uint32_t)exp << 10) | (mant >> 13), rem = mant & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
  return (uint16_t)(sign | h);
}
uint16_t f32ToBf16Bits(float f) {  // as hip_bfloat16
  uint32_t u; memcpy(&u, &f, 4);
  if (~u & 0x7f800000u) u += 0x7fffu + ((u >> 16) & 1u);
  else if (u & 0xffffu) u |= 0x10000u;
  return (uint16_t)(u >> 16);
}

// float -> the reference's 8-bit floats (bias 2^(we-1), 0x80 = NaN, round to nearest even, clipped;
// src/include/rccl_float8.h), used for the average's 1/nranks scalar.
uint8_t f32ToF8Bits(float f, int wm) {
  const int we = 7 - wm, bias = 1 << (we - 1), maxExp = (1 << we) - 1;
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return 0x80;
  const uint32_t sign = (u >> 31) << 7;
  const float a = fabsf(f);
  if (a == 0.0f) return 0;
  int e2; (void)frexpf(a, &e2);
  int E = e2 - 1, expField, mant;
  if (E < 1 - bias) {
    const int q = (int)nearbyintf(ldexpf(a, -(1 - bias - wm)));
    if (q == 0) return 0;
    if (q >= (1 << wm)) { expField = 1; mant = q - (1 << wm); } else { expField = 0; mant = q; }
  } else {
    int q = (int)nearbyintf(ldexpf(a, -(E - wm)));
    if (q == (1 << (wm + 1))) { E++; q = 1 << wm; }
    expField = E + bias; mant = q - (1 << wm);
  }
  if (expField > maxExp) return (uint8_t)(sign | 0x7f);
  return (uint8_t)(sign | ((uint32_t)expField << wm) | (uint32_t)mant);
}

// ncclAvg over n ranks: integers sum and divide, floating point pre-multiplies by 1/n (src/enqueue.cc:1987-2010).
}  // namespace
ncclResult_t raAvgOp(int n, int dtype, int* devOp, uint64_t* arg) {
  switch (dtype) {
    case ncclInt8: case ncclInt32: case ncclInt64: case ncclUint8: case ncclUint32: case ncclUint64:
      *devOp = RA_OP_SUMPOSTDIV; *arg = (uint64_t)n; return ncclSuccess;
    case ncclFloat16: *devOp = RA_OP_PREMULSUM; *arg = f32ToF16Bits(float(1.0 / n)); return ncclSuccess;
    case ncclBfloat16: *devOp = RA_OP_PREMULSUM; *arg = f32ToBf16Bits(float(1.0 / n)); return ncclSuccess;
    case ncclFp8E4M3: *devOp = RA_OP_PREMULSUM; *arg = f32ToF8Bits(float(1.0 / n), 3); return ncclSuccess;
    case ncclFp8E5M2: *devOp = RA_OP_PREMULSUM; *arg = f32ToF8Bits(float(1.0 / n), 2); return ncclSuccess;
    case ncclFloat32: { float s = float(1.0 / n); uint32_t u; memcpy(&u, &s, 4); *devOp = RA_OP_PREMULSUM; *arg = u; return ncclSuccess; }
    case ncclFloat64: { double s = 1.0 / n; uint64_t u; memcpy(&u, &s, 8); *devOp = RA_OP_PREMULSUM; *arg = u; return ncclSuccess; }
    default: return ncclInvalidArgument;
  }
}

namespace {
// ncclRedOp_t + datatype -> device op and its scalar (src/enqueue.cc:1962-2057).
ncclResult_t translateOp(const ncclComm* comm, int op, int dtype, int* devOp, uint64_t* arg, const void** argPtr) {
  *arg = 0; *argPtr = nullptr;
  switch (op) {
    case ncclSum: *devOp = RA_OP_SUM; return ncclSuccess;
    case ncclProd: *devOp = RA_OP_PROD; return ncclSuccess;
    case ncclMax: *devOp = RA_OP_MAX; return ncclSuccess;
    case ncclMin: *devOp = RA_OP_MIN; return ncclSuccess;
    case ncclAvg: return raAvgOp(comm->nRanks, dtype, devOp, arg);
    default: {
      const int ix = op - (int)ncclNumOps;
      if (ix < 0 || ix >= RA_MAX_USER_OPS || !comm->userOps[ix].used) { RA_WARN("invalid reduction operator %d", op); return ncclInvalidArgument; }
      if (comm->userOps[ix].dtype != dtype) {
        RA_WARN("Data type supplied to user-created ncclRedOp_t does not match type given to reduction operation");
        return ncclInvalidArgument;
      }
      *devOp = RA_OP_PREMULSUM; *arg = comm->userOps[ix].arg; *argPtr = comm->userOps[ix].argPtr;
      return ncclSuccess;
    }
  }
}

// ----- group state (thread-local, src/group.cc:19) -------------------------------------------
struct InitJob {
  ncclComm_t* out; int nranks; ncclUniqueId id; int rank; ncclConfig_t config; bool haveConfig; int device;
  ncclComm_t parent; int color, key;      // parent != null: an ncclCommSplit of `parent`
  ncclResult_t result;
};
thread_local int tGroupDepth = 0;
thread_local std::vector<RaCollArgs> tPending;
thread_local std::vector<InitJob> tInits;
thread_local ncclResult_t tGroupError = ncclSuccess;

// Two ranks, all-pairs all-reduce, an even number of channels: the scatter and gather workgroups serve two channels
// each (a reducer step moves twice their bytes), so the launch has 2 x channels workgroups instead of 3 x.
bool pairsRoles(int n, int coll, int algo, int nCh) {
  return n == 2 && coll == RA_COLL_ALLREDUCE && algo == RA_ALGO_DIRECT && nCh >= 2 && nCh % 2 == 0;
}
int gridFor(int n, int coll, int algo, int nCh) {
  if (algo == RA_ALGO_RING) return nCh;
  if (algo == RA_ALGO_TREE) return 2 * nCh;
  if (pairsRoles(n, coll, algo, nCh)) return 2 * nCh;
  return (coll == RA_COLL_ALLREDUCE ? 3 : 2) * nCh;
}

ncclResult_t launchOne(const RaCollArgs& a) {
  ncclComm* c = a.comm;
  const size_t es = raTypeSize(a.dtype);
  int devOp = RA_OP_SUM; uint64_t arg = 0; const void* argPtr = nullptr;
  const bool reduces = a.coll != RA_COLL_ALLGATHER && a.coll != RA_COLL_BROADCAST;
  if (reduces) RA_CHECK(translateOp(c, a.op, a.dtype, &devOp, &arg, &argPtr));
  DeviceGuard guard;
  guard.set(c->device);
  // A device-resident scalar (ncclScalarDevice) travels as a pointer: the kernel reads it when it runs, so nothing
  // here waits for the stream and a captured launch replays with the scalar's value at that time.
  // Operations of one communicator run in issue order even when the streams differ.
  if (c->haveLast && c->lastStream != a.stream) {
    RA_HIP(hipEventRecord(c->orderEvent, c->lastStream));
    RA_HIP(hipStreamWaitEvent(a.stream, c->orderEvent, 0));
  }
  c->lastStream = a.stream; c->haveLast = true;

  if (c->nRanks == 1) {
    // One rank: a copy, or nothing when in place; pre-multiply ops still scale
    // (src/device/onerank.cu:55-90).
    const bool scale = reduces && devOp == RA_OP_PREMULSUM;
    if (a.send == a.recv && !scale) return ncclSuccess;
    const void* srcs[1] = {a.send};
    const size_t bytes = a.count * es;
    int grid = raLocalGrid(bytes, 1);
    hipError_t e = scale ? raLaunchLocal(a.recv, srcs, 1, 0, false, arg, argPtr, a.count, a.dtype, RA_OP_PREMULSUM, grid, a.stream)
                         : raLaunchLocal(a.recv, srcs, 1, -1, false, 0, nullptr, bytes, ncclInt8, RA_OP_SUM, grid, a.stream);
    if (e != hipSuccess) { RA_WARN("local kernel launch failed: %s", hipGetErrorString(e)); return ncclUnhandledCudaError; }
    return ncclSuccess;
  }

  RaPlan p;
  RA_CHECK(raPlan(c, a.coll, a.count, a.dtype, &p));
  RaWork w;
  memset(&w, 0, sizeof w);
  w.coll = a.coll; w.algo = p.algo; w.nChannels = p.nChannels; w.elemSize = (int)es;
  w.send = (const char*)a.send; w.recv = (char*)a.recv; w.count = a.count;
  w.chunkElems = p.chunkElems; w.lastChunkElems = p.lastChunkElems; w.opArg = arg; w.opArgPtr = argPtr;
  w.stepBytes = (uint32_t)(p.chunkElems * es);
  w.root = a.root;
  w.pairRoles = p.pairRoles;
  w.oneShot = p.oneShot;
  // Registered receive buffers: the all-pairs all-reduce at every FIFO size; its ring and chain forms and both
  // all-gathers from RA_AG_ANNOUNCE_BYTES on (the size from which their kernels exchange the registration words).
  if (a.coll == RA_COLL_ALLREDUCE && p.proto == 0 && !c->net && c->nRanks > 1 &&
      (p.algo == RA_ALGO_DIRECT || a.count * es >= RA_AG_ANNOUNCE_BYTES))
    RA_CHECK(raRegPrepare(c, a.recv, a.count * es, a.stream, &w.regWord, &w.peerRegs));
  if (a.coll == RA_COLL_ALLGATHER && p.proto == 0 && !c->net && c->nRanks > 1 && a.count * es >= RA_AG_ANNOUNCE_BYTES)
    RA_CHECK(raRegPrepare(c, a.recv, a.count * es * (size_t)c->nRanks, a.stream, &w.regWord, &w.peerRegs));
  // Reduce-scatter on the all-pairs FIFOs: a registered SEND buffer is read by the peers' reducers directly.
  if (a.coll == RA_COLL_REDUCESCATTER && p.algo == RA_ALGO_DIRECT && p.proto == 0 && !c->net && c->nRanks > 1 &&
      a.count * es >= RA_AG_ANNOUNCE_BYTES)
    RA_CHECK(raRegPrepare(c, a.send, a.count * es * (size_t)c->nRanks, a.stream, &w.regWord, &w.peerRegs));
  RA_TRACE("rank %d coll %d count %zu dtype %d op %d algo %d ch %d chunk %zu last %zu grid %d", c->rank, a.coll, a.count,
           a.dtype, devOp, p.algo, p.nChannels, p.chunkElems, p.lastChunkElems, p.grid);
  hipError_t e;
  if (p.proto == 2 && a.coll == RA_COLL_ALLGATHER) {
    w.count = a.count * es;                                // bytes: the copy is the same for every element type
    w.elemSize = 1;
    e = raLaunchColl(c->devComm, w, ncclInt8, RA_OP_SUM, p.grid, p.block, a.stream);
  } else if (p.proto == 1 && (a.coll == RA_COLL_ALLGATHER || a.coll == RA_COLL_BROADCAST)) {
    w.count = a.count * es;                                // bytes: the copy is the same for every element type
    e = raLaunchLL(c->devComm, w, ncclInt8, RA_OP_SUM, a.stream);
  } else if (p.proto == 1) {
    e = raLaunchLL(c->devComm, w, a.dtype, devOp, a.stream);
  } else {
    e = raLaunchColl(c->devComm, w, a.dtype, devOp, p.grid, p.block, a.stream);
  }
  if (e != hipSuccess) { RA_WARN("collective kernel launch failed: %s", hipGetErrorString(e)); return ncclUnhandledCudaError; }
  return ncclSuccess;
}

ncclResult_t checkArgs(const RaCollArgs& a, const char* name) {
  // src/misc/argcheck.cc:45-84
  RA_CHECK(raCommCheck(a.comm, name));
  if (a.dtype < 0 || a.dtype >= ncclNumTypes) { RA_WARN("%s: invalid type %d", name, a.dtype); return ncclInvalidArgument; }
  if (a.coll != RA_COLL_ALLGATHER && a.coll != RA_COLL_BROADCAST) {
    if (a.op < 0 || (a.op >= ncclNumOps && (a.op - ncclNumOps >= RA_MAX_USER_OPS || !a.comm->userOps[a.op - ncclNumOps].used))) {
      RA_WARN("%s: invalid reduction operation %d", name, a.op);
      return ncclInvalidArgument;
    }
  }
  if (a.coll == RA_COLL_BROADCAST || a.coll == RA_COLL_REDUCE) {
    if (a.root < 0 || a.root >= a.comm->nRanks) { RA_WARN("%s: invalid root %d (root should be in the 0..%d range)", name, a.root, a.comm->nRanks); return ncclInvalidArgument; }
    if (a.comm->nRanks > RA_MAX_SRCS) { RA_WARN("%s: supported up to %d ranks", name, RA_MAX_SRCS); return ncclInvalidUsage; }
  }
  if (a.count == 0) return ncclSuccess;
  // sendbuff may be NULL off the root of a broadcast, recvbuff off the root of a reduce (nccl.h.in:444,488)
  const bool needSend = !(a.coll == RA_COLL_BROADCAST && a.comm->rank != a.root);
  const bool needRecv = !(a.coll == RA_COLL_REDUCE && a.comm->rank != a.root);
  if ((needSend && a.send == nullptr) || (needRecv && a.recv == nullptr)) { RA_WARN("%s: sendbuff/recvbuff is NULL", name); return ncclInvalidArgument; }
  return ncclSuccess;
}

}  // namespace

// Plan: algorithm, channels and chunk.  The reference prices (algorithm, protocol) pairs
// with an analytic model tuned for older parts (src/graph/tuning.cc:368-424); here the
// all-pairs schedule is the default wherever every rank has its own link to every other
// (it moves the same 2(n-1)/n bytes per link pair as a ring in 2 steps instead of 2(n-1)),
// and NCCL_ALGO selects the ring or chain schedules.
ncclResult_t raPlanFor(int n, int coll, size_t count, int dtype, int algoEnv, int maxChannels, size_t slotBytes,
                       size_t minChunkBytes, int blockThreads, int allowLL, size_t oneShotMaxBytes, size_t twoExMaxBytes, RaPlan* p) {
  const size_t es = raTypeSize(dtype);
  if (es == 0 || n < 1 || maxChannels < 1 || slotBytes < 16) return ncclInvalidArgument;
  int algo = algoEnv >= 0 ? algoEnv : RA_ALGO_DIRECT;
  if (coll == RA_COLL_BROADCAST || coll == RA_COLL_REDUCE) algo = RA_ALGO_DIRECT;   // one hop from / to the root
  if (algo == RA_ALGO_TREE && coll != RA_COLL_ALLREDUCE) algo = RA_ALGO_RING;   // no chain schedule for these two
  if (algo == RA_ALGO_DIRECT && n > RA_MAX_SRCS) algo = RA_ALGO_RING;
  p->algo = algo;
  p->proto = 0;
  p->oneShot = 0;
  p->block = blockThreads;
  const size_t unit = 16 / es;                                  // elements per 16-byte pack
  const size_t maxChunk = slotBytes / es / unit * unit;         // one FIFO slot
  size_t minChunk = alignUp(minChunkBytes / es, unit);
  if (minChunk > maxChunk) minChunk = maxChunk;
  // elements each owner handles in total
  size_t perOwner;
  if (coll == RA_COLL_ALLREDUCE && algo != RA_ALGO_TREE) perOwner = divUp(count, (size_t)n);
  else perOwner = count;
  // channels: enough that each has at least one minimum chunk
  size_t nCh = divUp(perOwner, minChunk);
  if (nCh < 1) nCh = 1;
  if (nCh > (size_t)maxChannels) nCh = (size_t)maxChannels;
  // chunk: about four rounds per channel, never above one slot.  Where that gives steps above 256 KiB the pipeline
  // of a channel (scatter -> reduce -> gather) is better fed by eight rounds of at least 256 KiB (1 GiB over 56
  // channels at 8 ranks: 8 rounds of 293 KiB instead of 4.6 of 512 KiB; measured with ranks sharing one GPU, 128 MiB
  // over 8 channels: 1.79 ms against 1.90 ms), while small steps are dominated by their fixed cost and stay at four
  // rounds (16 MiB at 2 ranks: 0.107 ms with 64 KiB steps, 0.164 ms with 32 KiB steps).
  // Whole 1 KiB rows (one 16-byte pack per lane of a wave), so that pieces start on cache-line boundaries
  // (a 349536-byte chunk measured 10 % slower than a 262144-byte one).
  static const size_t rounds = []() { const char* e = getenv("RCCL_AMD_ROUNDS_PER_CHANNEL"); long v = e ? atol(e) : 0; return (size_t)(v > 0 ? v : 0); }();
  size_t chunk;
  if (rounds) {
    chunk = alignUp(divUp(perOwner, nCh * rounds), unit);
  } else if (n == 2) {
    // Two ranks: a step moves one piece, and below a full slot its cost is mostly fixed (flag round trip, waiting for
    // the stores to be acknowledged), so one round with the largest piece that fits beats any pipelining
    // (ranks sharing one GPU: 8 MiB 91 -> 53 us, 16 MiB 104 -> 78 us, 128 MiB 371 -> 344 us; equal from 256 MiB on).
    chunk = alignUp(divUp(perOwner, nCh), unit);
  } else {
    chunk = alignUp(divUp(perOwner, nCh * 4), unit);
    const size_t big = (256u << 10) / es;
    if (chunk > big) {
      chunk = alignUp(divUp(perOwner, nCh * 8), unit);
      if (chunk < big) chunk = big;
    }
  }
  if (chunk * es >= 4096) chunk = alignUp(chunk, 1024 / es);
  // Large steps in whole tiles of the reduce-copy loops (1024 threads x 16 bytes x packs per thread: 64 KiB per source
  // with three sources and more, 64 KiB and 128 KiB with two): what is left over of a step runs one pack per thread.
  const size_t tileBytes = n <= 2 ? (128u << 10) : (64u << 10);
  if (chunk * es >= 4 * tileBytes) chunk = alignUp(chunk, tileBytes / es);
  if (chunk < minChunk) chunk = minChunk;
  if (chunk > maxChunk) {
    // More than a slot per step: as many rounds per channel as that takes, of about the same size (1 MiB per channel
    // and 768 KiB slots: two rounds of 512 KiB, not 768 + 256), in whole 128 KiB tiles of the copy loops: a step that
    // ends in a ragged tile is slower than a slightly larger one that does not (1 GiB, two ranks sharing a GPU:
    // 11 steps of 745 KiB 1.80 ms, 768 KiB steps 1.69 ms).
    const size_t perChannel = divUp(perOwner, nCh);
    const size_t nr = divUp(perChannel, maxChunk);
    chunk = alignUp(divUp(perChannel, nr), (128u << 10) / es);
    if (chunk > maxChunk) chunk = maxChunk;
  }
  p->nChannels = (int)nCh;
  p->chunkElems = chunk;
  p->lastChunkElems = chunk;
  if (coll == RA_COLL_ALLREDUCE && algo != RA_ALGO_TREE && count > 0) {
    const size_t roundElems = chunk * (size_t)n;
    const size_t nRounds = divUp(count, roundElems);
    const size_t rem = count - (nRounds - 1) * roundElems;
    size_t last = alignUp(divUp(rem, (size_t)n), unit);         // cf. src/enqueue.cc:277-297
    if (last > chunk) last = chunk;
    p->lastChunkElems = last;
  }
  p->grid = gridFor(n, coll, algo, p->nChannels);
  p->pairRoles = pairsRoles(n, coll, algo, p->nChannels) ? 1 : 0;
  // Small all-reduces on the all-pairs schedule take the one-shot low-latency protocol
  // (the reference switches to LL below 256-512 KiB, src/graph/tuning.cc:632-671).
  // Reduce-scatter and all-gather take it up to the same payload per rank: like the all-reduce they need every
  // peer's lines to finish, which is the flow control of the two alternating line buffers.  Broadcast and reduce
  // get that property from one presence line per peer and take it up to RA_LL_DEFAULT_BYTES.
  const bool rooted = coll == RA_COLL_BROADCAST || coll == RA_COLL_REDUCE;
  // Limit: measured switch-over with ranks sharing one GPU is above 1 MiB at 2 ranks (23 us against 37 us) and about
  // 1 MiB at 8; the one-shot exchange sends n-1 copies of the payload, so larger groups keep the smaller limit.
  static const long llEnv = []() { const char* e = getenv("RCCL_AMD_LL_MAX_BYTES"); return e ? atol(e) : 0L; }();
  size_t llMax = n <= 4 ? (size_t)RA_LL_MAX_BYTES : (size_t)RA_LL_DEFAULT_BYTES;
  if (llEnv > 0) llMax = llEnv > (long)RA_LL_MAX_BYTES ? (size_t)RA_LL_MAX_BYTES : (size_t)llEnv;
  if (rooted && llMax > (size_t)RA_LL_DEFAULT_BYTES) llMax = (size_t)RA_LL_DEFAULT_BYTES;
  // Where the one-exchange protocol takes over, it does so from 512 KiB: it moves plain packs (half the bytes of the
  // line protocol) and measured 14.7 us against 18 us at 1 MiB (two ranks sharing a GPU), 13 us against 10 us at 512 KiB.
  if (!rooted && llEnv <= 0 && oneShotMaxBytes >= (size_t)RA_LL_MAX_BYTES && llMax > (size_t)RA_LL_DEFAULT_BYTES)
    llMax = (size_t)RA_LL_DEFAULT_BYTES;
  // allowLL 2 (NCCL_PROTO=LL asked for by name): small payloads take the line protocol under NCCL_ALGO=Ring|Tree too.
  // The reference runs its LL primitives along the ring or tree there (src/device/all_reduce.h:358-551, prims_ll.h);
  // here the line protocol exists as the one-hop exchange on the all-pairs links only, so that is what such a launch
  // becomes (same result on every rank, rank-order fold).
  // NCCL_ALGO=Tree with NCCL_PROTO=LL: an all-reduce of up to RA_LL_DEFAULT_BYTES (its presence tokens sit beyond that
  // in the line buffers) runs the line protocol along the chain (raLLChainKernel), folded in the chain's order.
  // NCCL_ALGO=Ring with NCCL_PROTO=LL: the same around the ring (raLLRingKernel), one round of n pieces.
  const bool chainLL = allowLL == 2 && algo == RA_ALGO_TREE && coll == RA_COLL_ALLREDUCE && n <= RA_MAX_RANKS;
  const bool ringLL = allowLL == 2 && algo == RA_ALGO_RING && coll == RA_COLL_ALLREDUCE && n <= RA_MAX_RANKS;
  if ((chainLL || ringLL) && llMax > (size_t)RA_LL_DEFAULT_BYTES) llMax = (size_t)RA_LL_DEFAULT_BYTES;
  if (allowLL && (algo == RA_ALGO_DIRECT || allowLL == 2) && n > 1 && count * es <= llMax) {
    p->algo = chainLL ? RA_ALGO_TREE : ringLL ? RA_ALGO_RING : RA_ALGO_DIRECT;
    if (chainLL) { p->chunkElems = count; p->lastChunkElems = count; }
    if (ringLL) { p->chunkElems = alignUp(divUp(count, (size_t)n), unit); p->lastChunkElems = p->chunkElems; }
    p->proto = 1;
    p->pairRoles = 0;
    p->nChannels = RA_LL_GRID;
    p->grid = RA_LL_GRID;
    p->block = RA_LL_THREADS;
    return ncclSuccess;
  }
  // Mid-size all-reduces on the all-pairs schedule run as ONE exchange (proto 2): every rank pushes its whole input
  // into every peer's window and folds the n contributions itself, in rank order.  The three flag-synchronised phases
  // of the FIFO schedule become one push and one fold, for (n-1) x S link bytes per rank instead of 2(n-1)/n x S:
  // the same bytes with two ranks, twice as many with four, so the limit shrinks as the group grows (comm.cc).
  // Reduce-scatter and all-gather move the same link bytes either way, so they take the exchange whenever a rank's
  // block fits the window region (which is sized by the all-reduce limit).
  // (All-gathers of RA_AG_ANNOUNCE_BYTES and more stay on the FIFOs: that is where registered receive buffers are
  // written directly, and every rank has to choose the protocol from the size alone.)
  const bool exchange = coll == RA_COLL_ALLREDUCE || coll == RA_COLL_REDUCESCATTER ||
                        (coll == RA_COLL_ALLGATHER && count * es < RA_AG_ANNOUNCE_BYTES);
  if (exchange && algo == RA_ALGO_DIRECT && n > 1 && n <= RA_MAX_SRCS && count * es <= oneShotMaxBytes) {
    p->proto = 2;
    p->oneShot = 1;
    p->pairRoles = 0;
    p->nChannels = 64;
    p->grid = 64;
    p->block = 1024;
    p->chunkElems = count;
    p->lastChunkElems = count;
  } else if (coll == RA_COLL_ALLREDUCE && algo == RA_ALGO_DIRECT && n > 2 && n <= RA_MAX_SRCS && count * es <= twoExMaxBytes) {
    // Two exchanges (proto 3): reduce-scatter exchange, fold, all-gather exchange on the same window region and flags;
    // 2(n-1)/n x S link bytes like the three-phase schedule, two flag exchanges instead of three FIFO steps.
    p->proto = 3;
    p->oneShot = 2;
    p->pairRoles = 0;
    p->nChannels = 64;
    p->grid = 64;
    p->block = 1024;
    p->chunkElems = count;
    p->lastChunkElems = count;
  }
  return ncclSuccess;
}

ncclResult_t raPlan(const ncclComm* c, int coll, size_t count, int dtype, RaPlan* p) {
  // Two ranks, large payloads, no NCCL_ALGO: the ring.  With two ranks the ring and the all-pairs schedule move the same
  // bytes over the same link, but the ring's one workgroup per channel walks scatter, reduce and gather of its rounds
  // itself, so no role waits for another to fill or empty its pipeline (two ranks sharing one GPU, all 256 compute units
  // busy either way: all-reduce of 32 MiB 59 us against 71 us, 256 MiB 0.37 ms against 0.45 ms, 1 GiB 1.52 ms against
  // 1.58 ms; reduce-scatter / all-gather of 16 MiB per rank 36 / 42 us against 42 / 51 us; below these sizes the
  // one-exchange protocols win).  RCCL_AMD_TWO_RANK_RING=0 keeps the all-pairs schedule.
  static const bool twoRankRing = []() { const char* e = getenv("RCCL_AMD_TWO_RANK_RING"); return !e || atoi(e) != 0; }();
  int algoEnv = c->algoEnv;
  if (algoEnv < 0 && c->nRanks == 2 && !c->net && twoRankRing) {
    const size_t bytes = count * raTypeSize(dtype);
    if ((coll == RA_COLL_ALLREDUCE && bytes >= (24u << 20)) ||
        ((coll == RA_COLL_REDUCESCATTER || coll == RA_COLL_ALLGATHER) && bytes >= (16u << 20)))
      algoEnv = RA_ALGO_RING;
  }
  RA_CHECK(raPlanFor(c->nRanks, coll, count, dtype, algoEnv, c->maxChannels, c->lay.slotBytes, c->minChunkBytes,
                     c->blockThreads, c->net || !c->allowLL ? 0 : c->explicitLL ? 2 : 1, c->lay.osBytes ? c->oneShotMaxBytes : 0,
                     c->lay.osBytes ? c->twoExMaxBytes : 0, p));
  // RCCL_AMD_FUSED_ROLES=1: the all-pairs all-reduce with one workgroup per channel that walks scatter, reduce and gather
  // itself (directAllReduce, role 3) instead of three workgroups per channel.
  if (c->fusedRoles && coll == RA_COLL_ALLREDUCE && p->algo == RA_ALGO_DIRECT && p->proto == 0 && !p->oneShot && !p->pairRoles && !c->net) {
    p->pairRoles = 2;
    p->grid = p->nChannels;
  }
  if (p->proto == 1) { p->nChannels = c->llGrid; p->grid = c->llGrid; }
  if (p->proto >= 2) { p->nChannels = c->osGrid; p->grid = c->osGrid; }
  return ncclSuccess;
}

int raGroupDepth() { return tGroupDepth; }

ncclResult_t raGroupStart() {
  tGroupDepth++;
  return ncclSuccess;
}

void raGroupAddInit(ncclComm_t* comm, int nranks, const ncclUniqueId& id, int rank, const ncclConfig_t* config, int device) {
  InitJob j;
  j.out = comm; j.nranks = nranks; j.id = id; j.rank = rank; j.haveConfig = config != nullptr; j.device = device;
  if (config) j.config = *config;
  j.parent = nullptr; j.color = 0; j.key = 0;
  j.result = ncclSuccess;
  tInits.push_back(j);
}

void raGroupAddSplit(ncclComm_t parent, int color, int key, ncclComm_t* newcomm, const ncclConfig_t* config) {
  InitJob j;
  memset(&j.id, 0, sizeof j.id);
  j.out = newcomm; j.nranks = 0; j.rank = 0; j.haveConfig = config != nullptr; j.device = parent->device;
  if (config) j.config = *config;
  j.parent = parent; j.color = color; j.key = key;
  j.result = ncclSuccess;
  tInits.push_back(j);
}

// NCCL_LAUNCH_MODE (src/group.cc:128-190).  PARALLEL, the reference's default and this library's: every launch goes out as
// its turn comes.  GROUP: the communicators of one process that belong to one job (siblings: made by one
// ncclCommInitAll, or by ncclCommInitRank calls of one process) launch in rounds, one operation per communicator and
// round, and nobody launches a round before every sibling has arrived for it, whichever threads drive them.
bool launchModeGroup() {
  static const bool group = []() { const char* e = getenv("NCCL_LAUNCH_MODE"); return e && !strcasecmp(e, "GROUP"); }();
  return group;
}

ncclResult_t launchAll(const std::vector<RaCollArgs>& ops) {
  ncclResult_t res = ncclSuccess;
  if (!launchModeGroup()) {
    for (const auto& a : ops) {
      const ncclResult_t r = launchOne(a);
      if (r != ncclSuccess && res == ncclSuccess) res = r;
    }
    return res;
  }
  // the communicators of this call in first-use order, each with its operations in issue order
  std::vector<ncclComm*> comms;
  std::vector<std::vector<const RaCollArgs*>> queue;
  for (const auto& a : ops) {
    size_t i = 0;
    while (i < comms.size() && comms[i] != a.comm) i++;
    if (i == comms.size()) { comms.push_back(a.comm); queue.emplace_back(); }
    queue[i].push_back(&a);
  }
  std::vector<size_t> next(comms.size(), 0);
  std::vector<uint64_t> round(comms.size(), 0);
  for (size_t i = 0; i < comms.size(); i++)
    if (comms[i]->clique) round[i] = comms[i]->clique->barrierIn(1);
  while (true) {
    bool more = false;
    for (size_t i = 0; i < comms.size(); i++) {
      int sum = next[i] < queue[i].size() ? 1 : 0;
      if (comms[i]->clique) RA_CHECK(comms[i]->clique->barrierOut(round[i], &sum));
      if (sum == 0) continue;                            // no sibling has anything left: this communicator is done
      more = true;
      if (next[i] < queue[i].size()) {
        const ncclResult_t r = launchOne(*queue[i][next[i]++]);
        if (r != ncclSuccess && res == ncclSuccess) res = r;
      }
      if (comms[i]->clique) round[i] = comms[i]->clique->barrierIn(next[i] < queue[i].size() ? 1 : 0);
    }
    if (!more) break;
  }
  return res;
}

ncclResult_t raGroupEnd() {
  if (tGroupDepth == 0) { RA_WARN("ncclGroupEnd: not in a group call"); return ncclInvalidUsage; }
  if (--tGroupDepth > 0) return ncclSuccess;
  ncclResult_t res = tGroupError;
  tGroupError = ncclSuccess;
  // Communicator creation inside a group: every rank's init blocks on the others, so each
  // runs on its own thread (src/group.cc:311-349).
  if (!tInits.empty()) {
    std::vector<InitJob> jobs;
    jobs.swap(tInits);
    std::vector<std::thread> th;
    for (auto& j : jobs)
      th.emplace_back([&j]() {
        if (hipSetDevice(j.device) != hipSuccess) { j.result = ncclUnhandledCudaError; return; }
        const ncclConfig_t* cfg = j.haveConfig ? &j.config : nullptr;
        j.result = j.parent ? raCommSplit(j.parent, j.color, j.key, j.out, cfg) : raCommInitRank(j.out, j.nranks, j.id, j.rank, cfg);
      });
    for (auto& t : th) t.join();
    // errors first; otherwise ncclInProgress when a non-blocking communicator is still being set up (src/group.cc:412-435)
    bool pending = false;
    for (auto& j : jobs) {
      if (j.result == ncclInProgress) pending = true;
      else if (j.result != ncclSuccess && res == ncclSuccess) res = j.result;
    }
    if (res == ncclSuccess && pending) res = ncclInProgress;
  }
  std::vector<RaCollArgs> ops;
  ops.swap(tPending);
  if (res != ncclSuccess && res != ncclInProgress) return res;   // an error inside the group cancels its launches
  const ncclResult_t lr = launchAll(ops);
  return res == ncclSuccess ? lr : res;
}

ncclResult_t raEnqueue(const RaCollArgs& a) {
  static const char* names[] = {"AllReduce", "ReduceScatter", "AllGather", "Broadcast", "Reduce"};
  ncclResult_t r = checkArgs(a, names[a.coll]);
  if (r == ncclSuccess && a.comm) {
    r = raCommPollError(a.comm);
    if (r == ncclInProgress) {             // src/init.cc ncclCommEnsureReady
      RA_WARN("%s: attempt to use the communicator before its creation returned ncclSuccess", names[a.coll]);
      r = ncclInvalidArgument;
    } else if (r != ncclSuccess) RA_WARN("%s: the communicator has a pending asynchronous error", names[a.coll]);
  }
  if (r != ncclSuccess) {
    if (tGroupDepth > 0 && tGroupError == ncclSuccess) tGroupError = r;
    return r;
  }
  if (a.count == 0) return ncclSuccess;           // src/enqueue.cc:2078
  RA_INFO("%s: rank %d sendbuff %p recvbuff %p count %zu datatype %d op %d stream %p", names[a.coll], a.comm->rank,
          a.send, a.recv, a.count, a.dtype, a.op, (void*)a.stream);
  if (tGroupDepth > 0) { tPending.push_back(a); return ncclSuccess; }
  if (launchModeGroup()) return launchAll(std::vector<RaCollArgs>(1, a));
  return launchOne(a);
}

// Host-side communicator state.
#pragma once
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "bootstrap.h"
#include "device_types.h"
#include "rccl_amd.h"

#define RA_COMM_MAGIC 0x52434d4d414d4431ull
#define RA_MAX_USER_OPS 64

// Ranks of one communicator that live in the same process reach each other's windows by
// plain pointers (the reference's same-process P2P path, src/transport/p2p.cc:318-344).
// The clique owns those windows so that none is freed while a sibling may still write it.
struct RaClique {
  std::mutex m;
  uint64_t key = 0;
  int nRanks = 0;
  int joined = 0, left = 0;
  std::vector<char*> window;   // per rank, null for ranks of other processes
  std::vector<int> device;
  // All-gather between the ranks of an ncclCommInitAll communicator (they have no rendezvous connection): every
  // rank deposits `bytes`, the last one to arrive releases the round.  Used by ncclCommSplit.
  std::condition_variable xcv;
  std::vector<char> xIn, xOut;
  int xCount = 0;
  uint64_t xRound = 0;
  ncclResult_t gather(int rank, const void* mine, void* all, size_t bytes);
  // Launch barrier of NCCL_LAUNCH_MODE=GROUP (src/group.cc:128-190, ncclCommIntraBarrierIn / Out): every sibling adds a
  // number, the last one to arrive closes the round, everybody reads the round's sum.
  int bCount = 0, bSum = 0, bResult = 0;
  uint64_t bRound = 0;
  uint64_t barrierIn(int x);                           // returns the round this arrival belongs to
  ncclResult_t barrierOut(uint64_t round, int* sum);   // waits for that round to close
  ~RaClique();
};

struct RaNet;
struct RaRegState;

struct RaUserOp {
  bool used = false;
  int dtype = 0;
  uint64_t arg = 0;          // scalar bits (host-immediate)
  const void* argPtr = nullptr;  // device-resident scalar
};

struct ncclComm {
  ncclComm() { for (int i = 0; i < RA_MAX_RANKS * RA_MAX_RANKS; i++) linkAdj[i] = (i / RA_MAX_RANKS) != (i % RA_MAX_RANKS); }
  uint64_t magic = RA_COMM_MAGIC;
  int rank = 0, nRanks = 1, device = 0;
  bool blocking = true;

  // tunables (environment, read once at init)
  int maxChannels = 16;
  int cfgMinCTAs = 0, cfgMaxCTAs = 0;   // ncclConfig_t::minCTAs / maxCTAs of this communicator (0: not given)
  int blockThreads = 512;
  int algoEnv = -1;          // NCCL_ALGO: -1 unset
  bool allowLL = true;       // NCCL_PROTO permits the low-latency protocol
  bool explicitLL = false;   // NCCL_PROTO names LL and nothing else: small payloads take it under NCCL_ALGO=Ring|Tree too
  bool fusedRoles = false;   // RCCL_AMD_FUSED_ROLES: one workgroup per channel for the all-pairs all-reduce
  int llGrid = RA_LL_GRID;   // workgroups of a low-latency launch (the same on every rank of the communicator)
  size_t minChunkBytes = 32 << 10;
  size_t oneShotMaxBytes = 0;  // all-reduces up to this size (and above the low-latency limit) run as one exchange
  size_t twoExMaxBytes = 0;    // ... and up to this size as two exchanges (bandwidth-optimal, more than two ranks)

  // device resources
  RaWindowLayout lay{};
  char* window = nullptr;            // my window (owned by the clique)
  char* peerWin[RA_MAX_RANKS] = {};
  bool peerIsIpc[RA_MAX_RANKS] = {};
  RaDevComm* devComm = nullptr;
  uint64_t* steps = nullptr;         // sendStep + recvStep
  uint32_t* llEpoch = nullptr;       // per low-latency workgroup launch counters (device)
  uint64_t* osEpoch = nullptr;       // per one-exchange workgroup launch counters (device)
  int osGrid = 64;                   // workgroups of a one-exchange launch (the same on every rank)
  uint32_t* hostFlags = nullptr;     // pinned, mapped: [0] abort request, [1] device error
  uint32_t* hostFlagsDev = nullptr;
  RaRegSlot* regTable = nullptr;     // device copy of the peers' registered allocations as mapped here
  uint64_t* stats = nullptr;         // device counters (RaDevComm::stats)
  RaRegState* reg = nullptr;         // registration cache and boards (reg.cc)
  int nRings = 0;
  int8_t ringRanks[RA_MAX_CHANNELS][RA_MAX_RANKS];
  int linkAdj[RA_MAX_RANKS * RA_MAX_RANKS];   // [a * RA_MAX_RANKS + b] = 1: ranks a and b are joined by one xGMI hop (or share a GPU)
  bool allDirect = true;                      // every pair is: the all-pairs schedule is the default

  std::shared_ptr<RaClique> clique;
  RaBootstrap boot;
  bool multiProcess = false;
  bool peersElsewhere = false;       // some rank of this communicator lives in another process
  RaNet* net = nullptr;              // host-staged TCP transport state, when selected

  // launch ordering between user streams
  hipStream_t lastStream = nullptr;
  bool haveLast = false;
  hipEvent_t orderEvent = nullptr;

  ncclResult_t asyncError = ncclSuccess;
  RaUserOp userOps[RA_MAX_USER_OPS];
  bool finalized = false;

  // Non-blocking creation (ncclConfig_t::blocking = 0, src/init.cc:2281-2346, src/group.cc:412-435): the handle is
  // returned at once, the set-up runs on initThread, and initState is ncclInProgress until it has ended.
  std::atomic<int> initState{ncclSuccess};
  std::thread initThread;
};

// comm.cc
ncclResult_t raCommInitAll(ncclComm_t* comms, int ndev, const int* devlist);
ncclResult_t raCommInitRank(ncclComm_t* comm, int nranks, const ncclUniqueId& id, int rank, const ncclConfig_t* config);
ncclResult_t raCommSplit(ncclComm_t comm, int color, int key, ncclComm_t* newcomm, const ncclConfig_t* config);
ncclResult_t raCommDestroy(ncclComm_t comm, bool abort);
ncclResult_t raCommCheck(const ncclComm* comm, const char* what);
ncclResult_t raCommPollError(ncclComm* comm);
ncclResult_t raCommEnsureReady(const ncclComm* comm, const char* what);   // raCommCheck + the creation has ended well

// The window layout a communicator of nRanks ranks gets for *channels channels (updated when the fit reduces them).
void raWindowLayoutFor(int nRanks, int* channels, RaWindowLayout* lay);

// enqueue.cc
struct RaPlan {
  int algo, proto, nChannels, grid, block;
  int pairRoles;             // RaWork::pairRoles
  int oneShot;               // RaWork::oneShot (proto 2)
  size_t chunkElems, lastChunkElems;
};
struct RaCollArgs {
  int coll;
  const void* send;
  void* recv;
  size_t count;
  int dtype;
  int op;                    // ncclRedOp_t value (may be a user op handle)
  ncclComm* comm;
  hipStream_t stream;
  int root = 0;              // broadcast / reduce
};
ncclResult_t raPlan(const ncclComm* comm, int coll, size_t count, int dtype, RaPlan* plan);
ncclResult_t raPlanFor(int nRanks, int coll, size_t count, int dtype, int algoEnv, int maxChannels, size_t slotBytes,
                       size_t minChunkBytes, int blockThreads, int allowLL, size_t oneShotMaxBytes, size_t twoExMaxBytes, RaPlan* plan);
ncclResult_t raEnqueue(const RaCollArgs& a);
ncclResult_t raGroupStart();
ncclResult_t raGroupEnd();
int raGroupDepth();
void raGroupAddInit(ncclComm_t* comm, int nranks, const ncclUniqueId& id, int rank, const ncclConfig_t* config, int device);
void raGroupAddSplit(ncclComm_t parent, int color, int key, ncclComm_t* newcomm, const ncclConfig_t* config);
size_t raTypeSize(int dtype);
// ncclAvg over n ranks in element type dtype: the device operator and its scalar.
ncclResult_t raAvgOp(int n, int dtype, int* devOp, uint64_t* arg);

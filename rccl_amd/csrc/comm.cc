// Communicator lifecycle: windows, peer mappings, device-side tables.
//
// Replaces, for one node, the reference's init path (src/init.cc:1188-1874
// initTransportsRank: peer-info all-gather, transport setup, devCommSetup) and its P2P
// transport setup/connect (src/transport/p2p.cc:354-561).  There is one transport here —
// peer-mapped receive windows over xGMI — so "connect" is: allocate one fine-grained
// window, export it (HIP IPC between processes, plain pointer inside a process), map
// everybody else's, and hand the table to the kernels.
#include "comm.h"

#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stddef.h>
#include <string.h>
#include <strings.h>
#include <unistd.h>

#include <algorithm>
#include <map>
#include <thread>

#include "debug.h"
#include "net.h"
#include "reg.h"
#include "topo.h"

namespace {

std::mutex gCliqueMutex;
std::map<uint64_t, std::weak_ptr<RaClique>> gCliques;

long envLong(const char* name, long def) {
  const char* e = getenv(name);
  if (!e || !*e) return def;
  char* end = nullptr;
  long v = strtol(e, &end, 0);
  return end == e ? def : v;
}

size_t alignUp(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct DeviceGuard {
  int old = -1;
  DeviceGuard() { if (hipGetDevice(&old) != hipSuccess) old = -1; }
  ~DeviceGuard() { if (old >= 0) (void)hipSetDevice(old); }
};

// Limits of the one-exchange and two-exchange all-reduce protocols for this group size.
void exchangeLimits(ncclComm* c) {
  // One-exchange all-reduce (enqueue.cc, proto 2): it sends (n-1) x S per rank instead of 2(n-1)/n x S, so the default
  // limit follows the group size: with two ranks the link bytes are the same and only HBM traffic grows (5 S against
  // 4 S per rank), with four they double, with eight they are four times as many.  Priced against one xGMI link per
  // peer (about 120 GB/s achievable) and the roughly 30 us the two extra flag-synchronised phases of the three-phase
  // schedule cost, the break-even is near 7 MB at four ranks and 5 MB at eight; the defaults stay below that (4 MiB,
  // 2 MiB).  With ranks sharing one GPU (no links, HBM only) one exchange measured faster up to 16 MiB at four ranks
  // and 8 MiB at eight (tools/oneshot_limits.sh).
  {
    const long dflt = c->nRanks == 2 ? (64L << 20) : c->nRanks <= 4 ? (4L << 20) : (2L << 20);
    long v = envLong("RCCL_AMD_ONESHOT_MAX_BYTES", dflt);
    const long most = (256L << 20) / c->nRanks;        // two buffers for every sender: at most 512 MiB of window
    if (v > most) { RA_INFO("RCCL_AMD_ONESHOT_MAX_BYTES=%ld reduced to %ld (window space)", v, most); v = most; }
    c->oneShotMaxBytes = v > 0 ? (size_t)v : 0;
    // Above that, up to RCCL_AMD_TWO_EXCHANGE_MAX_BYTES, an all-reduce of more than two ranks runs as two exchanges:
    // the link bytes of the three-phase schedule without its FIFOs and credits (gradient-bucket sizes; the default
    // needs 2 x 2 x 64 MiB of window).  Larger payloads keep the pipelined FIFO schedule, whose phases overlap on the
    // links; here a workgroup's push, fold and copy follow one another.  With ranks sharing one GPU two exchanges
    // measured 2.5 times faster at 8 MiB, 1.7 times at 32 MiB and still 1.3 times at 128 MiB (8 ranks).
    long t = envLong("RCCL_AMD_TWO_EXCHANGE_MAX_BYTES", c->nRanks > 2 ? (64L << 20) : 0);
    if (t > (128L << 20)) t = 128L << 20;
    c->twoExMaxBytes = (c->nRanks > 2 && t > 0) ? (size_t)t : 0;
  }
}

void readTunables(ncclComm* c) {
  // NCCL_MAX_NCHANNELS / NCCL_MIN_NCHANNELS as in src/graph/connect.cc:521-545; the native
  // knob is RCCL_AMD_NCHANNELS.
  // Default: 8 channels per xGMI link in use, between 32 and 56 (56 = 8 x 7 links at 8 ranks; the
  // reference also ends up with 56 channels on an 8-GPU all-to-all node, src/graph/connect.cc:638-660, and
  // recommends 32 channels for 2 and 24 for 4 MI300X, README.md:194-203).
  long dflt = 8L * (c->nRanks - 1);
  if (dflt < 32) dflt = 32;
  if (dflt > 56) dflt = 56;
  long ch = envLong("RCCL_AMD_NCHANNELS", -1);
  if (ch < 0) ch = envLong("NCCL_MAX_NCHANNELS", dflt);
  long mn = envLong("NCCL_MIN_NCHANNELS", 1);
  // ncclConfig_t::maxCTAs / minCTAs bound the channels of this one communicator (src/init.cc:2091-2100: a workgroup per
  // channel there); a small maxCTAs is how a communicator made by ncclCommSplit is kept cheap (its window follows).
  if (c->cfgMaxCTAs > 0 && ch > c->cfgMaxCTAs) ch = c->cfgMaxCTAs;
  if (c->cfgMinCTAs > 0 && mn < c->cfgMinCTAs) mn = c->cfgMinCTAs;
  if (ch < mn) ch = mn;
  if (ch < 1) ch = 1;
  if (ch > RA_MAX_CHANNELS) ch = RA_MAX_CHANNELS;
  // Every workgroup of a launch waits for workgroups on other GPUs, so all of them must be resident at once:
  // one 1024-thread workgroup per compute unit, three roles per channel in the all-reduce (with two ranks the
  // scatter and gather workgroups serve two channels each: two workgroups per channel).
  int cus = 0;
  const int perChannel = c->nRanks == 2 ? 2 : 3;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && cus >= perChannel &&
      ch > cus / perChannel) {
    RA_INFO("device %d has %d compute units: %ld channels reduced to %d", c->device, cus, ch, cus / perChannel);
    ch = cus / perChannel;
  }
  (void)hipGetLastError();
  c->maxChannels = (int)ch;
  long nt = envLong("RCCL_AMD_NTHREADS", envLong("NCCL_NTHREADS", 1024));
  c->blockThreads = nt <= 256 ? 256 : nt <= 512 ? 512 : 1024;
  c->fusedRoles = envLong("RCCL_AMD_FUSED_ROLES", 0) != 0;
  c->minChunkBytes = (size_t)envLong("RCCL_AMD_MIN_CHUNK_BYTES", 32 << 10);
  if (c->minChunkBytes < 1024) c->minChunkBytes = 1024;
  exchangeLimits(c);
  // NCCL_PROTO (src/graph/tuning.cc:476-485): "Simple" keeps small all-reduces off the low-latency path
  const char* pr = getenv("NCCL_PROTO");
  c->allowLL = !(pr && *pr && !strcasestr(pr, "LL"));
  c->explicitLL = pr && *pr && strcasestr(pr, "LL") && !strcasestr(pr, "Simple");
  c->algoEnv = -1;
  const char* a = getenv("NCCL_ALGO");    // src/graph/tuning.cc:476-485
  if (a && *a) {
    if (!strcasecmp(a, "ring")) c->algoEnv = RA_ALGO_RING;
    else if (!strcasecmp(a, "tree")) c->algoEnv = RA_ALGO_TREE;
    else if (!strcasecmp(a, "direct")) c->algoEnv = RA_ALGO_DIRECT;
    else RA_WARN("NCCL_ALGO=%s is not one of Ring, Tree, Direct: ignored", a);
  }
}

// Lays the window out for the communicator's channel count, slot geometry and exchange limits.
void layoutFor(ncclComm* c, long slots, long slotBytes, bool staged) {
  RaWindowLayout& l = c->lay;
  l.nRanks = c->nRanks;
  l.nChannels = c->maxChannels;
  l.nSlots = (int)slots;
  l.slotBytes = (uint32_t)alignUp((size_t)slotBytes, 4096);
  l.dataOffset = alignUp(raFlagBytes(l), 4096);
  l.llOffset = alignUp(l.dataOffset + raDataBytes(l), 4096);          // one past the last FIFO
  l.llBytes = (size_t)RA_LL_MAX_BYTES * 2;                          // 16-byte line per 8 bytes of payload
  l.osFlagOffset = alignUp(raLLOffset(l, 2, 0), 4096);
  // one sender's share of a buffer: a whole payload of the one-exchange limit, or two pieces (one per exchange) of
  // the two-exchange limit
  size_t share = c->oneShotMaxBytes;
  if (c->twoExMaxBytes) share = std::max(share, 2 * (c->twoExMaxBytes / (size_t)c->nRanks + 2048));
  l.osBytes = staged ? 0 : alignUp(share, 4096);
  l.osDataOffset = l.osFlagOffset + (l.osBytes ? alignUp((size_t)2 * RA_OS_MAX_GRID * l.nRanks * RA_FLAG_STRIDE, 4096) : 0);
  l.totalBytes = raOsDataOffset(l, 2, 0);
}

void makeLayout(ncclComm* c) {
  // Two deep, 768 KiB slots between GPUs: the roles of a channel run in different workgroups, so one slot being
  // filled while the other is drained is all the overlap there is to have, and a larger step spreads the per-step
  // costs (measured: 1 GiB all-reduce, ranks sharing one GPU, 4 x 256 KiB -> 2 x 512 KiB: 2.81 -> 2.36 ms at 2 ranks,
  // 12.4 -> 11.9 ms at 8; 512 KiB -> 768 KiB with the steps of a channel balanced: 1.73 -> 1.68 ms at 2 ranks,
  // 5.53 -> 5.09 ms at 4, 10.85 -> 10.46 ms at 8; 1 MiB slots are slower again, 1.81 ms at 2 ranks).  Above 8 ranks
  // 512 KiB.  The host-staged transport keeps 4 x 256 KiB: its proxy threads want more steps in flight.
  const bool staged = raNetRequested() && c->multiProcess;
  long slots = envLong("RCCL_AMD_NSLOTS", staged ? 4 : 2);
  if (slots < 2) slots = 2;
  if (slots > 16) slots = 16;
  long slotBytes = envLong("RCCL_AMD_SLOT_BYTES", -1);
  const bool slotsFromEnv = slotBytes > 0 || envLong("NCCL_BUFFSIZE", -1) > 0;
  if (slotBytes <= 0) {
    long buff = envLong("NCCL_BUFFSIZE", -1);      // whole FIFO, as in src/init.cc:837-875
    slotBytes = buff > 0 ? buff / slots : (staged ? (256 << 10) : c->nRanks <= 8 ? (768 << 10) : (512 << 10));
  }
  if (slotBytes < 4096) slotBytes = 4096;
  layoutFor(c, slots, slotBytes, staged);
  // Head-room.  A window shared between processes has to stay below the 2 GiB an IPC mapping is good for (comm.cc
  // refuses more), and every communicator of a process holds one, so the default geometry is fitted into a budget
  // (RCCL_AMD_WINDOW_MAX_BYTES, default 1.5 GiB) whatever channel count was asked for: the two-exchange region
  // shrinks first (down to 16 MiB payloads), then the one-exchange limit (down to 1 MiB), then the slots in 64 KiB
  // steps (down to 256 KiB; not when the environment fixed them), last the channels.  Every rank does the same sums.
  long budget = envLong("RCCL_AMD_WINDOW_MAX_BYTES", 1536L << 20);
  if (budget > (2047L << 20)) budget = 2047L << 20;
  if (budget < (64L << 20)) budget = 64L << 20;
  const size_t before = c->lay.totalBytes;
  while (c->lay.totalBytes > (size_t)budget) {
    if (c->twoExMaxBytes > ((size_t)16 << 20)) c->twoExMaxBytes /= 2;
    else if (c->oneShotMaxBytes > ((size_t)1 << 20)) c->oneShotMaxBytes /= 2;
    else if (!slotsFromEnv && slotBytes > (256 << 10)) slotBytes -= 64 << 10;
    else if (c->maxChannels > 8) c->maxChannels--;
    else break;
    layoutFor(c, slots, slotBytes, staged);
  }
  if (c->lay.totalBytes != before)
    RA_INFO("rank %d: window fitted into %ld MiB: %zu -> %zu MiB (two-exchange limit %zu MiB, one-exchange limit %zu KiB, "
            "slots %u KiB, %d channels)", c->rank, budget >> 20, before >> 20, c->lay.totalBytes >> 20, c->twoExMaxBytes >> 20,
            c->oneShotMaxBytes >> 10, c->lay.slotBytes >> 10, c->maxChannels);
}

// Low-latency launches use 32 workgroups when every rank has a GPU to itself and 16 when ranks share one (8 ranks x 16
// workgroups of 1024 threads still fit a GPU together); RCCL_AMD_LL_GRID overrides.  Every rank decides from the same data.
void setLLGrid(ncclComm* c, bool shared) {
  long g = envLong("RCCL_AMD_LL_GRID", shared ? RA_LL_GRID : RA_LL_MAX_GRID);
  if (g < 1) g = 1;
  if (g > RA_LL_MAX_GRID) g = RA_LL_MAX_GRID;
  c->llGrid = (int)g;
  // One-exchange launches: a workgroup that waits for its peer keeps its compute unit, so when ranks share a GPU all of
  // their workgroups must fit it together (128 in all); a rank with a GPU to itself uses 128.
  long og = envLong("RCCL_AMD_ONESHOT_GRID", shared ? 128 / c->nRanks : 128);
  if (og < 1) og = 1;
  if (og > RA_OS_MAX_GRID) og = RA_OS_MAX_GRID;
  c->osGrid = (int)og;
}

uint64_t busIdOf(int device) {
  char id[64] = "";
  if (hipDeviceGetPCIBusId(id, (int)sizeof id, device) != hipSuccess) { (void)hipGetLastError(); return (uint64_t)device + 1; }
  uint64_t h = 1469598103934665603ull;
  for (const char* p = id; *p; p++) h = (h ^ (uint64_t)(unsigned char)*p) * 1099511628211ull;
  return h;
}

ncclResult_t allocWindow(ncclComm* c) {
  if (c->nRanks == 1) return ncclSuccess;
  const char* kind = getenv("RCCL_AMD_WINDOW_MEM");
  void* p = nullptr;
  hipError_t e = hipErrorUnknown;
  // Same preference as the reference's FIFO allocation: uncached, else fine-grained
  // (src/include/alloc.h:34-36, src/transport/p2p.cc:231-235).
  if (!kind || !strcasecmp(kind, "uncached")) e = hipExtMallocWithFlags(&p, c->lay.totalBytes, hipDeviceMallocUncached);
  if (e != hipSuccess && (!kind || strcasecmp(kind, "coarse")))
    e = hipExtMallocWithFlags(&p, c->lay.totalBytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) e = hipMalloc(&p, c->lay.totalBytes);
  if (e != hipSuccess) {
    RA_WARN("cannot allocate the %zu-byte receive window: %s", c->lay.totalBytes, hipGetErrorString(e));
    return ncclUnhandledCudaError;
  }
  (void)hipGetLastError();
  c->window = (char*)p;
  RA_HIP(hipMemset(p, 0, c->lay.dataOffset));   // all flag words start at zero
  RA_HIP(hipMemset((char*)p + c->lay.llOffset, 0, c->lay.osDataOffset - c->lay.llOffset));   // no line carries a live epoch, no flag is up
  RA_HIP(hipDeviceSynchronize());
  RA_INFO("rank %d: window %p, %zu MiB = flags %zu KiB + FIFOs %zu MiB (%d channels x %d peers x %d kinds x %d slots x %u KiB) "
          "+ line buffers %zu MiB + exchange region %zu MiB (RCCL_AMD_ONESHOT_MAX_BYTES / RCCL_AMD_TWO_EXCHANGE_MAX_BYTES)",
          c->rank, p, c->lay.totalBytes >> 20, c->lay.dataOffset >> 10, raDataBytes(c->lay) >> 20, c->lay.nChannels, c->nRanks - 1,
          RA_NKINDS, c->lay.nSlots, c->lay.slotBytes >> 10, (c->lay.osFlagOffset - c->lay.llOffset) >> 20,
          (c->lay.totalBytes - c->lay.osFlagOffset) >> 20);
  return ncclSuccess;
}

// Rings over the links the machine really has (c->linkAdj: 1 where two ranks' GPUs are joined by one xGMI hop, or are
// the same GPU).  The reference builds its rings from the same facts (src/graph/xml.cc:780-791 -> search.cc:970-986).
void computeRings(ncclComm* c) {
  const int n = c->nRanks;
  std::vector<int> adj((size_t)n * n, 1);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) adj[(size_t)i * n + j] = i == j ? 0 : c->linkAdj[i * RA_MAX_RANKS + j];
  std::vector<std::vector<int>> rings;
  raTopoRings(n, adj.data(), RA_MAX_CHANNELS, rings);
  if (rings.empty()) {
    std::vector<int> r;
    for (int i = 0; i < n; i++) r.push_back(i);
    rings.push_back(r);
  }
  c->nRings = (int)rings.size();
  for (int r = 0; r < c->nRings; r++)
    for (int i = 0; i < n; i++) c->ringRanks[r][i] = (int8_t)rings[(size_t)r][(size_t)i];
}

// The all-pairs schedule is the default only where every pair of ranks has its own link; otherwise the rings above,
// which use the links there are, take over (as if NCCL_ALGO=Ring had been set; an explicit NCCL_ALGO stands).
void chooseDefaultSchedule(ncclComm* c) {
  const int n = c->nRanks;
  int missing = 0;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) missing += i != j && !c->linkAdj[i * RA_MAX_RANKS + j];
  c->allDirect = missing == 0;
  if (!c->allDirect && c->algoEnv < 0) {
    RA_WARN("rank %d: %d of %d rank pairs are not joined by a direct xGMI link: the ring schedule over the links that "
            "exist replaces the all-pairs default (set NCCL_ALGO to choose otherwise)", c->rank, missing / 2, n * (n - 1) / 2);
    c->algoEnv = RA_ALGO_RING;
  }
}

ncclResult_t buildDevComm(ncclComm* c) {
  const size_t nSteps = (size_t)c->lay.nChannels * c->nRanks * RA_NKINDS;
  RA_HIP(hipMalloc((void**)&c->steps, 2 * nSteps * sizeof(uint64_t)));
  RA_HIP(hipMemset(c->steps, 0, 2 * nSteps * sizeof(uint64_t)));
  RA_HIP(hipMalloc((void**)&c->llEpoch, RA_LL_MAX_GRID * sizeof(uint32_t)));
  RA_HIP(hipMemset(c->llEpoch, 0, RA_LL_MAX_GRID * sizeof(uint32_t)));
  RA_HIP(hipMalloc((void**)&c->osEpoch, RA_OS_MAX_GRID * sizeof(uint64_t)));
  RA_HIP(hipMemset(c->osEpoch, 0, RA_OS_MAX_GRID * sizeof(uint64_t)));
  RA_HIP(hipMalloc((void**)&c->regTable, sizeof(RaRegSlot) * RA_MAX_RANKS * RA_MAX_REGS));
  RA_HIP(hipMemset(c->regTable, 0, sizeof(RaRegSlot) * RA_MAX_RANKS * RA_MAX_REGS));
  RA_HIP(hipMalloc((void**)&c->stats, 32 * sizeof(uint64_t)));
  RA_HIP(hipMemset(c->stats, 0, 32 * sizeof(uint64_t)));
  RA_HIP(hipHostMalloc((void**)&c->hostFlags, 64, hipHostMallocMapped | hipHostMallocCoherent));   // both sides poll it: never cached on the GPU
  memset(c->hostFlags, 0, 64);
  RA_HIP(hipHostGetDevicePointer((void**)&c->hostFlagsDev, c->hostFlags, 0));
  RA_HIP(hipEventCreateWithFlags(&c->orderEvent, hipEventDisableTiming));
  computeRings(c);

  RaDevComm h;
  memset(&h, 0, sizeof h);
  h.rank = c->rank;
  h.nRanks = c->nRanks;
  h.lay = c->lay;
  for (int p = 0; p < c->nRanks; p++) h.win[p] = c->peerWin[p];
  h.sendStep = c->steps;
  h.recvStep = c->steps + nSteps;
  h.llEpoch = c->llEpoch;
  h.osEpoch = c->osEpoch;
  h.abortFlag = c->hostFlagsDev;
  h.errorFlag = c->hostFlagsDev + 1;
  // wall_clock64() ticks at 100 MHz on gfx950
  long ms = envLong("RCCL_AMD_SPIN_TIMEOUT_MS", 120000);
  if (ms < 1) ms = 1;
  h.timeoutTicks = (uint64_t)ms * 100000ull;
  h.regTable = c->regTable;
  h.stats = c->stats;
  h.netMode = c->net ? 1u : 0u;
  h.tuneFlags = (uint32_t)envLong("RCCL_AMD_TUNE_FLAGS", 0);
  h.nRings = c->nRings;
  memcpy(h.ringRanks, c->ringRanks, sizeof h.ringRanks);
  RA_HIP(hipMalloc((void**)&c->devComm, sizeof h));
  RA_HIP(hipMemcpy(c->devComm, &h, sizeof h, hipMemcpyHostToDevice));
  RA_HIP(hipDeviceSynchronize());
  return ncclSuccess;
}

// One line per peer about the link that joins the two GPUs (the reference builds its topology from the same
// facts: xGMI links of one hop become <xgmi> edges, src/graph/xml.cc:780-791).  The all-pairs schedule assumes a
// direct xGMI link to every peer; anything else still works and is slower, so it is said out loud.
// Returns 1 when the two GPUs are joined by one xGMI hop (or are one GPU), 0 when they are not, and says so.
int describeLink(ncclComm* c, int peerRank, int peerDev) {
  if (peerDev == c->device) return 1;
  uint32_t type = 0, hops = 0;
  if (peerDev < 0 || hipExtGetLinkTypeAndHopCount(c->device, peerDev, &type, &hops) != hipSuccess) {
    (void)hipGetLastError();
    // (a peer whose GPU this process cannot see, or a runtime that does not say: nothing speaks against the link)
    RA_INFO("rank %d (device %d) -> rank %d (device %d): link type unknown, taken to be direct", c->rank, c->device, peerRank, peerDev);
    return 1;
  }
  const int direct = raTopoLinkIsDirect(type, hops);
  if (direct) RA_INFO("rank %d (device %d) -> rank %d (device %d): xGMI, 1 hop", c->rank, c->device, peerRank, peerDev);
  else RA_WARN("rank %d (device %d) -> rank %d (device %d): link type %u, %u hop(s): not a direct xGMI link",
               c->rank, c->device, peerRank, peerDev, type, hops);
  return direct;
}

ncclResult_t enablePeer(int myDev, int peerDev) {
  if (myDev == peerDev) return ncclSuccess;
  int can = 0;
  RA_HIP(hipDeviceCanAccessPeer(&can, myDev, peerDev));
  if (!can) {
    RA_WARN("device %d cannot access device %d peer-to-peer", myDev, peerDev);
    return ncclSystemError;
  }
  hipError_t e = hipDeviceEnablePeerAccess(peerDev, 0);
  if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
    RA_WARN("hipDeviceEnablePeerAccess(%d) from %d: %s", peerDev, myDev, hipGetErrorString(e));
    return ncclUnhandledCudaError;
  }
  (void)hipGetLastError();
  return ncclSuccess;
}

void freeLocal(ncclComm* c) {
  if (c->devComm) (void)hipFree(c->devComm);
  if (c->steps) (void)hipFree(c->steps);
  if (c->llEpoch) (void)hipFree(c->llEpoch);
  if (c->osEpoch) (void)hipFree(c->osEpoch);
  c->osEpoch = nullptr;
  if (c->regTable) (void)hipFree(c->regTable);
  if (c->stats) (void)hipFree(c->stats);
  c->llEpoch = nullptr; c->regTable = nullptr; c->stats = nullptr;
  raRegShutdown(c);
  if (c->hostFlags) (void)hipHostFree(c->hostFlags);
  if (c->orderEvent) (void)hipEventDestroy(c->orderEvent);
  c->devComm = nullptr; c->steps = nullptr; c->hostFlags = nullptr; c->orderEvent = nullptr;
}

}  // namespace

void raWindowLayoutFor(int nRanks, int* channels, RaWindowLayout* lay) {
  ncclComm c;
  c.nRanks = nRanks;
  c.maxChannels = *channels;
  c.multiProcess = true;
  exchangeLimits(&c);
  makeLayout(&c);
  *channels = c.maxChannels;
  *lay = c.lay;
}

RaClique::~RaClique() {
  DeviceGuard g;
  for (size_t r = 0; r < window.size(); r++)
    if (window[r]) {
      (void)hipSetDevice(device[r]);
      (void)hipFree(window[r]);
    }
}

ncclResult_t raCommCheck(const ncclComm* comm, const char* what) {
  // src/misc/argcheck.cc:36-43
  if (comm == nullptr) { RA_WARN("%s: communicator is NULL", what); return ncclInvalidArgument; }
  if (comm->magic != RA_COMM_MAGIC) { RA_WARN("%s: communicator is corrupted or already destroyed", what); return ncclInvalidArgument; }
  return ncclSuccess;
}

// src/init.cc ncclCommEnsureReady: an entry point that touches what the set-up builds (window layout, device tables,
// registration state) refuses a communicator whose non-blocking creation has not ended, or has failed.
ncclResult_t raCommEnsureReady(const ncclComm* comm, const char* what) {
  RA_CHECK(raCommCheck(comm, what));
  const int init = comm->initState.load(std::memory_order_acquire);
  if (init == ncclInProgress) {
    RA_WARN("%s: attempt to use the communicator before its creation returned ncclSuccess", what);
    return ncclInvalidArgument;
  }
  if (init != ncclSuccess) { RA_WARN("%s: the communicator's creation failed", what); return (ncclResult_t)init; }
  return ncclSuccess;
}

ncclResult_t raCommPollError(ncclComm* c) {
  const int init = c->initState.load(std::memory_order_acquire);
  if (init != ncclSuccess) return (ncclResult_t)init;      // still being set up (ncclInProgress), or the set-up failed
  if (c->asyncError == ncclSuccess && c->hostFlags) {
    const uint32_t e = __atomic_load_n(&c->hostFlags[1], __ATOMIC_RELAXED);
    if (e == RA_ERR_TIMEOUT) {
      c->asyncError = ncclRemoteError;
      RA_WARN("rank %d: a device-side wait for a peer timed out; the communicator is unusable", c->rank);
    } else if (e == RA_ERR_ABORTED) {
      c->asyncError = ncclInternalError;
    }
  }
  return c->asyncError;
}

ncclResult_t raCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
  // src/init.cc:2365-2420
  if (!comms || ndev < 1) { RA_WARN("ncclCommInitAll: invalid arguments"); return ncclInvalidArgument; }
  if (ndev > RA_MAX_RANKS) { RA_WARN("ncclCommInitAll: at most %d ranks per node", RA_MAX_RANKS); return ncclInvalidArgument; }
  int ngpu = 0;
  RA_HIP(hipGetDeviceCount(&ngpu));
  std::vector<int> devs((size_t)ndev);
  const bool allowDup = envLong("RCCL_AMD_MULTI_RANK_GPU", 0) != 0;
  for (int i = 0; i < ndev; i++) {
    devs[(size_t)i] = devlist ? devlist[i] : i;
    if (devs[(size_t)i] < 0 || devs[(size_t)i] >= ngpu) { RA_WARN("ncclCommInitAll: invalid device %d", devs[(size_t)i]); return ncclInvalidArgument; }
    for (int j = 0; j < i; j++)
      if (devs[(size_t)j] == devs[(size_t)i] && !allowDup) {
        RA_WARN("ncclCommInitAll: device %d is listed twice (set RCCL_AMD_MULTI_RANK_GPU=1 to allow it)", devs[(size_t)i]);
        return ncclInvalidArgument;
      }
  }
  bool shared = false;
  for (int i = 0; i < ndev; i++)
    for (int j = 0; j < i; j++) shared = shared || devs[(size_t)j] == devs[(size_t)i];
  DeviceGuard guard;
  if (ndev > 1 && raNetRequested()) {
    // Peer-to-peer is switched off: the ranks are set up like ranks of separate processes (rendezvous, sockets,
    // host-staged windows), one thread each because every rank's set-up waits for the others.
    ncclUniqueId id;
    RA_CHECK(raBootstrapCreate(&id));
    std::vector<ncclResult_t> results((size_t)ndev, ncclSuccess);
    std::vector<std::thread> th;
    for (int i = 0; i < ndev; i++) {
      comms[i] = nullptr;
      th.emplace_back([&, i]() {
        if (hipSetDevice(devs[(size_t)i]) != hipSuccess) { results[(size_t)i] = ncclUnhandledCudaError; return; }
        results[(size_t)i] = raCommInitRank(&comms[i], ndev, id, i, nullptr);
      });
    }
    for (auto& t : th) t.join();
    ncclResult_t res = ncclSuccess;
    for (ncclResult_t r : results) if (r != ncclSuccess && res == ncclSuccess) res = r;
    if (res != ncclSuccess)
      for (int i = 0; i < ndev; i++) if (comms[i]) { (void)raCommDestroy(comms[i], true); comms[i] = nullptr; }
    return res;
  }
  auto clique = std::make_shared<RaClique>();
  clique->nRanks = ndev;
  clique->window.assign((size_t)ndev, nullptr);
  clique->device = devs;
  const uint64_t regKey = raRegLocalKey();
  std::vector<ncclComm*> cs((size_t)ndev, nullptr);
  ncclResult_t res = ncclSuccess;
  for (int i = 0; i < ndev && res == ncclSuccess; i++) {
    ncclComm* c = new ncclComm;
    cs[(size_t)i] = c;
    c->rank = i; c->nRanks = ndev; c->device = devs[(size_t)i]; c->clique = clique;
    readTunables(c);
    setLLGrid(c, shared);
    if (hipSetDevice(c->device) != hipSuccess) { res = ncclUnhandledCudaError; break; }
    makeLayout(c);
    res = allocWindow(c);
    clique->window[(size_t)i] = c->window;
    if (res == ncclSuccess) res = raRegInit(c, regKey);
  }
  for (int i = 0; i < ndev && res == ncclSuccess; i++) {
    ncclComm* c = cs[(size_t)i];
    if (hipSetDevice(c->device) != hipSuccess) { res = ncclUnhandledCudaError; break; }
    for (int j = 0; j < ndev && res == ncclSuccess; j++) {
      c->peerWin[j] = cs[(size_t)j]->window;
      res = enablePeer(c->device, cs[(size_t)j]->device);
    }
    // the same link matrix on every rank (both directions of a pair must be direct)
    for (int a = 0; a < ndev; a++)
      for (int b = 0; b < ndev; b++) {
        if (a == b) { c->linkAdj[a * RA_MAX_RANKS + b] = 0; continue; }
        uint32_t t1 = 0, h1 = 0, t2 = 0, h2 = 0;
        const int da = devs[(size_t)a], db = devs[(size_t)b];
        int direct = 1;
        if (da != db) {
          if (hipExtGetLinkTypeAndHopCount(da, db, &t1, &h1) == hipSuccess && hipExtGetLinkTypeAndHopCount(db, da, &t2, &h2) == hipSuccess)
            direct = raTopoLinkIsDirect(t1, h1) && raTopoLinkIsDirect(t2, h2);
          else (void)hipGetLastError();
        }
        c->linkAdj[a * RA_MAX_RANKS + b] = direct;
      }
    chooseDefaultSchedule(c);
    if (res == ncclSuccess) res = buildDevComm(c);
  }
  if (res != ncclSuccess) {
    for (ncclComm* c : cs) if (c) { freeLocal(c); c->magic = 0; delete c; }
    return res;
  }
  clique->joined = ndev;
  for (int i = 0; i < ndev; i++) comms[i] = cs[(size_t)i];
  return ncclSuccess;
}

namespace {
struct PeerInfo {            // one per rank, all-gathered at init (cf. src/include/transport.h:36-50)
  int32_t rank, pid, device, maxChannels;
  uint32_t slotBytes; int32_t nSlots;
  uint64_t osBytes;
  uint64_t hostHash;
  uint64_t busId;            // hash of the device's PCI bus id: equal on two ranks = they share a GPU
  char pciBusId[24];         // the id itself: a peer finds its own ordinal for that GPU with it
  uint64_t windowPtr;        // meaningful inside the owner's process only
  hipIpcMemHandle_t handle;
  int32_t haveHandle, pad;
};
uint64_t hostHash() {
  char name[256] = "";
  (void)gethostname(name, sizeof name - 1);
  uint64_t h = 1469598103934665603ull;
  for (const char* p = name; *p; p++) h = (h ^ (uint64_t)(unsigned char)*p) * 1099511628211ull;
  return h;
}
}  // namespace

namespace {
// ncclCommInitRank behind its argument checks: bootstrap, window, peer mappings, device tables.  `c` carries rank,
// nRanks and device.  On failure the communicator holds no resources any more (the caller still owns the object).
ncclResult_t initRankBody(ncclComm* c, const ncclUniqueId& id) {
  const int nranks = c->nRanks, rank = c->rank;
  ncclResult_t res = ncclSuccess;
  std::vector<PeerInfo> all((size_t)nranks);
  PeerInfo me;
  bool windowOwned = false;      // the clique has taken the window over (it frees it when the last sibling leaves)
  const uint64_t key = raBootstrapKey(id);
  do {
    readTunables(c);
    if (nranks == 1) {
      makeLayout(c);
      c->clique = std::make_shared<RaClique>();
      if ((res = raRegInit(c, key)) != ncclSuccess) break;
      res = buildDevComm(c);
      break;
    }
    c->multiProcess = true;
    if ((res = c->boot.connect(id, rank, nranks)) != ncclSuccess) break;
    if (raNetRequested()) {
      // Host-staged transport: few channels (one CPU thread pumps them all), windows in pinned host memory.
      if (!getenv("RCCL_AMD_NCHANNELS") && !getenv("NCCL_MAX_NCHANNELS")) c->maxChannels = 2;
      makeLayout(c);
      c->clique = std::make_shared<RaClique>();
      if ((res = raNetSetup(c)) != ncclSuccess) break;
      if ((res = raRegInit(c, key)) != ncclSuccess) break;
      if ((res = buildDevComm(c)) != ncclSuccess) break;
      res = c->boot.barrier();
      break;
    }
    makeLayout(c);
    if (c->lay.totalBytes >= (1ull << 31)) {
      // hipIpcOpenMemHandle did not return for allocations of 2 GiB and more when tried (ROCm 7.2): refuse rather than hang
      RA_WARN("the receive window would be %zu MiB; windows shared between processes must stay below 2 GiB "
              "(fewer channels, NCCL_MAX_NCHANNELS, or a smaller NCCL_BUFFSIZE)", c->lay.totalBytes >> 20);
      res = ncclInvalidUsage;
      break;
    }
    if ((res = allocWindow(c)) != ncclSuccess) break;
    memset(&me, 0, sizeof me);
    me.rank = rank; me.pid = (int32_t)getpid(); me.device = c->device; me.maxChannels = c->maxChannels;
    me.slotBytes = c->lay.slotBytes; me.nSlots = c->lay.nSlots; me.osBytes = c->lay.osBytes; me.hostHash = hostHash(); me.busId = busIdOf(c->device);
    if (hipDeviceGetPCIBusId(me.pciBusId, (int)sizeof me.pciBusId, c->device) != hipSuccess) { (void)hipGetLastError(); me.pciBusId[0] = 0; }
    me.windowPtr = (uint64_t)(uintptr_t)c->window;
    me.haveHandle = hipIpcGetMemHandle(&me.handle, c->window) == hipSuccess;
    (void)hipGetLastError();
    if ((res = c->boot.allGather(&me, all.data(), sizeof me)) != ncclSuccess) break;

    // Same-process ranks share a clique that owns their windows.
    {
      std::lock_guard<std::mutex> g(gCliqueMutex);
      auto it = gCliques.find(key);
      if (it != gCliques.end()) c->clique = it->second.lock();
      if (!c->clique) {
        c->clique = std::make_shared<RaClique>();
        c->clique->key = key; c->clique->nRanks = nranks;
        c->clique->window.assign((size_t)nranks, nullptr);
        c->clique->device.assign((size_t)nranks, 0);
        gCliques[key] = c->clique;
      }
      c->clique->window[(size_t)rank] = c->window;
      c->clique->device[(size_t)rank] = c->device;
      c->clique->joined++;
      windowOwned = true;
    }
    int8_t myRow[RA_MAX_RANKS];
    for (int p = 0; p < RA_MAX_RANKS; p++) myRow[p] = 1;
    for (int p = 0; p < nranks && res == ncclSuccess; p++) {
      const PeerInfo& pi = all[(size_t)p];
      if (pi.hostHash != me.hostHash) { RA_WARN("rank %d is on another host: only one node is supported", p); res = ncclInvalidUsage; break; }
      if (pi.maxChannels != me.maxChannels || pi.slotBytes != me.slotBytes || pi.nSlots != me.nSlots || pi.osBytes != me.osBytes) {
        RA_WARN("rank %d uses different channel/slot settings (%d ch, %u B, %d slots) than rank %d (%d, %u, %d)", p,
                pi.maxChannels, pi.slotBytes, pi.nSlots, rank, me.maxChannels, me.slotBytes, me.nSlots);
        res = ncclInvalidUsage; break;
      }
      if (p == rank) { c->peerWin[p] = c->window; continue; }
      if (pi.pid != me.pid) c->peersElsewhere = true;
      if (pi.busId != me.busId) {
        // the peer's device number is its own process's; this process's number for the same GPU comes from the bus id
        int local = -1;
        if (pi.pid == me.pid) local = pi.device;
        else if (!pi.pciBusId[0] || hipDeviceGetByPCIBusId(&local, pi.pciBusId) != hipSuccess) { (void)hipGetLastError(); local = -1; }
        myRow[p] = (int8_t)describeLink(c, p, local);
      }
      if (pi.pid == me.pid) {
        c->peerWin[p] = (char*)(uintptr_t)pi.windowPtr;
        res = enablePeer(c->device, pi.device);
      } else {
        if (!pi.haveHandle) { RA_WARN("rank %d could not export its window (hipIpcGetMemHandle failed)", p); res = ncclSystemError; break; }
        void* mapped = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&mapped, pi.handle, hipIpcMemLazyEnablePeerAccess);   // src/transport/p2p.cc:292
        if (e != hipSuccess) { RA_WARN("hipIpcOpenMemHandle for rank %d's window: %s", p, hipGetErrorString(e)); res = ncclUnhandledCudaError; break; }
        c->peerWin[p] = (char*)mapped;
        c->peerIsIpc[p] = true;
      }
    }
    if (res != ncclSuccess) break;
    {
      bool shared = false;
      for (int p = 0; p < nranks; p++)
        for (int q = 0; q < p; q++) shared = shared || all[(size_t)p].busId == all[(size_t)q].busId;
      setLLGrid(c, shared);
    }
    {
      // every rank's view of its links, so that all of them build the same rings: a pair counts as linked when both
      // ends say so
      std::vector<int8_t> rows((size_t)nranks * RA_MAX_RANKS);
      if ((res = c->boot.allGather(myRow, rows.data(), sizeof myRow)) != ncclSuccess) break;
      for (int i = 0; i < nranks; i++)
        for (int j = 0; j < nranks; j++)
          c->linkAdj[i * RA_MAX_RANKS + j] = i != j && rows[(size_t)i * RA_MAX_RANKS + j] && rows[(size_t)j * RA_MAX_RANKS + i];
      chooseDefaultSchedule(c);
    }
    if ((res = raRegInit(c, key)) != ncclSuccess) break;
    if ((res = buildDevComm(c)) != ncclSuccess) break;
    res = c->boot.barrier();      // nobody launches before every window is mapped everywhere
  } while (0);
  if (res != ncclSuccess) {
    RA_WARN("ncclCommInitRank failed for rank %d of %d", rank, nranks);
    c->boot.close();
    raNetShutdown(c, false);
    for (int p = 0; p < nranks; p++) if (c->peerIsIpc[p]) (void)hipIpcCloseMemHandle(c->peerWin[p]);
    freeLocal(c);
    if (windowOwned) {
      std::lock_guard<std::mutex> g(gCliqueMutex);
      c->clique->left++;
      if (c->clique->left >= c->clique->joined) gCliques.erase(c->clique->key);
    } else if (c->window && !c->net) {
      (void)hipFree(c->window);      // allocated, never handed to a clique
    }
    c->clique.reset();
    c->window = nullptr;
    c->multiProcess = false;
    for (int p = 0; p < nranks; p++) { c->peerIsIpc[p] = false; c->peerWin[p] = nullptr; }
  }
  return res;
}

bool wantsBlocking(const ncclConfig_t* config) {
  bool blocking = !(config && config->blocking != NCCL_CONFIG_UNDEF_INT && config->blocking == 0);
  const long env = envLong("NCCL_COMM_BLOCKING", -1);      // overrides the configuration, src/init.cc:2128-2130
  if (env == 0 || env == 1) blocking = env == 1;
  return blocking;
}
}  // namespace

ncclResult_t raCommInitRank(ncclComm_t* out, int nranks, const ncclUniqueId& id, int rank, const ncclConfig_t* config) {
  // src/init.cc:2281-2346 (argument checks), 1188-1874 (transport setup)
  if (!out) { RA_WARN("ncclCommInitRank: comm is NULL"); return ncclInvalidArgument; }
  if (nranks < 1 || rank < 0 || rank >= nranks) { RA_WARN("ncclCommInitRank: invalid rank %d of %d", rank, nranks); return ncclInvalidArgument; }
  if (nranks > RA_MAX_RANKS) { RA_WARN("ncclCommInitRank: at most %d ranks per node", RA_MAX_RANKS); return ncclInvalidArgument; }
  if (config && (config->magic != 0xcafebeef || config->size < sizeof(size_t) + 2 * sizeof(unsigned))) {
    RA_WARN("ncclCommInitRankConfig: config was not initialised with NCCL_CONFIG_INITIALIZER");
    return ncclInvalidArgument;
  }
  int device = 0;
  RA_HIP(hipGetDevice(&device));
  ncclComm* c = new ncclComm;
  c->rank = rank; c->nRanks = nranks; c->device = device;
  c->blocking = wantsBlocking(config);
  if (config && config->size >= offsetof(ncclConfig_t, maxCTAs) + sizeof(int)) {
    if (config->minCTAs != NCCL_CONFIG_UNDEF_INT && config->minCTAs > 0) c->cfgMinCTAs = config->minCTAs;
    if (config->maxCTAs != NCCL_CONFIG_UNDEF_INT && config->maxCTAs > 0) c->cfgMaxCTAs = config->maxCTAs;
    if (c->cfgMinCTAs && c->cfgMaxCTAs && c->cfgMinCTAs > c->cfgMaxCTAs) {
      RA_WARN("ncclCommInitRankConfig: minCTAs %d is larger than maxCTAs %d", c->cfgMinCTAs, c->cfgMaxCTAs);
      c->magic = 0; delete c;
      return ncclInvalidArgument;
    }
  }
  if (c->blocking) {
    const ncclResult_t res = initRankBody(c, id);
    if (res != ncclSuccess) { c->magic = 0; delete c; return res; }
    *out = c;
    return ncclSuccess;
  }
  // Non-blocking: the caller gets the handle now and polls ncclCommGetAsyncError, which says ncclInProgress until the
  // set-up has ended (src/group.cc:412-435).  A failed set-up leaves the handle valid for ncclCommDestroy / Abort.
  c->initState.store(ncclInProgress, std::memory_order_release);
  *out = c;
  c->initThread = std::thread([c, id]() {
    ncclResult_t res = hipSetDevice(c->device) == hipSuccess ? initRankBody(c, id) : ncclUnhandledCudaError;
    c->initState.store(res, std::memory_order_release);
  });
  return ncclInProgress;
}

// ncclCommSplit (src/init.cc:1909-1960, 2753-2820): ranks that pass the same colour form a new communicator, ordered by
// key and then by their rank in the parent; NCCL_SPLIT_NOCOLOR takes part in the exchange and gets no communicator.
// Two all-gathers over the parent (its rendezvous connection, or the in-process exchange of an ncclCommInitAll
// communicator): colours and keys first, then the ncclUniqueId the first member of every new group has made.
namespace {
ncclResult_t commGather(ncclComm* c, const void* mine, void* all, size_t bytes) {
  if (c->nRanks == 1) { memcpy(all, mine, bytes); return ncclSuccess; }
  if (c->multiProcess) return c->boot.allGather(mine, all, bytes);
  return c->clique->gather(c->rank, mine, all, bytes);
}
}  // namespace

ncclResult_t RaClique::gather(int rank, const void* mine, void* all, size_t bytes) {
  std::unique_lock<std::mutex> lk(m);
  const uint64_t round = xRound;
  if (xCount == 0) xIn.assign(bytes * (size_t)nRanks, 0);
  if (xIn.size() != bytes * (size_t)nRanks) { RA_WARN("ranks of one communicator disagree about an exchange"); return ncclInvalidUsage; }
  memcpy(xIn.data() + (size_t)rank * bytes, mine, bytes);
  if (++xCount == nRanks) {
    xOut.swap(xIn);
    xCount = 0;
    xRound++;
    xcv.notify_all();
  } else if (!xcv.wait_for(lk, std::chrono::seconds(envLong("RCCL_AMD_BOOTSTRAP_TIMEOUT_S", 120)), [&]() { return xRound != round; })) {
    xCount--;
    RA_WARN("rank %d waited in vain for the other ranks of its communicator (inside one thread, group the calls)", rank);
    return ncclRemoteError;
  }
  // xOut stays as it is until every rank has come back for the next round, which needs this rank to return first
  memcpy(all, xOut.data(), bytes * (size_t)nRanks);
  return ncclSuccess;
}

uint64_t RaClique::barrierIn(int x) {
  std::lock_guard<std::mutex> lk(m);
  const uint64_t round = bRound;
  bSum += x;
  if (++bCount >= joined - left) {
    bResult = bSum; bSum = 0; bCount = 0; bRound++;
    xcv.notify_all();
  }
  return round;
}

ncclResult_t RaClique::barrierOut(uint64_t round, int* sum) {
  std::unique_lock<std::mutex> lk(m);
  if (!xcv.wait_for(lk, std::chrono::seconds(envLong("RCCL_AMD_BOOTSTRAP_TIMEOUT_S", 120)), [&]() { return bRound > round; })) {
    RA_WARN("NCCL_LAUNCH_MODE=GROUP: the other ranks of this process did not reach the launch barrier");
    return ncclRemoteError;
  }
  // bResult stays as it is until every sibling has arrived for the next round, which needs this one to return first
  *sum = bResult;
  return ncclSuccess;
}

ncclResult_t raCommSplit(ncclComm_t parent, int color, int key, ncclComm_t* newcomm, const ncclConfig_t* config) {
  struct Entry { int32_t color, key; };
  const int n = parent->nRanks;
  Entry me{color, key};
  std::vector<Entry> all((size_t)n);
  RA_CHECK(commGather(parent, &me, all.data(), sizeof me));
  std::vector<int> members;                 // parent ranks of my group in their new order
  if (color != NCCL_SPLIT_NOCOLOR)
    for (int i = 0; i < n; i++) {
      if (all[(size_t)i].color != color) continue;
      size_t at = 0;
      while (at < members.size() && all[(size_t)members[at]].key <= all[(size_t)i].key) at++;
      members.insert(members.begin() + (long)at, i);
    }
  int myRank = 0;
  for (size_t i = 0; i < members.size(); i++) if (members[i] == parent->rank) myRank = (int)i;
  ncclUniqueId id;
  memset(&id, 0, sizeof id);
  ncclResult_t idRes = ncclSuccess;
  if (members.size() > 1 && members[0] == parent->rank) idRes = raBootstrapCreate(&id);
  std::vector<ncclUniqueId> ids((size_t)n);
  RA_CHECK(commGather(parent, &id, ids.data(), sizeof id));    // also when making the id failed: the others are waiting
  RA_CHECK(idRes);
  if (color == NCCL_SPLIT_NOCOLOR) {
    RA_INFO("ncclCommSplit: rank %d passed NCCL_SPLIT_NOCOLOR and gets no communicator", parent->rank);
    return ncclSuccess;
  }
  ncclConfig_t inherited = NCCL_CONFIG_INITIALIZER;           // no config: the child is created like its parent
  inherited.blocking = parent->blocking ? 1 : 0;
  DeviceGuard guard;
  RA_HIP(hipSetDevice(parent->device));
  RA_INFO("ncclCommSplit: rank %d of %d -> colour %d, rank %d of %zu", parent->rank, n, color, myRank, members.size());
  return raCommInitRank(newcomm, (int)members.size(), ids[(size_t)members[0]], myRank, config ? config : &inherited);
}

ncclResult_t raCommDestroy(ncclComm_t c, bool abort) {
  RA_CHECK(raCommCheck(c, abort ? "ncclCommAbort" : "ncclCommDestroy"));
  if (c->initThread.joinable()) {
    // a set-up still under way: an abort cuts its rendezvous short, a destroy lets it end
    if (abort && c->initState.load(std::memory_order_acquire) == ncclInProgress) c->boot.interrupt();
    c->initThread.join();
  }
  DeviceGuard guard;
  (void)hipSetDevice(c->device);
  if (abort && c->hostFlags) __atomic_store_n(&c->hostFlags[0], 1u, __ATOMIC_RELAXED);   // spins give up
  // Everything this rank enqueued must have left the GPU before the tables go away.
  if (c->haveLast) (void)hipStreamSynchronize(c->lastStream);
  (void)hipGetLastError();
  if (c->multiProcess) {
    // Peers in other processes may still be posting credits into this window: leave together.  Ranks of one process
    // need no such wait (their clique keeps every window alive until the last of them has left), so one thread may
    // destroy its communicators one after the other, as the reference allows.
    if (!abort && c->peersElsewhere) (void)c->boot.barrier();
    c->boot.close();
    raNetShutdown(c, !abort);
    for (int p = 0; p < c->nRanks; p++)
      if (c->peerIsIpc[p]) (void)hipIpcCloseMemHandle(c->peerWin[p]);
  }
  freeLocal(c);
  {
    std::lock_guard<std::mutex> g(gCliqueMutex);
    if (c->clique && c->clique->key) {
      c->clique->left++;
      if (c->clique->left >= c->clique->joined) gCliques.erase(c->clique->key);
    }
    c->clique.reset();     // the last sibling frees the windows
  }
  c->magic = 0;
  delete c;
  return ncclSuccess;
}

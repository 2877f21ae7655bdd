// Registration cache and the shared-memory boards that publish it to the other ranks (see reg.h).
#include "reg.h"

#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <map>
#include <mutex>
#include <vector>

#include "comm.h"
#include "debug.h"

namespace {

#define RA_BOARD_MAGIC 0x52414247u
// Largest peer allocation mapped through HIP IPC (RCCL_AMD_REG_MAX_IPC_BYTES overrides).  With the runtime's legacy IPC
// mode allocations of 2 GiB and more did not come back from hipIpcOpenMemHandle within a minute (ROCm 7.2, two processes
// on one MI355X; 1.5 GiB mapped in milliseconds).  With dmabuf IPC (HSA_ENABLE_IPC_MODE_LEGACY=0) a bare process maps 2 and
// 3 GiB in under a millisecond (tools/ipc_big_probe.py), but two benchmark processes (torch, a communicator, 2 GiB send
// buffers registered) with the limit lifted did not get through their first launch within two minutes, so the limit stays.
uint64_t maxIpcBytes() {
  static const uint64_t v = []() -> uint64_t {
    if (const char* e = getenv("RCCL_AMD_REG_MAX_IPC_BYTES")) { if (*e) return strtoull(e, nullptr, 0); }
    return 1ull << 31;
  }();
  return v;
}

// One published registration.  `serial` is a per-entry sequence lock: 0 while the owner rewrites the
// entry (and for an empty one), a fresh non-zero value once it is complete.
struct RegEntry {
  uint64_t base, size;         // the whole device allocation, in the owner's address space
  hipIpcMemHandle_t handle;
  uint32_t serial, pad;
};
struct RegBoard {
  uint32_t magic;
  int32_t pid;
  uint64_t generation;         // bumped after every change
  RegEntry e[RA_MAX_REGS];
};

struct RegLocal {              // what a ncclCommRegister handle points to
  uintptr_t addr;
  size_t size;
  int refs;
  int slot;                    // board entry, -1 when the allocation cannot be exported
  uint32_t serial;
};
struct PeerMap {               // one peer allocation mapped here (shared by its registrations)
  uint64_t base;
  void* mapped;
  int refs;
  bool ipc;
};

}  // namespace

struct RaRegState {
  std::mutex m;
  uint64_t key = 0;
  char myName[64] = "";
  RegBoard* mine = nullptr;
  const RegBoard* peer[RA_MAX_RANKS] = {};
  uint64_t seenGen[RA_MAX_RANKS] = {};
  uint32_t seenSerial[RA_MAX_RANKS][RA_MAX_REGS] = {};
  uint64_t seenBase[RA_MAX_RANKS][RA_MAX_REGS] = {};
  std::vector<PeerMap> maps[RA_MAX_RANKS];
  RaRegSlot table[RA_MAX_RANKS][RA_MAX_REGS] = {};
  bool dirty = false;
  int mapped = 0;              // table entries that are live
  uint32_t nextSerial = 1;
  std::vector<RegLocal*> local;
  RaRegSlot* stage[4] = {};    // pinned copies of `table` in flight to the device
  unsigned stageNext = 0;
};

namespace {

void boardName(char* out, size_t n, uint64_t key, int rank) {
  snprintf(out, n, "/rccl_amd-%016llx-%d", (unsigned long long)key, rank);
}

void releaseMap(RaRegState* s, int p, uint64_t base) {
  for (size_t i = 0; i < s->maps[p].size(); i++) {
    PeerMap& m = s->maps[p][i];
    if (m.base != base) continue;
    if (--m.refs == 0) {
      if (m.ipc) { (void)hipIpcCloseMemHandle(m.mapped); (void)hipGetLastError(); }
      s->maps[p].erase(s->maps[p].begin() + (long)i);
    }
    return;
  }
}

void* acquireMap(RaRegState* s, int p, const RegEntry& e, bool samePid) {
  for (PeerMap& m : s->maps[p])
    if (m.base == e.base) { m.refs++; return m.mapped; }
  PeerMap m;
  m.base = e.base; m.refs = 1; m.ipc = !samePid; m.mapped = nullptr;
  if (samePid) {
    m.mapped = (void*)(uintptr_t)e.base;
  } else if (e.size >= maxIpcBytes()) {
    RA_INFO("rank %d's registered allocation of %llu bytes is not mapped (limit %llu: without dmabuf IPC, allocations of 2 GiB and "
            "more did not come back from hipIpcOpenMemHandle when tried; RCCL_AMD_REG_MAX_IPC_BYTES): its pieces keep going through "
            "the FIFO", p, (unsigned long long)e.size, (unsigned long long)maxIpcBytes());
    return nullptr;
  } else {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    RA_INFO("mapping the %llu-byte allocation rank %d registered", (unsigned long long)e.size, p);
    hipError_t err = hipIpcOpenMemHandle(&m.mapped, e.handle, hipIpcMemLazyEnablePeerAccess);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    RA_INFO("mapped in %.1f ms (%s)", (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) / 1e6, hipGetErrorString(err));
    if (err != hipSuccess) {
      (void)hipGetLastError();
      RA_INFO("cannot map a buffer registered by rank %d (%s): its pieces keep going through the FIFO", p, hipGetErrorString(err));
      return nullptr;
    }
  }
  s->maps[p].push_back(m);
  return m.mapped;
}

// Bring this rank's view of peer p's board up to date.
void refreshPeer(ncclComm* c, RaRegState* s, int p) {
  if (!s->peer[p]) {
    char name[64];
    boardName(name, sizeof name, s->key, p);
    int fd = shm_open(name, O_RDONLY, 0);
    if (fd < 0) return;
    void* m = mmap(nullptr, sizeof(RegBoard), PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return;
    s->peer[p] = (const RegBoard*)m;
  }
  const RegBoard* b = s->peer[p];
  if (__atomic_load_n(&b->magic, __ATOMIC_ACQUIRE) != RA_BOARD_MAGIC) return;
  const uint64_t gen = __atomic_load_n(&b->generation, __ATOMIC_ACQUIRE);
  if (gen == s->seenGen[p]) return;
  const bool samePid = b->pid == (int32_t)getpid();
  bool complete = true;
  for (int i = 0; i < RA_MAX_REGS; i++) {
    const uint32_t s1 = __atomic_load_n(&b->e[i].serial, __ATOMIC_ACQUIRE);
    RegEntry e;
    memcpy(&e, (const void*)&b->e[i], sizeof e);
    std::atomic_thread_fence(std::memory_order_acquire);
    const uint32_t s2 = __atomic_load_n(&b->e[i].serial, __ATOMIC_ACQUIRE);
    if (s1 != s2) { complete = false; continue; }          // the owner is rewriting it: look again next time
    if (s1 == s->seenSerial[p][i]) continue;
    if (s->seenSerial[p][i] != 0) {
      if (s->table[p][i].base) s->mapped--;
      releaseMap(s, p, s->seenBase[p][i]);
      s->table[p][i].base = 0; s->table[p][i].serial = 0;
    }
    s->seenSerial[p][i] = s1;
    s->seenBase[p][i] = e.base;
    if (s1 != 0) {
      void* mapped = acquireMap(s, p, e, samePid);
      s->table[p][i].base = (uint64_t)(uintptr_t)mapped;
      s->table[p][i].serial = s1 & 0xffffu;
      if (mapped) s->mapped++;
      else s->seenBase[p][i] = 0;                           // nothing to release later
      RA_INFO("rank %d: rank %d's registration %d (%llu bytes) is mapped at %p", c->rank, p, i, (unsigned long long)e.size, mapped);
    }
    s->dirty = true;
  }
  if (complete) s->seenGen[p] = gen;
}

}  // namespace

// ----- peer-writable memory -------------------------------------------------------------------
namespace {
std::mutex gMemMutex;
std::map<uintptr_t, size_t> gMemAllocs;      // allocations made by ncclMemAlloc in this process
}  // namespace

ncclResult_t raMemAlloc(void** ptr, size_t size) {
  if (!ptr) { RA_WARN("ncclMemAlloc: ptr is NULL"); return ncclInvalidArgument; }
  *ptr = nullptr;
  if (size == 0) return ncclSuccess;
  // Same preference as for the library's own windows (and the reference's peer-written buffers, src/include/alloc.h:34-36).
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, size, hipDeviceMallocUncached);
  if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&p, size, hipDeviceMallocFinegrained); }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    RA_WARN("ncclMemAlloc: cannot allocate %zu bytes of fine-grained device memory: %s", size, hipGetErrorString(e));
    return ncclUnhandledCudaError;
  }
  {
    std::lock_guard<std::mutex> g(gMemMutex);
    gMemAllocs[(uintptr_t)p] = size;
  }
  *ptr = p;
  return ncclSuccess;
}

ncclResult_t raMemFree(void* ptr) {
  if (!ptr) return ncclSuccess;
  {
    std::lock_guard<std::mutex> g(gMemMutex);
    gMemAllocs.erase((uintptr_t)ptr);
  }
  RA_HIP(hipFree(ptr));
  return ncclSuccess;
}

bool raMemPeerWritable(const void* base, size_t size) {
  const uintptr_t a = (uintptr_t)base;
  {
    std::lock_guard<std::mutex> g(gMemMutex);
    auto it = gMemAllocs.upper_bound(a);
    if (it != gMemAllocs.begin()) {
      --it;
      if (a >= it->first && a + size <= it->first + it->second) return true;
    }
  }
  // Not ours: ask the runtime which pool the allocation came from (hipExtMallocWithFlags with the uncached or
  // fine-grained flag allocates from the device's fine-grained pool).
  hsa_amd_pointer_info_t info;
  memset(&info, 0, sizeof info);
  info.size = sizeof info;
  if (hsa_amd_pointer_info(base, &info, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS) return false;
  if (info.type != HSA_EXT_POINTER_TYPE_HSA) return false;
  return (info.global_flags & (HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_FINE_GRAINED |
                               HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_EXTENDED_SCOPE_FINE_GRAINED)) != 0;
}

uint64_t raRegLocalKey() {
  static std::atomic<uint32_t> counter{0};
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ((uint64_t)getpid() << 40) ^ ((uint64_t)ts.tv_nsec << 8) ^ (uint64_t)ts.tv_sec ^ ((uint64_t)counter.fetch_add(1) << 56) ^
         0x4c4f43414c000000ull;
}

ncclResult_t raRegInit(ncclComm* c, uint64_t key) {
  RaRegState* s = new RaRegState;
  s->key = key;
  c->reg = s;
  if (c->nRanks == 1 || c->net) return ncclSuccess;          // nothing to publish: the cache stays local
  boardName(s->myName, sizeof s->myName, key, c->rank);
  int fd = shm_open(s->myName, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)sizeof(RegBoard)) != 0) {
    if (fd >= 0) close(fd);
    RA_INFO("rank %d: no shared-memory board (%s): registered buffers will not be written directly", c->rank, s->myName);
    s->myName[0] = 0;
    return ncclSuccess;
  }
  void* m = mmap(nullptr, sizeof(RegBoard), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) { shm_unlink(s->myName); s->myName[0] = 0; return ncclSuccess; }
  s->mine = (RegBoard*)m;
  memset(s->mine, 0, sizeof(RegBoard));
  s->mine->pid = (int32_t)getpid();
  __atomic_store_n(&s->mine->magic, RA_BOARD_MAGIC, __ATOMIC_RELEASE);
  return ncclSuccess;
}

void raRegShutdown(ncclComm* c) {
  RaRegState* s = c->reg;
  if (!s) return;
  for (int p = 0; p < RA_MAX_RANKS; p++) {
    for (PeerMap& m : s->maps[p])
      if (m.ipc) { (void)hipIpcCloseMemHandle(m.mapped); (void)hipGetLastError(); }
    if (s->peer[p]) munmap((void*)s->peer[p], sizeof(RegBoard));
  }
  if (s->mine) {
    __atomic_store_n(&s->mine->magic, 0u, __ATOMIC_RELEASE);
    munmap(s->mine, sizeof(RegBoard));
  }
  if (s->myName[0]) shm_unlink(s->myName);
  for (RaRegSlot* st : s->stage) if (st) (void)hipHostFree(st);
  for (RegLocal* r : s->local) delete r;    // src/register.cc:138-152 (cleanup of what the user left registered)
  delete s;
  c->reg = nullptr;
}

ncclResult_t raRegRegister(ncclComm* c, void* buff, size_t size, void** handle) {
  RaRegState* s = c->reg;
  if (!buff || !handle || size == 0) { RA_WARN("ncclCommRegister: buff/handle is NULL or size is 0"); return ncclInvalidArgument; }
  std::lock_guard<std::mutex> g(s->m);
  const uintptr_t a = (uintptr_t)buff;
  for (RegLocal* r : s->local)
    if (a >= r->addr && a + size <= r->addr + r->size) { r->refs++; *handle = r; return ncclSuccess; }   // src/register.cc:129-134
  int old = -1;
  (void)hipGetDevice(&old);
  if (old != c->device) (void)hipSetDevice(c->device);
  hipDeviceptr_t base = nullptr;
  size_t asize = 0;
  hipError_t e = hipMemGetAddressRange(&base, &asize, (hipDeviceptr_t)buff);
  if (e != hipSuccess || !base || a + size > (uintptr_t)base + asize) {
    (void)hipGetLastError();
    if (old >= 0 && old != c->device) (void)hipSetDevice(old);
    RA_WARN("ncclCommRegister: %p (+%zu bytes) is not inside one device allocation", buff, size);
    return ncclInvalidArgument;
  }
  RegLocal* r = new RegLocal;
  r->addr = (uintptr_t)base; r->size = asize; r->refs = 1; r->slot = -1; r->serial = 0;
  // Published (so that peers write into it) only when it is memory a peer may write: see reg.h.  RCCL_AMD_REG_COARSE=1
  // lifts the rule for experiments.
  static const bool anyType = []() { const char* e = getenv("RCCL_AMD_REG_COARSE"); return e && atoi(e) != 0; }();
  const bool writable = anyType || raMemPeerWritable((const void*)base, asize);
  if (s->mine && !writable)
    RA_INFO("ncclCommRegister: %p lies in ordinary (coarse-grained) device memory: registered locally only, collectives on "
            "it keep using the FIFOs (allocate with ncclMemAlloc for direct writes)", buff);
  if (s->mine && writable) {
    int slot = -1;
    for (int i = 0; i < RA_MAX_REGS && slot < 0; i++)
      if (s->mine->e[i].serial == 0) slot = i;
    hipIpcMemHandle_t h;
    if (slot >= 0 && hipIpcGetMemHandle(&h, base) == hipSuccess) {
      RegEntry& en = s->mine->e[slot];
      en.base = (uint64_t)(uintptr_t)base; en.size = asize; en.handle = h;
      uint32_t ser = s->nextSerial++ & 0xffffu;
      if (ser == 0) ser = s->nextSerial++ & 0xffffu;
      __atomic_store_n(&en.serial, ser, __ATOMIC_RELEASE);
      __atomic_fetch_add(&s->mine->generation, 1, __ATOMIC_RELEASE);
      r->slot = slot; r->serial = ser;
    } else {
      (void)hipGetLastError();
      RA_INFO("ncclCommRegister: %p is registered locally only (%s)", buff, slot < 0 ? "the board is full" : "the allocation cannot be exported");
    }
  }
  if (old >= 0 && old != c->device) (void)hipSetDevice(old);
  s->local.push_back(r);
  *handle = r;
  RA_INFO("rank %d: registered %p (+%zu) as allocation %p (+%zu), board entry %d", c->rank, buff, size, (void*)base, asize, r->slot);
  return ncclSuccess;
}

ncclResult_t raRegDeregister(ncclComm* c, void* handle) {
  RaRegState* s = c->reg;
  std::lock_guard<std::mutex> g(s->m);
  for (size_t i = 0; i < s->local.size(); i++) {
    RegLocal* r = s->local[i];
    if (r != handle) continue;
    if (--r->refs) return ncclSuccess;
    if (r->slot >= 0 && s->mine) {
      __atomic_store_n(&s->mine->e[r->slot].serial, 0u, __ATOMIC_RELEASE);
      __atomic_fetch_add(&s->mine->generation, 1, __ATOMIC_RELEASE);
    }
    s->local.erase(s->local.begin() + (long)i);
    delete r;
    return ncclSuccess;
  }
  RA_WARN("Deregister: Could not find handle");              // src/register.cc:169-172
  return ncclInvalidUsage;
}

ncclResult_t raRegPrepare(ncclComm* c, const void* recv, size_t bytes, hipStream_t stream, uint64_t* word, int* havePeerRegs) {
  RaRegState* s = c->reg;
  *word = 0; *havePeerRegs = 0;
  if (!s || !s->mine) return ncclSuccess;
  std::lock_guard<std::mutex> g(s->m);
  const uintptr_t a = (uintptr_t)recv;
  for (RegLocal* r : s->local)
    if (r->slot >= 0 && a >= r->addr && a + bytes <= r->addr + r->size && a - r->addr < (1ull << 41)) {
      *word = raRegWord(r->slot, r->serial, a - r->addr);
      break;
    }
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
  if (cap == hipStreamCaptureStatusNone) {
    for (int p = 0; p < c->nRanks; p++)
      if (p != c->rank) refreshPeer(c, s, p);
    if (s->dirty) {
      // The upload runs in stream order, so a kernel never sees the table change under it; each upload reads its
      // own pinned snapshot (four in rotation: an older upload has long finished when its snapshot is reused).
      RaRegSlot*& st = s->stage[s->stageNext++ % 4];
      if (!st) RA_HIP(hipHostMalloc((void**)&st, sizeof s->table, hipHostMallocDefault));
      memcpy(st, s->table, sizeof s->table);
      RA_HIP(hipMemcpyAsync(c->regTable, st, sizeof s->table, hipMemcpyHostToDevice, stream));
      s->dirty = false;
    }
  }
  *havePeerRegs = s->mapped > 0;
  return ncclSuccess;
}

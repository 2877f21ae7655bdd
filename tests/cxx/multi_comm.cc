// One thread that owns several communicators, written only against include/rccl_amd.h:
//   * ncclCommInitAll for four ranks + ncclGroupStart/End around ncclAllReduce / ncclReduceScatter / ncclAllGather
//     (the single-thread launch loop of src/group.cc:128-195 beyond two communicators),
//   * the reference's three ncclCommSplit tests (test/StandaloneTests.cpp:18-153: round-robin colours, one colour,
//     NCCL_SPLIT_NOCOLOR for the upper half), each followed by an all-reduce on the new communicators,
//   * non-blocking creation (ncclConfig_t::blocking = 0): ncclInProgress, then ncclCommGetAsyncError until it
//     settles, as test/common/TestBedChild.cpp:40-55,250-258 does.
// All ranks sit on device 0 (RCCL_AMD_MULTI_RANK_GPU=1) so that it runs on a one-GPU box; four ranks x three roles x
// 16 channels stay resident together there.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <chrono>
#include <vector>
#include "rccl_amd.h"

#define HIPCHECK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
#define NCCLCHECK(c) do { ncclResult_t r_ = (c); if (r_ != ncclSuccess) { printf("NCCL error %s (%s) at %s:%d\n", ncclGetErrorString(r_), ncclGetLastError(NULL), __FILE__, __LINE__); return 1; } } while (0)
#define EXPECT(cond) do { if (!(cond)) { printf("check failed: %s at %s:%d\n", #cond, __FILE__, __LINE__); return 1; } } while (0)

static const int kRanks = 4;

struct Bufs {
  std::vector<int*> in, out, big;
  std::vector<hipStream_t> stream;
  size_t n = 0;
};

static int makeBufs(Bufs& b, int nRanks, size_t n) {
  b.n = n;
  b.in.assign(nRanks, nullptr); b.out.assign(nRanks, nullptr); b.big.assign(nRanks, nullptr); b.stream.assign(nRanks, nullptr);
  std::vector<int> host(n);
  for (int r = 0; r < nRanks; r++) {
    // Streams of one process share a few hardware queues per priority level, and every rank's kernel must be running at
    // the same time: spread the ranks over the priority levels so that no two of them queue up behind each other.
    int lo = 0, hi = 0;
    HIPCHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    const char* pe = getenv("MULTI_COMM_PRIO");
    const int mode = pe ? atoi(pe) : 1;
    const int prio = mode == 0 ? 0 : (hi + r % (lo - hi + 1));
    HIPCHECK(hipStreamCreateWithPriority(&b.stream[r], hipStreamNonBlocking, prio));
    HIPCHECK(hipMalloc((void**)&b.in[r], n * sizeof(int)));
    HIPCHECK(hipMalloc((void**)&b.out[r], n * sizeof(int)));
    HIPCHECK(hipMalloc((void**)&b.big[r], (size_t)nRanks * n * sizeof(int)));
    for (size_t i = 0; i < n; i++) host[i] = (int)i + 1000 * r;
    HIPCHECK(hipMemcpy(b.in[r], host.data(), n * sizeof(int), hipMemcpyHostToDevice));
    HIPCHECK(hipMemset(b.out[r], 0xff, n * sizeof(int)));
  }
  return 0;
}
static int freeBufs(Bufs& b) {
  for (size_t r = 0; r < b.in.size(); r++) {
    HIPCHECK(hipFree(b.in[r])); HIPCHECK(hipFree(b.out[r])); HIPCHECK(hipFree(b.big[r]));
    HIPCHECK(hipStreamDestroy(b.stream[r]));
  }
  return 0;
}

// All-reduce over the communicators in `comms` (ranks[i] = the input pattern comms[i] contributes), grouped, checked.
static int allReduceCheck(const std::vector<ncclComm_t>& comms, const std::vector<int>& ranks, Bufs& b, const char* what) {
  const size_t n = b.n;
  NCCLCHECK(ncclGroupStart());
  for (size_t i = 0; i < comms.size(); i++)
    NCCLCHECK(ncclAllReduce(b.in[ranks[i]], b.out[ranks[i]], n, ncclInt, ncclSum, comms[i], b.stream[ranks[i]]));
  NCCLCHECK(ncclGroupEnd());
  long base = 0;
  for (int r : ranks) base += 1000L * r;
  std::vector<int> host(n);
  for (size_t i = 0; i < comms.size(); i++) {
    HIPCHECK(hipStreamSynchronize(b.stream[ranks[i]]));
    HIPCHECK(hipMemcpy(host.data(), b.out[ranks[i]], n * sizeof(int), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < n; k++) {
      const long want = (long)comms.size() * (long)k + base;
      if (host[k] != (int)want) { printf("%s: rank %d out[%zu] = %d, expected %ld\n", what, ranks[i], k, host[k], want); return 1; }
    }
    ncclResult_t async;
    NCCLCHECK(ncclCommGetAsyncError(comms[i], &async));
    EXPECT(async == ncclSuccess);
  }
  return 0;
}

static int groupLaunchFour(size_t n) {
  int devs[kRanks] = {0, 0, 0, 0};
  std::vector<ncclComm_t> comms(kRanks);
  NCCLCHECK(ncclCommInitAll(comms.data(), kRanks, devs));
  Bufs b;
  if (makeBufs(b, kRanks, n)) return 1;
  std::vector<int> ranks = {0, 1, 2, 3};
  for (int iter = 0; iter < 3; iter++)
    if (allReduceCheck(comms, ranks, b, "4-rank all-reduce")) return 1;
  // all-gather, then reduce-scatter of the gathered buffer: rank r ends with 4 x (block r of the gathered buffer)
  NCCLCHECK(ncclGroupStart());
  for (int r = 0; r < kRanks; r++) NCCLCHECK(ncclAllGather(b.in[r], b.big[r], n, ncclInt, comms[r], b.stream[r]));
  NCCLCHECK(ncclGroupEnd());
  NCCLCHECK(ncclGroupStart());
  for (int r = 0; r < kRanks; r++) NCCLCHECK(ncclReduceScatter(b.big[r], b.out[r], n, ncclInt, ncclSum, comms[r], b.stream[r]));
  NCCLCHECK(ncclGroupEnd());
  std::vector<int> host(n);
  for (int r = 0; r < kRanks; r++) {
    HIPCHECK(hipStreamSynchronize(b.stream[r]));
    HIPCHECK(hipMemcpy(host.data(), b.out[r], n * sizeof(int), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < n; k++)
      if (host[k] != kRanks * ((int)k + 1000 * r)) { printf("gather/scatter: rank %d out[%zu] = %d\n", r, k, host[k]); return 1; }
  }
  if (freeBufs(b)) return 1;
  for (auto c : comms) NCCLCHECK(ncclCommDestroy(c));      // one thread, one after the other
  printf("group launch over 4 communicators, %zu elements: ok\n", n);
  return 0;
}

// mode 0: colour = rank % 2 (test/StandaloneTests.cpp:18-58); 1: one colour (:63-102); 2: upper half NOCOLOR (:107-153)
static int splitCase(int mode) {
  int devs[kRanks] = {0, 0, 0, 0};
  std::vector<ncclComm_t> comms(kRanks), sub(kRanks, (ncclComm_t)0x1);
  NCCLCHECK(ncclCommInitAll(comms.data(), kRanks, devs));
  NCCLCHECK(ncclGroupStart());
  for (int r = 0; r < kRanks; r++) {
    const int color = mode == 0 ? r % 2 : mode == 1 ? 0 : (r < kRanks / 2 ? 0 : NCCL_SPLIT_NOCOLOR);
    NCCLCHECK(ncclCommSplit(comms[r], color, r, &sub[r], NULL));
  }
  NCCLCHECK(ncclGroupEnd());
  Bufs b;
  if (makeBufs(b, kRanks, 70000)) return 1;
  for (int r = 0; r < kRanks; r++) {
    if (mode == 2 && r >= kRanks / 2) { EXPECT(sub[r] == NULL); continue; }
    int subRank = -1, subN = -1, dev = -1;
    NCCLCHECK(ncclCommUserRank(sub[r], &subRank));
    NCCLCHECK(ncclCommCount(sub[r], &subN));
    NCCLCHECK(ncclCommCuDevice(sub[r], &dev));
    EXPECT(dev == 0);
    if (mode == 0) { EXPECT(subRank == r / 2); EXPECT(subN == kRanks / 2); }
    if (mode == 1) { EXPECT(subRank == r); EXPECT(subN == kRanks); }
    if (mode == 2) { EXPECT(subRank == r); EXPECT(subN == kRanks / 2); }
  }
  // the new communicators work: one group at a time (their kernels wait for each other)
  if (mode == 0) {
    for (int color = 0; color < 2; color++) {
      std::vector<ncclComm_t> g = {sub[color], sub[color + 2]};
      std::vector<int> ranks = {color, color + 2};
      if (allReduceCheck(g, ranks, b, "split by parity")) return 1;
    }
  } else if (mode == 1) {
    std::vector<int> ranks = {0, 1, 2, 3};
    if (allReduceCheck(sub, ranks, b, "split, one colour")) return 1;
  } else {
    std::vector<ncclComm_t> g = {sub[0], sub[1]};
    std::vector<int> ranks = {0, 1};
    if (allReduceCheck(g, ranks, b, "split, lower half")) return 1;
  }
  // the parents still work afterwards
  std::vector<int> all = {0, 1, 2, 3};
  if (allReduceCheck(comms, all, b, "parents after the split")) return 1;
  if (freeBufs(b)) return 1;
  for (auto c : sub) NCCLCHECK(ncclCommDestroy(c));         // NULL entries are accepted
  for (auto c : comms) NCCLCHECK(ncclCommDestroy(c));
  printf("ncclCommSplit case %d: ok\n", mode);
  return 0;
}

// A key order that differs from the rank order, and a new communicator of one rank.
static int splitByKey() {
  int devs[kRanks] = {0, 0, 0, 0};
  std::vector<ncclComm_t> comms(kRanks), sub(kRanks);
  NCCLCHECK(ncclCommInitAll(comms.data(), kRanks, devs));
  NCCLCHECK(ncclGroupStart());
  for (int r = 0; r < kRanks; r++) NCCLCHECK(ncclCommSplit(comms[r], r == 3 ? 7 : 5, r == 3 ? 0 : 10 - (r == 2 ? 0 : r), &sub[r], NULL));
  NCCLCHECK(ncclGroupEnd());
  // colour 5: keys 10 (rank 0), 9 (rank 1), 10 (rank 2) -> order 1, 0, 2 (equal keys keep the parent order)
  const int wantRank[kRanks] = {1, 0, 2, 0}, wantN[kRanks] = {3, 3, 3, 1};
  for (int r = 0; r < kRanks; r++) {
    int subRank = -1, subN = -1;
    NCCLCHECK(ncclCommUserRank(sub[r], &subRank));
    NCCLCHECK(ncclCommCount(sub[r], &subN));
    EXPECT(subRank == wantRank[r]); EXPECT(subN == wantN[r]);
  }
  ncclComm_t bad = NULL;
  EXPECT(ncclCommSplit(comms[0], -7, 0, &bad, NULL) == ncclInvalidArgument);
  EXPECT(ncclCommSplit(comms[0], 0, 0, NULL, NULL) == ncclInvalidArgument);
  for (auto c : sub) NCCLCHECK(ncclCommDestroy(c));
  for (auto c : comms) NCCLCHECK(ncclCommDestroy(c));
  printf("ncclCommSplit by key: ok\n");
  return 0;
}

static int nonBlocking() {
  const int nRanks = 2;
  ncclUniqueId id;
  NCCLCHECK(ncclGetUniqueId(&id));
  std::vector<ncclComm_t> comms(nRanks, NULL);
  ncclConfig_t config = NCCL_CONFIG_INITIALIZER;
  config.blocking = 0;
  HIPCHECK(hipSetDevice(0));
  NCCLCHECK(ncclGroupStart());
  for (int r = 0; r < nRanks; r++) NCCLCHECK(ncclCommInitRankConfig(&comms[r], nRanks, id, r, &config));
  ncclResult_t ge = ncclGroupEnd();
  EXPECT(ge == ncclSuccess || ge == ncclInProgress);
  int sawInProgress = ge == ncclInProgress;
  for (int r = 0; r < nRanks; r++) {
    EXPECT(comms[r] != NULL);                   // the handle is there at once
    ncclResult_t st = ncclInProgress;
    for (int spin = 0; spin < 600000 && st == ncclInProgress; spin++) {
      NCCLCHECK(ncclCommGetAsyncError(comms[r], &st));
      if (st == ncclInProgress) { sawInProgress = 1; usleep(100); }
    }
    EXPECT(st == ncclSuccess);
  }
  Bufs b;
  if (makeBufs(b, nRanks, 4096)) return 1;
  std::vector<int> ranks = {0, 1};
  if (allReduceCheck(comms, ranks, b, "non-blocking communicators")) return 1;
  if (freeBufs(b)) return 1;
  for (auto c : comms) { NCCLCHECK(ncclCommFinalize(c)); NCCLCHECK(ncclCommDestroy(c)); }
  // a creation that cannot end (the second rank never comes) is cut short by ncclCommAbort
  NCCLCHECK(ncclGetUniqueId(&id));
  ncclComm_t lonely = NULL;
  ncclResult_t r0 = ncclCommInitRankConfig(&lonely, 2, id, 0, &config);
  EXPECT(r0 == ncclInProgress);
  EXPECT(lonely != NULL);
  ncclResult_t st;
  NCCLCHECK(ncclCommGetAsyncError(lonely, &st));
  EXPECT(st == ncclInProgress);
  int* dummy = NULL;
  HIPCHECK(hipMalloc((void**)&dummy, 64));
  EXPECT(ncclAllReduce(dummy, dummy, 16, ncclInt, ncclSum, lonely, 0) == ncclInvalidArgument);   // not ready yet
  HIPCHECK(hipFree(dummy));
  NCCLCHECK(ncclCommAbort(lonely));
  printf("non-blocking creation: ok (ncclInProgress seen: %d)\n", sawInProgress);
  return 0;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const double t0 = now();
  setenv("RCCL_AMD_MULTI_RANK_GPU", "1", 1);
  setenv("RCCL_AMD_NCHANNELS", "16", 0);
  setenv("RCCL_AMD_SPIN_TIMEOUT_MS", "20000", 0);
  HIPCHECK(hipSetDevice(0));
  if (groupLaunchFour(1250)) return 1;            // one-shot low-latency protocol
  printf("[%.1f s]\n", now() - t0);
  if (groupLaunchFour(3 * 1048576 + 1000)) return 1;   // FIFO protocol, ragged
  printf("[%.1f s]\n", now() - t0);
  for (int mode = 0; mode < 3; mode++) { if (splitCase(mode)) return 1; printf("[%.1f s]\n", now() - t0); }
  if (splitByKey()) return 1;
  printf("[%.1f s]\n", now() - t0);
  if (nonBlocking()) return 1;
  printf("[%.1f s]\n", now() - t0);
  printf("multi-communicator test passed\n");
  return 0;
}

// A C++ caller written only against include/rccl_amd.h, the way a program written against
// <rccl/rccl.h> would be: ncclCommInitAll + ncclGroupStart/End + ncclAllReduce, then the same calls
// for ncclReduceScatter / ncclAllGather.  It replays the reference's known-answer test
// (test/StandaloneTests.cpp:191-246: 2 ranks, ncclInt, ncclSum, N = 1250, in[i] = i -> out[i] = 2*i),
// under the protocols that test cycles through (NCCL_PROTO = LL, Simple).
//
//   hipcc -I include tests/cxx/known_answer.cc -L rccl_amd -lrccl_amd -Wl,-rpath,'$ORIGIN/../../rccl_amd'
//
// Both ranks sit on device 0 (RCCL_AMD_MULTI_RANK_GPU=1) so that it runs on a one-GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "rccl_amd.h"

#define HIPCHECK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
#define NCCLCHECK(c) do { ncclResult_t r_ = (c); if (r_ != ncclSuccess) { printf("NCCL error %s (%s) at %s:%d\n", ncclGetErrorString(r_), ncclGetLastError(NULL), __FILE__, __LINE__); return 1; } } while (0)

static int runOnce(const char* proto) {
  setenv("NCCL_PROTO", proto, 1);
  const int nRanks = 2, N = 1250;
  int devs[nRanks] = {0, 0};
  ncclComm_t comms[nRanks];
  NCCLCHECK(ncclCommInitAll(comms, nRanks, devs));
  std::vector<int> in(N), expect(N), host(2 * N);
  for (int i = 0; i < N; i++) { in[i] = i; expect[i] = 2 * i; }
  int *dIn[nRanks], *dOut[nRanks], *dBig[nRanks];
  hipStream_t stream[nRanks];
  for (int r = 0; r < nRanks; r++) {
    HIPCHECK(hipStreamCreate(&stream[r]));
    HIPCHECK(hipMalloc((void**)&dIn[r], N * sizeof(int)));
    HIPCHECK(hipMalloc((void**)&dOut[r], N * sizeof(int)));
    HIPCHECK(hipMalloc((void**)&dBig[r], 2 * N * sizeof(int)));
    HIPCHECK(hipMemcpy(dIn[r], in.data(), N * sizeof(int), hipMemcpyHostToDevice));
    HIPCHECK(hipMemset(dOut[r], 0, N * sizeof(int)));
  }
  for (int iter = 0; iter < 25; iter++) {
    NCCLCHECK(ncclGroupStart());
    for (int r = 0; r < nRanks; r++)
      NCCLCHECK(ncclAllReduce(dIn[r], dOut[r], N, ncclInt, ncclSum, comms[r], stream[r]));
    NCCLCHECK(ncclGroupEnd());
    for (int r = 0; r < nRanks; r++) HIPCHECK(hipStreamSynchronize(stream[r]));
    for (int r = 0; r < nRanks; r++) {
      HIPCHECK(hipMemcpy(host.data(), dOut[r], N * sizeof(int), hipMemcpyDeviceToHost));
      for (int i = 0; i < N; i++)
        if (host[i] != expect[i]) { printf("proto %s iter %d rank %d: out[%d] = %d, expected %d\n", proto, iter, r, i, host[i], expect[i]); return 1; }
    }
  }
  // all-gather then reduce-scatter of the gathered buffer: every rank ends with 2 * (its own block)
  NCCLCHECK(ncclGroupStart());
  for (int r = 0; r < nRanks; r++) NCCLCHECK(ncclAllGather(dIn[r], dBig[r], N, ncclInt, comms[r], stream[r]));
  NCCLCHECK(ncclGroupEnd());
  NCCLCHECK(ncclGroupStart());
  for (int r = 0; r < nRanks; r++) NCCLCHECK(ncclReduceScatter(dBig[r], dOut[r], N, ncclInt, ncclSum, comms[r], stream[r]));
  NCCLCHECK(ncclGroupEnd());
  for (int r = 0; r < nRanks; r++) {
    HIPCHECK(hipStreamSynchronize(stream[r]));
    HIPCHECK(hipMemcpy(host.data(), dOut[r], N * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < N; i++)
      if (host[i] != 2 * i) { printf("proto %s gather/scatter rank %d: out[%d] = %d\n", proto, r, i, host[i]); return 1; }
  }
  int count = 0, rank = -1, dev = -1;
  NCCLCHECK(ncclCommCount(comms[1], &count));
  NCCLCHECK(ncclCommUserRank(comms[1], &rank));
  NCCLCHECK(ncclCommCuDevice(comms[1], &dev));
  if (count != 2 || rank != 1 || dev != 0) { printf("communicator queries wrong: %d %d %d\n", count, rank, dev); return 1; }
  ncclResult_t async;
  NCCLCHECK(ncclCommGetAsyncError(comms[0], &async));
  if (async != ncclSuccess) { printf("asynchronous error %d\n", async); return 1; }
  for (int r = 0; r < nRanks; r++) {
    HIPCHECK(hipFree(dIn[r])); HIPCHECK(hipFree(dOut[r])); HIPCHECK(hipFree(dBig[r]));
    HIPCHECK(hipStreamDestroy(stream[r]));
    NCCLCHECK(ncclCommDestroy(comms[r]));
  }
  printf("proto %s: ok\n", proto);
  return 0;
}

int main() {
  setenv("RCCL_AMD_MULTI_RANK_GPU", "1", 1);
  setenv("RCCL_AMD_SPIN_TIMEOUT_MS", "10000", 0);
  int version = 0;
  NCCLCHECK(ncclGetVersion(&version));
  printf("NCCL API level %d\n", version);
  const char* protos[] = {"LL", "Simple"};
  for (const char* p : protos)
    if (runOnce(p)) return 1;
  printf("known-answer test passed\n");
  return 0;
}

// Out-of-band rendezvous between the ranks of one communicator.
//
// Role of the reference's bootstrap (src/bootstrap.cc:244-333 bootstrapInit, ring
// all-gather and intra-node barrier): the ncclUniqueId names a TCP endpoint; every rank
// connects to it and the ranks then exchange fixed-size blobs (window IPC handles, pids,
// devices).  Single node only, so a star around the id's creator is enough: the creator
// runs a small service thread that gathers one blob per rank and returns the
// concatenation to everyone.
#pragma once
#include <atomic>
#include <mutex>
#include <stddef.h>
#include <stdint.h>
#include "rccl_amd.h"

struct RaBootstrap {
  int fd = -1;
  std::mutex fdMutex;                   // close() and interrupt() may run on different threads: the descriptor is opened,
                                        // shut down and closed under this lock, so interrupt() never touches a reused number
  std::atomic<bool> cancelled{false};   // set by interrupt(): every wait of this connection gives up
  int rank = 0, nRanks = 0;
  // Connect to the endpoint encoded in `id`.  Blocks until all nRanks have joined.
  ncclResult_t connect(const ncclUniqueId& id, int rank_, int nRanks_);
  // all[r*bytes .. (r+1)*bytes) = rank r's `mine`.
  ncclResult_t allGather(const void* mine, void* all, size_t bytes);
  ncclResult_t barrier();
  void close();
  // From another thread: makes a connect/allGather that is waiting fail now instead of at its time limit.
  void interrupt();
};

// Creates the listening endpoint and its service thread, and writes the id.
ncclResult_t raBootstrapCreate(ncclUniqueId* id);
// 64-bit value that differs between ids; used to match ranks living in one process.
uint64_t raBootstrapKey(const ncclUniqueId& id);

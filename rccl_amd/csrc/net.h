// Host-staged TCP transport (see net.cc).
#pragma once
#include "comm.h"

// True when the environment turns the peer-to-peer transport off (NCCL_P2P_DISABLE=1).
bool raNetRequested();
// Allocates the pinned host windows, fills c->peerWin with their device addresses, connects the
// socket mesh over the bootstrap and starts the proxy threads.  c->lay must be final.
ncclResult_t raNetSetup(ncclComm* c);
// drain: ship what this rank's (finished) kernels still have pending before the sockets close; false on abort / failed set-up.
void raNetShutdown(ncclComm* c, bool drain);

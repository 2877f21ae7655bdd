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
// Proxy-thread timers and counters since the communicator was made (all zero without the transport): tx ns in send() for
// data, ns in send() for credits, ns idle, ns total, data messages, payload bytes, channel scans, scans skipped by the
// channel's stamp; rx ns in recv() for data, ns in poll(), ns total, data messages.
#define RA_NET_STATS 12
void raNetStats(ncclComm* c, uint64_t* out);

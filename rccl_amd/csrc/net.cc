// Host-staged transport over TCP (the "net" side case).
//
// Reference behaviour being reproduced (src/transport/net.cc:1123-1644 send/recvProxyProgress
// with the built-in socket plugin src/transport/net_socket.cc, selected by
// NCCL_P2P_DISABLE=1 NCCL_SHM_DISABLE=1 NCCL_NET=Socket): the kernels are the same as for the
// peer-to-peer transport, but their FIFOs live in pinned HOST memory; a CPU proxy thread ships
// every published slot to the peer over a socket and relays the credits back.
//
// Here: each rank owns one real window in pinned host memory (what its kernels poll and read)
// and, per peer, a local "shadow" of that peer's window with the same layout.  A kernel writes
// into shadow(p) exactly what it would have stored into p's window over xGMI: slot data, the slot's
// byte count, tails and heads.  The tx thread mirrors those changes to rank p over TCP; p's rx
// thread applies them to p's real window.  Every byte therefore crosses
// GPU -> pinned host -> TCP (loopback) -> pinned host -> GPU.
#include "net.h"

#include <arpa/inet.h>
#include <errno.h>
#include <hip/hip_runtime.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <sys/socket.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <thread>
#include <vector>

#include "debug.h"

namespace {

enum { MSG_DATA = 1, MSG_CREDIT = 2 };
#define RA_NET_IDLE_SLEEP_US 20
struct MsgHeader {
  uint32_t type, ch, kind, nbytes;
  uint64_t step;
};

bool sendAll(int fd, const void* p, size_t n) {
  const char* c = (const char*)p;
  while (n) {
    ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
    if (k <= 0) { if (k < 0 && (errno == EINTR || errno == EAGAIN)) continue; return false; }
    c += k; n -= (size_t)k;
  }
  return true;
}
bool recvAll(int fd, void* p, size_t n) {
  char* c = (char*)p;
  while (n) {
    ssize_t k = ::recv(fd, c, n, 0);
    if (k <= 0) { if (k < 0 && (errno == EINTR || errno == EAGAIN)) continue; return false; }
    c += k; n -= (size_t)k;
  }
  return true;
}

inline uint64_t* flagAt(char* win, const RaWindowLayout& l, int which, int ch, int peer, int kind) {
  return (uint64_t*)(win + raFlagOffset(l, which, ch, peer, kind));
}

inline uint64_t nowNs() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

}  // namespace

struct RaNet {
  ncclComm* c = nullptr;
  int rank = 0, n = 0;
  RaWindowLayout lay{};
  char* real = nullptr;                  // host address of this rank's window
  std::vector<char*> shadow;             // host addresses, per peer
  std::vector<int> sock;                 // per peer
  std::vector<uint64_t> sentTail, sentHead;   // [peer][ch][kind]
  std::thread tx, rx;
  std::atomic<bool> stop{false};
  std::atomic<bool> draining{false};     // shutdown has begun: tx leaves after a pass that found nothing left to ship
  std::atomic<uint64_t> bytesSent{0};
  // Where the two proxy threads spend their time (nanoseconds since the communicator was made; rcclAmdCommNetStats):
  // tx: inside send() for slot data / for credits, in passes that found nothing to ship (idle), everything else
  // (looking at dirty words and flags); rx: inside recv() for slot data, waiting in poll(), everything else.
  std::atomic<uint64_t> txSendNs{0}, txCreditNs{0}, txIdleNs{0}, txTotalNs{0}, txMsgs{0}, txScans{0}, txSkips{0};
  std::atomic<uint64_t> rxRecvNs{0}, rxPollNs{0}, rxTotalNs{0}, rxMsgs{0};

  size_t ix(int p, int ch, int k) const { return ((size_t)p * lay.nChannels + ch) * RA_NKINDS + k; }

  void fail(const char* what) {
    if (stop.load()) return;
    RA_WARN("host-staged transport: %s failed on rank %d: %s", what, rank, strerror(errno));
    if (c->hostFlags) __atomic_store_n(&c->hostFlags[0], 1u, __ATOMIC_RELAXED);   // kernels give up
  }

  // One look at everything this rank's kernels have posted on channel ch; ships what is new.  false: a send failed.
  bool shipChannel(int ch, bool& progress) {
    for (int p = 0; p < n; p++) {
      if (p == rank) continue;
      char* sh = shadow[(size_t)p];
      for (int k = 0; k < RA_NKINDS; k++) {
        // slots this rank's kernels published towards p
        const uint64_t tail = __atomic_load_n(flagAt(sh, lay, 0, ch, rank, k), __ATOMIC_ACQUIRE);
        uint64_t& st = sentTail[ix(p, ch, k)];
        while (st < tail) {
          const uint64_t step = st + 1;
          const int slot = (int)((step - 1) % (uint64_t)lay.nSlots);
          uint64_t nbytes = __atomic_load_n(flagAt(sh, lay, 2 + slot, ch, rank, k), __ATOMIC_RELAXED);
          if (nbytes > lay.slotBytes) nbytes = lay.slotBytes;
          MsgHeader h{MSG_DATA, (uint32_t)ch, (uint32_t)k, (uint32_t)nbytes, step};
          const char* data = sh + raDataOffset(lay, ch, p, rank, k) + (size_t)slot * lay.slotBytes;
          const uint64_t t0 = nowNs();
          if (!sendAll(sock[(size_t)p], &h, sizeof h) || !sendAll(sock[(size_t)p], data, (size_t)nbytes)) { fail("send"); return false; }
          txSendNs += nowNs() - t0;
          txMsgs++;
          bytesSent += nbytes;
          st = step;
          progress = true;
        }
        // slots this rank's kernels consumed from p: credit back
        const uint64_t head = __atomic_load_n(flagAt(sh, lay, 1, ch, rank, k), __ATOMIC_ACQUIRE);
        uint64_t& shd = sentHead[ix(p, ch, k)];
        if (shd < head) {
          MsgHeader h{MSG_CREDIT, (uint32_t)ch, (uint32_t)k, 0, head};
          const uint64_t t0 = nowNs();
          if (!sendAll(sock[(size_t)p], &h, sizeof h)) { fail("send"); return false; }
          txCreditNs += nowNs() - t0;
          shd = head;
          progress = true;
        }
      }
    }
    return true;
  }

  // The kernels stamp one word per channel in this rank's own window (the flag line a rank would use for a FIFO to
  // itself, which does not exist) after every batch of flag posts, behind the posts.  A pass looks at those nChannels
  // words and walks the n x 2 kinds x 2 flag lines of a channel only when its stamp has changed; every few milliseconds,
  // and while shutting down, a pass walks everything regardless.
  void txLoop() {
    std::vector<uint64_t> seen((size_t)lay.nChannels, 0);
    const char* fs = getenv("RCCL_AMD_NET_FULL_SCAN");          // 1: walk every flag line in every pass (for comparison)
    const bool always = fs && atoi(fs) != 0;
    // After a run of passes with nothing to ship the thread sleeps between passes instead of yielding (microseconds;
    // 0: always yield).  On a node with fewer cores than proxy threads a spinning tx thread takes the core the rx thread
    // of another rank needs.
    const char* sl = getenv("RCCL_AMD_NET_IDLE_SLEEP_US");
    const long idleSleepUs = sl ? atol(sl) : RA_NET_IDLE_SLEEP_US;
    int idlePasses = 0;
    const uint64_t start = nowNs();
    uint64_t lastFull = start;
    while (!stop.load(std::memory_order_relaxed)) {
      const uint64_t passStart = nowNs();
      const bool full = always || draining.load(std::memory_order_relaxed) || passStart - lastFull > 2000000ull;
      if (full) lastFull = passStart;
      bool progress = false;
      for (int ch = 0; ch < lay.nChannels; ch++) {
        const uint64_t stamp = __atomic_load_n(flagAt(real, lay, 0, ch, rank, 0), __ATOMIC_ACQUIRE);
        if (!full && stamp == seen[(size_t)ch]) { txSkips++; continue; }
        seen[(size_t)ch] = stamp;
        txScans++;
        if (!shipChannel(ch, progress)) { txTotalNs = nowNs() - start; return; }
      }
      if (!progress) {
        if (draining.load(std::memory_order_relaxed) && full) break;   // this rank's kernels are done and all of it is shipped
        if (idleSleepUs > 0 && ++idlePasses > 64) {
          struct timespec ts = {0, idleSleepUs * 1000};
          nanosleep(&ts, nullptr);
        } else {
          sched_yield();
        }
        txIdleNs += nowNs() - passStart;
      } else {
        idlePasses = 0;
      }
      txTotalNs = nowNs() - start;
    }
    txTotalNs = nowNs() - start;
  }

  void rxLoop() {
    std::vector<struct pollfd> pf;
    std::vector<int> who;
    for (int p = 0; p < n; p++)
      if (p != rank) { pf.push_back({sock[(size_t)p], POLLIN, 0}); who.push_back(p); }
    const uint64_t start = nowNs();
    while (!stop.load(std::memory_order_relaxed)) {
      const uint64_t p0 = nowNs();
      int r = ::poll(pf.data(), (nfds_t)pf.size(), 50);
      const uint64_t p1 = nowNs();
      rxPollNs += p1 - p0;
      rxTotalNs = p1 - start;
      if (r < 0) { if (errno == EINTR) continue; fail("poll"); return; }
      if (r == 0) continue;
      for (size_t i = 0; i < pf.size(); i++) {
        if (!(pf[i].revents & (POLLIN | POLLHUP | POLLERR))) continue;
        const int p = who[i];
        MsgHeader h;
        // A peer that has shipped everything closes its sending side: an end of stream on a message boundary is
        // that rank leaving in good order, not a failure (what it owed this rank has already arrived).
        ssize_t got = ::recv(pf[i].fd, &h, sizeof h, MSG_WAITALL);
        if (got == 0) { pf[i].fd = -1; continue; }
        if (got < 0 && (errno == EINTR || errno == EAGAIN)) continue;
        if (got != (ssize_t)sizeof h) { if (!stop.load()) fail("recv"); return; }
        if ((int)h.ch >= lay.nChannels || h.kind >= RA_NKINDS || h.nbytes > lay.slotBytes) { errno = EPROTO; fail("message check"); return; }
        if (h.type == MSG_DATA) {
          const int slot = (int)((h.step - 1) % (uint64_t)lay.nSlots);
          char* dst = real + raDataOffset(lay, (int)h.ch, rank, p, (int)h.kind) + (size_t)slot * lay.slotBytes;
          const uint64_t r0 = nowNs();
          if (h.nbytes && !recvAll(pf[i].fd, dst, h.nbytes)) { if (!stop.load()) fail("recv"); return; }
          rxRecvNs += nowNs() - r0;
          rxMsgs++;
          __atomic_store_n(flagAt(real, lay, 0, (int)h.ch, p, (int)h.kind), h.step, __ATOMIC_RELEASE);
        } else if (h.type == MSG_CREDIT) {
          __atomic_store_n(flagAt(real, lay, 1, (int)h.ch, p, (int)h.kind), h.step, __ATOMIC_RELEASE);
        } else {
          errno = EPROTO; fail("message type"); return;
        }
      }
    }
  }
};

bool raNetRequested() {
  // NCCL_P2P_DISABLE=1 turns the peer-to-peer transport off (src/graph/paths.cc:287-291).  The reference would try
  // shared memory next (transport order P2P, SHM, NET: src/transport.cc:14-53) unless NCCL_SHM_DISABLE=1
  // (src/transport/shm.cc:40-54); this library has no shared-memory transport, so with peer-to-peer off the socket
  // transport is what remains, whatever NCCL_SHM_DISABLE says.  NCCL_NET (src/init.cc:2150) can only name that one,
  // and a single node's ranks always meet over the loopback interface (NCCL_SOCKET_IFNAME, src/misc/socket.cc:331-357).
  const char* e = getenv("NCCL_P2P_DISABLE");
  const bool on = e && atoi(e) != 0;
  static std::atomic<bool> said{false};
  if (on && !said.exchange(true)) {
    const char* shm = getenv("NCCL_SHM_DISABLE");
    const char* net = getenv("NCCL_NET");
    const char* ifn = getenv("NCCL_SOCKET_IFNAME");
    if (!(shm && atoi(shm) != 0))
      RA_INFO("NCCL_P2P_DISABLE=1 without NCCL_SHM_DISABLE=1: there is no shared-memory transport here, the socket transport is used");
    if (net && *net && strcasecmp(net, "Socket"))
      RA_WARN("NCCL_NET=%s: the only network transport here is the built-in one (\"Socket\"); using it", net);
    if (ifn && *ifn && strcmp(ifn, "lo") && strcmp(ifn, "=lo"))
      RA_INFO("NCCL_SOCKET_IFNAME=%s: ranks of one node are connected over the loopback interface", ifn);
  }
  return on;
}

ncclResult_t raNetSetup(ncclComm* c) {
  RaNet* N = new RaNet;
  c->net = N;
  N->c = c; N->rank = c->rank; N->n = c->nRanks; N->lay = c->lay;
  const size_t bytes = c->lay.totalBytes;
  N->shadow.assign((size_t)N->n, nullptr);
  N->sock.assign((size_t)N->n, -1);
  N->sentTail.assign((size_t)N->n * c->lay.nChannels * RA_NKINDS, 0);
  N->sentHead.assign((size_t)N->n * c->lay.nChannels * RA_NKINDS, 0);
  // pinned, GPU-mapped host memory (the reference: ncclCudaHostCalloc, src/transport/net.cc:798-800)
  for (int p = 0; p < N->n; p++) {
    void* h = nullptr;
    RA_HIP(hipHostMalloc(&h, bytes, hipHostMallocMapped | hipHostMallocCoherent));   // CPU threads and kernels exchange data through it
    memset(h, 0, c->lay.dataOffset);
    void* d = nullptr;
    RA_HIP(hipHostGetDevicePointer(&d, h, 0));
    if (p == c->rank) N->real = (char*)h; else N->shadow[(size_t)p] = (char*)h;
    c->peerWin[p] = (char*)d;
  }
  RA_INFO("rank %d: host-staged transport, %d windows of %zu MiB in pinned host memory", c->rank, N->n, bytes >> 20);

  // full mesh of loopback sockets: ports are exchanged over the bootstrap, lower rank connects
  int lfd = ::socket(AF_INET, SOCK_STREAM, 0);
  if (lfd < 0) return ncclSystemError;
  struct sockaddr_in sa; memset(&sa, 0, sizeof sa);
  sa.sin_family = AF_INET; sa.sin_addr.s_addr = htonl(INADDR_LOOPBACK); sa.sin_port = 0;
  socklen_t sl = sizeof sa;
  if (::bind(lfd, (struct sockaddr*)&sa, sizeof sa) || ::listen(lfd, 64) || ::getsockname(lfd, (struct sockaddr*)&sa, &sl)) {
    RA_WARN("host-staged transport: cannot listen: %s", strerror(errno));
    ::close(lfd);
    return ncclSystemError;
  }
  std::vector<uint32_t> ports((size_t)N->n);
  uint32_t mine = ntohs(sa.sin_port);
  ncclResult_t res = c->boot.allGather(&mine, ports.data(), sizeof mine);
  if (res != ncclSuccess) { ::close(lfd); return res; }
  const int one = 1, big = 8 << 20;
  for (int p = N->rank + 1; p < N->n; p++) {          // connect to the higher ranks
    int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    struct sockaddr_in pa; memset(&pa, 0, sizeof pa);
    pa.sin_family = AF_INET; pa.sin_addr.s_addr = htonl(INADDR_LOOPBACK); pa.sin_port = htons((uint16_t)ports[(size_t)p]);
    if (fd < 0 || ::connect(fd, (struct sockaddr*)&pa, sizeof pa)) { RA_WARN("host-staged transport: connect to rank %d: %s", p, strerror(errno)); ::close(lfd); return ncclSystemError; }
    int32_t me = N->rank;
    if (!sendAll(fd, &me, sizeof me)) { ::close(lfd); return ncclSystemError; }
    N->sock[(size_t)p] = fd;
  }
  for (int k = 0; k < N->rank; k++) {                  // accept the lower ranks
    int fd = ::accept(lfd, nullptr, nullptr);
    int32_t who = -1;
    if (fd < 0 || !recvAll(fd, &who, sizeof who) || who < 0 || who >= N->rank || N->sock[(size_t)who] >= 0) {
      RA_WARN("host-staged transport: accept failed: %s", strerror(errno));
      ::close(lfd);
      return ncclSystemError;
    }
    N->sock[(size_t)who] = fd;
  }
  ::close(lfd);
  for (int p = 0; p < N->n; p++)
    if (N->sock[(size_t)p] >= 0) {
      setsockopt(N->sock[(size_t)p], IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
      setsockopt(N->sock[(size_t)p], SOL_SOCKET, SO_SNDBUF, &big, sizeof big);
      setsockopt(N->sock[(size_t)p], SOL_SOCKET, SO_RCVBUF, &big, sizeof big);
    }
  N->tx = std::thread([N]() { N->txLoop(); });
  N->rx = std::thread([N]() { N->rxLoop(); });
  return ncclSuccess;
}

void raNetStats(ncclComm* c, uint64_t* out) {
  RaNet* N = c->net;
  for (int i = 0; i < RA_NET_STATS; i++) out[i] = 0;
  if (!N) return;
  out[0] = N->txSendNs; out[1] = N->txCreditNs; out[2] = N->txIdleNs; out[3] = N->txTotalNs; out[4] = N->txMsgs;
  out[5] = N->bytesSent; out[6] = N->txScans; out[7] = N->txSkips;
  out[8] = N->rxRecvNs; out[9] = N->rxPollNs; out[10] = N->rxTotalNs; out[11] = N->rxMsgs;
}

void raNetShutdown(ncclComm* c, bool drain) {
  RaNet* N = c->net;
  if (!N) return;
  // The caller has waited for this rank's kernels.  What they published may still sit in the shadows: let the tx
  // thread ship it (it leaves after a pass with nothing to do), then close the sending side so that every peer
  // sees an orderly end of stream, and only then stop receiving.
  if (drain) N->draining.store(true); else N->stop.store(true);
  if (N->tx.joinable()) N->tx.join();
  for (int fd : N->sock) if (fd >= 0) ::shutdown(fd, SHUT_WR);
  N->stop.store(true);
  for (int fd : N->sock) if (fd >= 0) ::shutdown(fd, SHUT_RD);
  if (N->rx.joinable()) N->rx.join();
  for (int fd : N->sock) if (fd >= 0) ::close(fd);
  if (N->real) (void)hipHostFree(N->real);
  for (char* s : N->shadow) if (s) (void)hipHostFree(s);
  RA_INFO("rank %d: host-staged transport shipped %llu payload bytes in %llu messages; tx thread %.1f ms in send() for data, %.1f ms for "
          "credits, %.1f ms idle of %.1f ms (%llu channel scans, %llu skipped by their stamp); rx thread %.1f ms in recv() for data, "
          "%.1f ms in poll() of %.1f ms", c->rank, (unsigned long long)N->bytesSent.load(), (unsigned long long)N->txMsgs.load(),
          N->txSendNs / 1e6, N->txCreditNs / 1e6, N->txIdleNs / 1e6, N->txTotalNs / 1e6, (unsigned long long)N->txScans.load(),
          (unsigned long long)N->txSkips.load(), N->rxRecvNs / 1e6, N->rxPollNs / 1e6, N->rxTotalNs / 1e6);
  delete N;
  c->net = nullptr;
}

#include "bootstrap.h"

#include <arpa/inet.h>
#include <errno.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <unistd.h>

#include <random>
#include <mutex>
#include <thread>
#include <vector>

#include "debug.h"

namespace {

constexpr uint64_t kMagic = 0x52414d4443434c31ull;  // "RAMDCCL1"

struct IdPayload {      // lives inside ncclUniqueId::internal
  uint64_t magic;
  uint64_t nonce;
  uint32_t addr;        // network byte order
  uint16_t port;        // network byte order
  uint16_t pad;
};
static_assert(sizeof(IdPayload) <= NCCL_UNIQUE_ID_BYTES, "id payload must fit");

struct Hello {
  uint64_t magic, nonce;
  int32_t rank, nRanks;
};

bool sendAll(int fd, const void* p, size_t n) {
  const char* c = (const char*)p;
  while (n) {
    ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
    if (k <= 0) { if (k < 0 && errno == EINTR) continue; return false; }
    c += k; n -= (size_t)k;
  }
  return true;
}
bool recvAll(int fd, void* p, size_t n) {
  char* c = (char*)p;
  while (n) {
    ssize_t k = ::recv(fd, c, n, 0);
    if (k <= 0) { if (k < 0 && errno == EINTR) continue; return false; }
    c += k; n -= (size_t)k;
  }
  return true;
}

int timeoutSeconds();
// A rank's receive: in slices of 100 ms so that RaBootstrap::interrupt() ends it, `limitS` seconds in all (0: no limit).
bool recvWatched(int fd, void* p, size_t n, const std::atomic<bool>& cancelled, int limitS) {
  char* c = (char*)p;
  int waitedMs = 0;
  while (n) {
    if (cancelled.load(std::memory_order_relaxed)) { errno = ECANCELED; return false; }
    struct pollfd pf; pf.fd = fd; pf.events = POLLIN; pf.revents = 0;
    const int r = ::poll(&pf, 1, 100);
    if (r < 0) { if (errno == EINTR) continue; return false; }
    if (r == 0) { waitedMs += 100; if (limitS > 0 && waitedMs >= limitS * 1000) { errno = ETIMEDOUT; return false; } continue; }
    const ssize_t k = ::recv(fd, c, n, 0);
    if (k <= 0) { if (k < 0 && (errno == EINTR || errno == EAGAIN)) continue; return false; }
    c += k; n -= (size_t)k;
  }
  return true;
}

int timeoutSeconds() {
  const char* e = getenv("RCCL_AMD_BOOTSTRAP_TIMEOUT_S");
  int t = e ? atoi(e) : 120;
  return t > 0 ? t : 120;
}
void setTimeouts(int fd) {
  struct timeval tv; tv.tv_sec = timeoutSeconds(); tv.tv_usec = 0;
  setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
  setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
}

// Service thread of one id: admit nRanks connections, then serve gather rounds until a
// rank hangs up.
void serveBody(int lfd, uint64_t nonce, std::vector<int>& fds);

// Runs detached: nothing may escape it (an exception leaving a detached thread ends the whole process).
void serve(int lfd, uint64_t nonce) {
  std::vector<int> fds;
  try {
    serveBody(lfd, nonce, fds);
  } catch (...) {
    RA_WARN("bootstrap: the rendezvous service of one ncclUniqueId stopped on an exception");
  }
  for (int fd : fds) if (fd >= 0) ::close(fd);
}

void serveBody(int lfd, uint64_t nonce, std::vector<int>& fds) {
  int nRanks = -1, joined = 0;
  struct timeval tv; tv.tv_sec = 3600; tv.tv_usec = 0;   // an id nobody uses is dropped after an hour
  setsockopt(lfd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
  while (nRanks < 0 || joined < nRanks) {
    int fd = ::accept(lfd, nullptr, nullptr);
    if (fd < 0) { if (errno == EINTR) continue; break; }
    setTimeouts(fd);
    Hello h;
    if (!recvAll(fd, &h, sizeof h) || h.magic != kMagic || h.nonce != nonce || h.nRanks < 1 ||
        h.rank < 0 || h.rank >= h.nRanks || (nRanks >= 0 && h.nRanks != nRanks)) {
      ::close(fd);
      continue;
    }
    if (nRanks < 0) { nRanks = h.nRanks; fds.assign((size_t)nRanks, -1); }
    if (fds[(size_t)h.rank] >= 0) { ::close(fd); continue; }   // duplicate rank
    fds[(size_t)h.rank] = fd;
    joined++;
  }
  ::close(lfd);
  bool ok = nRanks > 0 && joined == nRanks;
  std::vector<char> buf;
  while (ok) {
    uint64_t len = 0;
    for (int r = 0; r < nRanks && ok; r++) {
      uint64_t l = 0;
      // a round may start long after the previous one: wait without a limit here
      struct timeval none; none.tv_sec = 0; none.tv_usec = 0;
      setsockopt(fds[(size_t)r], SOL_SOCKET, SO_RCVTIMEO, &none, sizeof none);
      ok = recvAll(fds[(size_t)r], &l, sizeof l);
      setTimeouts(fds[(size_t)r]);
      if (!ok) break;
      // the length is checked before anything is sized by it: a bogus value ends the service, it does not throw
      if (l > (64u << 20) || (r > 0 && l != len)) { ok = false; break; }
      if (r == 0) { len = l; buf.resize((size_t)len * (size_t)nRanks); }
      ok = len == 0 || recvAll(fds[(size_t)r], buf.data() + (size_t)r * len, (size_t)len);
    }
    for (int r = 0; r < nRanks && ok; r++)
      ok = sendAll(fds[(size_t)r], &len, sizeof len) && (len == 0 || sendAll(fds[(size_t)r], buf.data(), buf.size()));
  }
}

}  // namespace

uint64_t raBootstrapKey(const ncclUniqueId& id) {
  IdPayload p; memcpy(&p, id.internal, sizeof p);
  return p.nonce;
}

ncclResult_t raBootstrapCreate(ncclUniqueId* id) {
  memset(id, 0, sizeof *id);
  int lfd = ::socket(AF_INET, SOCK_STREAM, 0);
  if (lfd < 0) { RA_WARN("bootstrap: socket() failed: %s", strerror(errno)); return ncclSystemError; }
  int one = 1;
  setsockopt(lfd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
  struct sockaddr_in sa; memset(&sa, 0, sizeof sa);
  sa.sin_family = AF_INET;
  const char* host = getenv("RCCL_AMD_BOOTSTRAP_HOST");
  if (!host || inet_pton(AF_INET, host, &sa.sin_addr) != 1) sa.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
  sa.sin_port = 0;
  socklen_t sl = sizeof sa;
  if (::bind(lfd, (struct sockaddr*)&sa, sizeof sa) || ::listen(lfd, 256) ||
      ::getsockname(lfd, (struct sockaddr*)&sa, &sl)) {
    RA_WARN("bootstrap: bind/listen failed: %s", strerror(errno));
    ::close(lfd);
    return ncclSystemError;
  }
  std::random_device rd;
  IdPayload p; memset(&p, 0, sizeof p);
  p.magic = kMagic;
  p.nonce = ((uint64_t)rd() << 32) ^ (uint64_t)rd() ^ ((uint64_t)getpid() << 16);
  p.addr = sa.sin_addr.s_addr;
  p.port = sa.sin_port;
  memcpy(id->internal, &p, sizeof p);
  std::thread(serve, lfd, p.nonce).detach();
  return ncclSuccess;
}

ncclResult_t RaBootstrap::connect(const ncclUniqueId& id, int rank_, int nRanks_) {
  IdPayload p; memcpy(&p, id.internal, sizeof p);
  if (p.magic != kMagic) { RA_WARN("bootstrap: the ncclUniqueId was not produced by ncclGetUniqueId"); return ncclInvalidArgument; }
  rank = rank_; nRanks = nRanks_;
  if (cancelled.load()) return ncclSystemError;
  {
    std::lock_guard<std::mutex> g(fdMutex);
    fd = ::socket(AF_INET, SOCK_STREAM, 0);
  }
  if (fd < 0) return ncclSystemError;
  setTimeouts(fd);
  struct sockaddr_in sa; memset(&sa, 0, sizeof sa);
  sa.sin_family = AF_INET; sa.sin_addr.s_addr = p.addr; sa.sin_port = p.port;
  int tries = 0;
  while (::connect(fd, (struct sockaddr*)&sa, sizeof sa)) {
    if ((errno == ECONNREFUSED || errno == EINTR) && ++tries < 200 && !cancelled.load()) { usleep(50000); continue; }
    RA_WARN("bootstrap: connect failed: %s", strerror(errno));
    close();
    return ncclSystemError;
  }
  Hello h; h.magic = kMagic; h.nonce = p.nonce; h.rank = rank; h.nRanks = nRanks;
  if (!sendAll(fd, &h, sizeof h)) { close(); return ncclSystemError; }
  return barrier();   // returns once every rank has joined
}

ncclResult_t RaBootstrap::allGather(const void* mine, void* all, size_t bytes) {
  if (fd < 0) return ncclInternalError;
  uint64_t len = bytes, back = 0;
  const int limit = timeoutSeconds();
  if (!sendAll(fd, &len, sizeof len) || (bytes && !sendAll(fd, mine, bytes)) || !recvWatched(fd, &back, sizeof back, cancelled, limit) ||
      back != len || (bytes && !recvWatched(fd, all, bytes * (size_t)nRanks, cancelled, limit))) {
    RA_WARN("bootstrap: exchange with the other ranks failed (rank %d of %d): %s", rank, nRanks, strerror(errno));
    return ncclRemoteError;
  }
  return ncclSuccess;
}

ncclResult_t RaBootstrap::barrier() { return allGather(nullptr, nullptr, 0); }

void RaBootstrap::close() {
  std::lock_guard<std::mutex> g(fdMutex);
  if (fd >= 0) { ::close(fd); fd = -1; }
}

void RaBootstrap::interrupt() {
  cancelled.store(true);
  std::lock_guard<std::mutex> g(fdMutex);
  if (fd >= 0) ::shutdown(fd, SHUT_RDWR);
}

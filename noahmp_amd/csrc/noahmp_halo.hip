// Halo exchange of the MMF lateral-flow ring through the C-ABI: noahmp_hip_halo_init / noahmp_hip_exchange_halo.
//
// LATERALFLOW (gw:231-292) reads WTD, FDEPTH, TOPO and ISLTYP on the 1-cell ring around a rank's tile, corners included.  The
// reference fills such rings with mpp_land_comlr_real followed by mpp_land_comub_real, flag 99 (mpp:344-369, 603-642): first
// left/right over the tile's rows, then down/up over whole memory rows, which by then carry the columns received in the first
// phase -- two DEPENDENT message phases.  This file delivers the same cells in ONE phase: the four tile edges go to the four edge
// neighbours and the four corner cells to the four diagonal ranks, all at once (the exchange is latency-bound: ~25 KB per edge
// at the config-4 grid on 8 ranks).  For a caller with one rank per GPU (mpp_land_get_nprocsxy's rank grid, mpp:124-141;
// neighbours as mpp:93-107 plus the diagonals); needs neither MPI nor torch:
//   * rendezvous over TCP (rank 0 listens on master_addr:master_port; every rank learns its neighbours' listeners);
//   * transport NOAHMP_HALO_RCCL: ncclSend / ncclRecv of the packed edges to / from all <= 8 neighbours in ONE group on the caller's
//     stream (RCCL over xGMI, GPU-direct; librccl is loaded with dlopen at init, its unique id travels over the rendezvous);
//   * transport NOAHMP_HALO_TCP: the packed edges travel over the rendezvous sockets (host planes directly, device planes through a
//     pinned staging buffer) -- the form the CPU tests and single-GPU checks use, and a fallback where RCCL is not available.
// All planes of a call share the messages.
#include <arpa/inet.h>
#include <dlfcn.h>
#include <errno.h>
#include <hip/hip_runtime.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <time.h>
#include <unistd.h>
#include <string>
#include <vector>
#include "noahmp_hip.h"
#include "nmp_engine_host.hpp"

using nmp_host::g;

namespace {

// ---- RCCL entry points (resolved at run time: the library is not a link dependency)
typedef struct { char internal[128]; } NcclUniqueId;
typedef void* NcclComm;
struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
constexpr int kNcclInt32 = 2;      // ncclInt32 / ncclInt (nccl.h ncclDataType_t): planes travel as 4-byte words

struct Halo {
  bool up = false;
  int rank = 0, nranks = 1, npx = 1, npy = 1, transport = NOAHMP_HALO_TCP;
  // left, right, down, up, down-left, up-right, down-right, up-left: edge types first, inside a type the lower-ranked peer first, so
  // that the blocking pairwise transfers of the socket transport follow ONE global order of the links (no cyclic wait)
  int nb[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
  int sock[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
  Rccl rccl;
  NcclComm comm = nullptr;
  // staging
  uint32_t* d_send = nullptr; uint32_t* d_recv = nullptr; size_t d_words = 0;
  uint32_t* h_send = nullptr; uint32_t* h_recv = nullptr; size_t h_words = 0; bool h_pinned = false;
} H;

int fail(const std::string& m, int rc = -109) { g.last_error = "noahmp_hip_halo: " + m; return rc; }

int load_rccl() {
  if (H.rccl.Send) return 0;
  H.rccl.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!H.rccl.handle) H.rccl.handle = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!H.rccl.handle) return fail("librccl.so not found (dlopen)");
#define NMP_SYM(field, name) *(void**)&H.rccl.field = dlsym(H.rccl.handle, name); if (!H.rccl.field) return fail(std::string("librccl: no symbol ") + name);
  NMP_SYM(GetUniqueId, "ncclGetUniqueId") NMP_SYM(CommInitRank, "ncclCommInitRank") NMP_SYM(CommDestroy, "ncclCommDestroy")
  NMP_SYM(GroupStart, "ncclGroupStart") NMP_SYM(GroupEnd, "ncclGroupEnd") NMP_SYM(Send, "ncclSend") NMP_SYM(Recv, "ncclRecv")
#undef NMP_SYM
  *(void**)&H.rccl.GetErrorString = dlsym(H.rccl.handle, "ncclGetErrorString");
  return 0;
}

// ---- sockets
bool send_all(int fd, const void* p, size_t n) {
  const char* c = (const char*)p;
  while (n) { ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL); if (k <= 0) { if (errno == EINTR) continue; return false; } c += k; n -= k; }
  return true;
}
bool recv_all(int fd, void* p, size_t n) {
  char* c = (char*)p;
  while (n) { ssize_t k = ::recv(fd, c, n, 0); if (k <= 0) { if (k < 0 && errno == EINTR) continue; return false; } c += k; n -= k; }
  return true;
}
// Every wait of the rendezvous and of the socket transport has a deadline (default 120 s, NMP_HALO_TIMEOUT_S): a rank that dies or
// never arrives ends the others with an error instead of parking them on a GPU box.
double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
double halo_timeout_s() {
  const char* e = getenv("NMP_HALO_TIMEOUT_S");
  const double v = e ? atof(e) : 120.0;
  return v > 0.0 ? v : 120.0;
}
// the DATA path has its own, longer deadline (NMP_HALO_IO_TIMEOUT_S, default 1800 s): a neighbour that is late for a step -- its
// first step compiles a kernel, it sorts again, it writes a restart file -- is not a dead neighbour
double halo_io_timeout_s() {
  const char* e = getenv("NMP_HALO_IO_TIMEOUT_S");
  const double v = e ? atof(e) : 1800.0;
  return v > 0.0 ? v : 1800.0;
}
void set_io_timeout(int fd, double sec) {
  timeval tv; tv.tv_sec = (long)sec; tv.tv_usec = (long)((sec - (double)tv.tv_sec) * 1e6);
  setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
  setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
}
// accept() with a deadline (poll); the accepted socket gets the I/O timeout.  -1: nobody came in time
int accept_deadline(int lfd, double deadline, sockaddr_in* from) {
  pollfd p; p.fd = lfd; p.events = POLLIN; p.revents = 0;
  for (;;) {
    const double left = deadline - now_s();
    if (left <= 0.0) return -1;
    const int r = poll(&p, 1, (int)(left * 1000.0) + 1);
    if (r < 0 && errno == EINTR) continue;
    if (r <= 0) return -1;
    sockaddr_in a; socklen_t l = sizeof a;
    const int fd = accept(lfd, (sockaddr*)&a, &l);
    if (fd >= 0) { if (from) *from = a; set_io_timeout(fd, halo_timeout_s()); return fd; }
    if (errno == EINTR || errno == EAGAIN || errno == ECONNABORTED) continue;
    return -1;
  }
}
int listen_on(int port, int* port_out) {
  int fd = socket(AF_INET, SOCK_STREAM, 0);
  if (fd < 0) return -1;
  int one = 1;
  setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
  sockaddr_in a; memset(&a, 0, sizeof a);
  a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_ANY); a.sin_port = htons((uint16_t)port);
  if (bind(fd, (sockaddr*)&a, sizeof a) || listen(fd, 64)) { close(fd); return -1; }
  socklen_t l = sizeof a;
  getsockname(fd, (sockaddr*)&a, &l);
  if (port_out) *port_out = ntohs(a.sin_port);
  return fd;
}
int connect_to(uint32_t ip_be, int port, double timeout_s) {
  const double t0 = now_s();
  for (;;) {
    int fd = socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) return -1;
    sockaddr_in a; memset(&a, 0, sizeof a);
    a.sin_family = AF_INET; a.sin_addr.s_addr = ip_be; a.sin_port = htons((uint16_t)port);
    if (connect(fd, (sockaddr*)&a, sizeof a) == 0) {
      int one = 1; setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
      set_io_timeout(fd, halo_timeout_s());
      return fd;
    }
    close(fd);
    if (now_s() - t0 > timeout_s) return -1;
    usleep(50000);                         // the master may not be listening yet
  }
}
uint32_t resolve(const char* host) {
  in_addr a;
  if (inet_aton(host, &a)) return a.s_addr;
  hostent* he = gethostbyname(host);
  if (he && he->h_addrtype == AF_INET) return *(uint32_t*)he->h_addr_list[0];
  return htonl(INADDR_LOOPBACK);
}

// mpp_land_get_nprocsxy (mpp:124-141): most-square factorisation, first best wins
void nprocs_xy(int n, int& nx, int& ny) {
  int best = n; nx = n; ny = 1;
  for (int j = 1; j <= n; j++) if (n % j == 0) { const int i = n / j; const int d = i > j ? i - j : j - i; if (d < best) { best = d; nx = i; ny = j; } }
}

struct Peer { uint32_t ip; int port; };

// ---- edge packing.  An edge = `count` words of a plane starting at `first`, `stride` apart; n planes share a message.
struct EdgeDesc { long first, stride; int count; long buf_off; };     // buf_off: word offset of plane 0's copy in the staging buffer
constexpr int kMaxPlanes = 64, kMaxEdges = 8;
struct PackArgs { void* planes[kMaxPlanes]; int n; EdgeDesc e[kMaxEdges]; long start[kMaxEdges + 1]; int nedge; };   // by value: kernel argument

__global__ void __launch_bounds__(256) halo_pack_kernel(const PackArgs k, uint32_t* buf, int unpack) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= k.start[k.nedge]) return;
  int ed = 0;
  while (ed + 1 < k.nedge && t >= k.start[ed + 1]) ed++;               // <= 8 edges
  const EdgeDesc& e = k.e[ed];
  const long r = t - k.start[ed];
  const int q = (int)(r % e.count), p = (int)(r / e.count);
  uint32_t* plane = (uint32_t*)k.planes[p];
  uint32_t* slot = buf + e.buf_off + (long)p * e.count + q;
  if (unpack) plane[e.first + (long)q * e.stride] = *slot;
  else *slot = plane[e.first + (long)q * e.stride];
}

void host_pack(const PackArgs& k, uint32_t* buf, int unpack) {
  for (int ed = 0; ed < k.nedge; ed++)
    for (int p = 0; p < k.n; p++) {
      uint32_t* plane = (uint32_t*)k.planes[p];
      uint32_t* slot = buf + k.e[ed].buf_off + (long)p * k.e[ed].count;
      for (int q = 0; q < k.e[ed].count; q++) {
        if (unpack) plane[k.e[ed].first + (long)q * k.e[ed].stride] = slot[q];
        else slot[q] = plane[k.e[ed].first + (long)q * k.e[ed].stride];
      }
    }
}

int ensure_staging(size_t words, bool device) {
  if (H.h_words < words) {
    if (H.h_send) { if (H.h_pinned) { hipHostFree(H.h_send); hipHostFree(H.h_recv); } else { free(H.h_send); free(H.h_recv); } }
    H.h_send = H.h_recv = nullptr; H.h_words = 0;
    if (device) {
      HIPCHK(hipHostMalloc((void**)&H.h_send, words * 4, hipHostMallocDefault));
      HIPCHK(hipHostMalloc((void**)&H.h_recv, words * 4, hipHostMallocDefault));
      H.h_pinned = true;
    } else {
      H.h_send = (uint32_t*)malloc(words * 4); H.h_recv = (uint32_t*)malloc(words * 4); H.h_pinned = false;
      if (!H.h_send || !H.h_recv) return fail("out of memory");
    }
    H.h_words = words;
  }
  if (device && H.d_words < words) {
    if (H.d_send) { hipFree(H.d_send); hipFree(H.d_recv); }
    H.d_send = H.d_recv = nullptr; H.d_words = 0;
    HIPCHK(hipMalloc((void**)&H.d_send, words * 4));
    HIPCHK(hipMalloc((void**)&H.d_recv, words * 4));
    H.d_words = words;
  }
  return 0;
}

// the one phase: send[d] goes to neighbour slot d, recv[d] comes from it (absent neighbours are skipped)
int exchange_all(int n, void* const* planes, bool device, hipStream_t s, const EdgeDesc* send, const EdgeDesc* recv) {
  PackArgs pk; memset(&pk, 0, sizeof pk);
  PackArgs up; memset(&up, 0, sizeof up);
  for (int p = 0; p < n; p++) pk.planes[p] = up.planes[p] = planes[p];
  pk.n = up.n = n;
  long off[8]; int slot_of[8];
  long words = 0;
  for (int d = 0; d < 8; d++) {
    off[d] = words;
    if (H.nb[d] < 0 || send[d].count <= 0) continue;
    const int e = pk.nedge;
    slot_of[e] = d;
    pk.e[e] = send[d]; pk.e[e].buf_off = words; pk.start[e] = words;
    up.e[e] = recv[d]; up.e[e].buf_off = words; up.start[e] = words;
    words += (long)n * send[d].count;
    pk.nedge = up.nedge = e + 1;
  }
  if (!pk.nedge) return 0;
  pk.start[pk.nedge] = up.start[up.nedge] = words;
  int rc = ensure_staging((size_t)words, device);
  if (rc) return rc;
  const unsigned nblk = (unsigned)((words + 255) / 256);
  if (device) hipLaunchKernelGGL(halo_pack_kernel, dim3(nblk), dim3(256), 0, s, pk, H.d_send, 0);
  else host_pack(pk, H.h_send, 0);
  if (H.transport == NOAHMP_HALO_RCCL) {
    if (!device) return fail("the RCCL transport exchanges device-resident planes");
    int e = H.rccl.GroupStart();
    for (int i = 0; i < pk.nedge && !e; i++) {
      const int peer = H.nb[slot_of[i]];
      const size_t cnt = (size_t)n * pk.e[i].count;
      e = H.rccl.Send(H.d_send + pk.e[i].buf_off, cnt, kNcclInt32, peer, H.comm, s);
      if (!e) e = H.rccl.Recv(H.d_recv + pk.e[i].buf_off, cnt, kNcclInt32, peer, H.comm, s);
    }
    const int e2 = H.rccl.GroupEnd();
    if (e || e2) return fail(std::string("RCCL send/recv: ") + (H.rccl.GetErrorString ? H.rccl.GetErrorString(e ? e : e2) : "error"));
  } else {
    if (device) {
      HIPCHK(hipMemcpyAsync(H.h_send, H.d_send, (size_t)words * 4, hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
    }
    for (int i = 0; i < pk.nedge; i++) {              // links in their global order, on a link the lower rank sends first: no deadlock with blocking sockets
      const int peer = H.nb[slot_of[i]];
      const int fd = H.sock[slot_of[i]];
      const size_t bytes = (size_t)n * pk.e[i].count * 4;
      uint32_t* sb = H.h_send + pk.e[i].buf_off; uint32_t* rb = H.h_recv + pk.e[i].buf_off;
      errno = 0;
      bool ok;
      if (H.rank < peer) ok = send_all(fd, sb, bytes) && recv_all(fd, rb, bytes);
      else ok = recv_all(fd, rb, bytes) && send_all(fd, sb, bytes);
      if (!ok) return fail(std::string((errno == EAGAIN || errno == EWOULDBLOCK) ? "timeout (NMP_HALO_IO_TIMEOUT_S) in the" : "failed") +
                           " socket transfer with rank " + std::to_string(peer));
    }
    if (device) HIPCHK(hipMemcpyAsync(H.d_recv, H.h_recv, (size_t)words * 4, hipMemcpyHostToDevice, s));
  }
  if (device) hipLaunchKernelGGL(halo_pack_kernel, dim3(nblk), dim3(256), 0, s, up, H.d_recv, 1);
  else host_pack(up, H.h_recv, 1);
  return 0;
}

}  // namespace

extern "C" {

// Sockets that only live during noahmp_hip_halo_init: closed on every way out
struct FdSet {
  std::vector<int> fds;
  int add(int fd) { if (fd >= 0) fds.push_back(fd); return fd; }
  void close_one(int fd) { for (int& f : fds) if (f == fd) { close(f); f = -1; } }
  ~FdSet() { for (int f : fds) if (f >= 0) close(f); }
};
// a failed init leaves nothing behind: neighbour sockets closed, state reset (H.up = false)
static int init_failed(const std::string& m, int rc = -109) {
  noahmp_hip_halo_finalize();
  return fail(m, rc);
}

int noahmp_hip_halo_init(int rank, int nranks, const char* master_addr, int master_port, int transport) {
  if (H.up) noahmp_hip_halo_finalize();
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail("bad rank / nranks");
  if (transport != NOAHMP_HALO_TCP && transport != NOAHMP_HALO_RCCL) return fail("unknown transport");
  H.rank = rank; H.nranks = nranks; H.transport = transport;
  nprocs_xy(nranks, H.npx, H.npy);
  const int ipx = rank % H.npx, ipy = rank / H.npx;                               // rank = iprocy*nprocx + iprocx (mpp:93-107)
  {
    static const int dx[8] = {-1, 1, 0, 0, -1, 1, 1, -1}, dy[8] = {0, 0, -1, 1, -1, 1, -1, 1};
    for (int d = 0; d < 8; d++) {
      const int x = ipx + dx[d], y = ipy + dy[d];
      H.nb[d] = (x >= 0 && x < H.npx && y >= 0 && y < H.npy) ? y * H.npx + x : -1;
    }
  }
  H.up = true;
  if (nranks == 1) return 0;
  const double timeout = halo_timeout_s(), deadline = now_s() + timeout;
  FdSet tmp;
  // ---- rendezvous: every rank opens a listener; rank 0 collects (ip, port, status) of all and hands the table out.  A rank that cannot
  // take part (no librccl for the RCCL transport) says so in its hello, and rank 0 tells everybody: all ranks fail together, none waits.
  int my_port = 0;
  const int lfd = tmp.add(listen_on(0, &my_port));
  if (lfd < 0) return init_failed("cannot open a listening socket");
  std::vector<Peer> table(nranks);
  NcclUniqueId uid; memset(&uid, 0, sizeof uid);
  int my_abort = 0;
  std::string my_reason;
  if (transport == NOAHMP_HALO_RCCL) {
    if (load_rccl()) { my_abort = 1; my_reason = g.last_error; }
    else if (rank == 0 && H.rccl.GetUniqueId(&uid)) { my_abort = 1; my_reason = "ncclGetUniqueId failed"; }
  }
  int abort_all = my_abort;
  if (rank == 0) {
    const int mfd = tmp.add(listen_on(master_port, nullptr));
    if (mfd < 0) return init_failed("rank 0 cannot listen on the master port " + std::to_string(master_port));
    table[0] = Peer{htonl(INADDR_LOOPBACK), my_port};
    std::vector<int> fds(nranks, -1);
    for (int i = 1; i < nranks; i++) {
      sockaddr_in a;
      const int fd = tmp.add(accept_deadline(mfd, deadline, &a));
      if (fd < 0) return init_failed("rendezvous: only " + std::to_string(i) + " of " + std::to_string(nranks) + " ranks arrived within " +
                                     std::to_string((int)timeout) + " s");
      int hello[3];
      if (!recv_all(fd, hello, sizeof hello) || hello[0] < 1 || hello[0] >= nranks || fds[hello[0]] >= 0) return init_failed("rendezvous: bad hello");
      fds[hello[0]] = fd;
      table[hello[0]] = Peer{a.sin_addr.s_addr, hello[1]};
      if (hello[2]) abort_all = 1;
    }
    // rank 0 as the others see it: the master address they connected to
    table[0].ip = resolve(master_addr && *master_addr ? master_addr : "127.0.0.1");
    bool ok = true;
    for (int i = 1; i < nranks; i++)
      ok = send_all(fds[i], &abort_all, sizeof abort_all) && send_all(fds[i], table.data(), sizeof(Peer) * nranks) && send_all(fds[i], &uid, sizeof uid) && ok;
    if (!ok) return init_failed("rendezvous: table send");
  } else {
    const int fd = tmp.add(connect_to(resolve(master_addr && *master_addr ? master_addr : "127.0.0.1"), master_port, timeout));
    if (fd < 0) return init_failed("cannot reach rank 0 at " + std::string(master_addr ? master_addr : "127.0.0.1") + ":" + std::to_string(master_port));
    int hello[3] = {rank, my_port, my_abort};
    const bool ok = send_all(fd, hello, sizeof hello) && recv_all(fd, &abort_all, sizeof abort_all) &&
                    recv_all(fd, table.data(), sizeof(Peer) * nranks) && recv_all(fd, &uid, sizeof uid);
    if (!ok) return init_failed("rendezvous: no table from rank 0 (a rank is missing or rank 0 gave up)");
  }
  if (abort_all) return init_failed(my_abort ? my_reason : "another rank cannot use the requested transport (see its error): all ranks stop");
  // ---- neighbour links (TCP transport; also a liveness check for RCCL): connect to lower-ranked neighbours, accept the higher ones
  int expect = 0;
  for (int d = 0; d < 8; d++) if (H.nb[d] > rank) expect++;
  for (int d = 0; d < 8; d++) {
    if (H.nb[d] < 0 || H.nb[d] > rank) continue;
    const int fd = connect_to(table[H.nb[d]].ip, table[H.nb[d]].port, timeout);
    if (fd < 0) return init_failed("cannot connect to neighbour " + std::to_string(H.nb[d]));
    H.sock[d] = fd;                          // from here on noahmp_hip_halo_finalize() owns it
    if (!send_all(fd, &rank, sizeof rank)) return init_failed("cannot greet neighbour " + std::to_string(H.nb[d]));
  }
  for (int i = 0; i < expect; i++) {
    const int fd = accept_deadline(lfd, deadline + timeout, nullptr);
    int who = -1;
    if (fd < 0) return init_failed("a higher-ranked neighbour did not connect within the time limit");
    if (!recv_all(fd, &who, sizeof who)) { close(fd); return init_failed("neighbour accept"); }
    int one = 1; setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    bool placed = false;
    for (int d = 0; d < 8; d++) if (H.nb[d] == who && H.sock[d] < 0) { H.sock[d] = fd; placed = true; break; }
    if (!placed) { close(fd); return init_failed("unexpected neighbour " + std::to_string(who)); }
  }
  for (int d = 0; d < 8; d++) if (H.sock[d] >= 0) set_io_timeout(H.sock[d], halo_io_timeout_s());   // from here on: the data path's deadline
  if (transport == NOAHMP_HALO_RCCL) {
    int rc = nmp_host::ensure_init();
    if (rc) { const std::string m = g.last_error; noahmp_hip_halo_finalize(); g.last_error = m; return rc; }
    const int e = H.rccl.CommInitRank(&H.comm, nranks, uid, rank);
    if (e) return init_failed(std::string("ncclCommInitRank: ") + (H.rccl.GetErrorString ? H.rccl.GetErrorString(e) : "error"));
  }
  return 0;
}

int noahmp_hip_exchange_halo(int n, void* const* planes, const int32_t* index8, int mem, void* stream) {
  if (!H.up) return fail("noahmp_hip_halo_init has not been called");
  if (n <= 0 || H.nranks == 1) return 0;
  if (n > kMaxPlanes) return fail("at most 64 planes per call");
  const int ims = index8[0], ime = index8[1], jms = index8[2], jme = index8[3], its = index8[4], ite = index8[5], jts = index8[6], jte = index8[7];
  const long ni = ime - ims + 1;
  const int i0 = its - ims, i1 = ite - ims, j0 = jts - jms, j1 = jte - jms;
  // a side that has a neighbour must have its ring cell inside the memory block
  if ((H.nb[0] >= 0 && i0 < 1) || (H.nb[1] >= 0 && ite + 1 > ime) || (H.nb[2] >= 0 && j0 < 1) || (H.nb[3] >= 0 && jte + 1 > jme))
    return fail("the memory block (ims:ime, jms:jme) does not hold the 1-cell ring towards every neighbour", -103);
  const bool device = mem == NOAHMP_MEM_DEVICE;
  hipStream_t s = nullptr;
  if (device) {
    int rc = nmp_host::ensure_init();
    if (rc) return rc;
    s = stream ? (hipStream_t)stream : g.own_stream;
  }
  const int nrow = j1 - j0 + 1, ncol = i1 - i0 + 1;
  // what goes out: the tile cells that touch a side; what comes in: the ring cells on that side (same slot order as H.nb)
  EdgeDesc snd[8], rcv[8];
  snd[0] = EdgeDesc{(long)j0 * ni + i0, ni, nrow, 0};        rcv[0] = EdgeDesc{(long)j0 * ni + i0 - 1, ni, nrow, 0};          // left
  snd[1] = EdgeDesc{(long)j0 * ni + i1, ni, nrow, 0};        rcv[1] = EdgeDesc{(long)j0 * ni + i1 + 1, ni, nrow, 0};          // right
  snd[2] = EdgeDesc{(long)j0 * ni + i0, 1, ncol, 0};         rcv[2] = EdgeDesc{(long)(j0 - 1) * ni + i0, 1, ncol, 0};         // down
  snd[3] = EdgeDesc{(long)j1 * ni + i0, 1, ncol, 0};         rcv[3] = EdgeDesc{(long)(j1 + 1) * ni + i0, 1, ncol, 0};         // up
  snd[4] = EdgeDesc{(long)j0 * ni + i0, 1, 1, 0};            rcv[4] = EdgeDesc{(long)(j0 - 1) * ni + i0 - 1, 1, 1, 0};        // down-left
  snd[5] = EdgeDesc{(long)j1 * ni + i1, 1, 1, 0};            rcv[5] = EdgeDesc{(long)(j1 + 1) * ni + i1 + 1, 1, 1, 0};        // up-right
  snd[6] = EdgeDesc{(long)j0 * ni + i1, 1, 1, 0};            rcv[6] = EdgeDesc{(long)(j0 - 1) * ni + i1 + 1, 1, 1, 0};        // down-right
  snd[7] = EdgeDesc{(long)j1 * ni + i0, 1, 1, 0};            rcv[7] = EdgeDesc{(long)(j1 + 1) * ni + i0 - 1, 1, 1, 0};        // up-left
  int rc = exchange_all(n, planes, device, s, snd, rcv);
  if (rc) return rc;
  if (device) HIPCHK(hipGetLastError());
  return 0;
}

// Self-test of the RCCL plumbing on ONE GPU (the exchange itself needs one GPU per rank): load librccl, create a communicator of
// size 1, send `words` 4-byte words to self through ncclSend / ncclRecv in one group on the engine's stream, compare.  0 = ok.
int noahmp_hip_halo_selftest_rccl(int words) {
  int rc = nmp_host::ensure_init();
  if (rc) return rc;
  if (load_rccl()) return -109;
  NcclUniqueId uid; memset(&uid, 0, sizeof uid);
  int e = H.rccl.GetUniqueId(&uid);
  if (e) return fail("ncclGetUniqueId");
  NcclComm comm = nullptr;
  e = H.rccl.CommInitRank(&comm, 1, uid, 0);
  if (e) return fail(std::string("ncclCommInitRank: ") + (H.rccl.GetErrorString ? H.rccl.GetErrorString(e) : "error"));
  std::vector<uint32_t> h(words), back(words, 0);
  for (int i = 0; i < words; i++) h[i] = 0x9E3779B9u * (uint32_t)(i + 1);
  uint32_t *a = nullptr, *b = nullptr;
  HIPCHK(hipMalloc((void**)&a, words * 4));
  HIPCHK(hipMalloc((void**)&b, words * 4));
  HIPCHK(hipMemcpy(a, h.data(), words * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(b, 0, words * 4));
  e = H.rccl.GroupStart();
  if (!e) e = H.rccl.Send(a, (size_t)words, kNcclInt32, 0, comm, g.own_stream);
  if (!e) e = H.rccl.Recv(b, (size_t)words, kNcclInt32, 0, comm, g.own_stream);
  const int e2 = H.rccl.GroupEnd();
  if (e || e2) { H.rccl.CommDestroy(comm); return fail(std::string("RCCL self send/recv: ") + (H.rccl.GetErrorString ? H.rccl.GetErrorString(e ? e : e2) : "error")); }
  HIPCHK(hipStreamSynchronize(g.own_stream));
  HIPCHK(hipMemcpy(back.data(), b, words * 4, hipMemcpyDeviceToHost));
  hipFree(a); hipFree(b);
  H.rccl.CommDestroy(comm);
  for (int i = 0; i < words; i++) if (back[i] != h[i]) return fail("RCCL self send/recv returned other data");
  return 0;
}

int noahmp_hip_halo_finalize(void) {
  for (int d = 0; d < 8; d++) { if (H.sock[d] >= 0) close(H.sock[d]); H.sock[d] = -1; }
  if (H.comm && H.rccl.CommDestroy) H.rccl.CommDestroy(H.comm);
  H.comm = nullptr;
  if (H.d_send) { hipFree(H.d_send); hipFree(H.d_recv); }
  if (H.h_send) { if (H.h_pinned) { hipHostFree(H.h_send); hipHostFree(H.h_recv); } else { free(H.h_send); free(H.h_recv); } }
  H = Halo();
  return 0;
}

}  // extern "C"

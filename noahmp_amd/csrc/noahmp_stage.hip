// Engine-owned staging of pageable host memory (nmp_stage.hpp).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "nmp_engine_host.hpp"
#include "nmp_stage.hpp"
#include "nmp_copy_pool.hpp"

namespace nmp_host {
namespace {

constexpr size_t kPiece = 32u << 20;      // one bounce buffer: 32 MiB (0.6 ms of DMA at 55 GB/s -- long enough to hide the refill, short enough to start early)
constexpr int kBuffers = 4;

struct Stage {
  void* buf[kBuffers] = {};
  hipEvent_t ev[kBuffers] = {};
  bool busy[kBuffers] = {};
  bool ready = false;
  CopyPool pool;
  unsigned long long staged = 0, direct = 0;
};
// Never destroyed: a process that exits without noahmp_hip_finalize() still has the copy threads parked on the pool's condition variable, and
// destroying a condition variable with waiters blocks for ever (glibc's pthread_cond_destroy waits for them) -- the process would hang in its
// static destructors.  noahmp_hip_finalize() stops the threads and frees the buffers; process exit just ends them.
Stage& S = *new Stage;

int stage_init() {
  if (S.ready) return 0;
  for (int b = 0; b < kBuffers; b++) {
    HIPCHK(hipHostMalloc(&S.buf[b], kPiece, hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&S.ev[b], hipEventDisableTiming));
    S.busy[b] = false;
  }
  S.ready = true;
  return 0;
}

// page-locked already?  The engine's own registrations ("pin_host_arrays") first, then the runtime's view of the pointer
bool page_locked(const void* p, size_t bytes) {
  const char* lo = (const char*)p;
  for (auto& kv : g.host_regs) {
    const char* olo = (const char*)kv.first;
    if (kv.second.state == 1 && lo >= olo && lo + bytes <= olo + kv.second.bytes) return true;
  }
  hipPointerAttribute_t at;
  memset(&at, 0, sizeof at);
  const hipError_t e = hipPointerGetAttributes(&at, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }       // (older runtimes: an error for memory they do not know)
  return at.type == hipMemoryTypeHost;
}

// one bounce buffer's worth of work: pieces of segments, back to back in the buffer
struct Part { int seg; size_t off, len, at; };

}  // namespace

bool host_page_locked(const void* p, size_t bytes) { return page_locked(p, bytes); }

int copy_segments(const CopySeg* segs, int n, bool to_device, hipStream_t s) {
  // direct copies first (asynchronous, on `s`), the rest through the bounce buffers
  std::vector<int> staged;
  for (int i = 0; i < n; i++) {
    if (!segs[i].bytes) continue;
    if (page_locked(segs[i].host, segs[i].bytes)) {
      if (to_device) HIPCHK(hipMemcpyAsync(segs[i].dev, segs[i].host, segs[i].bytes, hipMemcpyHostToDevice, s));
      else HIPCHK(hipMemcpyAsync(segs[i].host, segs[i].dev, segs[i].bytes, hipMemcpyDeviceToHost, s));
      S.direct += segs[i].bytes;
    } else {
      staged.push_back(i);
      S.staged += segs[i].bytes;
    }
  }
  if (staged.empty()) {
    if (!to_device) HIPCHK(hipStreamSynchronize(s));
    return 0;
  }
  int rc = stage_init();
  if (rc) return rc;
  // the fills of the buffers, in order: small segments share a buffer, large ones span several
  std::vector<std::vector<Part>> fills(1);
  size_t used = 0;
  for (int i : staged) {
    size_t off = 0;
    while (off < segs[i].bytes) {
      if (used == kPiece) { fills.emplace_back(); used = 0; }
      const size_t len = segs[i].bytes - off < kPiece - used ? segs[i].bytes - off : kPiece - used;
      fills.back().push_back(Part{i, off, len, used});
      used += (len + 255) & ~(size_t)255;          // keep the DMA sources 256-byte aligned
      if (used > kPiece) used = kPiece;
      off += len;
    }
  }
  const int nf = (int)fills.size();
  if (to_device) {
    for (int f = 0; f < nf; f++) {
      const int b = f % kBuffers;
      if (S.busy[b]) HIPCHK(hipEventSynchronize(S.ev[b]));        // its previous DMA has read it
      for (const Part& p : fills[f]) S.pool.copy((char*)S.buf[b] + p.at, (const char*)segs[p.seg].host + p.off, p.len);
      for (const Part& p : fills[f])
        HIPCHK(hipMemcpyAsync((char*)segs[p.seg].dev + p.off, (char*)S.buf[b] + p.at, p.len, hipMemcpyHostToDevice, s));
      HIPCHK(hipEventRecord(S.ev[b], s));
      S.busy[b] = true;
    }
    return 0;                // the DMAs out of the bounce buffers may still run; the caller's arrays have been read
  }
  // device -> host: DMA fill f + kBuffers - 1 while fill f is drained by the copy threads
  auto drain = [&](int f) -> int {
    const int b = f % kBuffers;
    HIPCHK(hipEventSynchronize(S.ev[b]));
    for (const Part& p : fills[f]) S.pool.copy((char*)segs[p.seg].host + p.off, (const char*)S.buf[b] + p.at, p.len);
    S.busy[b] = false;
    return 0;
  };
  for (int f = 0; f < nf; f++) {
    const int b = f % kBuffers;
    if (f >= kBuffers) { if ((rc = drain(f - kBuffers))) return rc; }
    else if (S.busy[b]) { HIPCHK(hipEventSynchronize(S.ev[b])); S.busy[b] = false; }     // an earlier upload still reading this buffer
    for (const Part& p : fills[f])
      HIPCHK(hipMemcpyAsync((char*)S.buf[b] + p.at, (const char*)segs[p.seg].dev + p.off, p.len, hipMemcpyDeviceToHost, s));
    HIPCHK(hipEventRecord(S.ev[b], s));
    S.busy[b] = true;
  }
  for (int f = nf > kBuffers ? nf - kBuffers : 0; f < nf; f++)
    if ((rc = drain(f))) return rc;
  HIPCHK(hipStreamSynchronize(s));             // the direct segments
  return 0;
}

void stage_finalize() {
  S.pool.stop();
  if (S.ready) {
    for (int b = 0; b < kBuffers; b++) {
      if (S.busy[b]) hipEventSynchronize(S.ev[b]);
      hipHostFree(S.buf[b]);
      hipEventDestroy(S.ev[b]);
      S.buf[b] = nullptr; S.busy[b] = false;
    }
    S.ready = false;
  }
}

void stage_stats(unsigned long long* staged_bytes, unsigned long long* direct_bytes) {
  if (staged_bytes) *staged_bytes = S.staged;
  if (direct_bytes) *direct_bytes = S.direct;
}

}  // namespace nmp_host

extern "C" {
// bytes of caller memory that travelled through the engine's bounce buffers / were copied directly (page-locked memory) since the library was loaded
void noahmp_hip_debug_copy_stats(unsigned long long* staged_bytes, unsigned long long* direct_bytes) { nmp_host::stage_stats(staged_bytes, direct_bytes); }
}

// Copies between the caller's PAGEABLE host arrays and the device that never hand a pageable pointer to the HIP runtime.
//
// Why (profiles/r05_experiments.md section 3, profiles/r06_experiments.md): for a pageable buffer of ~2 MiB and more the runtime page-locks
// the caller's pages IN PLACE and caches the mapping; on this stack a cached mapping of heap memory that has since been freed and reused
// faults in a later copy ("Memory access fault by GPU ... Write access to a read-only page": 7 of 10 runs of the GPU test suite).  Round 5
// kept the runtime off that path with an environment variable set from a library constructor (a process-wide side effect that only works
// when this library is loaded before HIP initialises).  Round 6: the engine owns the staging.  Every host <-> device copy of caller
// memory goes through copy_segments(): memory the engine page-locked itself ("pin_host_arrays") or that the runtime reports as
// page-locked is copied directly; everything else travels through engine-owned hipHostMalloc bounce buffers (kBuffers x kPiece), filled /
// drained by a small pool of copy threads while the DMA of the previous piece runs.  The caller-owned arrays of drv:11-44 stay the caller's
// (SURVEY 8b "Ownership"): the engine never registers, maps or keeps anything of them on this path.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace nmp_host {

struct CopySeg { void* host; void* dev; size_t bytes; };

// Copy n segments host -> device (to_device) or device -> host, ordered on `s` behind the work already enqueued there.
// to_device: returns when every host segment has been READ (the caller may overwrite its arrays; the last DMAs may still be running on `s`).
// !to_device: returns when every host segment holds its data.
// 0 or -100 (g.last_error set).
int copy_segments(const CopySeg* segs, int n, bool to_device, hipStream_t s);

// one segment
inline int copy_h2d(void* dev, const void* host, size_t bytes, hipStream_t s) {
  CopySeg g{const_cast<void*>(host), dev, bytes};
  return copy_segments(&g, 1, true, s);
}
inline int copy_d2h(void* host, const void* dev, size_t bytes, hipStream_t s) {
  CopySeg g{host, const_cast<void*>(dev), bytes};
  return copy_segments(&g, 1, false, s);
}

// page-locked already (by the engine's own "pin_host_arrays" registration or as the runtime reports it)?
bool host_page_locked(const void* p, size_t bytes);

void stage_finalize();            // bounce buffers, events, copy threads
// counters for tests / INTEGRATION: bytes that went through the bounce buffers and bytes copied directly since the library was loaded
void stage_stats(unsigned long long* staged_bytes, unsigned long long* direct_bytes);

}  // namespace nmp_host

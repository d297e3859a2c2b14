// Column order of a device-resident tile: sort key, stable radix sort, staleness count and the
// per-step forcing-scatter plan, all on the device.  BASELINE.json north_star: "option-branch divergence handled by
// sorting columns by (vegetation type, snow-layer count) before launch".  The reference has no counterpart (its ILOOP
// visits columns in (i,j) order, drv:397-424); columns are independent for every option except the MMF lateral flow,
// so which lane computes which column is the engine's choice.  The snow-layer count changes while a run accumulates
// or melts snow (lsm:7044 COMBINE, 7110 DIVIDE, 7177 COMBO, 7294-7343 SNOWH2O), so a run re-sorts when
// noahmp_hip_sort_staleness() says enough columns left their bucket.
#include <string.h>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "noahmp_hip.h"
#include "nmp_engine_host.hpp"

using nmp_host::g;

namespace {

struct KeyArgs {
  const float* xland; const float* xice; const float* tsk;   // tsk: the temperature plane of the key (TSK, or level 1 of T3D)
  int t_nk, t_k, ni;                                          // its levels per row, the level taken, the row length
  const int* ivgtyp; const int* isnow;
  const int* band;                                            // optional caller-defined sub-key plane (noahmp_hip_sort_set_band), or NULL
  const unsigned char* cost;                                  // the engine's cost record of the last step (NOAHMP_SORT_COST), or NULL
  float xice_thres, inv_bin;
  int isice, flags;
  long n;
  unsigned char veg_rank[64];                                 // vegetation category -> its place in the key (noahmp_hip_sort_set_veg_order)
};

// key = class(2) | vegetation type and snow-layer count (8, in the order the flags ask for) | band(5) | cost(4) | tsk bin(8): class 0 land,
// 1 land ice, 2 skipped (the classification of drv:426-441); skipped columns carry no sub-key.  band: a caller-defined static sub-key, 0..31
// (noahmp_hip_sort_set_band; e.g. the 15-degree longitude band of a lat/lon grid, so that a wavefront's columns share their local solar time).
// cost (NOAHMP_SORT_COST): a bucket of the column's own trip counts in the step before the sort -- iterations of VEGE_FLUX's canopy loop and
// STOMATA's bisection steps, the two loops whose length differs from column to column (a wavefront runs as long as its slowest lane).
constexpr int kClsShift = 25, kHiShift = 17, kBandShift = 12, kCostShift = 8;
// one canopy iteration ~ 1070 vector instructions, one bisection step ~ 130 (tools/isa_loops.py): cost units of one bisection step
__device__ __forceinline__ unsigned cost_bucket(unsigned iters, unsigned bisections) {
  const unsigned c = 8u * iters + bisections;                 // 0 (no canopy) .. 200
  return c == 0u ? 0u : min(1u + c / 14u, 15u);
}
__device__ __forceinline__ unsigned column_key(const KeyArgs& k, long p) {
  const float xland = k.xland[p], xice = k.xice[p];
  const int ivg = k.ivgtyp[p];
  const unsigned cls = ((xland - 1.5f) >= 0.f || xice >= k.xice_thres) ? 2u : (ivg == k.isice ? 1u : 0u);
  if (cls == 2u) return 2u << kClsShift;
  unsigned veg = 0, sn = 0, tb = 0, band = 0, cost = 0;
  if (cls == 0u && (k.flags & NOAHMP_SORT_VEG)) veg = k.veg_rank[min(max(ivg, 0), 63)];
  if (k.flags & NOAHMP_SORT_SNOW) sn = (unsigned)min(max(-k.isnow[p], 0), 3);
  if (k.band) band = (unsigned)min(max(k.band[p], 0), 31);
  if (k.cost && cls == 0u) cost = cost_bucket(k.cost[2 * p], k.cost[2 * p + 1]);
  if (k.inv_bin > 0.f) {
    float t = k.t_nk == 1 ? k.tsk[p] : k.tsk[((size_t)(p / k.ni) * k.t_nk + k.t_k) * k.ni + p % k.ni];
    if (!(t == t)) t = 250.f;
    tb = (unsigned)min(max((int)((t - 230.0f) * k.inv_bin), 0), 255);
  }
  const unsigned hi = (k.flags & NOAHMP_SORT_SNOW_FIRST) ? (sn << 6 | veg) : (veg << 2 | sn);
  return cls << kClsShift | hi << kHiShift | band << kBandShift | cost << kCostShift | tb;
}
constexpr unsigned kTskMask = 0xFFFu;       // what a staleness count ignores: temperature bin and cost bucket (they change every step)
constexpr int kKeyBits = 27;
// host-side: the plane the NEXT sort / staleness call of this thread reads (device pointer, the store's column order); consumed by that
// call (fill_key_args), so that a failed or forgotten call leaves nothing behind
thread_local const int* g_band_plane = nullptr;
// vegetation category -> place in the key; identity until noahmp_hip_sort_set_veg_order names another order.  Process-wide like the engine:
// a staleness count must be taken with the order the layout was sorted by.
struct VegRank { unsigned char r[64]; VegRank() { for (int v = 0; v < 64; v++) r[v] = (unsigned char)v; } } g_veg_rank;

__global__ void __launch_bounds__(256) sort_key_kernel(const KeyArgs k, unsigned* keys, int* idx) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= k.n) return;
  keys[p] = column_key(k, p);
  idx[p] = (int)p;
}

// number of columns whose key (without the temperature bin) differs from the key they were sorted by
__global__ void __launch_bounds__(256) sort_stale_kernel(const KeyArgs k, const unsigned* sorted_keys, unsigned long long* slots) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  bool changed = false;
  if (p < k.n) changed = ((column_key(k, p) ^ sorted_keys[p]) & ~kTskMask) != 0;
  const unsigned long long m = __ballot(changed);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&slots[blockIdx.x & 255], (unsigned long long)__popcll(m));
}

// class boundaries of the sorted keys: out[0] = land columns, out[1] = land-ice columns
__global__ void sort_bounds_kernel(const unsigned* keys, long n, long* out) {
  if (threadIdx.x > 1) return;
  const unsigned want = (threadIdx.x + 1u) << kClsShift;          // first key >= want
  long lo = 0, hi = n;
  while (lo < hi) { const long mid = (lo + hi) >> 1; if (keys[mid] < want) lo = mid + 1; else hi = mid; }
  out[threadIdx.x] = lo;
}

// the 256 counters summed into ONE word of mapped host memory: a D2H copy command would make the host wait for the stream to reach it
__global__ void __launch_bounds__(256) stale_sum_kernel(unsigned long long* slots, volatile long* host_word) {
  __shared__ unsigned long long part[4];
  unsigned long long v = slots[threadIdx.x];
  slots[threadIdx.x] = 0;                      // ready for the next count
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) { *host_word = (long)(part[0] + part[1] + part[2] + part[3]); __threadfence_system(); }
}

__global__ void __launch_bounds__(256) invert_perm_kernel(const int* perm, int* inv, long n) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) inv[perm[p]] = (int)p;
}

// Scatter plan (noahmp_hip_scatter_fields): per chunk of 1024 consecutive tile columns, the columns ordered by their
// destination in the sorted layout.  One workgroup sorts one chunk in LDS.
constexpr int kChunk = 1024;
__global__ void __launch_bounds__(256) scatter_plan_kernel(const int* inv, long n, unsigned short* order, int* dpos) {
  using Sort = rocprim::block_radix_sort<int, 256, kChunk / 256, unsigned short>;
  __shared__ typename Sort::storage_type st;
  const long base = (long)blockIdx.x * kChunk;
  int key[kChunk / 256];
  unsigned short val[kChunk / 256];
#pragma unroll
  for (int r = 0; r < kChunk / 256; r++) {
    const int off = threadIdx.x * (kChunk / 256) + r;
    const long q = base + off;
    key[r] = q < n ? inv[q] : 0x7FFFFFFF;
    val[r] = (unsigned short)off;
  }
  Sort().sort(key, val, st);
#pragma unroll
  for (int r = 0; r < kChunk / 256; r++) {
    const long q = base + threadIdx.x * (kChunk / 256) + r;
    if (q < n) { order[q] = val[r]; dpos[q] = key[r] == 0x7FFFFFFF ? -1 : key[r]; }
  }
}

// the same plan for any chunk size: keys = destinations, values = offsets inside the chunk, one stable segmented radix sort
__global__ void __launch_bounds__(256) plan_slots_kernel(unsigned short* slot, long n, int chunk) {
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) slot[q] = (unsigned short)(q % chunk);
}
struct SegOffset {
  unsigned chunk, n, shift;
  __host__ __device__ unsigned operator()(unsigned i) const { const unsigned long long o = ((unsigned long long)i + shift) * chunk; return o < n ? (unsigned)o : n; }
};

int fill_key_args(KeyArgs& k, const noahmp_step_args* a, int flags, int tsk_bin_mk) {
  if (a->ims != a->its || a->ime != a->ite || a->jms != a->jts || a->jme != a->jte) {
    g.last_error = "column sort: the memory block must be the tile (a tile that carries a halo keeps its (i,j) order)";
    g_band_plane = nullptr;
    return -105;
  }
  k.xland = a->xland; k.xice = a->xice; k.tsk = a->tsk; k.ivgtyp = a->ivgtyp; k.isnow = a->isnowxy;
  k.t_nk = 1; k.t_k = 0; k.ni = a->ime - a->ims + 1;
  if (flags & NOAHMP_SORT_TAIR) { k.tsk = a->t3d; k.t_nk = a->kme - a->kms + 1; k.t_k = 1 - a->kms; }   // the forcing air temperature (level 1)
  k.xice_thres = a->xice_thres; k.isice = a->isice; k.flags = flags;
  k.band = g_band_plane;
  g_band_plane = nullptr;                 // one-shot (noahmp_hip_sort_set_band)
  memcpy(k.veg_rank, g_veg_rank.r, sizeof k.veg_rank);
  k.n = (long)(a->ime - a->ims + 1) * (a->jme - a->jms + 1);
  // the cost record is in the order the last step found the tile in: usable only if no permutation happened since
  k.cost = ((flags & NOAHMP_SORT_COST) && g.cost_fresh && g.d_cost && g.cost_cols == k.n) ? g.d_cost : nullptr;
  k.inv_bin = tsk_bin_mk > 0 ? 1000.0f / (float)tsk_bin_mk : 0.f;
  k.n = (long)(a->ime - a->ims + 1) * (a->jme - a->jms + 1);
  return 0;
}

struct SortScratch {
  unsigned* keys_in = nullptr; int* idx_in = nullptr; unsigned* keys_out = nullptr; void* tmp = nullptr; int* inv = nullptr;
  size_t keys_in_b = 0, idx_in_b = 0, keys_out_b = 0, tmp_b = 0, inv_b = 0;
  unsigned short* slot_in = nullptr; size_t slot_in_b = 0;     // scatter plan: chunk offsets before the segmented sort
  void* seg_tmp = nullptr; size_t seg_tmp_b = 0;
  unsigned long long* slots = nullptr;      // 256 staleness counters
  long* h_bounds = nullptr;                 // pinned: 2 class boundaries + 256 staleness slots
  unsigned long long* slots_async = nullptr; long* h_async = nullptr; long* d_async_word = nullptr; hipEvent_t ev_async = nullptr; bool async_pending = false;   // noahmp_hip_sort_staleness_async
  long* d_bounds = nullptr;
} sc;

}  // namespace

namespace nmp_host {
void sort_finalize() {
  hipFree(sc.slot_in); hipFree(sc.seg_tmp);
  hipFree(sc.keys_in); hipFree(sc.idx_in); hipFree(sc.keys_out); hipFree(sc.tmp); hipFree(sc.inv); hipFree(sc.slots); hipFree(sc.d_bounds);
  if (sc.h_bounds) hipHostFree(sc.h_bounds);
  hipFree(sc.slots_async); if (sc.h_async) hipHostFree(sc.h_async); if (sc.ev_async) hipEventDestroy(sc.ev_async);
  sc.async_pending = false;
  sc = SortScratch();
  g_veg_rank = VegRank();
}
}  // namespace nmp_host

extern "C" {

int noahmp_hip_sort_set_band(const int32_t* band_plane) {
  g_band_plane = band_plane;
  return 0;
}

// Order of the vegetation categories inside the land range: rank[v] (0..63) = place of category v in the key's vegetation field, for
// v = 0 .. n-1 (categories >= n keep their own number); NULL or n <= 0: the categories' own numbers (the default).  Workgroups start in
// key order, so the categories named first run first: a caller that names the EXPENSIVE categories first (forests before barren ground)
// leaves the cheap waves for the tail of the launch, where wave slots idle while the last workgroups finish -- the longest-first rule of
// list scheduling; what a tile of an 8-rank run (6.75 rounds of the chip's 2 048 wave slots) loses there is measured in
// profiles/r06_experiments.md.  Pure function of the category, so staleness logic and results are untouched.  Stays in force for every
// later sort / staleness call of the process (a layout must be checked with the order it was sorted by).
int noahmp_hip_sort_set_veg_order(const int32_t* rank, int n) {
  for (int v = 0; v < 64; v++) g_veg_rank.r[v] = (unsigned char)v;
  if (!rank || n <= 0) return 0;
  for (int v = 0; v < n && v < 64; v++) {
    if (rank[v] < 0 || rank[v] > 63) {
      for (int w = 0; w < 64; w++) g_veg_rank.r[w] = (unsigned char)w;
      g.last_error = "noahmp_hip_sort_set_veg_order: ranks must be 0..63";
      return -105;
    }
    g_veg_rank.r[v] = (unsigned char)rank[v];
  }
  return 0;
}

// buffers of noahmp_hip_sort_staleness_async: allocated when a layout is sorted (allocations synchronise the device -- not inside a run)
static int ensure_async_count() {
  if (sc.slots_async && sc.h_async && sc.d_async_word && sc.ev_async) return 0;
  auto fail = [](const char* what, hipError_t e) {          // all or nothing: a later call must not find half of the set
    char b[200]; snprintf(b, sizeof b, "noahmp_hip_sort: %s failed: %s", what, hipGetErrorString(e));
    g.last_error = b;
    if (sc.slots_async) hipFree(sc.slots_async);
    if (sc.h_async) hipHostFree(sc.h_async);
    if (sc.ev_async) hipEventDestroy(sc.ev_async);
    sc.slots_async = nullptr; sc.h_async = nullptr; sc.d_async_word = nullptr; sc.ev_async = nullptr;
    return -100;
  };
  hipError_t e;
  if ((e = hipMalloc(&sc.slots_async, 256 * sizeof(unsigned long long))) != hipSuccess) { sc.slots_async = nullptr; return fail("hipMalloc", e); }
  if ((e = hipMemset(sc.slots_async, 0, 256 * sizeof(unsigned long long))) != hipSuccess) return fail("hipMemset", e);
  if ((e = hipHostMalloc((void**)&sc.h_async, sizeof(long), hipHostMallocMapped)) != hipSuccess) { sc.h_async = nullptr; return fail("hipHostMalloc", e); }
  if ((e = hipHostGetDevicePointer((void**)&sc.d_async_word, sc.h_async, 0)) != hipSuccess) return fail("hipHostGetDevicePointer", e);
  if ((e = hipEventCreateWithFlags(&sc.ev_async, hipEventDisableTiming)) != hipSuccess) { sc.ev_async = nullptr; return fail("hipEventCreate", e); }
  return 0;
}

int noahmp_hip_sort_columns(const noahmp_step_args* a, int flags, int tsk_bin_mk, int32_t* perm_out, uint32_t* keys_out,
                            int64_t* class_counts, void* stream) {
  int rc = nmp_host::ensure_init();
  if (rc) return rc;
  KeyArgs k;
  rc = fill_key_args(k, a, flags, tsk_bin_mk);
  if (rc) return rc;
  if (!perm_out) { g.last_error = "noahmp_hip_sort_columns: perm_out is required"; return -105; }
  hipStream_t s = stream ? (hipStream_t)stream : g.own_stream;
  const long n = k.n;
  if (n <= 0) { if (class_counts) class_counts[0] = class_counts[1] = class_counts[2] = 0; return 0; }
  if (n > 0x7FFFFFFFL) { g.last_error = "noahmp_hip_sort_columns: more than 2^31 columns"; return -105; }
  if ((rc = nmp_host::ensure_bytes((void**)&sc.keys_in, &sc.keys_in_b, n * 4))) return rc;
  if ((rc = nmp_host::ensure_bytes((void**)&sc.idx_in, &sc.idx_in_b, n * 4))) return rc;
  unsigned* kout = keys_out;
  if (!kout) { if ((rc = nmp_host::ensure_bytes((void**)&sc.keys_out, &sc.keys_out_b, n * 4))) return rc; kout = sc.keys_out; }
  if (!sc.d_bounds) { HIPCHK(hipMalloc(&sc.d_bounds, 2 * sizeof(long))); HIPCHK(hipHostMalloc((void**)&sc.h_bounds, 258 * sizeof(long), hipHostMallocDefault)); }
  if ((rc = ensure_async_count())) return rc;
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(sort_key_kernel, dim3(nb), dim3(256), 0, s, k, sc.keys_in, sc.idx_in);
  size_t need = 0;
  HIPCHK(rocprim::radix_sort_pairs(nullptr, need, sc.keys_in, kout, sc.idx_in, perm_out, (size_t)n, 0, kKeyBits, s));
  if ((rc = nmp_host::ensure_bytes(&sc.tmp, &sc.tmp_b, need))) return rc;
  HIPCHK(rocprim::radix_sort_pairs(sc.tmp, need, sc.keys_in, kout, sc.idx_in, perm_out, (size_t)n, 0, kKeyBits, s));   // LSD radix: stable
  if (class_counts) {
    hipLaunchKernelGGL(sort_bounds_kernel, dim3(1), dim3(64), 0, s, kout, n, sc.d_bounds);
    HIPCHK(hipMemcpyAsync(sc.h_bounds, sc.d_bounds, 2 * sizeof(long), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    class_counts[0] = sc.h_bounds[0];
    class_counts[1] = sc.h_bounds[1] - sc.h_bounds[0];
    class_counts[2] = n - sc.h_bounds[1];
  }
  HIPCHK(hipGetLastError());
  return 0;
}

int noahmp_hip_sort_staleness(const noahmp_step_args* a, int flags, const uint32_t* sorted_keys, int64_t* changed, void* stream) {
  int rc = nmp_host::ensure_init();
  if (rc) return rc;
  KeyArgs k;
  rc = fill_key_args(k, a, flags, 0);
  if (rc) return rc;
  if (!sorted_keys || !changed) { g.last_error = "noahmp_hip_sort_staleness: sorted_keys and changed are required"; return -105; }
  hipStream_t s = stream ? (hipStream_t)stream : g.own_stream;
  *changed = 0;
  if (k.n <= 0) return 0;
  if (!sc.slots) HIPCHK(hipMalloc(&sc.slots, 256 * sizeof(unsigned long long)));
  if (!sc.d_bounds) { HIPCHK(hipMalloc(&sc.d_bounds, 2 * sizeof(long))); HIPCHK(hipHostMalloc((void**)&sc.h_bounds, 258 * sizeof(long), hipHostMallocDefault)); }
  HIPCHK(hipMemsetAsync(sc.slots, 0, 256 * sizeof(unsigned long long), s));
  hipLaunchKernelGGL(sort_stale_kernel, dim3((unsigned)((k.n + 255) / 256)), dim3(256), 0, s, k, sorted_keys, sc.slots);
  HIPCHK(hipMemcpyAsync(sc.h_bounds + 2, sc.slots, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  long tot = 0;
  for (int i = 0; i < 256; i++) tot += sc.h_bounds[2 + i];
  *changed = tot;
  return 0;
}

// The same count without a wait: a check every K steps that drains the stream costs a run ~0.4 ms of idle GPU each time (the host has to
// refill the queue); enqueued here, the count is read K steps later, when it has long arrived.
int noahmp_hip_sort_staleness_async(const noahmp_step_args* a, int flags, const uint32_t* sorted_keys, void* stream) {
  int rc = nmp_host::ensure_init();
  if (rc) return rc;
  KeyArgs k;
  rc = fill_key_args(k, a, flags, 0);
  if (rc) return rc;
  if (!sorted_keys) { g.last_error = "noahmp_hip_sort_staleness_async: sorted_keys is required"; return -105; }
  hipStream_t s = stream ? (hipStream_t)stream : g.own_stream;
  if ((rc = ensure_async_count())) return rc;
  if (sc.async_pending) HIPCHK(hipEventSynchronize(sc.ev_async));      // an unread earlier count is overwritten, never raced
  if (k.n > 0) hipLaunchKernelGGL(sort_stale_kernel, dim3((unsigned)((k.n + 255) / 256)), dim3(256), 0, s, k, sorted_keys, sc.slots_async);
  hipLaunchKernelGGL(stale_sum_kernel, dim3(1), dim3(256), 0, s, sc.slots_async, (volatile long*)sc.d_async_word);
  HIPCHK(hipEventRecord(sc.ev_async, s));
  sc.async_pending = true;
  return 0;
}

// 0: *changed = the count of the last noahmp_hip_sort_staleness_async; 1: not there yet (wait == 0); -105: nothing was asked for.
int noahmp_hip_sort_staleness_result(int64_t* changed, int wait) {
  if (!sc.async_pending || !changed) { g.last_error = "noahmp_hip_sort_staleness_result: no pending count"; return -105; }
  if (wait) HIPCHK(hipEventSynchronize(sc.ev_async));
  else {
    hipError_t q = hipEventQuery(sc.ev_async);
    if (q == hipErrorNotReady) return 1;
    HIPCHK(q);
  }
  *changed = *(volatile long*)sc.h_async;
  sc.async_pending = false;
  return 0;
}

int noahmp_hip_scatter_plan(const int32_t* perm, int ni, int nj, uint16_t* order_out, int32_t* dpos_out, void* stream) {
  int rc = nmp_host::ensure_init();
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : g.own_stream;
  const long n = (long)ni * nj;
  if (n <= 0) return 0;
  if ((rc = nmp_host::ensure_bytes((void**)&sc.inv, &sc.inv_b, n * 4))) return rc;
  hipLaunchKernelGGL(invert_perm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, perm, sc.inv, n);
  const int chunk = noahmp_hip_scatter_chunk_of(ni, nj);
  if (chunk == kChunk) {
    hipLaunchKernelGGL(scatter_plan_kernel, dim3((unsigned)((n + kChunk - 1) / kChunk)), dim3(256), 0, s, sc.inv, n, order_out, dpos_out);
  } else {
    if ((rc = nmp_host::ensure_bytes((void**)&sc.slot_in, &sc.slot_in_b, n * 2))) return rc;
    hipLaunchKernelGGL(plan_slots_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sc.slot_in, n, chunk);
    const unsigned nseg = (unsigned)((n + chunk - 1) / chunk);
    auto cnt = rocprim::make_counting_iterator<unsigned>(0);
    auto beg = rocprim::make_transform_iterator(cnt, SegOffset{(unsigned)chunk, (unsigned)n, 0u});
    auto end = rocprim::make_transform_iterator(cnt, SegOffset{(unsigned)chunk, (unsigned)n, 1u});
    size_t need = 0;
    HIPCHK(rocprim::segmented_radix_sort_pairs(nullptr, need, (const unsigned*)sc.inv, (unsigned*)dpos_out, (const unsigned short*)sc.slot_in,
                                               (unsigned short*)order_out, (unsigned)n, nseg, beg, end, 0, 32, s));
    if ((rc = nmp_host::ensure_bytes(&sc.seg_tmp, &sc.seg_tmp_b, need))) return rc;
    HIPCHK(rocprim::segmented_radix_sort_pairs(sc.seg_tmp, need, (const unsigned*)sc.inv, (unsigned*)dpos_out, (const unsigned short*)sc.slot_in,
                                               (unsigned short*)order_out, (unsigned)n, nseg, beg, end, 0, 32, s));
  }
  HIPCHK(hipGetLastError());
  return 0;
}

}  // extern "C"

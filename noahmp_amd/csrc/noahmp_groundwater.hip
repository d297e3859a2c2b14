// Miguez-Macho & Fan groundwater for MI355X (gfx950): kernels + the C-ABI entry noahmp_hip_wtable_mmf(),
// the drop-in for WTABLE_mmf_noahmp (reference phys/module_sf_noahmp_groundwater.F90:14-198), called by the
// reference driver every STEPWTD steps when OPT_RUN == 5 (driver/module_hrldas_noahmp_driver.F90:420-436).
//
// Both kernels are pure streaming passes over (i,j) planes in the caller's Fortran layout (i fastest):
// ~216 algorithmic bytes per cell, a few dozen flops on most cells -> HBM-bound.
// 2-D blocks of 64 x 4 threads: one wavefront per row segment keeps every plane access a full 256-byte
// coalesced line, and the 9-point stencil's +-1 row neighbours are served from the same workgroup's L1/L2
// lines.  blockIdx is linearised row-major so consecutive workgroups (which land on different XCDs) walk
// along i; each XCD's L2 then holds a band of rows shared with its neighbours' halos only at the band edge.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <string.h>
#include "noahmp_hip.h"
#include "nmp_dev_groundwater.hpp"
#include "nmp_engine_host.hpp"
#include "nmp_stage.hpp"
#include <vector>

using namespace nmp;
using nmp_host::g;

namespace {

constexpr int BX = 64, BY = 4;

// KCELL / HEAD on the tile + ring rectangle (gw:237-252).  24 algorithmic bytes and one EXP per cell: pure streaming, so every
// thread takes HEAD_ILP cells (BX apart: each wave instruction still covers 256 contiguous bytes) and has their loads in flight together.
constexpr int HEAD_ILP = 4;
__global__ void __launch_bounds__(BX * BY) gw_head_kernel(const GwArgs k) {
  libm::libm_stage_tables();
  const int gj = k.hj0 + blockIdx.y * BY + threadIdx.y;
  if (gj > k.hj1) return;
  const int gi0 = k.hi0 + blockIdx.x * (BX * HEAD_ILP) + threadIdx.x;
  const size_t row = (size_t)(gj - k.a.jms) * k.ni;
  float fdepth[HEAD_ILP], wtd[HEAD_ILP], topo[HEAD_ILP];
  int st[HEAD_ILP];
#pragma unroll
  for (int u = 0; u < HEAD_ILP; u++) {
    const int gi = gi0 + u * BX;
    const bool in = gi <= k.hi1;
    const size_t x = row + (in ? gi - k.a.ims : k.hi0 - k.a.ims);
    fdepth[u] = k.a.fdepth[x]; wtd[u] = k.a.wtd[x]; topo[u] = k.a.topo[x]; st[u] = k.a.isltyp[x];
  }
#pragma unroll
  for (int u = 0; u < HEAD_ILP; u++) {
    const int gi = gi0 + u * BX;
    if (gi > k.hi1) continue;
    const size_t x = row + (gi - k.a.ims);
    float kc, hd;
    gw_cell_head_values(k, fdepth[u], wtd[u], topo[u], st[u], kc, hd);
    k.kcell[x] = kc;
    k.head[x] = hd;
  }
}

// Split form, first half: KCELL / HEAD and the QLAT stencil over the tile into the plane k.qlat (tile order), ONE launch.  The lateral half
// never writes WTD, so the old head need not be frozen in HBM planes first (the whole-call form below must: gw_head_kernel): a workgroup
// evaluates KCELL / HEAD (gw:237-252) of its BX x QY cells plus their 1-cell ring into LDS -- (BX+2)(QY+2) / (BX QY) = 1.29 evaluations
// per cell instead of one, but no 8 B / cell written and 72 B / cell read back through the caches, and one launch less per call (round 6:
// at the 1152 x 768 tile of an 8-rank run the two launches took 12.4 + 9.6 us, the fused one 14.8-15.1 us; at 4608 x 1536 ~95 -> 91 us).  Same gw_cell_head_values, same
// sum order (gw_qlat_sum): same bits.
constexpr int QY = 2 * BY;
__global__ void __launch_bounds__(BX * BY) gw_qlat_fused_kernel(const GwArgs k) {
  libm::libm_stage_tables();
  constexpr int LX = BX + 2, LY = QY + 2;
  __shared__ float s_kc[LY * LX], s_hd[LY * LX];
  const int gi0 = k.a.its + blockIdx.x * BX - 1, gj0 = k.a.jts + blockIdx.y * QY - 1;      // Fortran indices of the LDS tile's first cell
  const int tid = threadIdx.y * BX + threadIdx.x;
  for (int c = tid; c < LY * LX; c += BX * BY) {
    const int lj = c / LX, li = c - lj * LX;
    const int gi = gi0 + li, gj = gj0 + lj;
    float kc = 0.f, hd = 0.f;
    if (gi >= k.hi0 && gi <= k.hi1 && gj >= k.hj0 && gj <= k.hj1) {                      // the KCELL / HEAD rectangle, gw:231-234
      const size_t x = (size_t)(gj - k.a.jms) * k.ni + (gi - k.a.ims);
      gw_cell_head_values(k, k.a.fdepth[x], k.a.wtd[x], k.a.topo[x], k.a.isltyp[x], kc, hd);
    }
    s_kc[c] = kc; s_hd[c] = hd;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < QY / BY; r++) {
    const int li = threadIdx.x + 1, lj = threadIdx.y + r * BY + 1;
    const int gi = gi0 + li, gj = gj0 + lj;
    if (gi > k.a.ite || gj > k.a.jte) continue;
    const size_t x = (size_t)(gj - k.a.jms) * k.ni + (gi - k.a.ims);
    const bool inq = (gi >= k.qi0 && gi <= k.qi1 && gj >= k.qj0 && gj <= k.qj1);         // the QLAT rectangle, gw:254-257
    const bool land = gw_is_land(k.a, k.a.xland[x], k.a.xice[x], k.a.ivgtyp[x]);
    float q = 0.f;
    if (inq && land) {
      const int c = lj * LX + li, up = c + LX, dn = c - LX;
      q = gw_qlat_sum(s_kc[c], s_hd[c], s_kc[up - 1], s_kc[c - 1], s_kc[dn - 1], s_kc[up], s_kc[dn], s_kc[up + 1], s_kc[c + 1], s_kc[dn + 1],
                      s_hd[up - 1], s_hd[c - 1], s_hd[dn - 1], s_hd[up], s_hd[dn], s_hd[up + 1], s_hd[c + 1], s_hd[dn + 1], k.deltat, k.a.area[x]);
    }
    k.qlat[x] = q;
  }
}

// stencil + per-cell update over the tile (gw:105-195); STENCIL = false: the per-column half of the split form (any column order)
template <bool STENCIL>
__global__ void __launch_bounds__(BX * BY) gw_column_kernel(const GwArgs k) {
  libm::libm_stage_tables();
  const int gi = k.a.its + blockIdx.x * BX + threadIdx.x;
  const int gj = k.a.jts + blockIdx.y * BY + threadIdx.y;
  const bool in = (gi <= k.a.ite && gj <= k.a.jte);
  int land = 0;
  if (in) land = gw_column_t<STENCIL>(k, gi - k.a.ims, gj - k.a.jms, gi, gj);
  const unsigned long long m = __ballot(land != 0), mi = __ballot(in);
  if (((threadIdx.y * BX + threadIdx.x) & 63) == 0) {
    int* cnt = k.counts + ((blockIdx.y * gridDim.x + blockIdx.x) % nmp_host::kCountSlots) * nmp_host::kCountStride;
    if (m) atomicAdd(&cnt[0], __popcll(m));
    if (mi != m) atomicAdd(&cnt[2], __popcll(mi) - __popcll(m));
  }
}

// GROUNDWATER_INIT over its..itf x jts..jtf (drv:1330-1331: itf = min(ite, ide-1))
__global__ void __launch_bounds__(BX * BY) gw_init_kernel(const GwArgs k, int itf, int jtf, int iswater) {
  libm::libm_stage_tables();
  const int gi = k.a.its + blockIdx.x * BX + threadIdx.x;
  const int gj = k.a.jts + blockIdx.y * BY + threadIdx.y;
  if (gi > itf || gj > jtf) return;
  gw_init_column(k, gi - k.a.ims, gj - k.a.jms, gi, gj, iswater);
}

struct WField { const char* name; size_t off; int kind; int lev; int io; };
#define WF(n, kind, lev, io) {#n, offsetof(noahmp_wtable_args, n), kind, lev, io}
const WField kW[] = {
#include "nmp_wtable_fields.inc"
};
constexpr int kNW = sizeof(kW) / sizeof(kW[0]);

inline int imax(int a, int b) { return a > b ? a : b; }
inline int imin(int a, int b) { return a < b ? a : b; }

}  // namespace

extern "C" {

size_t noahmp_hip_sizeof_wtable_args(void) { return sizeof(noahmp_wtable_args); }

// part: 0 = the whole WTABLE_mmf_noahmp; 1 = KCELL / HEAD + the QLAT stencil into `qlat` (tile order); 2 = the per-column half with
// QLAT read from `qlat` (the block may hold the columns in any order)
static int gw_call(const noahmp_wtable_args* a, int mem, void* stream, noahmp_status* st, bool init, int iswater, bool enqueue_only = false,
                   int part = 0, float* qlat = nullptr);

int noahmp_hip_wtable_mmf(const noahmp_wtable_args* a, int mem, void* stream, noahmp_status* st) {
  return gw_call(a, mem, stream, st, false, 0);
}

// The same call for device-resident arrays without the host wait: the two kernels are enqueued on `stream` (ordered after the
// column steps and the halo exchange the caller put there) and the call returns; its tallies are not reported.
int noahmp_hip_wtable_mmf_async(const noahmp_wtable_args* a, void* stream) {
  return gw_call(a, NOAHMP_MEM_DEVICE, stream, nullptr, false, 0, true);
}

// WTABLE_mmf_noahmp in two halves for a run that keeps its columns in the sorted layout: only the QLAT stencil
// needs the (i,j) neighbourhood.  noahmp_hip_wtable_lateral_async: KCELL / HEAD and the stencil (gw:231-292) on TILE-order planes --
// it reads wtd, fdepth, topo, isltyp (tile + ring) and xland, xice, ivgtyp, area of block `a` and writes QLAT [m per call] of every
// tile cell into `qlat` (shaped like the memory block; zero on non-land cells and outside the QLAT rectangle).
// noahmp_hip_wtable_columns_async: everything else of the call (river flux, deep recharge, UPDATEWTD, accumulators: gw:105-195) for a
// block whose columns are in ANY order, with QLAT taken from `qlat` in that order.  One plane each way instead of twelve.
int noahmp_hip_wtable_lateral_async(const noahmp_wtable_args* a, float* qlat, void* stream) {
  if (!qlat) { g.last_error = "noahmp_hip_wtable_lateral_async: qlat is required"; return -105; }
  return gw_call(a, NOAHMP_MEM_DEVICE, stream, nullptr, false, 0, true, 1, qlat);
}
int noahmp_hip_wtable_columns_async(const noahmp_wtable_args* a, const float* qlat, void* stream) {
  if (!qlat) { g.last_error = "noahmp_hip_wtable_columns_async: qlat is required"; return -105; }
  return gw_call(a, NOAHMP_MEM_DEVICE, stream, nullptr, false, 0, true, 2, const_cast<float*>(qlat));
}
int noahmp_hip_groundwater_init(const noahmp_wtable_args* a, int iswater, int mem, void* stream, noahmp_status* st) {
  return gw_call(a, mem, stream, st, true, iswater);
}

}  // extern "C"

static int gw_call(const noahmp_wtable_args* a, int mem, void* stream, noahmp_status* st, bool init, int iswater, bool enqueue_only,
                   int part, float* qlat) {
  if (st) memset(st, 0, sizeof(*st));
  int rc = nmp_host::ensure_init();
  if (rc) return rc;
  if (!g.have_tables) { g.last_error = "noahmp_hip_set_tables() has not been called"; return -102; }
  if (a->nsoil != NOAHMP_NSOIL) { if (st) st->code = NOAHMP_ERR_NSOIL_UNSUPPORTED; return NOAHMP_ERR_NSOIL_UNSUPPORTED; }
  hipStream_t s = stream ? (hipStream_t)stream : g.own_stream;

  GwArgs k;
  memset(&k, 0, sizeof(k));
  k.a = *a;
  k.T = g.d_tables;
  k.ni = a->ime - a->ims + 1;
  const int nj = a->jme - a->jms + 1;
  k.deltat = a->wtddt * 60.f;                                                      // gw:89
  k.zsoil[0] = 0.f;                                                                // gw:91-95
  k.zsoil[1] = -a->dzs[0];
  for (int l = 2; l <= NOAHMP_NSOIL; l++) k.zsoil[l] = -a->dzs[l - 1] + k.zsoil[l - 1];
  for (int l = 0; l < NOAHMP_NSOIL; l++) k.dzs[l] = a->dzs[l];
  k.a.dzs = nullptr;
  k.hi0 = imax(a->its - 1, a->ids); k.hi1 = imin(a->ite + 1, a->ide - 1);         // gw:231-234
  k.hj0 = imax(a->jts - 1, a->jds); k.hj1 = imin(a->jte + 1, a->jde - 1);
  k.qi0 = imax(a->its, a->ids + 1); k.qi1 = imin(a->ite, a->ide - 2);             // gw:254-257
  k.qj0 = imax(a->jts, a->jds + 1); k.qj1 = imin(a->jte, a->jde - 2);
  // The reference indexes KCELL/HEAD(ims:ime,...) at these bounds too; a tile whose ring is not inside the
  // caller's memory is a caller bug there (out-of-bounds) and a refused call here.
  if (part == 2) {                  // no stencil: the block needs no ring (and its columns no (i,j) meaning)
    k.hi0 = k.qi0 = a->its; k.hi1 = k.qi1 = a->ite; k.hj0 = k.qj0 = a->jts; k.hj1 = k.qj1 = a->jte;
  }
  k.qlat = qlat;
  if (k.hi0 < a->ims || k.hi1 > a->ime || k.hj0 < a->jms || k.hj1 > a->jme ||
      a->its < a->ims || a->ite > a->ime || a->jts < a->jms || a->jte > a->jme) {
    g.last_error = "noahmp_hip_wtable_mmf: memory dims (ims:ime,jms:jme) do not hold the tile plus its 1-cell ring";
    return -103;
  }
  const size_t plane = (size_t)k.ni * nj * sizeof(float);
  if (part == 0) {                   // the KCELL / HEAD planes of the whole-call form (the lateral half keeps them in LDS)
    size_t have = g.gw_plane_bytes;
    rc = nmp_host::ensure_bytes((void**)&g.gw_kcell, &have, plane);
    if (rc) return rc;
    have = g.gw_plane_bytes;
    rc = nmp_host::ensure_bytes((void**)&g.gw_head, &have, plane);
    if (rc) return rc;
    g.gw_plane_bytes = have;
  }
  k.kcell = g.gw_kcell;
  k.head = g.gw_head;
  k.err = g.d_err;
  k.counts = g.d_counts;
  if (enqueue_only) {        // pending asynchronous column steps own d_counts: this call's tallies go to a buffer nobody reads
    if (!g.d_gw_counts) HIPCHK(hipMalloc(&g.d_gw_counts, nmp_host::kCountSlots * nmp_host::kCountStride * sizeof(int)));
    k.counts = g.d_gw_counts;
  }

  if (mem == NOAHMP_MEM_HOST) {
    // a resident column state (noahmp_hip_step, "resident_state") shares SMOIS / SH2O / ZWTXY ... with this call: bring the host
    // arrays up to date first, and let the next column step upload them again
    if (g.resident_dirty) { rc = noahmp_hip_fetch(nullptr); if (rc) return rc; }
    g.resident_valid = false;
    if (g.gw_mirror.empty()) { g.gw_mirror.assign(kNW, nullptr); g.gw_mirror_bytes.assign(kNW, 0); }
    std::vector<nmp_host::CopySeg> up;
    for (int f = 0; f < kNW; f++) {
      const size_t bytes = plane * (kW[f].lev == 2 ? a->nsoil : 1);
      rc = nmp_host::ensure_bytes(&g.gw_mirror[f], &g.gw_mirror_bytes[f], bytes);
      if (rc) return rc;
      up.push_back(nmp_host::CopySeg{*(void* const*)((const char*)a + kW[f].off), g.gw_mirror[f], bytes});
      *(void**)((char*)&k.a + kW[f].off) = g.gw_mirror[f];
    }
    if ((rc = nmp_host::copy_segments(up.data(), (int)up.size(), true, s))) return rc;       // pageable arrays: the engine's bounce buffers
  }

  const int hni = k.hi1 - k.hi0 + 1, hnj = k.hj1 - k.hj0 + 1;
  const int tni = a->ite - a->its + 1, tnj = a->jte - a->jts + 1;
  const bool timed = !enqueue_only && tni > 0 && tnj > 0;       // an empty tile: no kernel, no events
  if (!enqueue_only) HIPCHK(hipMemsetAsync(g.d_counts, 0, nmp_host::kCountSlots * nmp_host::kCountStride * sizeof(int), s));
  if (timed) HIPCHK(hipEventRecord(g.ev0, s));
  if (hni > 0 && hnj > 0 && part == 0)
    hipLaunchKernelGGL(gw_head_kernel, dim3((hni + BX * HEAD_ILP - 1) / (BX * HEAD_ILP), (hnj + BY - 1) / BY), dim3(BX, BY), 0, s, k);
  if (init) {
    const int itf = imin(a->ite, a->ide - 1), jtf = imin(a->jte, a->jde - 1);
    const int ini_ = itf - a->its + 1, inj_ = jtf - a->jts + 1;
    if (ini_ > 0 && inj_ > 0)
      hipLaunchKernelGGL(gw_init_kernel, dim3((ini_ + BX - 1) / BX, (inj_ + BY - 1) / BY), dim3(BX, BY), 0, s, k, itf, jtf,
                         iswater);
  } else if (tni > 0 && tnj > 0) {
    const dim3 grid((tni + BX - 1) / BX, (tnj + BY - 1) / BY), block(BX, BY);
    if (part == 1) hipLaunchKernelGGL(gw_qlat_fused_kernel, dim3((tni + BX - 1) / BX, (tnj + QY - 1) / QY), block, 0, s, k);
    else if (part == 2) hipLaunchKernelGGL(gw_column_kernel<false>, grid, block, 0, s, k);
    else hipLaunchKernelGGL(gw_column_kernel<true>, grid, block, 0, s, k);
  }
  HIPCHK(hipGetLastError());
  if (enqueue_only) return 0;
  if (timed) HIPCHK(hipEventRecord(g.ev1, s));
  HIPCHK(hipMemcpyAsync(g.h_counts, g.d_counts, nmp_host::kCountSlots * nmp_host::kCountStride * sizeof(int), hipMemcpyDeviceToHost, s));
  if (mem == NOAHMP_MEM_HOST) {
    std::vector<nmp_host::CopySeg> down;
    for (int f = 0; f < kNW; f++) {
      if (kW[f].io == 0 && !(init && !strcmp(kW[f].name, "smoiseq"))) continue;   // GROUNDWATER_INIT writes SMOISEQ
      const size_t bytes = plane * (kW[f].lev == 2 ? a->nsoil : 1);
      down.push_back(nmp_host::CopySeg{*(void* const*)((const char*)a + kW[f].off), g.gw_mirror[f], bytes});
    }
    if ((rc = nmp_host::copy_segments(down.data(), (int)down.size(), false, s))) return rc;
  }
  HIPCHK(hipStreamSynchronize(s));
  if (st) {
    float ms = 0.f;
    if (timed) hipEventElapsedTime(&ms, g.ev0, g.ev1);
    st->kernel_ms = ms;
    noahmp_status t;
    nmp_host::status_counts(&t);
    st->n_land = t.n_land;
    st->n_skipped = t.n_skipped;
  }
  return 0;
}

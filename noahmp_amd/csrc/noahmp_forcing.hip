// Forcing preparation on the device: C-ABI entry noahmp_hip_forcing_prep(), the device-resident replacement of
// driver/module_hrldas_noahmp_driver.F90:336-354 (+ CALC_DECLIN, hdrv:813-863).  The uniform part of CALC_DECLIN
// (Julian day, solar declination: four libm calls per step) is evaluated here on the host with the host's libm,
// exactly as the reference does; the per-cell part runs in the kernel with the same libm algorithms (nmp_libm.hpp).
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include "noahmp_hip.h"
#include "nmp_dev_forcing.hpp"
#include "nmp_engine_host.hpp"

using namespace nmp;
using nmp_host::g;

namespace {
// Grid-stride loops over a CAPPED grid (kForcingBlocks workgroups): a driver enqueues these kernels for a LATER step on a second stream
// while a column kernel runs (bench.py, config 5).  Beside the land kernel's ~110 000 waves a second queue gets about every other wave
// slot that comes free, so a kernel of 100 000 short waves (one cell per thread at 6.5 M cells) lasts as long as the land kernel itself
// and ends behind it -- and the command processor does not start the next land kernel before that (kernel trace, profiles/r05_experiments.md
// section 4).  16 384 longer waves are through in the first tenth of the land kernel.  Alone on the GPU the kernels take the same time.
constexpr unsigned kForcingBlocks = 4096;
__global__ void __launch_bounds__(256) noahmp_forcing_kernel(const ForcingArgs k, int nti, int ntj) {
  const long n = (long)nti * ntj, stride = (long)gridDim.x * blockDim.x;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
    const int tj = (int)(t / nti), ti = (int)(t - (long)tj * nti);
    forcing_cell(k, k.a.its - k.a.ims + ti, k.a.jts - k.a.jms + tj);
  }
}
__global__ void __launch_bounds__(256) noahmp_interp_kernel(const InterpArgs k, int nti, int ntj) {
  const long n = (long)nti * ntj, stride = (long)gridDim.x * blockDim.x;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
    const int tj = (int)(t / nti), ti = (int)(t - (long)tj * nti);
    interp_cell(k, k.a.its - k.a.ims + ti, k.a.jts - k.a.jms + tj);
  }
}
// interpolation and preparation of one step in ONE launch (noahmp_hip_forcing_interpolate_prep): the same two cell functions, one after
// the other in the same thread -- what the second reads of the first (level 1 of the five atmospheric fields, the rain rate) comes
// out of the cache, and a launch with its gap is gone (config 5, 6.5 M cells: 0.20 ms for the two launches, 0.13 ms fused)
__global__ void __launch_bounds__(256) noahmp_interp_forcing_kernel(const InterpArgs ki, const ForcingArgs kf, int nti, int ntj) {
  const long n = (long)nti * ntj, stride = (long)gridDim.x * blockDim.x;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
    const int tj = (int)(t / nti), ti = (int)(t - (long)tj * nti);
    interp_cell(ki, ki.a.its - ki.a.ims + ti, ki.a.jts - ki.a.jms + tj);
    forcing_cell(kf, kf.a.its - kf.a.ims + ti, kf.a.jts - kf.a.jms + tj);
  }
}
}  // namespace

// ---- column permutation of fields (sorted device-resident layout) ---------------------------
// dst column p <- src column perm[p], for up to 32 fields of 1..8 levels each, one thread per destination column:
// writes are coalesced, reads gather.  Used per step for the forcing of a run whose state is kept sorted by
// (class, vegetation type), and once for the state itself.
namespace {
constexpr int kMaxGather = 32;
struct GatherArgs {
  void* dst[kMaxGather];
  const void* src[kMaxGather];
  int nlev[kMaxGather];
  const int* perm;                  // NULL = identity
  const int* vegtyp;                // output packing only: IVGTYP in SOURCE order, or NULL
  int iswater;
  unsigned mask;                    // bit f: field f gets -1.E33 on water points (put_var_2d / put_var_3d)
  int n, ni, nj;
};
__global__ void __launch_bounds__(256) noahmp_gather_kernel(const GatherArgs k) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long ncol = (long)k.ni * k.nj;
  if (p >= ncol) return;
  const long gsrc = k.perm ? k.perm[p] : p;
  const int pj = (int)(p / k.ni), pi = (int)(p - (long)pj * k.ni);
  const int gj = (int)(gsrc / k.ni), gi = (int)(gsrc - (long)gj * k.ni);
  for (int f = 0; f < k.n; f++) {
    const int nk = k.nlev[f];
    const uint32_t* s = (const uint32_t*)k.src[f];
    uint32_t* d = (uint32_t*)k.dst[f];
    for (int l = 0; l < nk; l++) d[((size_t)pj * nk + l) * k.ni + pi] = s[((size_t)gj * nk + l) * k.ni + gi];
  }
  if (k.vegtyp && k.mask && k.vegtyp[gsrc] == k.iswater) {
    const uint32_t missing = __float_as_uint(-1.E33f);              // netcdf_io:1971, 2041
    for (int f = 0; f < k.n; f++) {
      if (!((k.mask >> f) & 1u)) continue;
      const int nk = k.nlev[f];
      uint32_t* d = (uint32_t*)k.dst[f];
      for (int l = 0; l < nk; l++) d[((size_t)pj * nk + l) * k.ni + pi] = missing;
    }
  }
}
}  // namespace

// The same permutation as a chunked scatter with sorted write order: a workgroup loads CHUNK consecutive source columns
// of a field (coalesced) into LDS and writes them in ascending order of their destination, so that the columns that go
// to the same group leave as one contiguous run instead of 4 bytes at a time.  `order[c*CHUNK+q]` = offset inside chunk c
// of the column with the q-th smallest destination, `dpos[...]` = that destination (both prepared once per sort).
namespace {
constexpr int kChunk = 1024;
struct ScatterArgs {
  void* dst[kMaxGather];
  const void* src[kMaxGather];
  int nlev[kMaxGather];
  const unsigned short* order;
  const int* dpos;
  int n, ni, nj;
  // the tile-order side may be the interior of a larger memory block (a tile that carries the LATERALFLOW ring): cell (ti, tj)
  // of the tile sits at row tj + j_off, column ti + i_off of rows ni_mem long
  int ni_mem, i_off, j_off;
  int reverse;                     // 0: tile order -> sorted (src = tile side), 1: sorted -> tile order (dst = tile side)
};
__global__ void __launch_bounds__(256) noahmp_scatter_kernel(const ScatterArgs k) {
  constexpr int R = kChunk / 256;
  __shared__ uint32_t buf[2][kChunk];                 // double-buffered: one barrier per field level
  const long ncol = (long)k.ni * k.nj;
  const long base = (long)blockIdx.x * kChunk;
  int ord[R], sj[R], si[R], pj[R], pi[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const long q = base + r * 256 + threadIdx.x;
    const bool in = q < ncol;
    ord[r] = in ? k.order[q] : 0;
    const int dp = in ? k.dpos[q] : -1;
    pj[r] = dp >= 0 ? dp / k.ni : -1;
    pi[r] = dp >= 0 ? dp - pj[r] * k.ni : 0;
    sj[r] = in ? (int)(q / k.ni) : -1;
    si[r] = in ? (int)(q - (long)sj[r] * k.ni) + k.i_off : 0;
    if (in) sj[r] += k.j_off;
  }
  int phase = 0;
  for (int f = 0; f < k.n; f++) {
    const int nk = k.nlev[f] < 0 ? -k.nlev[f] : k.nlev[f];      // nlev < 0: |nlev| levels in memory, only the first one is moved
    const int nmove = k.nlev[f] < 0 ? 1 : nk;
    const uint32_t* s = (const uint32_t*)k.src[f];
    uint32_t* d = (uint32_t*)k.dst[f];
    for (int l = 0; l < nmove; l++, phase ^= 1) {
      if (!k.reverse) {          // coalesced reads of consecutive tile cells, writes in ascending sorted position
#pragma unroll
        for (int r = 0; r < R; r++)
          if (sj[r] >= 0) buf[phase][r * 256 + threadIdx.x] = s[((size_t)sj[r] * nk + l) * k.ni_mem + si[r]];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; r++)
          if (pj[r] >= 0) d[((size_t)pj[r] * nk + l) * k.ni + pi[r]] = buf[phase][ord[r]];
      } else {                   // reads in ascending sorted position, coalesced writes of consecutive tile cells
#pragma unroll
        for (int r = 0; r < R; r++)
          if (pj[r] >= 0) buf[phase][ord[r]] = s[((size_t)pj[r] * nk + l) * k.ni + pi[r]];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; r++)
          if (sj[r] >= 0) d[((size_t)sj[r] * nk + l) * k.ni_mem + si[r]] = buf[phase][r * 256 + threadIdx.x];
      }
    }
  }
}
// Large tiles: chunks of 8 K / 16 K columns per workgroup of 1024 threads (the whole chunk of one field level in LDS, up to 64 KB).
// Columns of one sort group keep their tile order (stable sort), so a chunk of C consecutive tile columns holds, for a group that owns
// the share s of the tile, a run of ~s C consecutive destinations: with 1024-column chunks and ~2000 groups nearly every 4-byte store
// goes to a line of its own (measured: 1.9 TB/s read + write); with 16 K columns the runs are 8+ columns.
// Per element two registers are kept (destination, LDS slot | destination row); the tile-side address advances incrementally.
template <int CHUNK>
__global__ void __launch_bounds__(1024) noahmp_scatter_big_kernel(const ScatterArgs k) {
  constexpr int T = 1024, R = CHUNK / T;
  __shared__ uint32_t buf[CHUNK];                   // single buffer, two barriers per field level (double buffering measured slower: 158 vs 125 us)
  const long ncol = (long)k.ni * k.nj;
  const long base = (long)blockIdx.x * CHUNK;
  int dp[R]; unsigned oj[R];                         // destination column (-1: none) | LDS slot << 16 | destination row
#pragma unroll
  for (int r = 0; r < R; r++) {
    const long q = base + r * T + threadIdx.x;
    const bool in = q < ncol;
    dp[r] = in ? k.dpos[q] : -1;
    const unsigned pj = dp[r] >= 0 ? (unsigned)(dp[r] / k.ni) : 0u;
    oj[r] = (in ? (unsigned)k.order[q] << 16 : 0u) | pj;
  }
  // tile-side cell of element r: q = base + r T + tid -> (row, column) by one division, then by increments of T
  const long q0 = base + threadIdx.x;
  const int sj0 = (int)(q0 / k.ni), si0 = (int)(q0 - (long)sj0 * k.ni);
#pragma unroll 1
  for (int f = 0; f < k.n; f++) {
    const int nk = k.nlev[f] < 0 ? -k.nlev[f] : k.nlev[f];      // nlev < 0: |nlev| levels in memory, only the first one is moved
    const int nmove = k.nlev[f] < 0 ? 1 : nk;
    const uint32_t* s = (const uint32_t*)k.src[f];
    uint32_t* d = (uint32_t*)k.dst[f];
#pragma unroll 1
    for (int l = 0; l < nmove; l++) {
      int sj = sj0, si = si0;
      asm volatile("" : "+v"(sj), "+v"(si));
      // elements go in groups of G with a scheduling barrier between groups: left alone, the compiler batches all R loads of a thread
      // (R values + R 64-bit addresses) and spills 320 registers
      constexpr int G = 8;
      if (!k.reverse) {          // coalesced reads of consecutive tile cells, writes in ascending sorted position
#pragma unroll
        for (int g0 = 0; g0 < R; g0 += G) {
          uint32_t v[G];
#pragma unroll
          for (int u = 0; u < G; u++) {
            v[u] = (base + (g0 + u) * T + threadIdx.x < ncol) ? s[((size_t)(sj + k.j_off) * nk + l) * k.ni_mem + si + k.i_off] : 0u;
            si += T; while (si >= k.ni) { si -= k.ni; sj++; }
          }
#pragma unroll
          for (int u = 0; u < G; u++) buf[(g0 + u) * T + threadIdx.x] = v[u];
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
#pragma unroll
        for (int g0 = 0; g0 < R; g0 += G) {
          uint32_t v[G];
#pragma unroll
          for (int u = 0; u < G; u++) v[u] = buf[oj[g0 + u] >> 16];
#pragma unroll
          for (int u = 0; u < G; u++) {
            int dpr = dp[g0 + u];
            asm volatile("" : "+v"(dpr));          // keeps the address arithmetic inside the loop (no 64-bit address per element live across levels)
            if (dpr >= 0) d[(size_t)dpr + ((size_t)(oj[g0 + u] & 0xFFFFu) * (nk - 1) + l) * k.ni] = v[u];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {                   // reads in ascending sorted position, coalesced writes of consecutive tile cells
#pragma unroll
        for (int g0 = 0; g0 < R; g0 += G) {
          uint32_t v[G];
#pragma unroll
          for (int u = 0; u < G; u++) {
            int dpr = dp[g0 + u];
            asm volatile("" : "+v"(dpr));
            v[u] = dpr >= 0 ? s[(size_t)dpr + ((size_t)(oj[g0 + u] & 0xFFFFu) * (nk - 1) + l) * k.ni] : 0u;
          }
#pragma unroll
          for (int u = 0; u < G; u++) if (dp[g0 + u] >= 0) buf[oj[g0 + u] >> 16] = v[u];
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
#pragma unroll
        for (int g0 = 0; g0 < R; g0 += G) {
          uint32_t v[G];
#pragma unroll
          for (int u = 0; u < G; u++) v[u] = buf[(g0 + u) * T + threadIdx.x];
#pragma unroll
          for (int u = 0; u < G; u++) {
            if (base + (g0 + u) * T + threadIdx.x < ncol) d[((size_t)(sj + k.j_off) * nk + l) * k.ni_mem + si + k.i_off] = v[u];
            si += T; while (si >= k.ni) { si -= k.ni; sj++; }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();           // the buffer is reused by the next level
    }
  }
}

void launch_scatter(const ScatterArgs& k, hipStream_t s) {
  const long ncol = (long)k.ni * k.nj;
  if (ncol <= 0 || k.n <= 0) return;
  const int chunk = noahmp_hip_scatter_chunk_of(k.ni, k.nj);
  const dim3 grid((unsigned)((ncol + chunk - 1) / chunk));
  if (chunk == 16384) hipLaunchKernelGGL(noahmp_scatter_big_kernel<16384>, grid, dim3(1024), 0, s, k);
  else if (chunk == 8192) hipLaunchKernelGGL(noahmp_scatter_big_kernel<8192>, grid, dim3(1024), 0, s, k);
  else if (chunk == 4096) hipLaunchKernelGGL(noahmp_scatter_big_kernel<4096>, grid, dim3(1024), 0, s, k);
  else if (chunk == 2048) hipLaunchKernelGGL(noahmp_scatter_big_kernel<2048>, grid, dim3(1024), 0, s, k);
  else hipLaunchKernelGGL(noahmp_scatter_kernel, grid, dim3(256), 0, s, k);
}
}  // namespace

extern "C" {

// Columns per workgroup of the chunked permutation for a tile of ni x nj columns (the plan of noahmp_hip_scatter_plan is built for it):
// as large as LDS allows while the launch still has about one workgroup per CU.
int noahmp_hip_scatter_chunk_of(int ni, int nj) {
  const long ncol = (long)ni * nj;
  if (nj > 65535) return kChunk;                 // the big kernel packs the destination row into 16 bits
  // the largest chunk that still gives the launch ~200 workgroups (256 CUs; a workgroup of 1024 threads fills a CU's LDS share);
  // 32768 would need 2 x 32 registers per thread for the plan: it spills
  for (int chunk = 16384; chunk >= 2048; chunk >>= 1)
    if (ncol >= 200L * chunk) return chunk;
  return kChunk;
}

int noahmp_hip_scatter_fields(int n, void* const* dst, const void* const* src, const int* nlev, const uint16_t* order,
                              const int32_t* dpos, int ni, int nj, void* stream) {
  int rc = nmp_host::ensure_init();
  if (rc) return rc;
  if (n < 0 || n > kMaxGather) { g.last_error = "noahmp_hip_scatter_fields: at most 32 fields per call"; return -107; }
  hipStream_t s = stream ? (hipStream_t)stream : g.own_stream;
  ScatterArgs k;
  memset(&k, 0, sizeof(k));
  for (int f = 0; f < n; f++) { k.dst[f] = dst[f]; k.src[f] = src[f]; k.nlev[f] = nlev[f]; }
  k.order = order; k.dpos = dpos; k.n = n; k.ni = ni; k.nj = nj; k.ni_mem = ni;
  launch_scatter(k, s);
  HIPCHK(hipGetLastError());
  return 0;
}

// The same plan between a SORTED store of a tile (nti x ntj columns, sorted positions) and planes in TILE order that may be the
// interior of a larger memory block (rows ni_mem long, tile origin at (i_off, j_off)): direction 0 tile -> sorted, 1 sorted -> tile.
// An OPT_RUN = 5 run keeps its columns sorted for the column kernel and returns the planes WTABLE_mmf_noahmp shares with it
// (SMOIS, SH2O, SMCWTD, ZWT, DEEPRECH, RECH) to (i,j) order around every groundwater call (the stencil of gw:259-292 needs it).
int noahmp_hip_sorted_exchange(int n, void* const* sorted, void* const* tile, const int* nlev, const uint16_t* order,
                               const int32_t* dpos, int nti, int ntj, int ni_mem, int i_off, int j_off, int direction, void* stream) {
  int rc = nmp_host::ensure_init();
  if (rc) return rc;
  if (n < 0 || n > kMaxGather) { g.last_error = "noahmp_hip_sorted_exchange: at most 32 fields per call"; return -107; }
  if (ni_mem < nti + i_off || i_off < 0 || j_off < 0) { g.last_error = "noahmp_hip_sorted_exchange: the tile does not fit the memory block"; return -105; }
  hipStream_t s = stream ? (hipStream_t)stream : g.own_stream;
  ScatterArgs k;
  memset(&k, 0, sizeof(k));
  for (int f = 0; f < n; f++) { k.dst[f] = direction ? tile[f] : sorted[f]; k.src[f] = direction ? sorted[f] : tile[f]; k.nlev[f] = nlev[f]; }
  k.order = order; k.dpos = dpos; k.n = n; k.ni = nti; k.nj = ntj; k.ni_mem = ni_mem; k.i_off = i_off; k.j_off = j_off;
  k.reverse = direction ? 1 : 0;
  launch_scatter(k, s);
  HIPCHK(hipGetLastError());
  return 0;
}

int noahmp_hip_scatter_chunk(void) { return kChunk; }

int noahmp_hip_gather_fields(int n, void* const* dst, const void* const* src, const int* nlev, const int32_t* perm, int ni,
                             int nj, void* stream) {
  int rc = nmp_host::ensure_init();
  if (rc) return rc;
  if (n < 0 || n > kMaxGather) { g.last_error = "noahmp_hip_gather_fields: at most 32 fields per call"; return -107; }
  hipStream_t s = stream ? (hipStream_t)stream : g.own_stream;
  GatherArgs k;
  memset(&k, 0, sizeof(k));
  for (int f = 0; f < n; f++) { k.dst[f] = dst[f]; k.src[f] = src[f]; k.nlev[f] = nlev[f]; }
  k.perm = perm; k.n = n; k.ni = ni; k.nj = nj;
  const long ncol = (long)ni * nj;
  if (ncol > 0 && n > 0)
    hipLaunchKernelGGL(noahmp_gather_kernel, dim3((unsigned)((ncol + 255) / 256)), dim3(256), 0, s, k);
  HIPCHK(hipGetLastError());
  return 0;
}

// Output / restart packing for a device-resident run: what put_var_2d / put_var_3d do to an array before nf90_put_var
// (driver/module_hrldas_netcdf_io.F90:1971-1975, 2039-2042: water points become -1.E33 -- in output files for every real
// field, in restart files for the layered fields only, netcdf_io:2347 restart_flag) fused with the return from the sorted
// layout to tile order.  dst column p <- src column perm[p]; ivgtyp_src is in SOURCE order.
int noahmp_hip_output_fields(int n, void* const* dst, const void* const* src, const int* nlev, const int32_t* perm,
                             const int32_t* ivgtyp_src, int iswater, uint32_t mask_fields, int ni, int nj, void* stream) {
  int rc = nmp_host::ensure_init();
  if (rc) return rc;
  if (n < 0 || n > kMaxGather) { g.last_error = "noahmp_hip_output_fields: at most 32 fields per call"; return -107; }
  if (mask_fields && !ivgtyp_src) { g.last_error = "noahmp_hip_output_fields: masking needs IVGTYP"; return -105; }
  hipStream_t s = stream ? (hipStream_t)stream : g.own_stream;
  GatherArgs k;
  memset(&k, 0, sizeof(k));
  for (int f = 0; f < n; f++) { k.dst[f] = dst[f]; k.src[f] = src[f]; k.nlev[f] = nlev[f]; }
  k.perm = perm; k.vegtyp = ivgtyp_src; k.iswater = iswater; k.mask = mask_fields;
  k.n = n; k.ni = ni; k.nj = nj;
  const long ncol = (long)ni * nj;
  if (ncol > 0 && n > 0)
    hipLaunchKernelGGL(noahmp_gather_kernel, dim3((unsigned)((ncol + 255) / 256)), dim3(256), 0, s, k);
  HIPCHK(hipGetLastError());
  return 0;
}

// hdrv:826-854: JULIAN and the solar declination; returns JULIAN, fills sin/cos of the declination
float noahmp_hip_declination(int iday, int ihour, float* sin_declin, float* cos_declin) {
  const float DEGRAD = 3.14159265f / 180.f, DPD = 360.f / 365.f;
  const float julian = (float)iday + (float)ihour / 24.f;
  const float obecl = 23.5f * DEGRAD;
  const float sinob = sinf(obecl);
  float sxlong;
  if (julian >= 80.f) sxlong = DPD * (julian - 80.f) * DEGRAD;
  else sxlong = DPD * (julian + 285.f) * DEGRAD;
  const float arg = sinob * sinf(sxlong);
  const float declin = asinf(arg);
  if (sin_declin) *sin_declin = sinf(declin);
  if (cos_declin) *cos_declin = cosf(declin);
  return julian;
}

}  // extern "C" (helpers follow)

// argument blocks of the two kernels (checked); 0 or a negative return code with g.last_error set
static int fill_forcing_args(ForcingArgs& k, const noahmp_step_args* a, const float* lon2d, const float* rain_rate, int iday, int ihour,
                             int iminute, int isecond, float zlvl, int flags, float* julian_out) {
  memset(&k, 0, sizeof(k));
  k.a = *a;
  k.lon = lon2d;
  k.rain_rate = rain_rate;
  const float julian = noahmp_hip_declination(iday, ihour, &k.sin_declin, &k.cos_declin);
  if (julian_out) *julian_out = julian;
  k.hour_utc = (float)ihour + (float)iminute / 60.0f + (float)isecond / 3600.0f;   // hdrv:856, left to right
  k.dt = a->dt;
  k.dz8w = 2.0f * zlvl;
  k.scale_vegfra = (flags & NOAHMP_PREP_SCALE_VEGFRA) ? 1 : 0;
  k.first_step = (flags & NOAHMP_PREP_FIRST_STEP) ? 1 : 0;
  k.ni = a->ime - a->ims + 1;
  k.nka = a->kme - a->kms + 1;
  k.k1 = 1 - a->kms;
  if (k.nka < 2) { g.last_error = "forcing_prep needs two atmospheric levels (kms:kme)"; return -105; }
  return 0;
}
static int fill_interp_args(InterpArgs& k, const noahmp_step_args* a, const noahmp_forcing_record* ra, const noahmp_forcing_record* rb,
                            int idts, int idts2, float* rain_rate_out) {
  if (!ra || !ra->t || !ra->q || !ra->u || !ra->v || !ra->p || !ra->lw || !ra->sw || !ra->pcp || !rain_rate_out) {
    g.last_error = "forcing_interpolate: record A needs t q u v p lw sw pcp, and rain_rate_out must be given";
    return -105;
  }
  if (rb && (!rb->t || !rb->q || !rb->u || !rb->v || !rb->p || !rb->lw || !rb->sw)) {
    g.last_error = "forcing_interpolate: record B needs t q u v p lw sw";
    return -105;
  }
  if (rb && (idts2 <= 0 || idts < 0 || idts > idts2)) {
    // hrldas_input_read stops unless lastread < target < nextread (netcdf_io:1286, 1297-1301); the end points are
    // accepted here (fraction 1 and 0).
    g.last_error = "forcing_interpolate: target date outside the bracketing records";
    return -105;
  }
  memset(&k, 0, sizeof(k));
  k.a = *a;
  k.ra = *ra;
  if (rb) k.rb = *rb;
  k.has_b = rb ? 1 : 0;
  k.rain_rate = rain_rate_out;
  k.fraction = rb ? (float)(idts2 - idts) / (float)idts2 : 1.0f;      // netcdf_io:1390
  k.one_minus = 1.0f - k.fraction;
  k.ni = a->ime - a->ims + 1;
  k.nka = a->kme - a->kms + 1;
  k.k1 = 1 - a->kms;
  return 0;
}
// launch `fn(grid)` between the timing events, wait and fill *st if it is given
template <class Launch>
static int run_forcing_launch(const noahmp_step_args* a, hipStream_t s, noahmp_status* st, Launch launch) {
  const int nti = a->ite - a->its + 1, ntj = a->jte - a->jts + 1;
  if (st) HIPCHK(hipEventRecord(g.ev0, s));
  if (nti > 0 && ntj > 0) {
    const long n = (long)nti * ntj;
    const unsigned nb = (unsigned)((n + 255) / 256);
    launch(dim3(nb < kForcingBlocks ? nb : kForcingBlocks), nti, ntj);
  }
  HIPCHK(hipGetLastError());
  if (st) {                                   // st == NULL: enqueue only (ordered on `stream`), no host wait
    HIPCHK(hipEventRecord(g.ev1, s));
    HIPCHK(hipStreamSynchronize(s));
    float ms = 0.f;
    hipEventElapsedTime(&ms, g.ev0, g.ev1);
    st->kernel_ms = ms;
    st->n_land = nti > 0 && ntj > 0 ? nti * ntj : 0;
  }
  return 0;
}

extern "C" {

int noahmp_hip_forcing_prep(const noahmp_step_args* a, const float* lon2d, const float* rain_rate, int iday, int ihour,
                            int iminute, int isecond, float zlvl, int flags, float* julian_out, int mem,
                            void* stream, noahmp_status* st) {
  if (st) memset(st, 0, sizeof(*st));
  int rc = nmp_host::ensure_init();
  if (rc) return rc;
  if (mem != NOAHMP_MEM_DEVICE) {
    g.last_error = "noahmp_hip_forcing_prep works on device-resident arrays only (a host caller keeps hdrv:336-354)";
    return -104;
  }
  hipStream_t s = stream ? (hipStream_t)stream : g.own_stream;
  ForcingArgs k;
  if ((rc = fill_forcing_args(k, a, lon2d, rain_rate, iday, ihour, iminute, isecond, zlvl, flags, julian_out))) return rc;
  return run_forcing_launch(a, s, st, [&](dim3 grid, int nti, int ntj) {
    hipLaunchKernelGGL(noahmp_forcing_kernel, grid, dim3(256), 0, s, k, nti, ntj);
  });
}

int noahmp_hip_forcing_interpolate(const noahmp_step_args* a, const noahmp_forcing_record* ra, const noahmp_forcing_record* rb,
                                   int idts, int idts2, float* rain_rate_out, int mem, void* stream, noahmp_status* st) {
  if (st) memset(st, 0, sizeof(*st));
  int rc = nmp_host::ensure_init();
  if (rc) return rc;
  if (mem != NOAHMP_MEM_DEVICE) {
    g.last_error = "noahmp_hip_forcing_interpolate works on device-resident arrays only (a host caller keeps hrldas_input_read)";
    return -104;
  }
  hipStream_t s = stream ? (hipStream_t)stream : g.own_stream;
  InterpArgs k;
  if ((rc = fill_interp_args(k, a, ra, rb, idts, idts2, rain_rate_out))) return rc;
  return run_forcing_launch(a, s, st, [&](dim3 grid, int nti, int ntj) {
    hipLaunchKernelGGL(noahmp_interp_kernel, grid, dim3(256), 0, s, k, nti, ntj);
  });
}

// noahmp_hip_forcing_interpolate followed by noahmp_hip_forcing_prep in one launch: rain_rate is the scratch plane the first fills and
// the second reads (RAINBL_tmp); the arguments of both, same results
int noahmp_hip_forcing_interpolate_prep(const noahmp_step_args* a, const noahmp_forcing_record* ra, const noahmp_forcing_record* rb,
                                        int idts, int idts2, float* rain_rate, const float* lon2d, int iday, int ihour, int iminute,
                                        int isecond, float zlvl, int flags, float* julian_out, int mem, void* stream, noahmp_status* st) {
  if (st) memset(st, 0, sizeof(*st));
  int rc = nmp_host::ensure_init();
  if (rc) return rc;
  if (mem != NOAHMP_MEM_DEVICE) {
    g.last_error = "noahmp_hip_forcing_interpolate_prep works on device-resident arrays only";
    return -104;
  }
  hipStream_t s = stream ? (hipStream_t)stream : g.own_stream;
  InterpArgs ki;
  ForcingArgs kf;
  if ((rc = fill_interp_args(ki, a, ra, rb, idts, idts2, rain_rate))) return rc;
  if ((rc = fill_forcing_args(kf, a, lon2d, rain_rate, iday, ihour, iminute, isecond, zlvl, flags, julian_out))) return rc;
  return run_forcing_launch(a, s, st, [&](dim3 grid, int nti, int ntj) {
    hipLaunchKernelGGL(noahmp_interp_forcing_kernel, grid, dim3(256), 0, s, ki, kf, nti, ntj);
  });
}

}  // extern "C"

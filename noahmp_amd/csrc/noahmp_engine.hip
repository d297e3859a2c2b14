// Noah-MP column engine for MI355X (gfx950): kernels + the C-ABI of include/noahmp_hip.h.
//
// Drop-in boundary: noahmp_hip_step() replaces the body of module_sf_noahmpdrv::noahmplsm
// (reference phys/module_sf_noahmpdrv.F90:11-844).  The JLOOP/ILOOP gather -> NOAHMP_SFLX ->
// scatter of drv:397-840 becomes one kernel launch: thread t owns column (i,j); every field is
// read / written in the caller's own Fortran layout, whose i-contiguous rows are already the
// coalesced structure-of-arrays the GPU wants.  No CPU fallback exists in this library.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "noahmp_hip.h"
#define NMP_WIDE_INDEX 1     // this unit's kernels (run-time options; land ice; skipped cells) address arrays of any size (nmp_dev_column.hpp)
#include "nmp_kernel.hpp"
#include "nmp_stage.hpp"

using namespace nmp;

namespace {

// --------------------------------------------------------------------------------------------
// host side
struct FieldDesc { const char* name; size_t off; int kind; int lev; int io; };   // kind 0 float*, 1 int*
// lev: 0 2-D, 1 atm, 2 soil, 3 snow, 4 snso ; io: 0 in, 1 inout, 2 out
#define FD(n, kind, lev, io) {#n, offsetof(noahmp_step_args, n), kind, lev, io}
const FieldDesc kFields[] = {
#include "nmp_fields.inc"
};
constexpr int kNumFields = sizeof(kFields) / sizeof(kFields[0]);

}  // namespace

namespace nmp_host {
Engine g;

int ensure_init() {
  if (g.d_err) return 0;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    g.last_error = "no HIP device visible: the Noah-MP HIP engine has no CPU fallback";
    return -101;
  }
  if (g.device < 0) { HIPCHK(hipGetDevice(&g.device)); }
  HIPCHK(hipMalloc(&g.d_err, sizeof(unsigned long long)));
  HIPCHK(hipMalloc(&g.d_counts, kCountSlots * kCountStride * sizeof(int)));
  HIPCHK(hipHostMalloc((void**)&g.h_err, sizeof(unsigned long long), hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&g.h_counts, kCountSlots * kCountStride * sizeof(int), hipHostMallocDefault));
  HIPCHK(hipEventCreate(&g.ev0));
  HIPCHK(hipEventCreate(&g.ev1));
  HIPCHK(hipStreamCreateWithFlags(&g.own_stream, hipStreamNonBlocking));
  g.mirror.assign(kNumFields, nullptr);
  g.mirror_bytes.assign(kNumFields, 0);
  return 0;
}

void sum_counts(long long* out) {
  for (int c = 0; c < 4; c++) out[c] = 0;
  for (int s = 0; s < kCountSlots; s++)
    for (int c = 0; c < 4; c++) out[c] += g.h_counts[s * kCountStride + c];
  for (int c = 0; c < 3; c++) g.last_counts[c] = out[c];
}
// the tallies as noahmp_status carries them (int32): one step always fits; a sync over many steps saturates and the caller
// reads the 64-bit sums through noahmp_hip_sync_counts
void status_counts(noahmp_status* st) {
  long long cnt[4];
  sum_counts(cnt);
  auto sat = [](long long v) { return (int32_t)(v > 0x7FFFFFFFll ? 0x7FFFFFFFll : v); };
  st->n_land = sat(cnt[0]); st->n_glacier = sat(cnt[1]); st->n_skipped = sat(cnt[2]);
}

int ensure_bytes(void** p, size_t* have, size_t need) {
  if (*have >= need) return 0;
  if (*p) HIPCHK(hipFree(*p));
  *p = nullptr; *have = 0;
  HIPCHK(hipMalloc(p, need));
  *have = need;
  return 0;
}
}  // namespace nmp_host

using nmp_host::g;
using nmp_host::ensure_init;
using nmp_host::kCountSlots;
using nmp_host::kCountStride;

namespace {

size_t field_elems(const FieldDesc& f, const noahmp_step_args* a) {
  size_t ni = a->ime - a->ims + 1, nj = a->jme - a->jms + 1;
  size_t nk = 1;
  switch (f.lev) {
    case 1: nk = a->kme - a->kms + 1; break;
    case 2: nk = a->nsoil; break;
    case 3: nk = 3; break;
    case 4: nk = a->nsoil + 3; break;
  }
  return ni * nk * nj;
}

template <int BLOCK>
void launch(const KArgs& k, long ncol, bool lds, hipStream_t st) {
  dim3 grid((unsigned)((ncol + BLOCK - 1) / BLOCK)), block(BLOCK);
  if (lds) hipLaunchKernelGGL((noahmp_column_kernel<BLOCK, true>), grid, block, 0, st, k);
  else hipLaunchKernelGGL((noahmp_column_kernel<BLOCK, false>), grid, block, 0, st, k);
}

}  // namespace

extern "C" {

int noahmp_hip_abi_version(void) { return NOAHMP_HIP_ABI_VERSION; }
int noahmp_hip_nsoil(void) { return NOAHMP_NSOIL; }
size_t noahmp_hip_sizeof_step_args(void) { return sizeof(noahmp_step_args); }
size_t noahmp_hip_sizeof_tables(void) { return sizeof(noahmp_tables); }

int noahmp_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int noahmp_hip_set_device(int device) {
  HIPCHK(hipSetDevice(device));
  g.device = device;
  return 0;
}

// Device memory for callers without a HIP binding of their own (a Fortran driver without hipfort): plain
// hipMalloc / hipMemcpy / hipFree on the engine's device.  kind: 0 = host -> device, 1 = device -> host, 2 = device -> device.
void* noahmp_hip_malloc(size_t bytes) {
  if (ensure_init()) return nullptr;
  void* p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { g.last_error = "noahmp_hip_malloc: hipMalloc failed"; return nullptr; }
  return p;
}
int noahmp_hip_memcpy(void* dst, const void* src, size_t bytes, int kind) {
  int rc = ensure_init();
  if (rc) return rc;
  if (!bytes) return 0;
  if (kind == 2) { HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice)); return 0; }
  // host memory of the caller: through the engine's staging unless it is page-locked (nmp_stage.hpp)
  rc = kind == 0 ? nmp_host::copy_h2d(dst, src, bytes, g.own_stream) : nmp_host::copy_d2h(dst, src, bytes, g.own_stream);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(g.own_stream));
  return 0;
}
void noahmp_hip_free(void* p) { if (p) hipFree(p); }

static void drop_host_regs();

// page-locked registrations of caller arrays that are alive right now ("pin_host_arrays"): the test suite asserts 0 after every test
int noahmp_hip_debug_live_host_registrations(void) {
  int n = 0;
  for (auto& kv : g.host_regs) n += kv.second.state == 1;
  return n;
}

int noahmp_hip_set_tables(const noahmp_tables* t) {
  int rc = ensure_init();
  if (rc) return rc;
  drop_host_regs();           // a new set of tables = a new run of the caller: nothing known about its arrays carries over
  g.out_mirror_valid = false;
  // the ABI struct followed by the per-type constants derived from it (Derived, nmp_dev_common.hpp), evaluated here on the host
  static nmp::TablesDev img;
  img.t = *t;
  nmp::derive_tables(img.t, img.d);
  if (!g.d_tables) HIPCHK(hipMalloc((void**)&g.d_tables, sizeof(nmp::TablesDev)));
  HIPCHK(hipMemcpy(g.d_tables, &img, sizeof(nmp::TablesDev), hipMemcpyHostToDevice));
  nmp::tab_scalars_pack(nmp::tab_scalars(img.t), g.ts_i, g.ts_f);      // the tables' scalars travel as kernel arguments (Ctx::ts)
  g.have_tables = true;
  return 0;
}

// set_option() has no channel for a status: a fatal column that a mode switch collected stays in g.deferred_code -- the next
// noahmp_hip_step / noahmp_hip_fetch returns it -- and is named in last_error
static void note_fetch(int rc) {
  if (rc > 0) {
    g.deferred_code = rc;
    g.last_error = std::string("a fatal column of the previous (deferred) step is pending: ") + noahmp_hip_error_string(rc);
  }
}

int noahmp_hip_set_option(const char* key, int value) {
  int prev = -1;
  if (!strcmp(key, "block")) { prev = g.block; if (value == 64 || value == 128 || value == 256) g.block = value; }
  else if (!strcmp(key, "lds")) { prev = g.use_lds; g.use_lds = value ? 1 : 0; }
  else if (!strcmp(key, "host_chunks")) {        // -2 (or -1): the engine picks by tile size (the default); a query then returns -2 -- not -1, which means "unknown option"
    prev = g.host_chunks_auto ? -2 : g.host_chunks;
    if (value == -1 || value == -2) g.host_chunks_auto = true;
    else if (value >= 0 && value <= 32) { g.host_chunks = value; g.host_chunks_auto = false; }
  }
  else if (!strcmp(key, "pin_host_arrays")) {
    prev = g.pin_host_arrays;
    if (value == 0 || value == 1) {
      g.pin_host_arrays = value;
      if (!value) drop_host_regs();               // leaving the mode: drop every registration (the arrays may be freed now)
    }
  }
  else if (!strcmp(key, "trust_out_mirror")) {
    prev = g.trust_out_mirror;
    if (value == 0 || value == 1) { g.trust_out_mirror = value; g.out_mirror_valid = false; }
  }
  else if (!strcmp(key, "jit_option_kernels")) { prev = g.jit_kernels; if (value == 0 || value == 1) g.jit_kernels = value; }
  else if (!strcmp(key, "jit_compile_only")) { prev = g.jit_compile_only; if (value == 0 || value == 1) g.jit_compile_only = value; }
  else if (!strcmp(key, "fixed_option_kernels")) { prev = g.fixed_kernels; if (value == 0 || value == 1) g.fixed_kernels = value; }
  else if (!strcmp(key, "sorted_land_columns")) { prev = (int)g.sorted_land; g.sorted_land = value; }
  else if (!strcmp(key, "sorted_glacier_columns")) { prev = (int)g.sorted_glacier; g.sorted_glacier = value; }
  else if (!strcmp(key, "resident_state")) {
    prev = g.resident_state;
    if (value == 0 || value == 1) {
      if (!value && (g.resident_dirty || g.deferred_pending)) note_fetch(noahmp_hip_fetch(nullptr));     // leaving the mode: bring the host arrays up to date
      g.resident_state = value; g.resident_valid = false;
    }
  }
  else if (!strcmp(key, "resident_sorted")) {
    prev = g.resident_sorted;
    if (value == 0 || value == 1) {
      if (value != g.resident_sorted && (g.resident_dirty || g.deferred_pending)) note_fetch(noahmp_hip_fetch(nullptr));
      if (value != g.resident_sorted) { g.resident_sorted = value; g.sorted_ok = false; g.resident_valid = false; }
    }
  }
  else if (!strcmp(key, "lazy_download")) {
    prev = g.lazy_download;
    if (value == 0 || value == 1) {
      if (!value && (g.resident_dirty || g.deferred_pending)) note_fetch(noahmp_hip_fetch(nullptr));
      g.lazy_download = value;
    }
  }
  else if (!strcmp(key, "overlap_class_kernels")) prev = 1;          // (rounds 2-5: a second stream for the small class kernels; one launch now)
  else if (!strcmp(key, "static_inputs")) {
    prev = g.static_inputs;
    if ((value == 0 || value == 1) && value != g.static_inputs) {
      // the declaration decides which IN arrays travel per call and whether the sorted set may exist (XLAND / XICE / IVGTYP place a
      // column in its class range): bring the host arrays up to date and rebuild both mirror sets at the next call
      if (g.resident_dirty || g.deferred_pending) note_fetch(noahmp_hip_fetch(nullptr));
      g.static_inputs = value; g.sorted_ok = false; g.sorted_newer = false; g.resident_valid = false;
    }
  }
  else if (!strcmp(key, "deferred_status")) {
    prev = g.deferred_status;
    if (value == 0 || value == 1) {
      if (g.deferred_pending || g.resident_dirty) note_fetch(noahmp_hip_fetch(nullptr));
      g.deferred_status = value; g.resident_valid = false;
    }
  }
  else if (!strcmp(key, "record_cost")) {
#ifdef NMP_COST_RECORD                            // (an experiment build, tools/build_variants.py cost=-DNMP_COST_RECORD; otherwise -1: unknown option)
    prev = g.record_cost;
    if (value == 0 || value == 1) { g.record_cost = value; if (!value) g.cost_fresh = false; }
#endif
  }
  else if (!strcmp(key, "exact_libm")) prev = NMP_EXACT_LIBM;   // read-only: how this library was built
  return prev;
}

}  // extern "C" (helpers follow)

// launch-uniform kernel arguments of one noahmplsm call
static void fill_kargs(KArgs& k, const noahmp_step_args* a) {
  memset(&k, 0, sizeof(k));
  k.a = *a;
  k.ni = a->ime - a->ims + 1;
  k.nka = a->kme - a->kms + 1;
  k.nti = a->ite - a->its + 1;
  k.ntj = a->jte - a->jts + 1;
  k.k1 = 1 - a->kms;
  k.kp_lo = a->kts - a->kms;
  k.kp_hi = a->kts + 1 - a->kms;
  k.yearlen = 365;                                                                 // drv:381-390
  if (a->yr % 4 == 0) { k.yearlen = 366; if (a->yr % 100 == 0) { k.yearlen = 365; if (a->yr % 400 == 0) k.yearlen = 366; } }
  k.c.T = g.d_tables;
  k.c.D = derived_of(g.d_tables);
  k.c.O = Opt{a->idveg, a->iopt_crs, a->iopt_btr, a->iopt_run, a->iopt_sfc, a->iopt_frz, a->iopt_inf,
              a->iopt_rad, a->iopt_alb, a->iopt_snf, a->iopt_tbot, a->iopt_stc};
  k.c.dt = a->dt;
  k.c.isurban = a->isurban;
  k.c.ts = tab_scalars_unpack(g.ts_i, g.ts_f);
  k.c.zsoil[L(1)] = -a->dzs[0];                                                    // drv:392-395
  for (int l = 2; l <= NOAHMP_NSOIL; l++) k.c.zsoil[L(l)] = -a->dzs[l - 1] + k.c.zsoil[L(l - 1)];
  ctx_fill_uniform(k.c);
  k.err = g.d_err;
  k.counts = g.d_counts;
  k.a.dzs = nullptr;
}

// the option-specialised translation units (noahmp_engine_d*_r*.hip): (DVEG, RUN) with the other ten options at the namelist values
#ifndef NMP_NO_FIXED_KERNELS
struct FixedKernel { int dveg, run; void (*launch)(const nmp_host::LaunchDesc&, int, hipStream_t, hipEvent_t, hipEvent_t); };
static const FixedKernel kFixed[] = {{1, 1, nmp_host::launch_fixed_d1_r1}, {3, 1, nmp_host::launch_fixed_d3_r1},
                                     {3, 5, nmp_host::launch_fixed_d3_r5}, {4, 1, nmp_host::launch_fixed_d4_r1},
                                     {4, 3, nmp_host::launch_fixed_d4_r3}};      // (4, 3): the WRF defaults
#endif
// 0: generic kernel; n > 0: kFixed[n-1] can serve this call
static int fixed_level(const KArgs& k) {
#ifdef NMP_NO_FIXED_KERNELS
  return 0;
#else
  const Opt& o = k.c.O;
  if (!g.fixed_kernels) return 0;
  if (noahmp_hip_index_width(k.ni, k.a.jme - k.a.jms + 1, k.nka) != 32) return 0;   // the specialised kernels use 32-bit byte offsets
  if (!(o.crs == 1 && o.btr == 1 && o.sfc == 1 && o.frz == 1 && o.inf == 1 && o.rad == 3 && o.alb == 2 &&
        o.snf == 1 && o.tbot == 2 && o.stc == 1)) return g.jit_kernels ? -1 : 0;
  for (int n = 0; n < (int)(sizeof(kFixed) / sizeof(kFixed[0])); n++)
    if (kFixed[n].dveg == o.dveg && kFixed[n].run == o.run) return n + 1;
  return g.jit_kernels ? -1 : 0;
#endif
}

// level > 0: ahead-of-time kernel kFixed[level-1]; level < 0: compile one for this option set at run time.  false: not available.
static bool launch_fixed(const KArgs& k, int level, int mode, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {
#ifdef NMP_NO_FIXED_KERNELS
  return false;
#else
  nmp_host::LaunchDesc d;
  memset(&d, 0, sizeof(d));
  d.a = k.a; d.tables = k.c.T; d.dt = k.c.dt; d.isurban = k.c.isurban;
  for (int n = 0; n < 6; n++) d.ts_i[n] = g.ts_i[n];
  for (int n = 0; n < 15; n++) d.ts_f[n] = g.ts_f[n];
  for (int l = 0; l < NL; l++) d.zsoil[l] = k.c.zsoil[l];
  d.ni = k.ni; d.nka = k.nka; d.nti = k.nti; d.ntj = k.ntj; d.k1 = k.k1; d.kp_lo = k.kp_lo; d.kp_hi = k.kp_hi; d.yearlen = k.yearlen;
  d.err = k.err; d.counts = k.counts; d.err_base = k.err_base; d.t_offset = k.t_offset; d.t_first = k.t_first; d.t_count = k.t_count;
  d.cost = k.c.cost;
  d.r_land = k.r_land; d.r_ice = k.r_ice; d.r_skip = k.r_skip;
  if (level > 0) { kFixed[level - 1].launch(d, mode, s, ev0, ev1); return true; }
  const Opt& o = k.c.O;
  const int opts[12] = {o.dveg, o.crs, o.btr, o.run, o.sfc, o.frz, o.inf, o.rad, o.alb, o.snf, o.tbot, o.stc};
  return nmp_host::launch_jit(opts, d, mode, s, ev0, ev1);
#endif
}

// The three class ranges of a sorted layout in one launch (noahmp_ranges_kernel, nmp_kernel.hpp): option-specialised ahead of time, at
// run time, or generic.  ev0 / ev1: the kernel's own start / stop events (attached to the dispatch: no packet of their own) or NULL.
static void launch_ranges(KArgs k, long n_land, long n_ice, long n_skip, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
  constexpr long B = NMP_FIXED_BLOCK;
  k.r_land = n_land; k.r_ice = n_ice; k.r_skip = n_skip;
  k.t_first = 0; k.t_count = n_land + n_ice + n_skip;
  const long nb_ice = (n_ice + B - 1) / B, nb = nb_ice + (n_land + B - 1) / B + (n_skip + B - 1) / B;
  if (nb <= 0) return;
  // "record_cost": the kernel indexes the record by its global thread number, and the land range's workgroups follow the land-ice ones
  if (k.c.cost) k.c.cost -= 2 * nb_ice * B;
  const int fx = fixed_level(k);
  if (fx && launch_fixed(k, fx, 4, s, ev0, ev1)) return;
  hipExtLaunchKernelGGL((noahmp_ranges_kernel<NMP_FIXED_BLOCK>), dim3((unsigned)nb), dim3(NMP_FIXED_BLOCK), 0, s, ev0, ev1, 0, k);
}

// class_ranges: the call is a whole device-resident tile, the only kind of call the declared class ranges can describe
// ev: the step's timing events (noahmp_hip_sync_timing): [0] start, [1] end of the step's column kernel ([2]: unused since round 6).
// kind 1 (class ranges, one launch): the events are the dispatch's own start / stop events -- an event RECORDED on the stream is one more
// packet between two kernels (~5 us each on this chip); kind 0 (tile order): recorded around the launch; -1: an empty tile, none.
static void launch_any(const KArgs& k_in, hipStream_t s, bool class_ranges = false, hipEvent_t* ev = nullptr) {
  KArgs k = k_in;
  const long ncol = (long)k.nti * k.ntj;
  g.last_launch_kind = 0;
  if (ncol <= 0) { g.last_launch_kind = -1; return; }          // an empty tile: nothing is enqueued, no event is recorded (kind -1: no times)
  // "record_cost": the land columns of a device-resident tile leave their two trip counts in an engine-owned plane (Ctx::cost), in the
  // tile's CURRENT column order; noahmp_hip_sort_columns(NOAHMP_SORT_COST) reads it.  Not for row chunks of the host path.
  k.c.cost = nullptr;
  if (g.record_cost && class_ranges && k.t_offset == 0) {
    if (nmp_host::ensure_bytes((void**)&g.d_cost, &g.d_cost_bytes, (size_t)ncol * 2) == 0) {
      if (g.cost_cols != ncol) (void)hipMemsetAsync(g.d_cost, 0, (size_t)ncol * 2, s);
      k.c.cost = g.d_cost; g.cost_cols = ncol; g.cost_fresh = true;
    }
  }
  // class-sorted layout whose ranges the caller declared ("sorted_land_columns", "sorted_glacier_columns"): ONE launch, every workgroup
  // runs the code of its class only (rounds 2-5: three kernels, the small ones on a second stream beside the land kernel -- a fork and a
  // join event per step; the kernel trace of round 6 showed ~18 us per step in which those packets kept the queue idle)
  if (class_ranges && g.sorted_land >= 0 && g.sorted_glacier >= 0 && g.sorted_land + g.sorted_glacier <= ncol && k.t_offset == 0 &&
      g.block == 256 && g.use_lds) {
    g.last_launch_kind = 1;
    launch_ranges(k, g.sorted_land, g.sorted_glacier, ncol - g.sorted_land - g.sorted_glacier, s, ev ? ev[0] : nullptr, ev ? ev[1] : nullptr);
    return;
  }
  if (ev) hipEventRecord(ev[0], s);
  struct EndEvent { hipEvent_t* e; hipStream_t s; ~EndEvent() { if (e) hipEventRecord(e[1], s); } } at_end{ev, s};
  if (g.block == 256 && g.use_lds) {                                  // the option-specialised mixed-class kernel (tile order)
    const int fx = fixed_level(k);
    if (fx) { k.t_first = 0; k.t_count = ncol; if (launch_fixed(k, fx, 0, s)) return; }
  }
  if (g.block == 256) launch<256>(k, ncol, g.use_lds, s);
  else if (g.block == 128) launch<128>(k, ncol, g.use_lds, s);
  else launch<64>(k, ncol, g.use_lds, s);
}

static int check_step_args(const noahmp_step_args* a, noahmp_status* st) {
  if (!g.have_tables) { g.last_error = "noahmp_hip_set_tables() has not been called"; return -102; }
  if (a->nsoil != NOAHMP_NSOIL) { if (st) st->code = NOAHMP_ERR_NSOIL_UNSUPPORTED; return NOAHMP_ERR_NSOIL_UNSUPPORTED; }
  if (a->iopt_sfc != 1 && a->iopt_sfc != 2) {
    if (st) st->code = NOAHMP_ERR_OPT_SFC_UNSUPPORTED;
    return NOAHMP_ERR_OPT_SFC_UNSUPPORTED;
  }
  return 0;
}

// hipHostRegister an array of the caller the second time it shows up at the same address (opt-in: the caller guarantees
// that such arrays outlive the engine or calls noahmp_hip_finalize() first).  A registration is dropped as soon as the engine can
// tell that it is stale: an array of another size at the same address, an array at another address that OVERLAPS a known range
// (two live arrays never overlap, so the known one was freed and its memory reused), noahmp_hip_set_tables, leaving the mode,
// noahmp_hip_finalize.  A stale registration of freed memory is dangerous beyond the engine: the runtime treats whatever the process
// maps there next as page-locked.
static void drop_host_regs() {
  bool any = false;
  for (auto& kv : g.host_regs) any = any || kv.second.state == 1;
  if (any) hipDeviceSynchronize();
  for (auto& kv : g.host_regs) if (kv.second.state == 1) hipHostUnregister(const_cast<void*>(kv.first));
  g.host_regs.clear();
  (void)hipGetLastError();
}
static void maybe_pin(const void* host, size_t bytes) {
  if (!g.pin_host_arrays || !host || !bytes) return;
  const char* lo = (const char*)host; const char* hi = lo + bytes;
  for (auto it = g.host_regs.begin(); it != g.host_regs.end();) {           // ~120 entries at most
    const char* olo = (const char*)it->first; const char* ohi = olo + it->second.bytes;
    if (it->first != host && olo < hi && lo < ohi) {
      if (it->second.state == 1) { hipDeviceSynchronize(); hipHostUnregister(const_cast<void*>(it->first)); }
      it = g.host_regs.erase(it);
    } else ++it;
  }
  auto it = g.host_regs.find(host);
  if (it == g.host_regs.end()) { g.host_regs[host] = nmp_host::Engine::HostReg{bytes, 1, 0}; (void)hipGetLastError(); return; }
  nmp_host::Engine::HostReg& r = it->second;
  if (r.bytes != bytes) {
    if (r.state == 1) { hipDeviceSynchronize(); hipHostUnregister(const_cast<void*>(host)); }
    r = nmp_host::Engine::HostReg{bytes, 1, 0};
    (void)hipGetLastError();
    return;
  }
  r.seen++;
  if (r.state == 0 && r.seen >= 2)
    r.state = hipHostRegister(const_cast<void*>(host), bytes, hipHostRegisterDefault) == hipSuccess ? 1 : -1;
  (void)hipGetLastError();
}

// Host-memory path for large tiles: the tile is advanced in row chunks, chunk c+1 uploading while chunk c computes and
// chunk c-1 downloads (three streams).  Results are those of the single launch: columns are independent, the tallies
// accumulate, and the error word orders columns by their index in the whole tile (KArgs::t_offset).
static int pipeline_chunks(const noahmp_step_args* a) {
  if (!g.host_chunks_auto) return g.host_chunks;
  const long ncol = (long)(a->ite - a->its + 1) * (a->jte - a->jts + 1);
  const long n = (ncol + 600000) / 1200000;
  return (int)(n < 3 ? 3 : (n > 8 ? 8 : n));
}
static int step_host_pipelined(const noahmp_step_args* a, hipStream_t s, noahmp_status* st) {
  KArgs k;
  fill_kargs(k, a);
  const int nj_mem = a->jme - a->jms + 1;
  if (!g.s_up) { HIPCHK(hipStreamCreateWithFlags(&g.s_up, hipStreamNonBlocking)); HIPCHK(hipStreamCreateWithFlags(&g.s_dn, hipStreamNonBlocking)); }
  const int want = pipeline_chunks(a);
  const int nreg = want < nj_mem ? want : nj_mem;
  // row boundaries of the chunks.  The download side is the longer one (119 words per column down, 87 up) and cannot start before the first
  // chunk is up and computed: the first regular chunk is split 1 : 3, so that the downloads start four times earlier
  std::vector<int> rows;
  rows.push_back(0);
  for (int c = 0; c < nreg; c++) {
    const int r0 = (int)((long)nj_mem * c / nreg), r1 = (int)((long)nj_mem * (c + 1) / nreg);
    if (c == 0 && g.host_chunks_auto && r1 - r0 >= 8) rows.push_back(r0 + (r1 - r0) / 4);
    rows.push_back(r1);
  }
  const int nchunk = (int)rows.size() - 1;
  while ((int)g.pipe_events.size() < 3 * nchunk) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); g.pipe_events.push_back(e); }
  if (g.pipe_host.size() != (size_t)kNumFields) { g.pipe_host.assign(kNumFields, nullptr); g.out_mirror_valid = false; }
  {                                              // ... and about their extents: another memory shape = another layout of the mirrors
    const int ext[9] = {a->ims, a->ime, a->jms, a->jme, a->kms, a->kme, a->nsoil, 0, 0};
    if (memcmp(ext, g.pipe_extents, sizeof ext)) { memcpy(g.pipe_extents, ext, sizeof ext); g.out_mirror_valid = false; }
  }
  for (int f = 0; f < kNumFields; f++) {       // "trust_out_mirror" speaks about the arrays of the previous call only
    const void* host = *(void* const*)((const char*)a + kFields[f].off);
    if (g.pipe_host[f] != host) { g.pipe_host[f] = host; g.out_mirror_valid = false; }
  }
  const bool upload_out = !(g.trust_out_mirror && g.out_mirror_valid);
  size_t rowbytes[kNumFields];
  for (int f = 0; f < kNumFields; f++) {
    const FieldDesc& fd = kFields[f];
    const size_t bytes = field_elems(fd, a) * 4;
    rowbytes[f] = bytes / nj_mem;
    if (g.mirror_bytes[f] < bytes) {
      if (g.mirror[f]) HIPCHK(hipFree(g.mirror[f]));
      HIPCHK(hipMalloc(&g.mirror[f], bytes));
      g.mirror_bytes[f] = bytes;
      g.out_mirror_valid = false;
    }
    *(void**)((char*)&k.a + fd.off) = g.mirror[f];
  }
  const bool up_out = upload_out || !g.out_mirror_valid;
  *g.h_err = ~0ULL;
  HIPCHK(hipMemsetAsync(g.d_err, 0xFF, sizeof(unsigned long long), s));
  HIPCHK(hipMemsetAsync(g.d_counts, 0, kCountSlots * kCountStride * sizeof(int), s));
  // uploads, in row order
  for (int c = 0; c < nchunk; c++) {
    const int r0 = rows[c], r1 = rows[c + 1];
    for (int f = 0; f < kNumFields; f++) {
      const FieldDesc& fd = kFields[f];
      if (fd.io == 2 && !up_out) continue;
      const char* host = (const char*)*(void* const*)((const char*)a + fd.off);
      HIPCHK(hipMemcpyAsync((char*)g.mirror[f] + rowbytes[f] * r0, host + rowbytes[f] * r0, rowbytes[f] * (r1 - r0),
                                       hipMemcpyHostToDevice, g.s_up));
    }
    HIPCHK(hipEventRecord(g.pipe_events[3 * c], g.s_up));
  }
  // kernels and downloads
  for (int c = 0; c < nchunk; c++) {
    const int r0 = rows[c], r1 = rows[c + 1];
    HIPCHK(hipStreamWaitEvent(s, g.pipe_events[3 * c], 0));
    KArgs kc = k;
    const int j0 = a->jms + r0 > a->jts ? a->jms + r0 : a->jts;
    const int j1 = a->jms + r1 - 1 < a->jte ? a->jms + r1 - 1 : a->jte;
    HIPCHK(hipEventRecord(g.pipe_events[3 * c + 1], s));
    if (j1 >= j0) {
      kc.a.jts = j0; kc.a.jte = j1;
      kc.ntj = j1 - j0 + 1;
      kc.t_offset = (long)(j0 - a->jts) * k.nti;
      launch_any(kc, s);
      HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(g.pipe_events[3 * c + 2], s));
    HIPCHK(hipStreamWaitEvent(g.s_dn, g.pipe_events[3 * c + 2], 0));
    for (int f = 0; f < kNumFields; f++) {
      const FieldDesc& fd = kFields[f];
      if (fd.io == 0) continue;
      char* host = (char*)*(void* const*)((const char*)a + fd.off);
      HIPCHK(hipMemcpyAsync(host + rowbytes[f] * r0, (char*)g.mirror[f] + rowbytes[f] * r0, rowbytes[f] * (r1 - r0),
                                       hipMemcpyDeviceToHost, g.s_dn));
    }
  }
  HIPCHK(hipMemcpyAsync(g.h_err, g.d_err, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(g.h_counts, g.d_counts, kCountSlots * kCountStride * sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipStreamSynchronize(g.s_dn));
  g.out_mirror_valid = true;
  int code = 0;
  if (st) {
    float ms = 0.f;
    for (int c = 0; c < nchunk; c++) { float one = 0.f; hipEventElapsedTime(&one, g.pipe_events[3 * c + 1], g.pipe_events[3 * c + 2]); ms += one; }
    st->kernel_ms = ms;
    nmp_host::status_counts(st);
  }
  if (*g.h_err != ~0ULL) {
    code = (int)(*g.h_err & 0xFF);
    const long t = (long)(*g.h_err >> 8) - 1;
    if (st) { st->code = code; st->i = a->its + (int)(t % k.nti); st->j = a->jts + (int)(t / k.nti); }
  }
  return code;
}

// Resident host path ("resident_state" = 1): the caller's arrays are host arrays, but between calls the STATE lives in the device
// mirrors.  The first call (or any call with other arrays / extents) uploads everything; later calls upload only the IN
// arrays (forcing and static fields) and, with "lazy_download" = 1, copy nothing back: noahmp_hip_fetch() does that when the
// caller needs the arrays (output / restart times, hdrv:440-441, 588).  The caller promises not to modify INOUT / OUT
// arrays in between without switching the option off and on again.
// IN arrays a driver sets once (hdrv:250-300: static fields and DZ8W = 2 ZLVL): with "static_inputs" they are uploaded only when the
// resident state is (re)built.  What the HRLDAS loop rewrites between calls (hdrv:331-415) -- forcing, COSZEN, VEGFRA -- always travels.
static bool is_static_in(const FieldDesc& fd) {
  static const char* const names[] = {"xlatin", "ivgtyp", "isltyp", "vegmax", "tmn", "xland", "xice", "dz8w"};
  for (const char* n : names) if (!strcmp(fd.name, n)) return true;
  return false;
}
// atmospheric arrays of which the kernel reads level 1 only (drv:449-466; P8W3D is read at two levels): the resident path uploads that level
static bool level1_only(const FieldDesc& fd) {
  static const char* const names[] = {"t3d", "qv3d", "u_phy", "v_phy", "dz8w"};
  for (const char* n : names) if (!strcmp(fd.name, n)) return true;
  return false;
}

static void fill_status(const noahmp_step_args* a, int nti, noahmp_status* st, int* code_out) {
  float ms = 0.f;
  if (g.ev_timed) hipEventElapsedTime(&ms, g.ev0, g.ev1);
  int code = 0;
  if (st) {
    st->kernel_ms = ms;
    nmp_host::status_counts(st);
  }
  if (*g.h_err != ~0ULL) {
    code = (int)(*g.h_err & 0xFF);
    long t = (long)(*g.h_err >> 8) - 1;
    if (g.last_step_sorted && g.s_perm && t >= 0 && (size_t)t < g.s_cols) {     // "resident_sorted": a sorted position -> the tile column
      int p = 0;
      if (hipMemcpy(&p, g.s_perm + t, sizeof p, hipMemcpyDeviceToHost) == hipSuccess) t = p;
    }
    if (st) { st->code = code; st->i = a->its + (int)(t % nti); st->j = a->jts + (int)(t / nti); }
  }
  *code_out = code;
}

// ---- "resident_sorted": helpers.  `tile` / `sorted`: the per-field device pointers of the two mirror sets.
static int nlev_of(const FieldDesc& fd, const noahmp_step_args* a) {
  const size_t plane = (size_t)(a->ime - a->ims + 1) * (a->jme - a->jms + 1);
  return (int)(field_elems(fd, a) / plane);
}
// (and "static_inputs": XLAND / XICE / IVGTYP decide a column's class range; a caller that may rewrite them between calls keeps tile order)
static bool resident_sorted_possible(const noahmp_step_args* a) {      // the column sort needs memory block == tile, 32-bit offsets
  // (kms == 1: the incremental exchange of level-1-only forcing arrays moves memory level 0, which is level 1 only then)
  return g.resident_sorted && g.static_inputs && a->kms == 1 && a->ims == a->its && a->ime == a->ite && a->jms == a->jts && a->jme == a->jte &&
         (long)(a->ite - a->its + 1) * (a->jte - a->jts + 1) > 0 && (long)(a->ite - a->its + 1) * (a->jte - a->jts + 1) < 0x7FFFFFFFL &&
         a->jme - a->jms + 1 <= 65535;
}
// INOUT / OUT arrays: sorted set -> tile-order mirrors (before a download)
static int sorted_to_tile(const noahmp_step_args* a, hipStream_t s) {
  const int ni = a->ime - a->ims + 1, nj = a->jme - a->jms + 1;
  void* sp[32]; void* tp[32]; int nl[32];
  int n = 0;
  for (int f = 0; f < kNumFields; f++) {
    const FieldDesc& fd = kFields[f];
    if (fd.io != 0) { sp[n] = g.smirror[f]; tp[n] = g.mirror[f]; nl[n] = nlev_of(fd, a); n++; }
    if (n == 32 || (f == kNumFields - 1 && n)) {
      int rc = noahmp_hip_sorted_exchange(n, sp, tp, nl, g.s_order, g.s_dpos, ni, nj, ni, 0, 0, 1, s);
      if (rc) return rc;
      n = 0;
    }
  }
  g.sorted_newer = false;
  return 0;
}
// (re)build the sorted set from the tile-order set `tile_args` (device pointers of every field, the IN arrays at their current targets)
static int build_sorted(const noahmp_step_args* a, const noahmp_step_args& tile_args, hipStream_t s) {
  const int ni = a->ime - a->ims + 1, nj = a->jme - a->jms + 1;
  const size_t ncol = (size_t)ni * nj;
  if (g.smirror.empty()) { g.smirror.assign(kNumFields, nullptr); g.smirror_bytes.assign(kNumFields, 0); }
  for (int f = 0; f < kNumFields; f++) {
    const size_t bytes = field_elems(kFields[f], a) * 4;
    if (g.smirror_bytes[f] != bytes) {
      if (g.smirror[f]) HIPCHK(hipFree(g.smirror[f]));
      g.smirror[f] = nullptr; g.smirror_bytes[f] = 0;
      HIPCHK(hipMalloc(&g.smirror[f], bytes));
      g.smirror_bytes[f] = bytes;
    }
  }
  if (g.s_cols != ncol) {
    if (g.s_perm) hipFree(g.s_perm);
    if (g.s_keys) hipFree(g.s_keys);
    if (g.s_order) hipFree(g.s_order);
    if (g.s_dpos) hipFree(g.s_dpos);
    g.s_perm = nullptr; g.s_keys = nullptr; g.s_order = nullptr; g.s_dpos = nullptr; g.s_cols = 0;
    HIPCHK(hipMalloc((void**)&g.s_perm, ncol * 4));
    HIPCHK(hipMalloc((void**)&g.s_keys, ncol * 4));
    HIPCHK(hipMalloc((void**)&g.s_order, ncol * 2));
    HIPCHK(hipMalloc((void**)&g.s_dpos, ncol * 4));
    g.s_cols = ncol;
  }
  noahmp_step_args sorted_args = tile_args;
  for (int f = 0; f < kNumFields; f++) *(void**)((char*)&sorted_args + kFields[f].off) = g.smirror[f];
  int64_t counts[3] = {0, 0, 0};
  int rc = noahmp_hip_sort_columns(&tile_args, NOAHMP_SORT_VEG | NOAHMP_SORT_SNOW, 1000, g.s_perm, g.s_keys, counts, s);   // waits for `s`
  if (rc) return rc;
  if ((rc = noahmp_hip_permute_step_arrays(&tile_args, &sorted_args, g.s_perm, s))) return rc;
  if ((rc = noahmp_hip_scatter_plan(g.s_perm, ni, nj, g.s_order, g.s_dpos, s))) return rc;
  g.s_land = (long)counts[0]; g.s_glacier = (long)counts[1];
  g.sorted_ok = true; g.sorted_newer = false; g.calls_since_sort = 0;
  return 0;
}


// "deferred_status": wait for the step the previous resident call left running and report it
// The code is also kept in g.deferred_code until a call has RETURNED it to the caller (take_deferred_code): paths that collect a
// pending step on the way to something else (set_option, a fetch with other arrays) must not drop a fatal column.
static int resident_collect(noahmp_status* st, int* code_out) {
  *code_out = g.deferred_code;
  if (!g.deferred_pending) return 0;
  HIPCHK(hipEventSynchronize(g.ev_kdone));
  g.deferred_pending = false;
  int code = 0;
  fill_status(&g.resident_args, g.resident_args.ite - g.resident_args.its + 1, st, &code);
  if (code && !g.deferred_code) g.deferred_code = code;
  *code_out = g.deferred_code;
  return 0;
}
static int take_deferred_code(int code) { if (code == g.deferred_code) g.deferred_code = 0; return code; }

static int step_host_resident(const noahmp_step_args* a, hipStream_t s, noahmp_status* st) {
  KArgs k;
  fill_kargs(k, a);
  if (g.mirror_host.empty()) g.mirror_host.assign(kNumFields, nullptr);
  if (g.mirror_b.empty()) { g.mirror_b.assign(kNumFields, nullptr); g.mirror_b_bytes.assign(kNumFields, 0); }
  const bool defer = g.deferred_status && g.lazy_download;
  bool valid = g.resident_valid;
  for (int f = 0; f < kNumFields; f++) {       // pass 1: are these the arrays (and extents) of the resident state?
    const FieldDesc& fd = kFields[f];
    if (g.mirror_bytes[f] != field_elems(fd, a) * 4 || g.mirror_host[f] != *(void* const*)((const char*)a + fd.off)) valid = false;
  }
  if (!valid && (g.resident_dirty || g.deferred_pending)) {   // other arrays than last time while results are still only on the device:
    int rc = noahmp_hip_fetch(nullptr);        // bring the PREVIOUS call's host arrays up to date before any mirror is touched
    if (rc) return rc;
  }
  for (int f = 0; f < kNumFields; f++) {       // pass 2: only now may the mirrors change size
    const size_t bytes = field_elems(kFields[f], a) * 4;
    if (g.mirror_bytes[f] != bytes) {
      if (g.mirror[f]) HIPCHK(hipFree(g.mirror[f]));
      g.mirror[f] = nullptr; g.mirror_bytes[f] = 0;
      HIPCHK(hipMalloc(&g.mirror[f], bytes));
      g.mirror_bytes[f] = bytes;
    }
    if (defer && kFields[f].io == 0 && g.mirror_b_bytes[f] != bytes) {      // second buffer of the IN arrays: step n+1 uploads while step n computes
      if (g.mirror_b[f]) HIPCHK(hipFree(g.mirror_b[f]));
      g.mirror_b[f] = nullptr; g.mirror_b_bytes[f] = 0;
      HIPCHK(hipMalloc(&g.mirror_b[f], bytes));
      g.mirror_b_bytes[f] = bytes;
      valid = false;
    }
  }
  if (defer && !g.s_up) { HIPCHK(hipStreamCreateWithFlags(&g.s_up, hipStreamNonBlocking)); HIPCHK(hipStreamCreateWithFlags(&g.s_dn, hipStreamNonBlocking)); }
  if (defer && !g.ev_up) { HIPCHK(hipEventCreate(&g.ev_up)); HIPCHK(hipEventCreate(&g.ev_kdone)); }
  hipStream_t up = defer ? g.s_up : s;
  const bool second = defer && (g.resident_calls & 1);
  const size_t ni = a->ime - a->ims + 1, nj = a->jme - a->jms + 1, nka = a->kme - a->kms + 1;
  std::vector<nmp_host::CopySeg> up_segs;
  for (int f = 0; f < kNumFields; f++) {
    const FieldDesc& fd = kFields[f];
    const size_t bytes = field_elems(fd, a) * 4;
    void* host = *(void* const*)((const char*)a + fd.off);
    maybe_pin(host, bytes);
    const bool stat = fd.io == 0 && is_static_in(fd);
    // static IN arrays and the state live in `mirror`; the IN arrays that change per call alternate between the two buffers
    void* target = (second && fd.io == 0 && !(g.static_inputs && stat)) ? g.mirror_b[f] : g.mirror[f];
    if (!valid || (fd.io == 0 && !(g.static_inputs && stat))) {
      if (valid && fd.lev == 1 && nka > 1 && level1_only(fd) && nmp_host::host_page_locked(host, bytes)) {   // only the level the kernel reads
        const size_t off = (size_t)k.k1 * ni * 4;
        HIPCHK(hipMemcpy2DAsync((char*)target + off, nka * ni * 4, (const char*)host + off, nka * ni * 4, ni * 4, nj, hipMemcpyHostToDevice, up));
      } else {
        up_segs.push_back(nmp_host::CopySeg{host, target, bytes});        // page-locked: direct; pageable: the engine's bounce buffers
      }
    }
    g.mirror_host[f] = host;
    *(void**)((char*)&k.a + fd.off) = target;
  }
  if (!up_segs.empty()) { int rc = nmp_host::copy_segments(up_segs.data(), (int)up_segs.size(), true, up); if (rc) return rc; }
  g.out_mirror_valid = false;
  int prev_code = 0;
  if (defer) {
    HIPCHK(hipEventRecord(g.ev_up, up));
    int rc = resident_collect(st, &prev_code);      // the previous step ran under this call's upload: wait for it, report it
    if (rc) return rc;
    HIPCHK(hipStreamWaitEvent(s, g.ev_up, 0));
  }
  // "resident_sorted": the kernels run on a second set of mirrors in the north-star column order.  The tile-order set above stays the landing
  // place of the uploads; what this call uploaded is permuted into the sorted set (one launch), the whole state when it is (re)built
  // or has not been sorted for 24 calls (snow layers appear and vanish).
  const bool run_sorted = resident_sorted_possible(a);
  if (run_sorted) {
    const noahmp_step_args tile_args = k.a;                   // device pointers of the tile-order set, the IN arrays at this call's targets
    if (!valid || !g.sorted_ok || g.calls_since_sort >= 24) {
      if (valid && g.sorted_ok && g.sorted_newer) { int rc = sorted_to_tile(a, s); if (rc) return rc; }
      int rc = build_sorted(a, tile_args, s);
      if (rc) return rc;
    } else {
      void* sp[32]; void* tp[32]; int nl[32];
      int n = 0;
      for (int f = 0; f < kNumFields; f++) {
        const FieldDesc& fd = kFields[f];
        if (fd.io == 0 && !(g.static_inputs && is_static_in(fd))) {
          sp[n] = g.smirror[f]; tp[n] = *(void* const*)((const char*)&tile_args + fd.off);
          nl[n] = nlev_of(fd, a);
          if (fd.lev == 1 && nka > 1 && level1_only(fd)) nl[n] = -nl[n];      // only the level that was uploaded (and that the kernel reads)
          n++;
        }
        if (n == 32 || (f == kNumFields - 1 && n)) {
          int rc = noahmp_hip_sorted_exchange(n, sp, tp, nl, g.s_order, g.s_dpos, (int)ni, (int)nj, (int)ni, 0, 0, 0, s);
          if (rc) return rc;
          n = 0;
        }
      }
    }
    for (int f = 0; f < kNumFields; f++) *(void**)((char*)&k.a + kFields[f].off) = g.smirror[f];
  } else if (g.sorted_ok) {
    // a call the sorted set cannot serve (a sub-tile of the same arrays, other extents): this kernel runs on the tile-order mirrors, so
    // they must hold the newest state first, and the sorted set is dropped -- it would be stale from here on (state AND static inputs)
    if (valid && g.sorted_newer) { int rc = sorted_to_tile(&g.resident_args, s); if (rc) return rc; }
    g.sorted_ok = false; g.sorted_newer = false;
  }
  *g.h_err = ~0ULL;
  HIPCHK(hipMemsetAsync(g.d_err, 0xFF, sizeof(unsigned long long), s));
  HIPCHK(hipMemsetAsync(g.d_counts, 0, kCountSlots * kCountStride * sizeof(int), s));
  g.ev_timed = (long)k.nti * k.ntj > 0;                // (an empty tile records no events: see noahmp_hip_step)
  if (g.ev_timed) HIPCHK(hipEventRecord(g.ev0, s));
  {
    const long sl = g.sorted_land, sg = g.sorted_glacier;     // the caller's own declaration (device-resident calls) is not touched
    if (run_sorted) { g.sorted_land = g.s_land; g.sorted_glacier = g.s_glacier; }
    launch_any(k, s, run_sorted);
    if (run_sorted) { g.sorted_land = sl; g.sorted_glacier = sg; g.sorted_newer = true; g.calls_since_sort++; }
    g.last_step_sorted = run_sorted;
  }
  HIPCHK(hipGetLastError());
  if (g.ev_timed) HIPCHK(hipEventRecord(g.ev1, s));
  HIPCHK(hipMemcpyAsync(g.h_err, g.d_err, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(g.h_counts, g.d_counts, kCountSlots * kCountStride * sizeof(int), hipMemcpyDeviceToHost, s));
  g.resident_calls++;
  g.resident_args = *a;
  g.resident_valid = true;
  g.resident_dirty = g.lazy_download != 0;
  if (defer) {                                   // return as soon as the caller may overwrite its forcing arrays; the kernel keeps running
    HIPCHK(hipEventRecord(g.ev_kdone, s));
    g.deferred_pending = true;
    HIPCHK(hipEventSynchronize(g.ev_up));
    return take_deferred_code(prev_code);
  }
  if (!g.lazy_download) {
    if (run_sorted) { int rc = sorted_to_tile(a, s); if (rc) return rc; }
    std::vector<nmp_host::CopySeg> down;
    for (int f = 0; f < kNumFields; f++) {
      const FieldDesc& fd = kFields[f];
      if (fd.io == 0) continue;
      down.push_back(nmp_host::CopySeg{*(void* const*)((const char*)a + fd.off), g.mirror[f], field_elems(fd, a) * 4});
    }
    int rc = nmp_host::copy_segments(down.data(), (int)down.size(), false, s);
    if (rc) return rc;
  }
  HIPCHK(hipStreamSynchronize(s));
  int code = 0;
  fill_status(a, k.nti, st, &code);
  return code;
}

extern "C" {

// Copy the INOUT and OUT arrays of the resident mirrors back into the host arrays of the last resident call (a = NULL) or
// of `a` (which must name the same arrays).  No-op when nothing is pending.
int noahmp_hip_fetch(const noahmp_step_args* a) {
  int rc = ensure_init();
  if (rc) return rc;
  if (!g.resident_valid) {
    if (g.resident_dirty) { g.last_error = "noahmp_hip_fetch: no resident state"; return -108; }
    return take_deferred_code(g.deferred_code);          // a code a mode switch collected (note_fetch)
  }
  int pending_code = 0;
  rc = resident_collect(nullptr, &pending_code);         // "deferred_status": the last step may still be running
  if (rc) return rc;
  const noahmp_step_args* r = &g.resident_args;
  if (a)
    for (int f = 0; f < kNumFields; f++)
      if (*(void* const*)((const char*)a + kFields[f].off) != g.mirror_host[f]) {
        g.last_error = "noahmp_hip_fetch: these are not the arrays of the resident state";
        return -108;
      }
  if (!g.resident_dirty) return take_deferred_code(pending_code);
  if (g.sorted_ok && g.sorted_newer) { rc = sorted_to_tile(r, g.own_stream); if (rc) return rc; }    // "resident_sorted": results back to tile order
  {
    std::vector<nmp_host::CopySeg> down;
    for (int f = 0; f < kNumFields; f++) {
      const FieldDesc& fd = kFields[f];
      if (fd.io == 0) continue;
      down.push_back(nmp_host::CopySeg{const_cast<void*>(g.mirror_host[f]), g.mirror[f], field_elems(fd, r) * 4});
    }
    rc = nmp_host::copy_segments(down.data(), (int)down.size(), false, g.own_stream);
    if (rc) return rc;
  }
  HIPCHK(hipStreamSynchronize(g.own_stream));
  g.resident_dirty = false;
  return take_deferred_code(pending_code);              // a fatal column of the step that was still running (0 = none)
}

int noahmp_hip_step(const noahmp_step_args* a, int mem, void* stream, noahmp_status* st) {
  if (st) memset(st, 0, sizeof(*st));
  int rc = ensure_init();
  if (rc) return rc;
  rc = check_step_args(a, st);
  if (rc) return rc;
  if (g.async_pending) { g.last_error = "noahmp_hip_step: asynchronous steps are pending, call noahmp_hip_sync() first"; return -106; }
  hipStream_t s = stream ? (hipStream_t)stream : g.own_stream;
  if (mem == NOAHMP_MEM_HOST && g.resident_state) return step_host_resident(a, s, st);
  if (g.deferred_pending || g.resident_dirty || g.deferred_code) {        // also for DEVICE memory: d_err / h_err are shared with the running step
    rc = noahmp_hip_fetch(nullptr);                 // leaving the resident path: host arrays up to date first
    if (rc < 0) return rc;
    if (rc > 0) { if (st) st->code = rc; g.last_error = "a fatal column of the previous (deferred) step"; return rc; }   // never drop a pending fatal
  }
  if (mem == NOAHMP_MEM_HOST) g.resident_valid = false;
  // "pin_host_arrays": page-lock what shows up a second time; the row-chunk pipeline needs EVERY array page-locked (its copies are
  // asynchronous DMAs out of / into the caller's memory) -- until then, and for pageable callers, the single-shot path below stages
  // through the engine's own bounce buffers (nmp_stage.hpp)
  bool all_locked = false;
  if (mem == NOAHMP_MEM_HOST && g.pin_host_arrays) {
    all_locked = true;
    for (int f = 0; f < kNumFields; f++) {
      const void* host = *(void* const*)((const char*)a + kFields[f].off);
      const size_t bytes = field_elems(kFields[f], a) * 4;
      maybe_pin(host, bytes);
      auto it = g.host_regs.find(host);
      if (bytes && !(it != g.host_regs.end() && it->second.state == 1)) all_locked = false;
    }
  }
  if (mem == NOAHMP_MEM_HOST && g.pin_host_arrays && all_locked && pipeline_chunks(a) > 1 &&
      (long)(a->ite - a->its + 1) * (a->jte - a->jts + 1) >= 32768 &&
      a->jme - a->jms + 1 >= 2 * pipeline_chunks(a))
    return step_host_pipelined(a, s, st);

  KArgs k;
  fill_kargs(k, a);

  if (mem == NOAHMP_MEM_HOST) {
    g.out_mirror_valid = false;
    // stage every array H2D into persistent device mirrors (caller's arrays stay the source of truth); pageable arrays travel through
    // the engine's own page-locked bounce buffers (nmp_stage.hpp), never through the runtime's pageable path
    std::vector<nmp_host::CopySeg> up;
    for (int f = 0; f < kNumFields; f++) {
      const FieldDesc& fd = kFields[f];
      size_t bytes = field_elems(fd, a) * 4;
      if (g.mirror_bytes[f] < bytes) {
        if (g.mirror[f]) HIPCHK(hipFree(g.mirror[f]));
        HIPCHK(hipMalloc(&g.mirror[f], bytes));
        g.mirror_bytes[f] = bytes;
      }
      void* host = *(void* const*)((const char*)a + fd.off);
      // OUT arrays are uploaded too: columns the call does not touch (open water, sea ice, cells
      // outside its:ite/jts:jte, a column that raised a fatal) must come back unchanged, exactly
      // as the reference leaves them.
      up.push_back(nmp_host::CopySeg{host, g.mirror[f], bytes});
      *(void**)((char*)&k.a + fd.off) = g.mirror[f];
    }
    rc = nmp_host::copy_segments(up.data(), (int)up.size(), true, s);
    if (rc) return rc;
  }

  *g.h_err = ~0ULL;
  HIPCHK(hipMemsetAsync(g.d_err, 0xFF, sizeof(unsigned long long), s));
  HIPCHK(hipMemsetAsync(g.d_counts, 0, kCountSlots * kCountStride * sizeof(int), s));
  const bool timed = (long)k.nti * k.ntj > 0;          // an empty tile: no kernel, no events (two records with nothing between them
  if (timed) HIPCHK(hipEventRecord(g.ev0, s));         // send hipEventElapsedTime down a marker-inserting path of the runtime)
  launch_any(k, s, mem == NOAHMP_MEM_DEVICE);
  HIPCHK(hipGetLastError());
  if (timed) HIPCHK(hipEventRecord(g.ev1, s));
  HIPCHK(hipMemcpyAsync(g.h_err, g.d_err, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(g.h_counts, g.d_counts, kCountSlots * kCountStride * sizeof(int), hipMemcpyDeviceToHost, s));

  if (mem == NOAHMP_MEM_HOST) {
    std::vector<nmp_host::CopySeg> down;
    for (int f = 0; f < kNumFields; f++) {
      const FieldDesc& fd = kFields[f];
      if (fd.io == 0) continue;
      down.push_back(nmp_host::CopySeg{*(void* const*)((const char*)a + fd.off), g.mirror[f], field_elems(fd, a) * 4});
    }
    rc = nmp_host::copy_segments(down.data(), (int)down.size(), false, s);
    if (rc) return rc;
  }
  HIPCHK(hipStreamSynchronize(s));
  float ms = 0.f;
  if (timed) hipEventElapsedTime(&ms, g.ev0, g.ev1);
  int code = 0;
  if (st) {
    st->kernel_ms = ms;
    nmp_host::status_counts(st);
  }
  if (*g.h_err != ~0ULL) {
    code = (int)(*g.h_err & 0xFF);
    long t = (long)(*g.h_err >> 8) - 1;
    if (st) { st->code = code; st->i = a->its + (int)(t % k.nti); st->j = a->jts + (int)(t / k.nti); }
  }
  return code;
}

// ---- asynchronous stepping for device-resident state: enqueue and return.  Fatal columns and tallies
// accumulate on the device until noahmp_hip_sync(); the error word carries the step ordinal above the column index,
// so the earliest step wins, then the first column in loop order -- the column the reference would have STOPped at.
int noahmp_hip_step_async(const noahmp_step_args* a, void* stream) {
  int rc = ensure_init();
  if (rc) return rc;
  rc = check_step_args(a, nullptr);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : g.own_stream;
  if (!g.async_pending) {
    HIPCHK(hipMemsetAsync(g.d_err, 0xFF, sizeof(unsigned long long), s));
    HIPCHK(hipMemsetAsync(g.d_counts, 0, kCountSlots * kCountStride * sizeof(int), s));
    g.async_nti = a->ite - a->its + 1; g.async_its = a->its; g.async_jts = a->jts;
  }
  KArgs k;
  fill_kargs(k, a);
  {
    // the device tallies are int32 per slot (a slot takes the workgroups b with b mod kCountSlots equal): refuse a step that could
    // wrap one before the next noahmp_hip_sync (7 M columns: ~77 000 pending steps); the sums over the slots are 64-bit
    // (noahmp_hip_sync_counts).  The step ordinal of the error word has 24 bits.
    const long long ncol = (long long)k.nti * k.ntj;
    const long long per_slot = ((ncol + 255) / 256 + kCountSlots - 1) / kCountSlots * 256;
    if ((long long)(g.async_pending + 1) * per_slot > 0x7FFFFFFFll || g.async_pending >= (1 << 24) - 1) {
      g.last_error = "noahmp_hip_step_async: too many pending steps for the device tallies, call noahmp_hip_sync() first";
      return -107;
    }
  }
  k.err_base = (unsigned long long)g.async_pending << 40;      // step ordinal since the last sync (columns < 2^32)
  // three events per step (launch_any): kernel_ms of noahmp_hip_sync is the sum of the column kernels' own durations, whatever
  // else the caller puts on the stream between them
  while ((int)g.async_events.size() < 3 * (g.async_pending + 1)) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    g.async_events.push_back(e);
  }
  launch_any(k, s, true, &g.async_events[3 * g.async_pending]);
  HIPCHK(hipGetLastError());
  if ((int)g.async_kind.size() <= g.async_pending) g.async_kind.resize(g.async_pending + 1);
  g.async_kind[g.async_pending] = (signed char)g.last_launch_kind;
  g.async_pending++;
  g.async_stream = s;
  bool known = false;
  for (hipStream_t q : g.async_streams) known = known || (q == s);
  if (!known) g.async_streams.push_back(s);
  return 0;
}

// Wait for the pending asynchronous steps.  st: tallies summed over them, kernel_ms = device time from the first to
// the last of them, code/i/j = the first fatal column (0 if none); returns that code.  st->n_skipped is reused for
// nothing else; the ordinal of the failing step (0-based since the previous sync) is returned through *step_out.
int noahmp_hip_sync(noahmp_status* st, int* step_out) {
  if (st) memset(st, 0, sizeof(*st));
  if (step_out) *step_out = -1;
  if (!g.async_pending) { g.sync_steps = 0; g.sync_step_ms.clear(); for (int c = 0; c < 3; c++) g.sync_class_ms[c] = 0.f; return 0; }
  hipStream_t s = g.async_stream;
  for (hipStream_t q : g.async_streams) if (q != s) HIPCHK(hipStreamSynchronize(q));   // steps may sit on several streams
  g.async_streams.clear();
  HIPCHK(hipMemcpyAsync(g.h_err, g.d_err, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(g.h_counts, g.d_counts, kCountSlots * kCountStride * sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  const int nsteps = g.async_pending;
  g.async_pending = 0;
  int code = 0;
  {                                         // per-step / per-class times of THIS sync, whether or not the caller takes the tallies
    float ms = 0.f;
    for (int c = 0; c < 3; c++) g.sync_class_ms[c] = 0.f;
    g.sync_step_ms.assign(nsteps, 0.f);
    for (int i = 0; i < nsteps; i++) {
      // kind 0: the tile-order kernel between two recorded events; 1: the class ranges in one launch, events 0 / 1 = the dispatch's own
      // start / stop.  Either way ONE kernel per step since round 6: out[1] (land ice + skipped beside the land kernel) stays 0.
      hipEvent_t* e = &g.async_events[3 * i];
      const int kind = i < (int)g.async_kind.size() ? g.async_kind[i] : 0;
      float land = 0.f;
      if (kind < 0) continue;                            // an empty tile: no kernel, no events
      hipEventElapsedTime(&land, e[0], e[1]);
      ms += land;
      g.sync_class_ms[0] += land;
      g.sync_step_ms[i] = land;
    }
    g.sync_steps = nsteps;
    if (st) {
      st->kernel_ms = ms;
      nmp_host::status_counts(st);
    }
  }
  if (*g.h_err != ~0ULL) {
    code = (int)(*g.h_err & 0xFF);
    const long t = (long)((*g.h_err >> 8) & 0xFFFFFFFFull) - 1;
    if (step_out) *step_out = (int)(*g.h_err >> 40);
    if (st) { st->code = code; st->i = g.async_its + (int)(t % g.async_nti); st->j = g.async_jts + (int)(t / g.async_nti); }
  }
  return code;
}

int noahmp_hip_stream_sync(void* stream) {
  int rc = ensure_init();
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(stream ? (hipStream_t)stream : g.own_stream));
  return 0;
}

// dst column p <- src column perm[p] for every array of the step block ((re-)sorting a device-resident run)
int noahmp_hip_permute_step_arrays(const noahmp_step_args* src, const noahmp_step_args* dst, const int32_t* perm, void* stream) {
  g.cost_fresh = false;          // the cost record ("record_cost") is in the order the last step found the tile in

  int rc = ensure_init();
  if (rc) return rc;
  if (src->ims != dst->ims || src->ime != dst->ime || src->jms != dst->jms || src->jme != dst->jme || src->nsoil != dst->nsoil ||
      src->kms != dst->kms || src->kme != dst->kme) {
    g.last_error = "noahmp_hip_permute_step_arrays: the two blocks have different extents";
    return -105;
  }
  const int ni = src->ime - src->ims + 1, nj = src->jme - src->jms + 1;
  void* d[32]; const void* s[32]; int nl[32];
  int n = 0;
  for (int f = 0; f < kNumFields; f++) {
    const FieldDesc& fd = kFields[f];
    s[n] = *(void* const*)((const char*)src + fd.off);
    d[n] = *(void* const*)((const char*)dst + fd.off);
    if (s[n] == d[n]) { g.last_error = std::string("noahmp_hip_permute_step_arrays: src and dst share array ") + fd.name; return -105; }
    nl[n] = (int)(field_elems(fd, src) / ((size_t)ni * nj));
    if (++n == 32 || f == kNumFields - 1) {
      rc = noahmp_hip_gather_fields(n, d, s, nl, perm, ni, nj, stream);
      if (rc) return rc;
      n = 0;
    }
  }
  return 0;
}

// Kernel time of the steps the last noahmp_hip_sync collected, by class range: out[0] = land kernel (or the mixed kernel of an
// unsorted tile), out[1] = land-ice kernel, out[2] = skipped-cell kernel [ms, summed over the steps]; returns the number of steps.
int noahmp_hip_sync_timing(float* out, int n) {
  for (int c = 0; c < n && c < 3; c++) out[c] = g.sync_class_ms[c];
  return g.sync_steps;
}

// 64-bit tallies of the last noahmp_hip_sync (or synchronous step): out[0..2] = land, land-ice, skipped columns summed over its
// steps (noahmp_status carries them as int32, saturated); returns the number of steps of that sync.
int noahmp_hip_sync_counts(int64_t* out, int n) {
  for (int c = 0; c < n && c < 3; c++) out[c] = (int64_t)g.last_counts[c];
  return g.sync_steps;
}

// The cost record of the last device-resident step ("record_cost"): two bytes per column of the tile in its current order (canopy-loop
// iterations, STOMATA bisection steps); returns the number of columns copied (0: nothing recorded), waits for `stream`.
long noahmp_hip_fetch_cost(uint8_t* host_out, long ncol, void* stream) {
  if (!g.d_cost || !g.cost_fresh || !host_out) return 0;
  const long n = ncol < g.cost_cols ? ncol : g.cost_cols;
  hipStream_t s = stream ? (hipStream_t)stream : g.own_stream;
  if (hipMemcpyAsync(host_out, g.d_cost, (size_t)n * 2, hipMemcpyDeviceToHost, s) != hipSuccess) return 0;
  if (hipStreamSynchronize(s) != hipSuccess) return 0;
  return n;
}

// The same per step: out[i] = land (or mixed) kernel time of step i of the last noahmp_hip_sync [ms]; returns the number of steps.
int noahmp_hip_sync_step_timing(float* out, int n) {
  for (int i = 0; i < n && i < (int)g.sync_step_ms.size(); i++) out[i] = g.sync_step_ms[i];
  return (int)g.sync_step_ms.size();
}

const char* noahmp_hip_error_string(int code) {
  switch (code) {
    case 0: return "ok";
    case NOAHMP_ERR_SOILTYP_RANGE: return "REDPRM: too many input soil types (lsm:9266)";
    case NOAHMP_ERR_VEGTYP_RANGE: return "REDPRM: too many input landuse types (lsm:9272)";
    case NOAHMP_ERR_NROOT_GT_NSOIL: return "REDPRM: too many root layers (lsm:9340)";
    case NOAHMP_ERR_DVEG_UNKNOWN: return "Namelist parameter DVEG unknown (lsm:842)";
    case NOAHMP_ERR_SW_BALANCE: return "Stop in Noah-MP: ERRSW (lsm:1185)";
    case NOAHMP_ERR_ENERGY_BALANCE: return "Energy budget problem in NOAHMP LSM (lsm:1196)";
    case NOAHMP_ERR_WATER_BALANCE: return "Water budget problem in NOAHMP LSM (lsm:1221)";
    case NOAHMP_ERR_FIRE_NONPOSITIVE: return "STOP in Noah-MP: emitted longwave <0 (lsm:1786)";
    case NOAHMP_ERR_HCAN_LE_ZPD: return "CRITICAL PROBLEM: HCAN <= ZPD (lsm:3289)";
    case NOAHMP_ERR_STABILITY_STOP: return "STOP in Noah-MP: ZLVL <= ZPD (lsm:4124)";
    case NOAHMP_ERR_OPT_SFC_UNSUPPORTED: return "OPT_SFC 3/4 unsupported: MYJ/YSU tables are never initialised offline";
    case NOAHMP_ERR_GLACIER_SW_BALANCE: return "glacier: ERRSW (gla:2939)";
    case NOAHMP_ERR_GLACIER_ENERGY_BALANCE: return "glacier: energy budget (gla:2948)";
    case NOAHMP_ERR_GLACIER_WATER_BALANCE: return "glacier: water budget (gla:2968)";
    case NOAHMP_ERR_GLACIER_FIRE_NONPOSITIVE: return "glacier: emitted longwave <0 (gla:541)";
    case NOAHMP_ERR_NSOIL_UNSUPPORTED: return "the library is built for another NSOIL (default 4; -DNOAHMP_NSOIL=n), NSNOW=3";
    case NOAHMP_ERR_CLASS_RANGE: return "a column is not of the class its range was declared to hold (sorted_land_columns / sorted_glacier_columns)";
    case NOAHMP_ERR_ISNOW_RANGE: return "ISNOWXY outside -NSNOW..0";
  }
  return "unknown";
}

const char* noahmp_hip_last_error(void) { return g.last_error.c_str(); }

// 32: every array of a call with these memory extents (ni x nj cells, the widest one having max(7, nk_atm) levels) ends below 4 GiB, so
// the option-specialised kernels (32-bit byte offsets, nmp_dev_column.hpp) can serve it; 64: the generic kernels do.
int noahmp_hip_index_width(int ni_mem, int nj_mem, int nk_atm) {
  const unsigned long long lev = nk_atm > 7 ? nk_atm : 7;
  return ((unsigned long long)ni_mem * (unsigned long long)nj_mem * lev * 4ull < (1ull << 32)) ? 32 : 64;
}

#ifdef NMP_PHASE_TIMERS
// profiling build only: read and clear the phase tick counters (summed over their 256 slots, over the generic and the
// option-specialised translation units: each has its own copy of the counters)
int noahmp_hip_debug_phase_ticks(unsigned long long* out, int n) {
  std::vector<unsigned long long> h(nmp::NMP_NPHASE * 256);
  HIPCHK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(nmp::g_nmp_prof), h.size() * 8));
  for (int p = 0; p < n && p < nmp::NMP_NPHASE; p++) {
    out[p] = 0;
    for (int s2 = 0; s2 < 256; s2++) out[p] += h[p * 256 + s2];
  }
  std::fill(h.begin(), h.end(), 0ull);
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(nmp::g_nmp_prof), h.data(), h.size() * 8));
#ifndef NMP_NO_FIXED_KERNELS
  nmp_host::prof_fixed_d1_r1(out, n); nmp_host::prof_fixed_d3_r1(out, n); nmp_host::prof_fixed_d3_r5(out, n);
  nmp_host::prof_fixed_d4_r1(out, n); nmp_host::prof_fixed_d4_r3(out, n);
#endif
  return 0;
}
#endif

void noahmp_hip_finalize(void) {
  for (auto& p : g.mirror) { if (p) hipFree(p); p = nullptr; }
  for (auto& p : g.mirror_b) { if (p) hipFree(p); p = nullptr; }
  if (g.ev_up) hipEventDestroy(g.ev_up);
  if (g.ev_kdone) hipEventDestroy(g.ev_kdone);
  for (auto& b : g.mirror_bytes) b = 0;
  g.resident_valid = false; g.resident_dirty = false; g.mirror_host.clear();
  g.sorted_land = -1; g.sorted_glacier = -1;
#ifndef NMP_NO_FIXED_KERNELS
  nmp_host::jit_finalize();
#endif
  nmp_host::sort_finalize();
  for (auto& p : g.gw_mirror) { if (p) hipFree(p); p = nullptr; }
  for (auto& p : g.init_mirror) { if (p) hipFree(p); p = nullptr; }
  for (auto e : g.async_events) hipEventDestroy(e);
  for (auto e : g.pipe_events) hipEventDestroy(e);
  drop_host_regs();
  nmp_host::stage_finalize();
  if (g.s_up) hipStreamDestroy(g.s_up);
  if (g.s_dn) hipStreamDestroy(g.s_dn);
  if (g.gw_kcell) hipFree(g.gw_kcell);
  if (g.gw_head) hipFree(g.gw_head);
  if (g.d_tables) hipFree(g.d_tables);
  if (g.d_err) hipFree(g.d_err);
  if (g.d_counts) hipFree(g.d_counts);
  for (auto& q : g.smirror) { if (q) hipFree(q); q = nullptr; }
  for (auto& b : g.smirror_bytes) b = 0;
  if (g.s_perm) hipFree(g.s_perm);
  if (g.s_keys) hipFree(g.s_keys);
  if (g.s_order) hipFree(g.s_order);
  if (g.s_dpos) hipFree(g.s_dpos);
  g.s_perm = nullptr; g.s_keys = nullptr; g.s_order = nullptr; g.s_dpos = nullptr; g.s_cols = 0;
  g.sorted_ok = false; g.sorted_newer = false; g.last_step_sorted = false;
  if (g.d_cost) hipFree(g.d_cost);
  g.d_cost = nullptr; g.d_cost_bytes = 0; g.cost_cols = 0; g.cost_fresh = false;
  if (g.d_gw_counts) hipFree(g.d_gw_counts);
  if (g.h_err) hipHostFree(g.h_err);
  if (g.h_counts) hipHostFree(g.h_counts);
  if (g.ev0) hipEventDestroy(g.ev0);
  if (g.ev1) hipEventDestroy(g.ev1);
  if (g.own_stream) hipStreamDestroy(g.own_stream);
  g = nmp_host::Engine();
}

}  // extern "C"

// Host-side engine state shared by the translation units of libnoahmp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <string>
#include <unordered_map>
#include <vector>
#include "noahmp_hip.h"

#ifndef NMP_FIXED_BLOCK
#define NMP_FIXED_BLOCK 64        // workgroup size of the option-specialised kernels (nmp_kernel.hpp)
#endif

namespace nmp_host {

// Tally counters (land / glacier / skipped columns) are spread over kCountSlots cache lines, indexed by
// workgroup id: device-scope atomics on ONE address are resolved at the memory side of the 8 XCDs and
// serialise (measured: 110k same-address atomicAdds cost 2.3 ms in the groundwater kernel, 40x its
// streaming time).  The host adds the slots up.
constexpr int kCountSlots = 256;
constexpr int kCountStride = 16;   // ints per slot = one 64-byte line

struct Engine {
  int device = -1;
  bool have_tables = false;
  noahmp_tables* d_tables = nullptr;
  int ts_i[6] = {0, 0, 0, 0, 0, 0}; float ts_f[15] = {};   // scalars of the tables last passed to noahmp_hip_set_tables
  unsigned long long* d_err = nullptr;
  int* d_counts = nullptr;
  int* d_gw_counts = nullptr;            // tallies of asynchronous groundwater calls (not reported)
  unsigned long long* h_err = nullptr;   // pinned
  int* h_counts = nullptr;               // pinned
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool ev_timed = false;                    // ev0 / ev1 were recorded by the resident-path call now pending (not for an empty tile)
  hipStream_t own_stream = nullptr;
  // host-mode mirrors of the noahmp_step_args arrays (one per field of nmp_fields.inc)
  std::vector<void*> mirror;
  std::vector<size_t> mirror_bytes;
  // groundwater: host-mode mirrors (one per pointer member of noahmp_wtable_args) + KCELL/HEAD planes
  std::vector<void*> gw_mirror;
  std::vector<size_t> gw_mirror_bytes;
  std::vector<void*> init_mirror;        // noahmp_hip_init host-mode mirrors (noahmp_init.hip)
  std::vector<size_t> init_mirror_bytes;
  float* gw_kcell = nullptr;
  float* gw_head = nullptr;
  size_t gw_plane_bytes = 0;
  int async_pending = 0;        // noahmp_hip_step_async calls since the last noahmp_hip_sync
  hipStream_t async_stream = nullptr;
  std::vector<hipStream_t> async_streams;
  std::vector<hipEvent_t> async_events;     // start/end of each pending step's kernel   // every stream that carries pending asynchronous steps
  int async_nti = 1, async_its = 1, async_jts = 1;
  float sync_class_ms[3] = {0.f, 0.f, 0.f};   // noahmp_hip_sync_timing
  int sync_steps = 0;
  std::vector<float> sync_step_ms;            // noahmp_hip_sync_step_timing: land (or mixed) kernel of each step of the last sync
  std::vector<signed char> async_kind;              // per pending step: launch_any's kind (0 one kernel, 1 class kernels in a row, 2 forked)
  int last_launch_kind = 0;
  // "record_cost": per-column trip counts of the last device-resident step, in the tile's current column order (Ctx::cost)
  unsigned char* d_cost = nullptr; size_t d_cost_bytes = 0; long cost_cols = 0; int record_cost = 0;
  bool cost_fresh = false;                    // written by a step since the last permutation of the state
  long long last_counts[3] = {0, 0, 0};       // 64-bit tallies behind the last status (noahmp_hip_sync_counts)
  int deferred_code = 0;                      // fatal code of a deferred step that no call has returned yet
  // host-memory path: row-chunk pipeline H2D | kernel | D2H on three streams, optional pinning of the caller's arrays
  hipStream_t s_up = nullptr, s_dn = nullptr;
  std::vector<hipEvent_t> pipe_events;
  struct HostReg { size_t bytes; int seen; int state; };     // state 0 pageable, 1 registered, -1 registration refused
  std::unordered_map<const void*, HostReg> host_regs;
  int host_chunks = 3;          // 0/1: single-shot staging (3 measured best at 1 M columns: fewer, larger copies)
  bool host_chunks_auto = true; // until set_option("host_chunks") names a count: 3 up to ~2.4 M columns, then one chunk per ~1.2 M, at most 8
                                // (7 M columns: 90.0 ms with 3, 85.1 with 6, 87.6 with 12, 97.3 with 24 -- tools/host_chunks_exp.py)
  int pin_host_arrays = 0;      // hipHostRegister arrays seen twice at the same address (caller guarantees their lifetime)
  int trust_out_mirror = 0;     // do not re-upload OUT arrays after the first call (caller leaves them alone between calls)
  bool out_mirror_valid = false;
  std::vector<const void*> pipe_host;      // the caller's arrays at the last row-chunk call: other arrays = the OUT mirrors say nothing about them
  int pipe_extents[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // ... and their memory extents (ims ime jms jme kms kme nsoil)
  // resident host path: the device mirrors ARE the state between calls; only IN arrays are uploaded, INOUT / OUT arrays come back
  // on request (noahmp_hip_fetch) unless lazy_download is off
  int resident_state = 0, lazy_download = 0;
  int static_inputs = 0;        // resident path: static IN arrays (XLATIN, IVGTYP, ... DZ8W) are uploaded only when the state is rebuilt
  int deferred_status = 0;      // resident + lazy path: a call returns once its forcing is uploaded; its status comes with the next call / fetch
  bool deferred_pending = false;
  unsigned resident_calls = 0;
  std::vector<void*> mirror_b;                           // second buffer of the IN arrays (deferred_status)
  std::vector<size_t> mirror_b_bytes;
  // "resident_sorted": the resident state additionally lives in a second set of mirrors in the north-star column order (class, vegetation
  // type, snow-layer count, TSK bin), the set the kernels run on (class-range kernels instead of the mixed tile-order one); the tile-order
  // mirrors stay the landing place of uploads and the source of downloads
  int resident_sorted = 0;
  std::vector<void*> smirror; std::vector<size_t> smirror_bytes;
  int* s_perm = nullptr; unsigned* s_keys = nullptr; unsigned short* s_order = nullptr; int* s_dpos = nullptr; size_t s_cols = 0;
  long s_land = -1, s_glacier = -1;
  bool sorted_ok = false;          // smirror + plan describe the resident state
  bool sorted_newer = false;       // the INOUT / OUT arrays are newer in smirror than in the tile-order mirrors
  bool last_step_sorted = false;   // the error word of the step in flight counts SORTED positions
  unsigned calls_since_sort = 0;
  hipEvent_t ev_up = nullptr, ev_kdone = nullptr;
  bool resident_valid = false, resident_dirty = false;   // dirty: the mirrors hold results the host arrays do not have yet
  std::vector<const void*> mirror_host;                  // the caller's array behind each mirror at the last resident call
  noahmp_step_args resident_args;                        // the argument block of that call (for fetch)
  int jit_kernels = 1;          // compile a specialised kernel at run time (hiprtc, cached on disk) for option sets without an ahead-of-time one
  int jit_compile_only = 0;     // test hook: compile, do not load or launch (works without a GPU)
  int fixed_kernels = 1;        // use the option-specialised kernels when a call's options are the reference's namelist values
  long sorted_land = -1, sorted_glacier = -1;   // class ranges of a sorted device-resident layout (-1: not declared)
  int block = 256;             // 4 waves per workgroup: ~1 % faster than 64 at 1 M columns (bench); 64 and 128 selectable
  int use_lds = 1;
  std::string last_error;
};
extern Engine g;

// Everything a column-kernel launch needs, in a form that does not depend on the physics headers' types: the generic
// translation unit hands it to an option-specialised one (nmp_engine_fixed.inc), which builds its own KArgs from it.
struct LaunchDesc {
  noahmp_step_args a;            // array members = device pointers
  const noahmp_tables* tables;
  int ts_i[6]; float ts_f[15];   // the tables' scalars (TabScalars of the physics headers: tab_scalars_pack / _unpack)
  float dt, zsoil[NOAHMP_NSNOW + NOAHMP_NSOIL + 1];
  int isurban, ni, nka, nti, ntj, k1, kp_lo, kp_hi, yearlen;
  unsigned long long* err;
  int* counts;
  unsigned long long err_base;
  long t_offset, t_first, t_count;
  unsigned char* cost;           // Ctx::cost of this launch (already offset to its first column) or NULL
  long r_land, r_ice, r_skip;    // mode 4 (noahmp_ranges_kernel): columns per class range
};
// mode: 0 mixed tile, 4 the three class ranges of a sorted layout in one launch; ev0 / ev1: the kernel's start / stop events or NULL;
// d<DVEG>_r<RUN>, the other options = namelist values
void launch_fixed_d1_r1(const LaunchDesc& d, int mode, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
#ifdef NMP_PHASE_TIMERS
void prof_fixed_d1_r1(unsigned long long*, int); void prof_fixed_d3_r1(unsigned long long*, int); void prof_fixed_d3_r5(unsigned long long*, int);
void prof_fixed_d4_r1(unsigned long long*, int); void prof_fixed_d4_r3(unsigned long long*, int);
#endif
void launch_fixed_d3_r1(const LaunchDesc& d, int mode, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
void launch_fixed_d3_r5(const LaunchDesc& d, int mode, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
void launch_fixed_d4_r1(const LaunchDesc& d, int mode, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
void launch_fixed_d4_r3(const LaunchDesc& d, int mode, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
// noahmp_jit.hip: the same for any option set o[12] = (DVEG, CRS, BTR, RUN, SFC, FRZ, INF, RAD, ALB, SNF, TBOT, STC), compiled on first use
bool launch_jit(const int* o, const LaunchDesc& d, int mode, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
void jit_finalize();
void jit_stats(int* out);
std::string cache_dir_public();
unsigned long long source_hash_public();
void sort_finalize();     // noahmp_sort.hip

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  char b_[256]; snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
  nmp_host::g.last_error = b_; return -100; } } while (0)

int ensure_init();
// sum the slots of h_counts into out[0..3]
void sum_counts(long long* out);
void status_counts(noahmp_status* st);     // ... into the int32 members of a status (saturated)
// grow-only device buffer
int ensure_bytes(void** p, size_t* have, size_t need);

}  // namespace nmp_host

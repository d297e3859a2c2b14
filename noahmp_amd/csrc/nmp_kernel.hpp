// The column kernel (one thread = one column-step), shared by the generic translation unit (noahmp_engine.hip: options are run-time
// values) and the option-specialised ones (nmp_engine_fixed.inc: options are compile-time constants).
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
#include "nmp_dev_column.hpp"
#ifndef __HIPCC_RTC__
#include "nmp_engine_host.hpp"
#else                      // run-time compilation (hiprtc): only what the kernel itself needs of the host header
namespace nmp_host { constexpr int kCountSlots = 256, kCountStride = 16; }
#endif

namespace {
using namespace nmp;

// Minimum waves per SIMD the register allocator must leave room for (2nd __launch_bounds__ argument
// = k*BLOCK/256 blocks of BLOCK threads per CU).  One wave alone on a SIMD issues a VALU instruction
// every 4 cycles, two or more every 2 (MI355X_MICROARCH.md), and this kernel is VALU-issue bound.
#ifndef NMP_WAVES_PER_EU
#define NMP_WAVES_PER_EU 2
#endif
// Workgroup size of the option-specialised kernels (ahead-of-time units and hiprtc units).  One wave per workgroup: a workgroup's LDS
// (the layer slots of its columns) is released when its LAST wave ends, and with 256-thread workgroups (59 KB each, two per CU) a
// SIMD whose wave has finished idles until the three other waves of that workgroup have -- their trip counts differ.  Round 5, A/B on one
// box: config 3 land kernel 3.278 (256) / 3.229 (128) / **3.186 ms (64)**; config 5 3.338 / 3.113 / **3.042**.  (Round 4 measured -0.4 % for
// 128 on a kernel that still waited more than it computed.)
#ifndef NMP_FIXED_BLOCK
#define NMP_FIXED_BLOCK 64
#endif

// One thread = one column-step (the ILOOP body, drv:424-837).
// MODE 0: the tile as it is (any mix of classes).  MODE 1 / 2 / 3: a range of a class-sorted layout that holds only land /
// only glacier / only skipped (open water, sea ice) columns -- kernels without the other classes' code; a column of another
// class in such a range raises NOAHMP_ERR_CLASS_RANGE (its class changed since the sort, e.g. sea ice: sort again).
// lds: the workgroup's layer slots (LAY_SLOTS * BLOCK floats; unused by MODE 3).  blk: the workgroup's index inside its launch (MODE 0) or
// inside its class range, which covers tile indices [first, first + count).
template <int BLOCK, bool USE_LDS, int MODE>
__device__ __forceinline__ void column_kernel_body(const KArgs& k, float* lds, const long blk, const long first, const long count) {
  constexpr int STRIDE = USE_LDS ? BLOCK : 1;
  float priv[(USE_LDS || MODE == 3) ? 1 : LAY_SLOTS];
  float* base = USE_LDS ? (lds + threadIdx.x) : priv;

  if (MODE != 3) libm::libm_stage_tables();
  const long tl = blk * BLOCK + threadIdx.x;
  const long t = first + tl;
  int ii = 0, jj = 0;
  nmp_ij_t ij = 0;
  int cls = 3, err = 0;
  if (MODE == 1 || MODE == 2) {
    // class range of a sorted layout: the gather does not wait for the classification (column_step<.., EARLY>)
    if (tl < count && column_index(k, t, ii, jj, ij)) {
      SimpleLoop runner;
      err = column_step<STRIDE, MODE, true>(k, -1, ii, jj, ij, base, runner, &cls);
    }
  } else {
    cls = (MODE != 0 && tl >= count) ? 3 : column_classify(k, t, ii, jj, ij);
  }
  {                                           // per-wave tallies (64-wide wavefront)
    unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1), m2 = __ballot(cls == 2);
    if ((threadIdx.x & 63) == 0) {
      int* cnt = k.counts + (blockIdx.x % nmp_host::kCountSlots) * nmp_host::kCountStride;   // see nmp_engine_host.hpp
      if (m0) atomicAdd(&cnt[0], __popcll(m0));
      if (m1) atomicAdd(&cnt[1], __popcll(m1));
      if (m2) atomicAdd(&cnt[2], __popcll(m2));
    }
  }
  if (MODE != 0 && cls != 3 && cls != MODE - 1) {       // not the class this range was declared to hold
    atomicMin(k.err, k.err_base | ((unsigned long long)(t + k.t_offset + 1) << 8) | (unsigned)NOAHMP_ERR_CLASS_RANGE);
    return;
  }
  if (MODE == 1 || MODE == 2) {
    if (err) atomicMin(k.err, k.err_base | ((unsigned long long)(t + k.t_offset + 1) << 8) | (unsigned)err);   // first column wins
    return;
  }
  if (cls > 1 || MODE == 3) return;
  SimpleLoop runner;
  err = column_step<STRIDE, (MODE == 3 ? 0 : MODE)>(k, cls, ii, jj, ij, base, runner);
  if (err) atomicMin(k.err, k.err_base | ((unsigned long long)(t + k.t_offset + 1) << 8) | (unsigned)err);   // first column wins
}

template <int BLOCK, bool USE_LDS, int MODE = 0>
__global__ void __launch_bounds__(BLOCK, NMP_WAVES_PER_EU) noahmp_column_kernel(const KArgs k) {
  __shared__ float lds[(USE_LDS && MODE != 3) ? LAY_SLOTS * BLOCK : 1];
  column_kernel_body<BLOCK, USE_LDS, MODE>(k, lds, (long)blockIdx.x, k.t_first, k.t_count);
}

// The three class ranges of a sorted layout in ONE launch (round 6): workgroups [0, nb_ice) advance the land-ice range, the next nb_land the
// land range, the rest the skipped cells -- every wave runs the code of one class only, as in the three separate kernels of rounds 2-5,
// whose side-by-side execution needed a second stream with a fork and a join event: three barrier packets per step on the run's queue,
// ~18 us in which no kernel ran (kernel trace, profiles/r06_experiments.md), 3 % of the step of an 8-rank tile.  Land ice goes FIRST: its
// waves are the longest (84 vs ~60 us) and there are few of them, the skipped cells (a few instructions per wave) last -- the order of
// longest-first list scheduling.  Registers: the maximum over the three bodies (they share nothing but the kernel arguments); LDS: one
// array for whichever body the workgroup runs.  KArgs::r_land / r_ice / r_skip = columns per class (land columns come first in the arrays).
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, NMP_WAVES_PER_EU) noahmp_ranges_kernel(const KArgs k) {
  __shared__ float lds[LAY_SLOTS * BLOCK];
  const long nb_ice = (k.r_ice + BLOCK - 1) / BLOCK, nb_land = (k.r_land + BLOCK - 1) / BLOCK;
  const long b = (long)blockIdx.x;
  if (b < nb_ice) column_kernel_body<BLOCK, true, 2>(k, lds, b, k.r_land, k.r_ice);
  else if (b < nb_ice + nb_land) column_kernel_body<BLOCK, true, 1>(k, lds, b - nb_ice, 0L, k.r_land);
  else column_kernel_body<BLOCK, true, 3>(k, lds, b - nb_ice - nb_land, k.r_land + k.r_ice, k.r_skip);
}

}  // namespace

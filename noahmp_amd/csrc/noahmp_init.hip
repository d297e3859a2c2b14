// Cold start on the device: C-ABI entry noahmp_hip_init(), the drop-in for the per-column part of NOAHMP_INIT
// (reference phys/module_sf_noahmpdrv.F90:988-1134) and SNOW_INIT (drv:1182-1283).  One streaming pass, one thread
// per column, called once per run: it exists so that a device-resident run (30-day spin-up) never
// needs a host pass over the state.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <string.h>
#include "noahmp_hip.h"
#include "nmp_dev_init.hpp"
#include "nmp_engine_host.hpp"
#include "nmp_stage.hpp"

using namespace nmp;
using nmp_host::g;

namespace {

__global__ void __launch_bounds__(256) noahmp_init_kernel(const InitArgs k, int nti, int ntj) {
  libm::libm_stage_tables();
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)nti * ntj) return;
  const int tj = (int)(t / nti), ti = (int)(t - (long)tj * nti);
  const int err = init_column(k, k.a.its - k.a.ims + ti, k.a.jts - k.a.jms + tj);
  if (err) atomicMin(k.err, ((unsigned long long)(t + 1) << 8) | (unsigned)err);
}

// the members of noahmp_step_args NOAHMP_INIT reads or writes: {offset, levels-kind, written}
struct IField { size_t off; int lev; int out; };   // lev: 0 2-D, 2 soil, 3 snow, 4 snso
#define IF_(n, lev, out) {offsetof(noahmp_step_args, n), lev, out}
const IField kI[] = {
    IF_(snow, 0, 1), IF_(snowh, 0, 1), IF_(canwat, 0, 1), IF_(isltyp, 0, 0), IF_(ivgtyp, 0, 0), IF_(tslb, 2, 1),
    IF_(smois, 2, 1), IF_(sh2o, 2, 1), IF_(tsk, 0, 0), IF_(isnowxy, 0, 1), IF_(tvxy, 0, 1), IF_(tgxy, 0, 1),
    IF_(canicexy, 0, 1), IF_(xice, 0, 0), IF_(canliqxy, 0, 1), IF_(eahxy, 0, 1), IF_(tahxy, 0, 1), IF_(cmxy, 0, 1),
    IF_(chxy, 0, 1), IF_(fwetxy, 0, 1), IF_(sneqvoxy, 0, 1), IF_(alboldxy, 0, 1), IF_(qsnowxy, 0, 1),
    IF_(wslakexy, 0, 1), IF_(zwtxy, 0, 1), IF_(waxy, 0, 1), IF_(wtxy, 0, 1), IF_(tsnoxy, 3, 1), IF_(zsnsoxy, 4, 1),
    IF_(snicexy, 3, 1), IF_(snliqxy, 3, 1), IF_(lfmassxy, 0, 1), IF_(rtmassxy, 0, 1), IF_(stmassxy, 0, 1),
    IF_(woodxy, 0, 1), IF_(stblcpxy, 0, 1), IF_(fastcpxy, 0, 1), IF_(xsaixy, 0, 1), IF_(t2mvxy, 0, 1),
    IF_(t2mbxy, 0, 1)};
constexpr int kNI = sizeof(kI) / sizeof(kI[0]);

}  // namespace

extern "C" int noahmp_hip_init(const noahmp_step_args* a, int iswater, int fndsnowh, int mem, void* stream,
                               noahmp_status* st) {
  (void)iswater;
  if (st) memset(st, 0, sizeof(*st));
  int rc = nmp_host::ensure_init();
  if (rc) return rc;
  if (!g.have_tables) { g.last_error = "noahmp_hip_set_tables() has not been called"; return -102; }
  if (a->nsoil != NOAHMP_NSOIL) { if (st) st->code = NOAHMP_ERR_NSOIL_UNSUPPORTED; return NOAHMP_ERR_NSOIL_UNSUPPORTED; }
  hipStream_t s = stream ? (hipStream_t)stream : g.own_stream;
  InitArgs k;
  memset(&k, 0, sizeof(k));
  k.a = *a;
  k.T = g.d_tables;
  k.ni = a->ime - a->ims + 1;
  const int nj = a->jme - a->jms + 1;
  k.itf = a->ite < a->ide - 1 ? a->ite : a->ide - 1;                                // drv:991-992
  k.jtf = a->jte < a->jde - 1 ? a->jte : a->jde - 1;
  k.fndsnowh = fndsnowh;
  k.zsoil[0] = -a->dzs[0];                                                          // drv:1139-1142
  for (int l = 1; l < NOAHMP_NSOIL; l++) k.zsoil[l] = k.zsoil[l - 1] - a->dzs[l];
  k.a.dzs = nullptr;
  k.err = g.d_err;
  const int nti = k.itf - a->its + 1, ntj = k.jtf - a->jts + 1;

  std::vector<void*>& mir = g.init_mirror;
  if (mem == NOAHMP_MEM_HOST) {
    if (mir.empty()) { mir.assign(kNI, nullptr); g.init_mirror_bytes.assign(kNI, 0); }
    std::vector<nmp_host::CopySeg> up;
    for (int f = 0; f < kNI; f++) {
      const size_t nk = kI[f].lev == 2 ? NOAHMP_NSOIL : kI[f].lev == 3 ? 3 : kI[f].lev == 4 ? NOAHMP_NSOIL + 3 : 1;
      const size_t bytes = (size_t)k.ni * nj * nk * 4;
      rc = nmp_host::ensure_bytes(&mir[f], &g.init_mirror_bytes[f], bytes);
      if (rc) return rc;
      up.push_back(nmp_host::CopySeg{*(void* const*)((const char*)a + kI[f].off), mir[f], bytes});      // outputs too: untouched cells survive
      *(void**)((char*)&k.a + kI[f].off) = mir[f];
    }
    if ((rc = nmp_host::copy_segments(up.data(), (int)up.size(), true, s))) return rc;        // pageable arrays: the engine's bounce buffers (nmp_stage.hpp)
  }
  *g.h_err = ~0ULL;
  HIPCHK(hipMemsetAsync(g.d_err, 0xFF, sizeof(unsigned long long), s));
  const bool timed = nti > 0 && ntj > 0;               // an empty tile: no kernel, no events
  if (timed) HIPCHK(hipEventRecord(g.ev0, s));
  if (nti > 0 && ntj > 0) {
    const long n = (long)nti * ntj;
    hipLaunchKernelGGL(noahmp_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, k, nti, ntj);
  }
  HIPCHK(hipGetLastError());
  if (timed) HIPCHK(hipEventRecord(g.ev1, s));
  HIPCHK(hipMemcpyAsync(g.h_err, g.d_err, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  if (mem == NOAHMP_MEM_HOST) {
    std::vector<nmp_host::CopySeg> down;
    for (int f = 0; f < kNI; f++) {
      if (!kI[f].out) continue;
      const size_t nk = kI[f].lev == 2 ? NOAHMP_NSOIL : kI[f].lev == 3 ? 3 : kI[f].lev == 4 ? NOAHMP_NSOIL + 3 : 1;
      down.push_back(nmp_host::CopySeg{*(void* const*)((const char*)a + kI[f].off), mir[f], (size_t)k.ni * nj * nk * 4});
    }
    if ((rc = nmp_host::copy_segments(down.data(), (int)down.size(), false, s))) return rc;
  }
  HIPCHK(hipStreamSynchronize(s));
  int code = 0;
  if (st) {
    float ms = 0.f;
    if (timed) hipEventElapsedTime(&ms, g.ev0, g.ev1);
    st->kernel_ms = ms;
    st->n_land = nti > 0 && ntj > 0 ? nti * ntj : 0;
  }
  if (*g.h_err != ~0ULL) {
    code = (int)(*g.h_err & 0xFF);
    const long t = (long)(*g.h_err >> 8) - 1;
    if (st) { st->code = code; st->i = a->its + (int)(t % nti); st->j = a->jts + (int)(t / nti); }
  }
  return code;
}

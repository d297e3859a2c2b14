// TEST INFRASTRUCTURE -- host-compiled emulation of the device source.
// The column physics in noahmp_amd/csrc/*.hpp is __host__ __device__; this harness runs the very
// same column_step() on the CPU (host libm) so that the restructured GPU source can be checked
// bit-for-bit against the oracle in a container that has no GPU.  Never shipped, never a fallback:
// the product library (libnoahmp_hip.so) contains no host path.
#include <hip/hip_runtime.h>
#include <string.h>
#include <vector>
#include "noahmp_hip.h"
#include "nmp_dev_column.hpp"

using namespace nmp;

static TablesDev g_img;
#define g_t g_img.t

extern "C" int emul_set_tables(const noahmp_tables* t) { g_img.t = *t; derive_tables(g_img.t, g_img.d); return 0; }

extern "C" int emul_step(const noahmp_step_args* a, noahmp_status* st) {
  memset(st, 0, sizeof(*st));
  KArgs k;
  memset(&k, 0, sizeof(k));
  k.a = *a;
  k.ni = a->ime - a->ims + 1;
  k.nka = a->kme - a->kms + 1;
  k.nti = a->ite - a->its + 1;
  k.ntj = a->jte - a->jts + 1;
  k.k1 = 1 - a->kms;
  k.kp_lo = a->kts - a->kms;
  k.kp_hi = a->kts + 1 - a->kms;
  k.yearlen = 365;
  if (a->yr % 4 == 0) { k.yearlen = 366; if (a->yr % 100 == 0) { k.yearlen = 365; if (a->yr % 400 == 0) k.yearlen = 366; } }
  k.c.T = &g_t;
  k.c.D = &g_img.d;
  k.c.O = Opt{a->idveg, a->iopt_crs, a->iopt_btr, a->iopt_run, a->iopt_sfc, a->iopt_frz, a->iopt_inf,
              a->iopt_rad, a->iopt_alb, a->iopt_snf, a->iopt_tbot, a->iopt_stc};
  k.c.dt = a->dt;
  k.c.isurban = a->isurban;
  k.c.ts = tab_scalars(g_t);
  k.c.zsoil[L(1)] = -a->dzs[0];
  for (int l = 2; l <= NOAHMP_NSOIL; l++) k.c.zsoil[L(l)] = -a->dzs[l - 1] + k.c.zsoil[L(l - 1)];
  ctx_fill_uniform(k.c);
  const long n = (long)k.nti * k.ntj;
  for (long t = 0; t < n; t++) {
    float base[LAY_SLOTS];
    int ii = 0, jj = 0;
    nmp_ij_t ij = 0;
    int cls = column_classify(k, t, ii, jj, ij);
    if (cls == 2) st->n_skipped++;
    if (cls > 1) continue;
    if (cls == 0) st->n_land++; else st->n_glacier++;
    SimpleLoop runner;
    int err = column_step<1>(k, cls, ii, jj, ij, base, runner);
    if (err && !st->code) { st->code = err; st->i = a->its + (int)(t % k.nti); st->j = a->jts + (int)(t / k.nti); }
  }
  return st->code;
}

// ---- groundwater (noahmp_groundwater.hip's two kernels as host loops over the same device functions)
#include "nmp_dev_groundwater.hpp"
extern "C" int emul_wtable_mmf(const noahmp_wtable_args* a, noahmp_status* st) {
  memset(st, 0, sizeof(*st));
  GwArgs k;
  memset(&k, 0, sizeof(k));
  k.a = *a;
  k.T = &g_t;
  k.ni = a->ime - a->ims + 1;
  const int nj = a->jme - a->jms + 1;
  k.deltat = a->wtddt * 60.f;
  k.zsoil[0] = 0.f;
  k.zsoil[1] = -a->dzs[0];
  for (int l = 2; l <= NOAHMP_NSOIL; l++) k.zsoil[l] = -a->dzs[l - 1] + k.zsoil[l - 1];
  for (int l = 0; l < NOAHMP_NSOIL; l++) k.dzs[l] = a->dzs[l];
  auto imax = [](int x, int y) { return x > y ? x : y; };
  auto imin = [](int x, int y) { return x < y ? x : y; };
  k.hi0 = imax(a->its - 1, a->ids); k.hi1 = imin(a->ite + 1, a->ide - 1);
  k.hj0 = imax(a->jts - 1, a->jds); k.hj1 = imin(a->jte + 1, a->jde - 1);
  k.qi0 = imax(a->its, a->ids + 1); k.qi1 = imin(a->ite, a->ide - 2);
  k.qj0 = imax(a->jts, a->jds + 1); k.qj1 = imin(a->jte, a->jde - 2);
  std::vector<float> kcell((size_t)k.ni * nj, 0.f), head((size_t)k.ni * nj, 0.f);
  k.kcell = kcell.data();
  k.head = head.data();
  for (int gj = k.hj0; gj <= k.hj1; gj++)
    for (int gi = k.hi0; gi <= k.hi1; gi++) gw_cell_head(k, (size_t)(gj - a->jms) * k.ni + (gi - a->ims));
  for (int gj = a->jts; gj <= a->jte; gj++)
    for (int gi = a->its; gi <= a->ite; gi++) {
      if (gw_column(k, gi - a->ims, gj - a->jms, gi, gj)) st->n_land++; else st->n_skipped++;
    }
  return 0;
}

// ---- cold start (noahmp_init.hip's kernel as a host loop over the same device function)
#include "nmp_dev_init.hpp"
extern "C" int emul_init(const noahmp_step_args* a, int iswater, int fndsnowh, noahmp_status* st) {
  (void)iswater;
  memset(st, 0, sizeof(*st));
  InitArgs k;
  memset(&k, 0, sizeof(k));
  k.a = *a;
  k.T = &g_t;
  k.ni = a->ime - a->ims + 1;
  k.itf = a->ite < a->ide - 1 ? a->ite : a->ide - 1;
  k.jtf = a->jte < a->jde - 1 ? a->jte : a->jde - 1;
  k.fndsnowh = fndsnowh;
  k.zsoil[0] = -a->dzs[0];
  for (int l = 1; l < NOAHMP_NSOIL; l++) k.zsoil[l] = k.zsoil[l - 1] - a->dzs[l];
  for (int j = a->jts; j <= k.jtf; j++)
    for (int i = a->its; i <= k.itf; i++) {
      int err = init_column(k, i - a->ims, j - a->jms);
      if (err && !st->code) { st->code = err; st->i = i; st->j = j; }
    }
  return st->code;
}

extern "C" int emul_groundwater_init(const noahmp_wtable_args* a, int iswater, noahmp_status* st) {
  memset(st, 0, sizeof(*st));
  GwArgs k;
  memset(&k, 0, sizeof(k));
  k.a = *a;
  k.T = &g_t;
  k.ni = a->ime - a->ims + 1;
  const int nj = a->jme - a->jms + 1;
  k.deltat = a->wtddt * 60.f;
  k.zsoil[0] = 0.f;
  k.zsoil[1] = -a->dzs[0];
  for (int l = 2; l <= NOAHMP_NSOIL; l++) k.zsoil[l] = -a->dzs[l - 1] + k.zsoil[l - 1];
  for (int l = 0; l < NOAHMP_NSOIL; l++) k.dzs[l] = a->dzs[l];
  auto imax = [](int x, int y) { return x > y ? x : y; };
  auto imin = [](int x, int y) { return x < y ? x : y; };
  k.hi0 = imax(a->its - 1, a->ids); k.hi1 = imin(a->ite + 1, a->ide - 1);
  k.hj0 = imax(a->jts - 1, a->jds); k.hj1 = imin(a->jte + 1, a->jde - 1);
  k.qi0 = imax(a->its, a->ids + 1); k.qi1 = imin(a->ite, a->ide - 2);
  k.qj0 = imax(a->jts, a->jds + 1); k.qj1 = imin(a->jte, a->jde - 2);
  std::vector<float> kcell((size_t)k.ni * nj, 0.f), head((size_t)k.ni * nj, 0.f);
  k.kcell = kcell.data();
  k.head = head.data();
  for (int gj = k.hj0; gj <= k.hj1; gj++)
    for (int gi = k.hi0; gi <= k.hi1; gi++) gw_cell_head(k, (size_t)(gj - a->jms) * k.ni + (gi - a->ims));
  const int itf = imin(a->ite, a->ide - 1), jtf = imin(a->jte, a->jde - 1);
  for (int gj = a->jts; gj <= jtf; gj++)
    for (int gi = a->its; gi <= itf; gi++) gw_init_column(k, gi - a->ims, gj - a->jms, gi, gj, iswater);
  return 0;
}

// ---- forcing preparation (noahmp_forcing.hip's kernel as a host loop; the uniform declination comes from the caller)
#include "nmp_dev_forcing.hpp"
extern "C" int emul_forcing_prep(const noahmp_step_args* a, const float* lon2d, const float* rain_rate, float hour_utc,
                                 float sin_declin, float cos_declin, float zlvl, int flags) {
  ForcingArgs k;
  memset(&k, 0, sizeof(k));
  k.a = *a; k.lon = lon2d; k.rain_rate = rain_rate; k.hour_utc = hour_utc; k.sin_declin = sin_declin; k.cos_declin = cos_declin;
  k.dt = a->dt; k.dz8w = 2.0f * zlvl; k.scale_vegfra = flags & 1; k.first_step = (flags & 2) ? 1 : 0;
  k.ni = a->ime - a->ims + 1; k.nka = a->kme - a->kms + 1; k.k1 = 1 - a->kms;
  for (int j = a->jts; j <= a->jte; j++)
    for (int i = a->its; i <= a->ite; i++) forcing_cell(k, i - a->ims, j - a->jms);
  return 0;
}

extern "C" int emul_forcing_interpolate(const noahmp_step_args* a, const noahmp_forcing_record* ra, const noahmp_forcing_record* rb,
                                        int idts, int idts2, float* rain_rate_out) {
  InterpArgs k;
  memset(&k, 0, sizeof(k));
  k.a = *a; k.ra = *ra; if (rb) k.rb = *rb;
  k.has_b = rb ? 1 : 0; k.rain_rate = rain_rate_out;
  k.fraction = rb ? (float)(idts2 - idts) / (float)idts2 : 1.0f;
  k.one_minus = 1.0f - k.fraction;
  k.ni = a->ime - a->ims + 1; k.nka = a->kme - a->kms + 1; k.k1 = 1 - a->kms;
  for (int j = a->jts; j <= a->jte; j++)
    for (int i = a->its; i <= a->ite; i++) interp_cell(k, i - a->ims, j - a->jms);
  return 0;
}

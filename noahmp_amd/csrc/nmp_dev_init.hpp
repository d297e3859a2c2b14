// Cold start on the device: the per-column part of NOAHMP_INIT (reference phys/module_sf_noahmpdrv.F90:988-1134,
// "drv", restart = .false.) and SNOW_INIT (drv:1182-1283), one thread per column.  The table readers that
// NOAHMP_INIT calls first (drv:979-987) stay in the caller's Fortran; GROUNDWATER_INIT (drv:1286) is not part of
// this entry.  Arrays are addressed through the same noahmp_step_args block as a time step.
#pragma once
#include "nmp_dev_common.hpp"

namespace nmp {

struct InitArgs {
  noahmp_step_args a;            // device pointers; only the members NOAHMP_INIT takes are read or written
  const noahmp_tables* __restrict__ T;
  float zsoil[NSOIL];            // drv:1139-1142
  int ni;
  int itf, jtf;                  // min(ite, ide-1), min(jte, jde-1): drv:991-992
  int fndsnowh;
  unsigned long long* err;
};

// returns 0 or NOAHMP_ERR_SOILTYP_RANGE (drv:1010-1020: "lsminit: out of range value of ISLTYP")
NMP_DEV int init_column(const InitArgs& k, int ii, int jj) {
  const noahmp_step_args& a = k.a;
  const size_t ij = (size_t)jj * k.ni + ii;
  const size_t plane = (size_t)k.ni;
  const size_t s0 = ((size_t)jj * NSOIL) * k.ni + ii;           // (i, 1, j) of a soil array
  const size_t n0 = ((size_t)jj * NSNOW) * k.ni + ii;           // (i, -2, j) of a snow array
  const size_t z0 = ((size_t)jj * (NSNOW + NSOIL)) * k.ni + ii; // (i, -2, j) of ZSNSOXY
  const float HLICE = 3.335E5f, GRAV0 = 9.81f, T0 = 273.15f;    // drv:965-967

  float snow = a.snow[ij];
  float snowh = k.fndsnowh ? a.snowh[ij] : snow * 0.005f;       // drv:997-1005
  const int isl = a.isltyp[ij];
  if (isl < 1) return NOAHMP_ERR_SOILTYP_RANGE;

  // ---- soil liquid water, drv:1032-1069
  if (a.ivgtyp[ij] == a.isice && a.xice[ij] <= 0.0f) {          // glacier starts all frozen
#pragma unroll
    for (int ns = 0; ns < NSOIL; ns++) {
      a.smois[s0 + ns * plane] = 1.0f;
      a.sh2o[s0 + ns * plane] = 0.0f;
      a.tslb[s0 + ns * plane] = fmin2(a.tslb[s0 + ns * plane], 263.15f);
    }
    snow = fmax2(snow, 10.0f);
    snowh = snow * 0.01f;
  } else {
    const int st = (isl <= 30) ? isl - 1 : 29;
    const float bx = k.T->bb[st], smcmax = k.T->maxsmc[st], psisat = k.T->satpsi[st];
    const bool ok = (bx > 0.0f) && (smcmax > 0.0f) && (psisat > 0.0f);
#pragma unroll
    for (int ns = 0; ns < NSOIL; ns++) {
      float sm = a.smois[s0 + ns * plane];
      if (sm > smcmax) sm = smcmax;
      a.smois[s0 + ns * plane] = sm;
      const float t = a.tslb[s0 + ns * plane];
      float sh = sm;
      if (ok && t < 273.149f) {                                  // explicit initial soil ice
        float fk = nmp_powf((HLICE / (GRAV0 * (-psisat))) * ((t - T0) / t), -1.0f / bx) * smcmax;
        fk = fmax2(fk, 0.02f);
        sh = fmin2(fk, sm);
      }
      a.sh2o[s0 + ns * plane] = sh;
    }
  }
  a.snow[ij] = snow;
  a.snowh[ij] = snowh;

  // ---- per-column scalars, drv:1073-1134
  const float tsk = a.tsk[ij];
  const float tsfc = (snow > 0.0f && tsk > 273.15f) ? 273.15f : tsk;
  a.tvxy[ij] = tsfc; a.tgxy[ij] = tsfc; a.tahxy[ij] = tsfc; a.t2mvxy[ij] = tsfc; a.t2mbxy[ij] = tsfc;
  a.canwat[ij] = 0.0f; a.canliqxy[ij] = 0.0f; a.canicexy[ij] = 0.0f;
  a.eahxy[ij] = 2000.f;
  a.cmxy[ij] = 0.0f; a.chxy[ij] = 0.0f; a.fwetxy[ij] = 0.0f; a.sneqvoxy[ij] = 0.0f;
  a.alboldxy[ij] = 0.65f; a.qsnowxy[ij] = 0.0f; a.wslakexy[ij] = 0.0f;
  if (a.iopt_run != 5) {
    a.waxy[ij] = 4900.f;
    a.wtxy[ij] = 4900.f;
    a.zwtxy[ij] = (25.f + 2.0f) - 4900.f / 1000.f / 0.2f;
  } else {
    a.waxy[ij] = 0.f;
    a.wtxy[ij] = 0.f;
  }
  a.lfmassxy[ij] = 50.f; a.stmassxy[ij] = 50.0f; a.rtmassxy[ij] = 500.0f; a.woodxy[ij] = 500.0f;
  a.stblcpxy[ij] = 1000.0f; a.fastcpxy[ij] = 1000.0f; a.xsaixy[ij] = 0.1f;

  // ---- SNOW_INIT, drv:1182-1283
  int isnow = 0;
  float dz2 = 0.f, dz1 = 0.f, dz0 = 0.f;                        // DZSNO(-2), DZSNO(-1), DZSNO(0)
  if (!(snowh < 0.025f)) {
    if (snowh >= 0.025f && snowh <= 0.05f) { isnow = -1; dz0 = snowh; }
    else if (snowh > 0.05f && snowh <= 0.10f) { isnow = -2; dz1 = snowh / 2.f; dz0 = snowh / 2.f; }
    else if (snowh > 0.10f && snowh <= 0.25f) { isnow = -2; dz1 = 0.05f; dz0 = snowh - dz1; }
    else if (snowh > 0.25f && snowh <= 0.45f) { isnow = -3; dz2 = 0.05f; dz1 = 0.5f * (snowh - dz2); dz0 = 0.5f * (snowh - dz2); }
    else if (snowh > 0.45f) { isnow = -3; dz2 = 0.05f; dz1 = 0.20f; dz0 = snowh - dz1 - dz2; }
    // (a NaN depth reaches wrf_error_fatal in the reference, drv:1245; here it keeps ISNOW = 0)
  }
  a.isnowxy[ij] = isnow;
  const float rho = snow / snowh;                                // SWE / SNODEP, used only where a layer exists
  const float dzs_[3] = {dz2, dz1, dz0};
  float run = 0.f;
#pragma unroll
  for (int iz = -2; iz <= 0; iz++) {
    const bool act = iz >= isnow + 1;
    a.tsnoxy[n0 + (iz + 2) * plane] = act ? tsfc : 0.f;
    a.snliqxy[n0 + (iz + 2) * plane] = 0.f;
    a.snicexy[n0 + (iz + 2) * plane] = act ? 1.00f * dzs_[iz + 2] * rho : 0.f;
    if (act) {                                                   // inactive ZSNSOXY entries are never written (drv:1275)
      run = (iz == isnow + 1) ? -dzs_[iz + 2] : run + (-dzs_[iz + 2]);
      a.zsnsoxy[z0 + (iz + 2) * plane] = run;
    }
  }
#pragma unroll
  for (int iz = 1; iz <= NSOIL; iz++) {
    const float dzsnso = (iz == 1) ? k.zsoil[0] : (k.zsoil[iz - 1] - k.zsoil[iz - 2]);
    run = (isnow == 0 && iz == 1) ? dzsnso : run + dzsnso;
    a.zsnsoxy[z0 + (iz + 2) * plane] = run;
  }
  return 0;
}

}  // namespace nmp

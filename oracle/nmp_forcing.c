/* TEST INFRASTRUCTURE (oracle) -- CPU restatement of the driver-side forcing preparation:
 * driver/module_hrldas_noahmp_driver.F90:336-354 ("hdrv") and CALC_DECLIN (hdrv:813-863).
 * COSZEN / JULIAN: PINNED.  CALC_DECLIN is an external subroutine that needs only util/module_date_utilities.F; `make -C oracle declin`
 * compiles it unmodified (oracle/_ref/libnoahmp_declin_ref.so) and tests/test_forcing.py holds this file to it bit for bit, live and
 * through tests/golden/golden_declin.npz.  The copies and unit scalings of hdrv:336-354 sit inside the module part of the driver file
 * (needs NetCDF: not buildable here) and are restated in float32 in source order; the temporal interpolation below likewise. */
#include <math.h>
#include <string.h>
#include "noahmp_oracle.h"

int nmp_oracle_forcing_prep(const noahmp_step_args* a, const float* lon2d, const float* rain_rate, int iday, int ihour,
                            int iminute, int isecond, float zlvl, int flags, float* julian_out) {
  const real DEGRAD = 3.14159265f / 180.f, DPD = 360.f / 365.f;                      /* hdrv:815-816 */
  const int ni = a->ime - a->ims + 1, nka = a->kme - a->kms + 1, k1 = 1 - a->kms;
  real* t3d = (real*)a->t3d; real* qv = (real*)a->qv3d; real* u = (real*)a->u_phy; real* v = (real*)a->v_phy;
  real* p = (real*)a->p8w3d; real* dz = (real*)a->dz8w; real* rainbl = (real*)a->rainbl; real* vegfra = (real*)a->vegfra;
  real* cosz = (real*)a->coszin;
  real julian = (real)iday + (real)ihour / 24.f;                                      /* hdrv:830 */
  real obecl = 23.5f * DEGRAD;
  real sinob = sinf(obecl);
  real sxlong = 0.f;
  if (julian >= 80.f) sxlong = DPD * (julian - 80.f) * DEGRAD;
  if (julian < 80.f) sxlong = DPD * (julian + 285.f) * DEGRAD;
  real arg = sinob * sinf(sxlong);
  real declin = asinf(arg);
  if (julian_out) *julian_out = julian;
  for (int j = a->jts; j <= a->jte; j++)
    for (int i = a->its; i <= a->ite; i++) {
      size_t ij = (size_t)(j - a->jms) * ni + (i - a->ims);
      size_t l1 = ((size_t)(j - a->jms) * nka + k1) * ni + (i - a->ims), l2 = l1 + ni;
      if (flags & NOAHMP_PREP_SCALE_VEGFRA) vegfra[ij] = vegfra[ij] * 100.0f;                             /* hdrv:337 */
      p[l2] = p[l1]; t3d[l2] = t3d[l1]; u[l2] = u[l1]; v[l2] = v[l1]; qv[l2] = qv[l1]; /* hdrv:339-343 */
      rainbl[ij] = rain_rate[ij] * a->dt;                                             /* hdrv:344 */
      dz[l1] = 2.0f * zlvl; dz[l2] = 2.0f * zlvl;                                     /* hdrv:345 */
      if (flags & NOAHMP_PREP_FIRST_STEP) {                                            /* hdrv:374-384 */
        ((real*)a->eahxy)[ij] = (p[l1] * qv[l1]) / (0.622f + qv[l1]);
        ((real*)a->tahxy)[ij] = t3d[l1];
        ((real*)a->chxy)[ij] = 0.1f;
        ((real*)a->cmxy)[ij] = 0.1f;
      }
      real tloctim = (real)ihour + (real)iminute / 60.0f + (real)isecond / 3600.0f + lon2d[ij] / 15.0f;
      tloctim = fmodf(tloctim + 24.0f, 24.0f);
      real hrang = 15.f * (tloctim - 12.f) * DEGRAD;
      cosz[ij] = sinf(a->xlatin[ij] * DEGRAD) * sinf(declin) + cosf(a->xlatin[ij] * DEGRAD) * cosf(declin) * cosf(hrang);
    }
  return 0;
}

/* hrldas_input_interpolate (driver/module_hrldas_netcdf_io.F90:1369-1403) / hrldas_input_copy (1351-1366) into the arrays
 * hrldas_input_read's caller passes (hdrv:331-335).  PARITY UNPINNED for the same reason (the module uses netcdf). */
int nmp_oracle_forcing_interpolate(const noahmp_step_args* a, const noahmp_forcing_record* ra, const noahmp_forcing_record* rb,
                                   int idts, int idts2, float* rain_rate_out) {
  const int ni = a->ime - a->ims + 1, nka = a->kme - a->kms + 1, k1 = 1 - a->kms;
  real fraction = rb ? (real)(idts2 - idts) / (real)idts2 : 1.0f;                      /* netcdf_io:1390 */
  real* dst3[5] = {(real*)a->t3d, (real*)a->qv3d, (real*)a->u_phy, (real*)a->v_phy, (real*)a->p8w3d};
  const real* sa3[5] = {ra->t, ra->q, ra->u, ra->v, ra->p};
  const real* sb3[5] = {rb ? rb->t : 0, rb ? rb->q : 0, rb ? rb->u : 0, rb ? rb->v : 0, rb ? rb->p : 0};
  for (int j = a->jts; j <= a->jte; j++)
    for (int i = a->its; i <= a->ite; i++) {
      size_t ij = (size_t)(j - a->jms) * ni + (i - a->ims);
      size_t l1 = ((size_t)(j - a->jms) * nka + k1) * ni + (i - a->ims);
      for (int f = 0; f < 5; f++)
        dst3[f][l1] = rb ? (sa3[f][ij] * fraction) + (sb3[f][ij] * (1.0f - fraction)) : sa3[f][ij];
      ((real*)a->glw)[ij] = rb ? (ra->lw[ij] * fraction) + (rb->lw[ij] * (1.0f - fraction)) : ra->lw[ij];
      ((real*)a->swdown)[ij] = rb ? (ra->sw[ij] * fraction) + (rb->sw[ij] * (1.0f - fraction)) : ra->sw[ij];
      rain_rate_out[ij] = ra->pcp[ij];                                                 /* netcdf_io:1398 */
      if (ra->fpar) ((real*)a->vegfra)[ij] = ra->fpar[ij];
      if (ra->lai) ((real*)a->xlaixy)[ij] = ra->lai[ij];
    }
  return 0;
}

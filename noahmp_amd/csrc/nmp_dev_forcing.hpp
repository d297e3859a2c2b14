// Forcing preparation on the device: what the reference driver does on the host between reading a
// forcing file and calling noahmplsm (driver/module_hrldas_noahmp_driver.F90:336-354, "hdrv"): the level-2 copies
// of the atmospheric fields, RAINBL = rate * DT, VEGFRA in percent, DZ8W = 2*ZLVL and the per-cell cosine of the
// solar zenith angle of CALC_DECLIN (hdrv:813-863).  One thread per cell, pure streaming.
#pragma once
#include "nmp_dev_common.hpp"

namespace nmp {

struct ForcingArgs {
  noahmp_step_args a;              // device pointers: t3d qv3d u_phy v_phy p8w3d (level 1 holds the data), rainbl, vegfra,
                                   // dz8w, coszin, xlatin
  const float* __restrict__ lon;   // LON2D (degrees east)
  const float* __restrict__ rain_rate;   // RAINBL_tmp [mm/s]
  float hour_utc;                  // IHOUR + IMINUTE/60 + ISECOND/3600 (hdrv:856), float32
  float sin_declin, cos_declin;    // of the solar declination (uniform over the grid, hdrv:839-854)
  float dt, dz8w;                  // model time step [s]; 2*ZLVL (hdrv:345-346)
  int scale_vegfra;                // VEGFRA arrives as a fraction and is stored in percent (hdrv:337)
  int first_step;                  // itime == 1 of a cold start: first guesses of EAH, TAH, CH, CM (hdrv:374-384)
  int ni, nka, k1;                 // memory extents and the slot of level 1, as in the column kernel
};

constexpr float NMP_DEGRAD = 3.14159265f / 180.f;   // hdrv:815

// hdrv:856-859 for one cell
NMP_DEV float forcing_cosz(float lat, float lon, float hour_utc, float sin_declin, float cos_declin) {
  float tloctim = hour_utc + lon / 15.0f;
  tloctim = fmodf(tloctim + 24.0f, 24.0f);
  const float hrang = 15.f * (tloctim - 12.f) * NMP_DEGRAD;
  return libm::sinf_(lat * NMP_DEGRAD) * sin_declin + libm::cosf_(lat * NMP_DEGRAD) * cos_declin * libm::cosf_(hrang);
}

NMP_DEV void forcing_cell(const ForcingArgs& k, int ii, int jj) {
  const noahmp_step_args& a = k.a;
  const size_t ij = (size_t)jj * k.ni + ii;
  const size_t l1 = ((size_t)jj * k.nka + k.k1) * k.ni + ii, l2 = l1 + k.ni;     // (i, 1, j), (i, 2, j)
  float* t3d = const_cast<float*>(a.t3d); float* qv3d = const_cast<float*>(a.qv3d);
  float* u = const_cast<float*>(a.u_phy); float* v = const_cast<float*>(a.v_phy);
  float* p = const_cast<float*>(a.p8w3d); float* dz = const_cast<float*>(a.dz8w);
  p[l2] = p[l1]; t3d[l2] = t3d[l1]; u[l2] = u[l1]; v[l2] = v[l1]; qv3d[l2] = qv3d[l1];   // hdrv:339-343
  const_cast<float*>(a.rainbl)[ij] = k.rain_rate[ij] * k.dt;                              // hdrv:344
  dz[l1] = k.dz8w; dz[l2] = k.dz8w;                                                       // hdrv:345
  if (k.scale_vegfra) const_cast<float*>(a.vegfra)[ij] = a.vegfra[ij] * 100.0f;           // hdrv:337
  if (k.first_step) {                                                                     // hdrv:376-383
    const_cast<float*>(a.eahxy)[ij] = (p[l1] * qv3d[l1]) / (0.622f + qv3d[l1]);
    const_cast<float*>(a.tahxy)[ij] = t3d[l1];
    const_cast<float*>(a.chxy)[ij] = 0.1f;
    const_cast<float*>(a.cmxy)[ij] = 0.1f;
  }
  const_cast<float*>(a.coszin)[ij] = forcing_cosz(a.xlatin[ij], k.lon[ij], k.hour_utc, k.sin_declin, k.cos_declin);
}

// Temporal interpolation between two forcing records (driver/module_hrldas_netcdf_io.F90:1369-1403, hrldas_input_interpolate;
// has_b = 0: hrldas_input_copy, netcdf_io:1351-1366).  Targets are the arrays hrldas_input_read fills (hdrv:331-335).
struct InterpArgs {
  noahmp_step_args a;              // device pointers: t3d qv3d u_phy v_phy p8w3d (level 1 is written), glw, swdown, vegfra, xlaixy (the driver's LAI, hdrv:403)
  noahmp_forcing_record ra, rb;
  float* __restrict__ rain_rate;   // RAINBL_tmp
  float fraction, one_minus;       // netcdf_io:1390 and (1.0-fraction), both float32
  int has_b;
  int ni, nka, k1;
};

NMP_DEV float interp2(float xa, float xb, float f, float g) { return (xa * f) + (xb * g); }   // netcdf_io:1391-1397

NMP_DEV void interp_cell(const InterpArgs& k, int ii, int jj) {
  const noahmp_step_args& a = k.a;
  const size_t ij = (size_t)jj * k.ni + ii;
  const size_t l1 = ((size_t)jj * k.nka + k.k1) * k.ni + ii;
  float t = k.ra.t[ij], q = k.ra.q[ij], u = k.ra.u[ij], v = k.ra.v[ij], p = k.ra.p[ij], lw = k.ra.lw[ij], sw = k.ra.sw[ij];
  if (k.has_b) {
    t = interp2(t, k.rb.t[ij], k.fraction, k.one_minus);
    q = interp2(q, k.rb.q[ij], k.fraction, k.one_minus);
    u = interp2(u, k.rb.u[ij], k.fraction, k.one_minus);
    v = interp2(v, k.rb.v[ij], k.fraction, k.one_minus);
    p = interp2(p, k.rb.p[ij], k.fraction, k.one_minus);
    lw = interp2(lw, k.rb.lw[ij], k.fraction, k.one_minus);
    sw = interp2(sw, k.rb.sw[ij], k.fraction, k.one_minus);
  }
  const_cast<float*>(a.t3d)[l1] = t; const_cast<float*>(a.qv3d)[l1] = q;
  const_cast<float*>(a.u_phy)[l1] = u; const_cast<float*>(a.v_phy)[l1] = v; const_cast<float*>(a.p8w3d)[l1] = p;
  const_cast<float*>(a.glw)[ij] = lw; const_cast<float*>(a.swdown)[ij] = sw;
  k.rain_rate[ij] = k.ra.pcp[ij];                                                    // netcdf_io:1398: not interpolated
  if (k.ra.fpar) const_cast<float*>(a.vegfra)[ij] = k.ra.fpar[ij];                   // netcdf_io:1399
  if (k.ra.lai) const_cast<float*>(a.xlaixy)[ij] = k.ra.lai[ij];                      // netcdf_io:1400
}

}  // namespace nmp

// Miguez-Macho & Fan groundwater step for MI355X -- device functions.
//
// Replaces WTABLE_mmf_noahmp (reference phys/module_sf_noahmp_groundwater.F90:14-198, "gw") with its
// LATERALFLOW (gw:201-295) and UPDATEWTD (gw:298-606).  This is the one place where Noah-MP columns talk
// to each other: a 9-point stencil over the water-table head.  Two kernels (noahmp_groundwater.hip):
//   1. gw_cell_head : KCELL / HEAD on the tile plus a 1-cell ring (gw:237-252) -> scratch planes in HBM
//   2. gw_column    : QLAT stencil (gw:259-292), river flux (gw:114-129), deep recharge (gw:147-161),
//                     UPDATEWTD and the accumulators (gw:186-195), one thread per cell.
// The split is a correctness requirement, not a convenience: kernel 2 overwrites WTD in place while its
// neighbours' stencils need the OLD water table, so the old head has to be frozen first (the reference
// freezes it in the KCELL/HEAD locals for the same reason).  The lateral half of the split form (sorted
// runs: gw_qlat_fused_kernel) writes QLAT into a plane of its own and no WTD, so it keeps KCELL / HEAD in LDS.
//
// The four soil layers stay in registers: UPDATEWTD's runtime layer indices (IWTD/KWTD) are resolved by
// fully unrolled, predicated sweeps instead of dynamically indexed arrays (which would go to scratch).
#pragma once
#include "nmp_dev_common.hpp"

namespace nmp {

struct GwArgs {
  noahmp_wtable_args a;
  const noahmp_tables* __restrict__ T;
  float* __restrict__ kcell;       // scratch planes, memory-tile shaped (ims:ime, jms:jme)
  float* __restrict__ head;
  float deltat;                    // gw:89
  float zsoil[NSOIL + 1];          // zsoil[0..4], gw:91-95
  float dzs[NSOIL];
  int ni;                          // ime-ims+1
  int hi0, hi1, hj0, hj1;          // KCELL/HEAD rectangle, gw:231-234
  int qi0, qi1, qj0, qj1;          // QLAT rectangle, gw:254-257
  unsigned long long* err;
  int* counts;
  float* qlat;                     // split form (sorted layout): QLAT plane written by gw_qlat_fused_kernel / read by gw_column_t<false>
};

// KLATFACTOR, gw:224-225 (indexed by soil category 1..19)
NMP_DEV float gw_klatfactor(int st) {
  switch (st) {
    case 1: return 2.f;   case 2: return 3.f;   case 3: return 4.f;   case 4: return 10.f;
    case 5: return 10.f;  case 6: return 12.f;  case 7: return 14.f;  case 8: return 20.f;
    case 9: return 24.f;  case 10: return 28.f; case 11: return 40.f; case 12: return 48.f;
    case 13: return 2.f;  case 14: return 0.f;  case 15: return 10.f; case 16: return 0.f;
    case 17: return 20.f; case 18: return 2.f;  case 19: return 2.f;
  }
  return 0.f;
}

// gw:239-250 for one cell
NMP_DEV void gw_cell_head_values(const GwArgs& g, float fdepth, float wtd, float topo, int st, float& kcell, float& head) {
  float kc = 0.f;
  if (fdepth > 0.f) {
    const float satdk = (st >= 1 && st <= 30 /* NSLTYPE, lsm:83 */) ? g.T->satdk[st - 1] : 0.f;
    const float klat = satdk * gw_klatfactor(st);
    if (wtd < -1.5f) kc = fdepth * klat * nmp_expf((wtd + 1.5f) / fdepth);
    else kc = klat * (wtd + 1.5f + fdepth);
  }
  kcell = kc;
  head = topo + wtd;
}
NMP_DEV void gw_cell_head(const GwArgs& g, size_t x) {
  float kc, hd;
  gw_cell_head_values(g, g.a.fdepth[x], g.a.wtd[x], g.a.topo[x], g.a.isltyp[x], kc, hd);
  g.kcell[x] = kc;
  g.head[x] = hd;
}

// register-resident 1-based views of the NSOIL=4 layer arrays
struct Soil4 { float v[NSOIL]; };

struct GwCol {            // UPDATEWTD in/out scalars
  float totwater, wtd, smcwtd, qspring;
  Soil4 smc, sh2o;
};

// The three "fill layers upward until the water is used up" loops of UPDATEWTD (gw:370-388, 412-428,
// 459-477).  Sweep k = kstart..1; `use_min` selects the MIN(...,ZSOIL(IWTD)) clamp that the first two
// carry and the deep-table one (gw:468) does not; `keep_first` reproduces gw:357-365, where the layer that
// holds the water table is tried before any WTD = ZSOIL(K) assignment.  Ends with wtd = zsoil(0) = 0 when
// every layer filled.
NMP_DEV void gw_fill_up(GwCol& c, const Soil4& smceq, const float* zsoil, const float* dzs, float smcmax,
                        int kstart, bool use_min, bool keep_first) {
  bool active = true;
#pragma unroll
  for (int k = NSOIL; k >= 1; k--) {
    if (active && k <= kstart) {
      if (!(keep_first && k == kstart)) c.wtd = zsoil[k];
      const float maxwatup = dzs[k - 1] * (smcmax - c.smc.v[k - 1]);
      if (c.totwater <= maxwatup) {
        float s = c.smc.v[k - 1] + c.totwater / dzs[k - 1];
        s = fmin2(s, smcmax);
        c.smc.v[k - 1] = s;
        if (s > smceq.v[k - 1]) {
          const float w = (s * dzs[k - 1] - smceq.v[k - 1] * zsoil[k - 1] + smcmax * zsoil[k]) /
                          (smcmax - smceq.v[k - 1]);
          c.wtd = use_min ? fmin2(w, zsoil[k - 1]) : w;
        }
        c.totwater = 0.f;
        active = false;
      } else {
        c.smc.v[k - 1] = smcmax;
        c.totwater = c.totwater - maxwatup;
      }
    }
  }
  if (active) c.wtd = zsoil[0];   // k == 0 exit: gw:371-373
}

NMP_DEV float gw_smceqdeep(float smcmax, float psisat, float bexp, float dz) {   // gw:395-398 and twins
  const float e = smcmax * nmp_powf(psisat / (psisat - dz), 1.f / bexp);
  return fmax2(e, 1.E-4f);
}

// UPDATEWTD, gw:298-606
NMP_DEV void gw_updatewtd(GwCol& c, const Soil4& smceq, const float* zsoil, const float* dzs, float smcmax,
                          float psisat, float bexp) {
  Soil4 sice;
#pragma unroll
  for (int k = 0; k < NSOIL; k++) sice.v[k] = c.smc.v[k] - c.sh2o.v[k];          // gw:340
  c.qspring = 0.f;
  const float zbot = zsoil[NSOIL], dzn = dzs[NSOIL - 1];

  if (c.totwater > 0.f) {                                                         // gw:345 rising
    if (c.wtd >= zbot) {                                                          // gw:348
      int iwtd = 0;                                                               // gw:350-353
#pragma unroll
      for (int k = 1; k <= NSOIL - 1; k++) if (c.wtd < zsoil[k]) iwtd = k;        // last k (largest) wins
      // the reference scans k = nsoil-1..1 and exits at the first (deepest) k with wtd < zsoil(k)
      gw_fill_up(c, smceq, zsoil, dzs, smcmax, iwtd + 1, true, true);                   // gw:357-390 (kwtd, then k1..1)
    } else if (c.wtd >= zbot - dzn) {                                             // gw:392
      const float smceqdeep = gw_smceqdeep(smcmax, psisat, bexp, dzn);
      const float maxwatup = (smcmax - c.smcwtd) * dzn;
      if (c.totwater <= maxwatup) {
        c.smcwtd = c.smcwtd + c.totwater / dzn;
        c.smcwtd = fmin2(c.smcwtd, smcmax);
        if (c.smcwtd > smceqdeep)
          c.wtd = fmin2((c.smcwtd * dzn - smceqdeep * zbot + smcmax * (zbot - dzn)) / (smcmax - smceqdeep), zbot);
        c.totwater = 0.f;
      } else {
        c.smcwtd = smcmax;
        c.totwater = c.totwater - maxwatup;
        gw_fill_up(c, smceq, zsoil, dzs, smcmax, NSOIL, true, false);                    // gw:412-428
      }
    } else {                                                                      // gw:432 deep table
      float maxwatup = (smcmax - c.smcwtd) * (zbot - dzn - c.wtd);
      if (c.totwater <= maxwatup) {
        c.wtd = c.wtd + c.totwater / (smcmax - c.smcwtd);
        c.totwater = 0.f;
      } else {
        c.totwater = c.totwater - maxwatup;
        c.wtd = zbot - dzn;
        maxwatup = (smcmax - c.smcwtd) * dzn;
        if (c.totwater <= maxwatup) {
          const float smceqdeep = gw_smceqdeep(smcmax, psisat, bexp, dzn);
          c.smcwtd = c.smcwtd + c.totwater / dzn;
          c.smcwtd = fmin2(c.smcwtd, smcmax);
          c.wtd = (c.smcwtd * dzn - smceqdeep * zbot + smcmax * (zbot - dzn)) / (smcmax - smceqdeep);
          c.totwater = 0.f;
        } else {
          c.smcwtd = smcmax;
          c.totwater = c.totwater - maxwatup;
          gw_fill_up(c, smceq, zsoil, dzs, smcmax, NSOIL, false, false);                 // gw:459-477
        }
      }
    }
    c.qspring = c.totwater;                                                       // gw:483
  } else if (c.totwater < 0.f) {                                                  // gw:486 falling
    bool deep = false;       // run the below-the-soil-column block (gw:525-552 / 556-583)
    if (c.wtd >= zbot) {                                                          // gw:489
      int iwtd = 0;
#pragma unroll
      for (int k = 1; k <= NSOIL - 1; k++) if (c.wtd < zsoil[k]) iwtd = k;
      const int k1 = iwtd + 1;
      bool active = true;
#pragma unroll
      for (int kw = 1; kw <= NSOIL; kw++) {                                       // gw:497-523; iwtd == kw-1 here
        if (active && kw >= k1) {
          const float maxwatdw = dzs[kw - 1] * (c.smc.v[kw - 1] - fmax2(smceq.v[kw - 1], sice.v[kw - 1]));
          if (-c.totwater <= maxwatdw) {
            const float s = c.smc.v[kw - 1] + c.totwater / dzs[kw - 1];
            c.smc.v[kw - 1] = s;
            if (s > smceq.v[kw - 1]) {
              c.wtd = (s * dzs[kw - 1] - smceq.v[kw - 1] * zsoil[kw - 1] + smcmax * zsoil[kw]) /
                      (smcmax - smceq.v[kw - 1]);
            } else {
              c.wtd = zsoil[kw];
              iwtd = iwtd + 1;
            }
            c.totwater = 0.f;
            active = false;
          } else {
            c.wtd = zsoil[kw];
            iwtd = iwtd + 1;
            if (maxwatdw >= 0.f) {
              c.smc.v[kw - 1] = c.smc.v[kw - 1] + maxwatdw / dzs[kw - 1];
              c.totwater = c.totwater + maxwatdw;
            }
          }
        }
      }
      deep = (iwtd == NSOIL && c.totwater < 0.f);                                 // gw:525
    } else if (c.wtd >= zbot - dzn) {                                             // gw:556
      deep = true;
    } else {                                                                      // gw:585-595
      float wgpmid = smcmax * nmp_powf(psisat / (psisat - (zbot - c.wtd)), 1.f / bexp);
      wgpmid = fmax2(wgpmid, 1.E-4f);
      const float syielddw = smcmax - wgpmid;
      const float wtdold = c.wtd;
      c.wtd = wtdold + c.totwater / syielddw;
      c.smcwtd = (c.smcwtd * (zbot - wtdold) + wgpmid * (wtdold - c.wtd)) / (zbot - c.wtd);
    }
    if (deep) {                                                                   // gw:526-550 == gw:560-583
      const float smceqdeep = gw_smceqdeep(smcmax, psisat, bexp, dzn);
      const float maxwatdw = dzn * (c.smcwtd - smceqdeep);
      if (-c.totwater <= maxwatdw) {
        c.smcwtd = c.smcwtd + c.totwater / dzn;
        c.wtd = fmax2((c.smcwtd * dzn - smceqdeep * zbot + smcmax * (zbot - dzn)) / (smcmax - smceqdeep),
                      zbot - dzn);
      } else {
        c.wtd = zbot - dzn;
        c.smcwtd = c.smcwtd + c.totwater / dzn;
        const float dzup = (smceqdeep - c.smcwtd) * dzn / (smcmax - smceqdeep);
        c.wtd = c.wtd - dzup;
        c.smcwtd = smceqdeep;
      }
    }
    c.qspring = 0.f;
  }
#pragma unroll
  for (int k = 0; k < NSOIL; k++) c.sh2o.v[k] = c.smc.v[k] - sice.v[k];          // gw:603
}

// One cell of gw:97-195.  (i,j) are 0-based offsets into the memory tile; x = j*ni + i.
// Returns 1 for a land cell, 0 otherwise.
//
// Written as load-everything / compute / store-everything: the ~47 plane reads of a cell are independent
// of each other, so they are all issued before the first use (one HBM round trip per wave instead of a
// chain of dependent ones behind the land-mask and regime branches), and nothing is stored until the end,
// so the compiler never has to order a load behind a possibly aliasing store.
// the 9-point stencil of LATERALFLOW (gw:259-292) for one cell: QLAT [m] over DELTAT; zero outside the QLAT rectangle (gw:254-257)
// the sum of gw:259-292 in the reference's order (ul, l, dl, u, d, ur, r, dr), times FANGLE * DELTAT / AREA
NMP_DEV float gw_qlat_sum(float kc, float hd, float k_ul, float k_l, float k_dl, float k_u, float k_d, float k_ur, float k_r, float k_dr,
                          float h_ul, float h_l, float h_dl, float h_u, float h_d, float h_ur, float h_r, float h_dr, float deltat, float area) {
  const float SQRT2 = 1.41421354f;          // SQRT(2.) in float32
  float q = 0.f;
  q = q + (k_ul + kc) * (h_ul - hd) / SQRT2;
  q = q + (k_l + kc) * (h_l - hd);
  q = q + (k_dl + kc) * (h_dl - hd) / SQRT2;
  q = q + (k_u + kc) * (h_u - hd);
  q = q + (k_d + kc) * (h_d - hd);
  q = q + (k_ur + kc) * (h_ur - hd) / SQRT2;
  q = q + (k_r + kc) * (h_r - hd);
  q = q + (k_dr + kc) * (h_dr - hd) / SQRT2;
  return 0.45508986056f * q * deltat / area;                                    // FANGLE, gw:229
}
NMP_DEV float gw_qlat_stencil(const GwArgs& g, size_t x, bool inq, float area) {
  const int ni = g.ni;
  // outside the QLAT rectangle the offsets collapse onto the cell itself so that the loads stay inside the caller's memory
  const size_t up = inq ? x + ni : x, dn = inq ? x - ni : x, e = inq ? 1 : 0;
  const float kc = g.kcell[x], hd = g.head[x];
  const float k_ul = g.kcell[up - e], k_l = g.kcell[x - e], k_dl = g.kcell[dn - e], k_u = g.kcell[up],
              k_d = g.kcell[dn], k_ur = g.kcell[up + e], k_r = g.kcell[x + e], k_dr = g.kcell[dn + e];
  const float h_ul = g.head[up - e], h_l = g.head[x - e], h_dl = g.head[dn - e], h_u = g.head[up],
              h_d = g.head[dn], h_ur = g.head[up + e], h_r = g.head[x + e], h_dr = g.head[dn + e];
  if (!inq) return 0.f;
  return gw_qlat_sum(kc, hd, k_ul, k_l, k_dl, k_u, k_d, k_ur, k_r, k_dr, h_ul, h_l, h_dl, h_u, h_d, h_ur, h_r, h_dr, g.deltat, area);
}
NMP_DEV bool gw_is_land(const noahmp_wtable_args& a, float xland, float xice, int ivgtyp) {
  return (xland - 1.5f < 0.f) && (xice < a.xice_threshold) && (ivgtyp != a.isice);                // gw:97-101
}
// STENCIL = true: the whole cell update (stencil included; tile order).  false: the per-column half of the split form -- QLAT comes from
// the plane g.qlat (same column order as the other arrays, any order), nothing else of the call depends on the neighbours.
template <bool STENCIL>
NMP_DEV int gw_column_t(const GwArgs& g, int i, int j, int gi, int gj) {
  const noahmp_wtable_args& a = g.a;
  const int ni = g.ni;
  const size_t x = (size_t)j * ni + i;
  const size_t plane = (size_t)ni, x3 = ((size_t)j * NSOIL) * ni + i;
  // ---- loads
  const float xland = a.xland[x], xice = a.xice[x];
  const int ivgtyp = a.ivgtyp[x], sl = a.isltyp[x];
  const float area = a.area[x], riverbed = a.riverbed[x], eqwtd = a.eqwtd[x], rivercond = a.rivercond[x],
              pexp = a.pexp[x];
  const float wtd = a.wtd[x], smcwtd0 = a.smcwtd[x], deeprech0 = a.deeprech[x], qspring0 = a.qspring[x];
  const float qslat0 = a.qslat[x], qrfs0 = a.qrfs[x], qsprings0 = a.qsprings[x], rech0 = a.rech[x];
  GwCol c;
  Soil4 smceq;
#pragma unroll
  for (int k = 0; k < NSOIL; k++) {
    c.smc.v[k] = a.smois[x3 + k * plane];
    c.sh2o.v[k] = a.sh2oxy[x3 + k * plane];
    smceq.v[k] = a.smoiseq[x3 + k * plane];
  }
  // stencil operands (whole form) or the QLAT plane (split form)
  const bool inq = STENCIL && (gi >= g.qi0 && gi <= g.qi1 && gj >= g.qj0 && gj <= g.qj1);
  const float qlat_in = STENCIL ? gw_qlat_stencil(g, x, inq, area) : g.qlat[x];

  // ---- compute
  const bool land = gw_is_land(a, xland, xice, ivgtyp);
  float qlat = 0.f, qrf = 0.f, deeprech = deeprech0;
  float qspring = qspring0;                     // non-land cells keep the caller's value (gw:172-174)
  if (land) {
    qlat = qlat_in;                                                               // gw:259-292
    {                                                                             // gw:116-124
      float rcond = rivercond;
      if (wtd > riverbed && eqwtd > riverbed) rcond = rcond * nmp_expf(pexp * (wtd - eqwtd));
      qrf = rcond * (wtd - riverbed) * g.deltat / area;
      qrf = fmax2(qrf, 0.f);
    }
    const int sli = (sl >= 1 && sl <= 30) ? sl - 1 : 0;
    const float bexp = g.T->bb[sli], dksat = g.T->satdk[sli];
    float smcmax = g.T->maxsmc[sli];
    const float psisat = -g.T->satpsi[sli];
    if (ivgtyp == a.isurban) smcmax = 0.45f;                                      // gw:141-144
    c.smcwtd = smcwtd0;
    const float zbot = g.zsoil[NSOIL], dzn = g.dzs[NSOIL - 1];
    if (wtd < zbot - dzn) {                                                       // gw:147-161
      const float ddz = zbot - wtd;
      const float smcwtdmid = 0.5f * (c.smcwtd + smcmax);
      const float psi = psisat * nmp_powf(smcmax / c.smcwtd, bexp);
      const float wcnddeep = dksat * nmp_powf(smcwtdmid / smcmax, 2.0f * bexp + 3.0f);
      float wfluxdeep = -g.deltat * wcnddeep * ((psisat - psi) / ddz - 1.f);
      c.smcwtd = c.smcwtd + (deeprech - wfluxdeep) / ddz;
      const float wplus = fmax2(c.smcwtd - smcmax, 0.0f) * ddz;
      const float wminus = fmax2(1.E-4f - c.smcwtd, 0.0f) * ddz;
      c.smcwtd = fmax2(fmin2(c.smcwtd, smcmax), 1.E-4f);
      wfluxdeep = wfluxdeep + wplus - wminus;
      deeprech = wfluxdeep;
    }
    c.totwater = qlat - qrf + deeprech;                                           // gw:165
    c.wtd = wtd;
    gw_updatewtd(c, smceq, g.zsoil, g.dzs, smcmax, psisat, bexp);                 // gw:172-174
    qspring = c.qspring;
  }
  // ---- stores
  if (land) {
#pragma unroll
    for (int k = 0; k < NSOIL; k++) {
      a.smois[x3 + k * plane] = c.smc.v[k];
      a.sh2oxy[x3 + k * plane] = c.sh2o.v[k];
    }
    a.wtd[x] = c.wtd;
    a.smcwtd[x] = c.smcwtd;
    a.qspring[x] = qspring;
  }
  a.qrf[x] = qrf;                                                                 // gw:122-126
  a.qslat[x] = qslat0 + qlat * 1.E3f;                                             // gw:188-193
  a.qrfs[x] = qrfs0 + qrf * 1.E3f;
  a.qsprings[x] = qsprings0 + qspring * 1.E3f;
  a.rech[x] = rech0 + deeprech * 1.E3f;
  a.deeprech[x] = 0.f;
  return land ? 1 : 0;
}
NMP_DEV int gw_column(const GwArgs& g, int i, int j, int gi, int gj) { return gw_column_t<true>(g, i, j, gi, gj); }


// ---- GROUNDWATER_INIT + EQSMOISTURE (reference phys/module_sf_noahmpdrv.F90:1286-1522, "drv"): the one-time
// equilibrium set-up of the MMF scheme.  Same two-kernel shape as the time step (KCELL/HEAD first, then one thread
// per cell); only the land mask differs (IVGTYP /= ISWATER and /= ISICE, drv:1340-1344).

// EQSMOISTURE drv:1477-1522: Newton solve for the equilibrium moisture of each soil layer
NMP_DEV void gw_eqsmoisture(const float* zsoil /*0..4*/, float smcmax, float dwsat, float dksat, float bexp, Soil4& smceq) {
#pragma unroll
  for (int k = 1; k <= NSOIL; k++) {
    float ddz;
    if (k == 1) ddz = -zsoil[k + 1] * 0.5f;
    else if (k < NSOIL) ddz = (zsoil[k - 1] - zsoil[k + 1]) * 0.5f;
    else ddz = zsoil[k - 1] - zsoil[k];
    const float expon = bexp + 1.f;
    const float aa = dwsat / ddz;
    const float bb = dksat / nmp_powf(smcmax, expon);
    float smc = 0.5f * smcmax;
#pragma unroll 1
    for (int iter = 1; iter <= 100; iter++) {
      const float func = (smc - smcmax) * aa + bb * nmp_powf(smc, expon);
      const float dfunc = aa + bb * expon * nmp_powf(smc, bexp);
      const float dx = func / dfunc;
      smc = smc - dx;
      if (fabsf(dx) < 1.E-6f) break;
    }
    smceq.v[k - 1] = fmin2(fmax2(smc, 1.E-4f), smcmax * 0.99f);
  }
}

// one cell of drv:1340-1470.  (i,j): offsets into the memory tile; (gi,gj): Fortran indices.
NMP_DEV void gw_init_column(const GwArgs& g, int i, int j, int gi, int gj, int iswater) {
  const noahmp_wtable_args& a = g.a;
  const int ni = g.ni;
  const size_t x = (size_t)j * ni + i;
  const size_t plane = (size_t)ni, x3 = ((size_t)j * NSOIL) * ni + i;
  float* smoiseq = const_cast<float*>(a.smoiseq);               // INOUT here (drv:1311), IN for the time step
  const int ivgtyp = a.ivgtyp[x], sl = a.isltyp[x];
  const bool land = (ivgtyp != iswater) && (ivgtyp != a.isice);
  float wtd = a.wtd[x];
  float qlat = 0.f, qrf = 0.f;
  if (land) {
    const float area = a.area[x];
    if (gi >= g.qi0 && gi <= g.qi1 && gj >= g.qj0 && gj <= g.qj1) {             // LATERALFLOW gw:259-292
      const float SQRT2 = 1.41421354f;
      const float kc = g.kcell[x], hd = g.head[x];
      const size_t up = x + ni, dn = x - ni;
      float q = 0.f;
      q = q + (g.kcell[up - 1] + kc) * (g.head[up - 1] - hd) / SQRT2;
      q = q + (g.kcell[x - 1] + kc) * (g.head[x - 1] - hd);
      q = q + (g.kcell[dn - 1] + kc) * (g.head[dn - 1] - hd) / SQRT2;
      q = q + (g.kcell[up] + kc) * (g.head[up] - hd);
      q = q + (g.kcell[dn] + kc) * (g.head[dn] - hd);
      q = q + (g.kcell[up + 1] + kc) * (g.head[up + 1] - hd) / SQRT2;
      q = q + (g.kcell[x + 1] + kc) * (g.head[x + 1] - hd);
      q = q + (g.kcell[dn + 1] + kc) * (g.head[dn + 1] - hd) / SQRT2;
      qlat = 0.45508986056f * q * g.deltat / area;
    }
    const float riverbed = a.riverbed[x], eqwtd = a.eqwtd[x];                  // drv:1356-1370
    float rcond = a.rivercond[x];
    if (wtd > riverbed && eqwtd > riverbed) rcond = rcond * nmp_expf(a.pexp[x] * (wtd - eqwtd));
    qrf = rcond * (wtd - riverbed) * g.deltat / area;
    qrf = fmax2(qrf, 0.f);
  }
  const int sli = (sl >= 1 && sl <= 30) ? sl - 1 : 0;
  const float bx = g.T->bb[sli], dwsat = g.T->satdw[sli], dksat = g.T->satdk[sli], psisat = -g.T->satpsi[sli];
  float smcmax = g.T->maxsmc[sli];
  if (ivgtyp == a.isurban) smcmax = 0.45f;                                     // drv:1378-1381
  float smcwtd;
  const float zbot = g.zsoil[NSOIL], dzn = g.dzs[NSOIL - 1];
  if (bx > 0.0f && smcmax > 0.0f && -psisat > 0.0f) {
    Soil4 smceq;
    gw_eqsmoisture(g.zsoil, smcmax, dwsat, dksat, bx, smceq);
#pragma unroll
    for (int k = 0; k < NSOIL; k++) smoiseq[x3 + k * plane] = smceq.v[k];
    if (wtd < zbot - dzn) {                                                    // drv:1393-1417 deep table: Newton
      const float expon = 2.f * bx + 3.f;
      const float ddz = zbot - wtd;
      const float cc = psisat / ddz;
      const float flux = (qlat - qrf) / g.deltat;
      float smc = 0.5f * smcmax;
#pragma unroll 1
      for (int iter = 1; iter <= 100; iter++) {
        const float dd = (smc + smcmax) / (2.f * smcmax);
        const float aa = -dksat * nmp_powf(dd, expon);
        const float bbb = cc * (nmp_powf(smcmax / smc, bx) - 1.f) + 1.f;
        const float func = aa * bbb - flux;
        const float dfunc = -dksat * (expon / (2.f * smcmax)) * nmp_powf(dd, expon - 1.f) * bbb +
                            aa * cc * (-bx) * nmp_powf(smcmax, bx) * nmp_powf(smc, -bx - 1.f);
        const float dx = func / dfunc;
        smc = smc - dx;
        if (fabsf(dx) < 1.E-6f) break;
      }
      smcwtd = fmax2(smc, 1.E-4f);
    } else if (wtd < zbot) {                                                   // drv:1419-1424
      float smceqdeep = smcmax * nmp_powf(psisat / (psisat - dzn), 1.f / bx);
      smceqdeep = fmax2(smceqdeep, 1.E-4f);
      smcwtd = smcmax * (wtd - (zbot - dzn)) + smceqdeep * (zbot - wtd);
    } else {                                                                   // drv:1426-1444 table inside the soil column
      smcwtd = smcmax;
      bool active = true;
#pragma unroll
      for (int k = NSOIL; k >= 2; k--) {
        if (active) {
          const float smk = a.smois[x3 + (k - 1) * plane];
          if (wtd >= g.zsoil[k - 1]) {
            const float frliq = a.sh2oxy[x3 + (k - 1) * plane] / smk;
            a.smois[x3 + (k - 1) * plane] = smcmax;
            a.sh2oxy[x3 + (k - 1) * plane] = smcmax * frliq;
          } else {
            if (smk < smceq.v[k - 1]) wtd = g.zsoil[k];
            else wtd = (smk * g.dzs[k - 1] - smceq.v[k - 1] * g.zsoil[k - 1] + smcmax * g.zsoil[k]) /
                       (smcmax - smceq.v[k - 1]);
            active = false;
          }
        }
      }
    }
  } else {                                                                     // drv:1446-1450
#pragma unroll
    for (int k = 0; k < NSOIL; k++) smoiseq[x3 + k * plane] = smcmax;
    smcwtd = smcmax;
    wtd = 0.f;
  }
  a.wtd[x] = wtd;
  a.smcwtd[x] = smcwtd;
  a.deeprech[x] = 0.f; a.rech[x] = 0.f; a.qslat[x] = 0.f; a.qrfs[x] = 0.f; a.qsprings[x] = 0.f;   // drv:1453-1457
}

}  // namespace nmp

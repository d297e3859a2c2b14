// Noah-MP column engine for MI355X -- water phase device code.
// Follows WATER and its callees in the reference (phys/module_sf_noahmplsm.F90, "lsm").
// The snow-layer bookkeeping (COMBINE / DIVIDE / COMPACT / SNOWH2O) is the branchy, tiny-arithmetic
// part: it addresses the layer arrays with run-time indices, which is what the [layer][thread]
// LDS layout of `Lay` is for.  The soil-moisture solve works on the 4 fixed soil rows and is
// fully unrolled into registers.
#pragma once
#include "nmp_dev_common.hpp"

namespace nmp {

// CANWATER lsm:6615-6865
NMP_DEV void canwater(const Ctx& c, const Parm& P, Col& s, float& qrain, float& snowhin) {
  const float dt = c.dt, sfctmp = s.sfctmp, fveg = s.fveg;
  const double rdt = c.u.dt;
  float fp = 0.0f, qintr, qdripr, qthror, qints, qdrips, qthros, qevac, qdewc, qsubc, qfroc;
  float fpice = 0.f;
  if (c.O.snf == 1) {
    if (sfctmp > TFRZ + 2.5f) fpice = 0.f;
    else if (sfctmp <= TFRZ + 0.5f) fpice = 1.0f;
    else if (sfctmp <= TFRZ + 2.f) fpice = 1.f - (-54.632f + 0.2f * sfctmp);
    else fpice = 0.6f;
  } else if (c.O.snf == 2) {
    fpice = (sfctmp >= TFRZ + 2.2f) ? 0.f : 1.0f;
  } else {
    fpice = (sfctmp >= TFRZ) ? 0.f : 1.0f;
  }
  s.fpice = fpice;
  float bdfall = nmp_min(120.f, 67.92f + 51.25f * nmp_expf(div_rc(sfctmp - TFRZ, NMP_RCC(2.59f))));
  float rain = (s.qprecc + s.qprecl) * (1.f - fpice);
  float snow = (s.qprecc + s.qprecl) * fpice;
  if (s.qprecc + s.qprecl > 0.f) fp = (s.qprecc + s.qprecl) / (10.f * s.qprecc + s.qprecl);
  const float vai = s.elai + s.esai;
  float maxliq = P.ch2op * vai;
  if (vai > 0.f) {
    qintr = fveg * rain * fp;
    qintr = nmp_min(qintr, div_rc(maxliq - s.canliq, rdt) * (1.f - nmp_expf(-rain * dt / maxliq)));
    qintr = nmp_max(qintr, 0.f);
    qdripr = fveg * rain - qintr;
    qthror = (1.f - fveg) * rain;
  } else {
    qintr = 0.f; qdripr = 0.f; qthror = rain;
  }
  if (!s.frozen_canopy) {
    const float fcev_hvap = div_rc(s.fcev, NMP_RCC(HVAP));
    s.etran = nmp_max(div_rc(s.fctr, NMP_RCC(HVAP)), 0.f);
    qevac = nmp_max(fcev_hvap, 0.f);
    qdewc = fabsf(nmp_min(fcev_hvap, 0.f));
    qsubc = 0.f; qfroc = 0.f;
  } else {
    const float fcev_hsub = div_rc(s.fcev, NMP_RCC(HSUB));
    s.etran = nmp_max(div_rc(s.fctr, NMP_RCC(HSUB)), 0.f);
    qevac = 0.f; qdewc = 0.f;
    qsubc = nmp_max(fcev_hsub, 0.f);
    qfroc = fabsf(nmp_min(fcev_hsub, 0.f));
  }
  qevac = nmp_min(div_rc(s.canliq, rdt), qevac);
  s.canliq = nmp_max(0.f, s.canliq + (qintr + qdewc - qevac) * dt);
  if (s.canliq <= 1.E-06f) s.canliq = 0.0f;
  float maxsno = 6.6f * (0.27f + 46.f / bdfall) * vai;
  if (vai > 0.f) {
    qints = fveg * snow * fp;
    qints = nmp_min(qints, div_rc(maxsno - s.canice, rdt) * (1.f - nmp_expf(-snow * dt / maxsno)));
    qints = nmp_max(qints, 0.f);
    float ft = nmp_max(0.0f, div_rc(s.tv - 270.15f, NMP_RCC(1.87E5f)));
    float fv = div_rc(sqrtf(s.uu * s.uu + s.vv * s.vv), NMP_RCC(1.56E5f));
    qdrips = nmp_max(0.f, s.canice) * (fv + ft);
    qthros = (1.0f - fveg) * snow + (fveg * snow - qints);
  } else {
    qints = 0.f; qdrips = 0.f; qthros = snow;
  }
  qsubc = nmp_min(div_rc(s.canice, rdt), qsubc);
  s.canice = nmp_max(0.f, s.canice + (qints - qdrips) * dt + (qfroc - qsubc) * dt);
  if (s.canice <= 1.E-6f) s.canice = 0.f;
  if (s.canice > 0.f) s.fwet = nmp_max(0.f, s.canice) / nmp_max(maxsno, 1.E-06f);
  else s.fwet = nmp_max(0.f, s.canliq) / nmp_max(maxliq, 1.E-06f);
  s.fwet = nmp_powf(nmp_min(s.fwet, 1.f), 0.667f);
  if (s.canice > 1.E-6f && s.tv > TFRZ) {
    float qmeltc = nmp_min(div_rc(s.canice, rdt), div_rc(div_rc((s.tv - TFRZ) * CICE * s.canice, NMP_RCC(DENICE)), c.u.dt_hfus));
    s.canice = nmp_max(0.f, s.canice - qmeltc * dt);
    s.canliq = nmp_max(0.f, s.canliq + qmeltc * dt);
    s.tv = s.fwet * TFRZ + (1.f - s.fwet) * s.tv;
  }
  if (s.canliq > 1.E-6f && s.tv < TFRZ) {
    float qfrzc = nmp_min(div_rc(s.canliq, rdt), div_rc(div_rc((TFRZ - s.tv) * CWAT * s.canliq, NMP_RCC(DENH2O)), c.u.dt_hfus));
    s.canliq = nmp_max(0.f, s.canliq - qfrzc * dt);
    s.canice = nmp_max(0.f, s.canice + qfrzc * dt);
    s.tv = s.fwet * TFRZ + (1.f - s.fwet) * s.tv;
  }
  s.ecan = qevac + qsubc - qdewc - qfroc;
  qrain = qdripr + qthror;
  s.qsnow = qdrips + qthros;
  snowhin = s.qsnow / bdfall;
}

// COMBO lsm:7375-7424
NMP_DEV void combo(float& dz, float& wliq, float& wice, float& t, float dz2, float wliq2, float wice2,
                   float t2) {
  float dzc = dz + dz2;
  float wicec = (wice + wice2);
  float wliqc = (wliq + wliq2);
  float h = (CICE * wice + CWAT * wliq) * (t - TFRZ) + HFUS * wliq;
  float h2 = (CICE * wice2 + CWAT * wliq2) * (t2 - TFRZ) + HFUS * wliq2;
  float hc = h + h2, tc;
  if (hc < 0.f) tc = TFRZ + hc / (CICE * wicec + CWAT * wliqc);
  else if (hc <= HFUS * wliqc) tc = TFRZ;
  else tc = TFRZ + (hc - HFUS * wliqc) / (CICE * wicec + CWAT * wliqc);
  dz = dzc; wice = wicec; wliq = wliqc; t = tc;
}

// COMBINE lsm:7065-7246.  GLAC selects COMBINE_GLACIER (gla:2403-2571): DZMIN /0.045,0.05,0.2/,
// collapse below 0.05 m, PONDING1/2 accumulate, no negative-ice branch.
template <bool GLAC, class A>
NMP_DEV void combine(Col& s, const Lay<A>& y) {
  int& isnow = s.isnow;
  int isnow_old = isnow;
#pragma unroll 1
  for (int j = isnow_old + 1; j <= 0; j++) {
    if (y.snice[L(j)] <= .1f) {
      if (j != 0) {
        y.snliq[L(j + 1)] = y.snliq[L(j + 1)] + y.snliq[L(j)];
        y.snice[L(j + 1)] = y.snice[L(j + 1)] + y.snice[L(j)];
      } else {
        if (isnow_old < -1) {
          y.snliq[L(j - 1)] = y.snliq[L(j - 1)] + y.snliq[L(j)];
          y.snice[L(j - 1)] = y.snice[L(j - 1)] + y.snice[L(j)];
        } else {
          if (GLAC) {
            s.ponding1 = s.ponding1 + y.snliq[L(j)];
            s.sneqv = y.snice[L(j)];
            s.snowh = y.dzsnso[L(j)];
          } else if (y.snice[L(j)] >= 0.f) {
            s.ponding1 = y.snliq[L(j)];
            s.sneqv = y.snice[L(j)];
            s.snowh = y.dzsnso[L(j)];
          } else {
            s.ponding1 = y.snliq[L(j)] + y.snice[L(j)];
            if (s.ponding1 < 0.f) {
              y.sice[L(1)] = nmp_max(0.0f, y.sice[L(1)] + s.ponding1 / (y.dzsnso[L(1)] * 1000.f));
              s.ponding1 = 0.0f;
            }
            s.sneqv = 0.0f;
            s.snowh = 0.0f;
          }
          y.snliq[L(j)] = 0.0f; y.snice[L(j)] = 0.0f; y.dzsnso[L(j)] = 0.0f;
        }
      }
      if (j > isnow + 1 && isnow < -1) {
#pragma unroll 1
        for (int i = j; i >= isnow + 2; i--) {
          y.stc[L(i)] = y.stc[L(i - 1)];
          y.snliq[L(i)] = y.snliq[L(i - 1)];
          y.snice[L(i)] = y.snice[L(i - 1)];
          y.dzsnso[L(i)] = y.dzsnso[L(i - 1)];
        }
      }
      isnow = isnow + 1;
    }
  }
  if (y.sice[L(1)] < 0.f) { y.sh2o[L(1)] = y.sh2o[L(1)] + y.sice[L(1)]; y.sice[L(1)] = 0.f; }
  if (isnow == 0) return;
  s.sneqv = 0.f; s.snowh = 0.f;
  float zwice = 0.f, zwliq = 0.f;
#pragma unroll 1
  for (int j = isnow + 1; j <= 0; j++) {
    s.sneqv = s.sneqv + y.snice[L(j)] + y.snliq[L(j)];
    s.snowh = s.snowh + y.dzsnso[L(j)];
    zwice = zwice + y.snice[L(j)];
    zwliq = zwliq + y.snliq[L(j)];
  }
  if (s.snowh < (GLAC ? 0.05f : 0.025f) && isnow < 0) {
    isnow = 0;
    s.sneqv = zwice;
    s.ponding2 = GLAC ? (s.ponding2 + zwliq) : zwliq;
    if (s.sneqv <= 0.f) s.snowh = 0.f;
  }
  if (isnow < -1) {
    isnow_old = isnow;
    int mssi = 1;
#pragma unroll 1
    for (int i = isnow_old + 1; i <= 0; i++) {
      float dzmin = GLAC ? ((mssi == 1) ? 0.045f : (mssi == 2) ? 0.05f : 0.2f)    // gla:2438
                         : ((mssi == 3) ? 0.1f : 0.025f);                        // lsm:7104
      if (y.dzsnso[L(i)] < dzmin) {
        int neibor, j, l;
        if (i == isnow + 1) neibor = i + 1;
        else if (i == 0) neibor = i - 1;
        else {
          neibor = i + 1;
          if ((y.dzsnso[L(i - 1)] + y.dzsnso[L(i)]) < (y.dzsnso[L(i + 1)] + y.dzsnso[L(i)])) neibor = i - 1;
        }
        if (neibor > i) { j = neibor; l = i; } else { j = i; l = neibor; }
        float dz = y.dzsnso[L(j)], wl = y.snliq[L(j)], wi = y.snice[L(j)], t = y.stc[L(j)];
        combo(dz, wl, wi, t, y.dzsnso[L(l)], y.snliq[L(l)], y.snice[L(l)], y.stc[L(l)]);
        y.dzsnso[L(j)] = dz; y.snliq[L(j)] = wl; y.snice[L(j)] = wi; y.stc[L(j)] = t;
        if (j - 1 > isnow + 1) {
#pragma unroll 1
          for (int k = j - 1; k >= isnow + 2; k--) {
            y.stc[L(k)] = y.stc[L(k - 1)];
            y.snice[L(k)] = y.snice[L(k - 1)];
            y.snliq[L(k)] = y.snliq[L(k - 1)];
            y.dzsnso[L(k)] = y.dzsnso[L(k - 1)];
          }
        }
        isnow = isnow + 1;
        if (isnow >= -1) break;
      } else {
        mssi = mssi + 1;
      }
    }
  }
}

// DIVIDE lsm:7248-7371 (working copies held as scalars: at most 3 snow layers)
template <bool GLAC, class A>
NMP_DEV void divide(Col& s, const Lay<A>& y) {
  const int isnow = s.isnow;
  int msno = -isnow;
  float dz1 = 0.f, dz2 = 0.f, dz3 = 0.f, wi1 = 0.f, wi2 = 0.f, wi3 = 0.f, wl1 = 0.f, wl2 = 0.f,
        wl3 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  dz1 = y.dzsnso[L(1 + isnow)]; wi1 = y.snice[L(1 + isnow)]; wl1 = y.snliq[L(1 + isnow)]; t1 = y.stc[L(1 + isnow)];
  if (msno >= 2) { dz2 = y.dzsnso[L(2 + isnow)]; wi2 = y.snice[L(2 + isnow)]; wl2 = y.snliq[L(2 + isnow)]; t2 = y.stc[L(2 + isnow)]; }
  if (msno >= 3) { dz3 = y.dzsnso[L(3 + isnow)]; wi3 = y.snice[L(3 + isnow)]; wl3 = y.snliq[L(3 + isnow)]; t3 = y.stc[L(3 + isnow)]; }
  if (msno == 1) {
    if (dz1 > 0.05f) {
      msno = 2;
      dz1 = dz1 / 2.f; wi1 = wi1 / 2.f; wl1 = wl1 / 2.f;
      dz2 = dz1; wi2 = wi1; wl2 = wl1; t2 = t1;
    }
  }
  if (msno > 1) {
    if (dz1 > 0.05f) {
      float drr = dz1 - 0.05f;
      float propor = drr / dz1;
      float zwice = propor * wi1;
      float zwliq = propor * wl1;
      propor = 0.05f / dz1;
      wi1 = propor * wi1;
      wl1 = propor * wl1;
      dz1 = 0.05f;
      combo(dz2, wl2, wi2, t2, drr, zwliq, zwice, t1);
      if (msno <= 2 && dz2 > (GLAC ? 0.10f : 0.20f)) {     // lsm:7321 / DIVIDE_GLACIER
        msno = 3;
        float dtdz = (t1 - t2) / ((dz1 + dz2) / 2.f);
        dz2 = dz2 / 2.f; wi2 = wi2 / 2.f; wl2 = wl2 / 2.f;
        dz3 = dz2; wi3 = wi2; wl3 = wl2;
        t3 = t2 - dtdz * dz2 / 2.f;
        if (t3 >= TFRZ) t3 = t2;
        else t2 = t2 + dtdz * dz2 / 2.f;
      }
    }
  }
  if (msno > 2) {
    if (dz2 > 0.2f) {
      float drr = dz2 - 0.2f;
      float propor = drr / dz2;
      float zwice = propor * wi2;
      float zwliq = propor * wl2;
      propor = 0.2f / dz2;
      wi2 = propor * wi2;
      wl2 = propor * wl2;
      dz2 = 0.2f;
      combo(dz3, wl3, wi3, t3, drr, zwliq, zwice, t2);
    }
  }
  s.isnow = -msno;
  NMP_ASSUME(s.isnow >= -NSNOW && s.isnow <= -1);         // DIVIDE runs on 1..3 layers and returns 1..3 (lsm:7263-7358)
  const int n = s.isnow;
  y.dzsnso[L(n + 1)] = dz1; y.snice[L(n + 1)] = wi1; y.snliq[L(n + 1)] = wl1; y.stc[L(n + 1)] = t1;
  if (msno >= 2) { y.dzsnso[L(n + 2)] = dz2; y.snice[L(n + 2)] = wi2; y.snliq[L(n + 2)] = wl2; y.stc[L(n + 2)] = t2; }
  if (msno >= 3) { y.dzsnso[L(n + 3)] = dz3; y.snice[L(n + 3)] = wi3; y.snliq[L(n + 3)] = wl3; y.stc[L(n + 3)] = t3; }
}

// COMPACT lsm:7427-7528
template <class A>
NMP_DEV void compact(const Ctx& c, const Col& s, const Lay<A>& y) {
  const float C2 = 21.e-3f, C3 = 2.5e-6f, C4 = 0.04f, C5 = 2.0f, DM = 100.0f, ETA0 = 0.8e+6f;
  float burden = 0.0f;
#pragma unroll
  for (int j = -2; j <= 0; j++) {
    if (j > s.isnow) {
      float snice = y.snice[L(j)], snliq = y.snliq[L(j)], dz = y.dzsnso[L(j)];
      float wx = snice + snliq;
      float fice = snice / wx;
      float void_ = 1.f - (div_rc(snice, NMP_RCC(DENICE)) + div_rc(snliq, NMP_RCC(DENH2O))) / dz;
      if (void_ > 0.001f && snice > 0.1f) {
        float bi = snice / dz;
        float td = nmp_max(0.f, TFRZ - y.stc[L(j)]);
        const float ca[3] = {-C4 * td, -46.0E-3f * (bi - DM), -0.08f * td - C2 * bi};
        float ce[3];
        nmp_expfN<3>(ca, ce);                  // (the second one is used only above DM: evaluating it anyway costs nothing extra)
        float dexpf = ce[0];
        float ddz1 = -C3 * dexpf, ddz3;
        if (bi > DM) ddz1 = ddz1 * ce[1];
        if (snliq > 0.01f * dz) ddz1 = ddz1 * C5;
        float ddz2 = div_rc(-(burden + 0.5f * wx) * ce[2], NMP_RCC(ETA0));
        if (y.imelt[L(j)] == 1.f) {
          float fo = y.ficeold[L(j)];
          ddz3 = nmp_max(0.f, (fo - fice) / nmp_max(1.E-6f, fo));
          ddz3 = div_rc(-ddz3, c.u.dt);
        } else {
          ddz3 = 0.f;
        }
        float pdzdtc = (ddz1 + ddz2 + ddz3) * c.dt;
        pdzdtc = nmp_max(-0.5f, pdzdtc);
        y.dzsnso[L(j)] = dz * (1.f + pdzdtc);
      }
      burden = burden + wx;
    }
  }
}

// SNOWH2O lsm:7530-7678
template <bool GLAC, class A>
NMP_DEV void snowh2o(const Ctx& c, Col& s, const Lay<A>& y, float qsnfro, float qsnsub, float qrain) {
  const float dt = c.dt;
  if (s.sneqv == 0.f) {
    y.sice[L(1)] = y.sice[L(1)] + (qsnfro - qsnsub) * dt / (y.dzsnso[L(1)] * 1000.f);
    if (!GLAC && y.sice[L(1)] < 0.f) { y.sh2o[L(1)] = y.sh2o[L(1)] + y.sice[L(1)]; y.sice[L(1)] = 0.f; }
  }
  if (s.isnow == 0 && s.sneqv > 0.f) {
    float temp = s.sneqv;
    s.sneqv = s.sneqv - qsnsub * dt + qsnfro * dt;
    float propor = s.sneqv / temp;
    s.snowh = nmp_max(0.f, propor * s.snowh);
    if (s.sneqv < 0.f) {
      y.sice[L(1)] = y.sice[L(1)] + s.sneqv / (y.dzsnso[L(1)] * 1000.f);
      s.sneqv = 0.f; s.snowh = 0.f;
    }
    if (y.sice[L(1)] < 0.f) { y.sh2o[L(1)] = y.sh2o[L(1)] + y.sice[L(1)]; y.sice[L(1)] = 0.f; }
  }
  if (s.snowh <= 1.E-8f || s.sneqv <= 1.E-6f) { s.snowh = 0.0f; s.sneqv = 0.0f; }
  if (s.isnow < 0) {
    float wgdif = y.snice[L(s.isnow + 1)] - qsnsub * dt + qsnfro * dt;
    y.snice[L(s.isnow + 1)] = wgdif;
    if (wgdif < 1.e-6f && s.isnow < 0) combine<GLAC>(s, y);
    if (s.isnow < 0) {
      float v = y.snliq[L(s.isnow + 1)] + qrain * dt;
      y.snliq[L(s.isnow + 1)] = nmp_max(0.f, v);
    }
  }
  float vol_liq[3], vol_ice[3], epore[3];
#pragma unroll
  for (int j = -2; j <= 0; j++) {
    vol_liq[j + 2] = 0.f; vol_ice[j + 2] = 0.f; epore[j + 2] = 0.f;
    if (j > s.isnow) {
      float dz = y.dzsnso[L(j)];
      vol_ice[j + 2] = nmp_min(1.f, y.snice[L(j)] / (dz * DENICE));
      epore[j + 2] = 1.f - vol_ice[j + 2];
      vol_liq[j + 2] = nmp_min(epore[j + 2], y.snliq[L(j)] / (dz * DENH2O));
    }
  }
  float qin = 0.f, qout = 0.f;
#pragma unroll
  for (int j = -2; j <= 0; j++) {
    if (j > s.isnow) {
      float snl = y.snliq[L(j)] + qin;
      float dz = y.dzsnso[L(j)];
      if (j <= -1) {
        const int jp = (j < 0) ? j + 1 : 0;
        if (epore[j + 2] < 0.05f || epore[jp + 2] < 0.05f) {
          qout = 0.f;
        } else {
          qout = nmp_max(0.f, (vol_liq[j + 2] - SSI * epore[j + 2]) * dz);
          qout = nmp_min(qout, (1.f - vol_ice[jp + 2] - vol_liq[jp + 2]) * y.dzsnso[L(jp)]);
        }
      } else {
        qout = nmp_max(0.f, (vol_liq[j + 2] - SSI * epore[j + 2]) * dz);
      }
      qout = qout * 1000.f;
      y.snliq[L(j)] = snl - qout;
      qin = qout;
    }
  }
  s.qsnbot = div_rc(qout, c.u.dt);
}

// Rebuild ZSNSO / DZSNSO after the snow-layer bookkeeping (lsm:6978-6994, gla:2219-2235);
// DZSNSO ends up as positive thickness.
template <class A>
NMP_DEV void rebuild_layers(const Ctx& c, const Col& s, const Lay<A>& y) {
  float run = 0.f;
#pragma unroll
  for (int iz = -2; iz <= NSOIL; iz++) {
    if (iz > s.isnow) {
      float d;
      if (iz <= 0) d = -y.dzsnso[L(iz)];
      else if (iz == 1) d = c.zsoil[L(1)];
      else d = (c.zsoil[L(iz)] - c.zsoil[L(iz > 1 ? iz - 1 : 1)]);
      run = (iz == s.isnow + 1) ? d : (run + d);
      y.zsnso[L(iz)] = run;
      y.dzsnso[L(iz)] = -d;
    }
  }
}

// SNOWWATER lsm:6868-6996 (SNOWFALL lsm:6998-7063 inlined)
template <class A>
NMP_DEV void snowwater(const Ctx& c, Col& s, const Lay<A>& y, float snowhin, float qsnfro,
                       float qsnsub, float qrain, float& snoflow) {
  const float dt = c.dt;
  snoflow = 0.0f; s.ponding1 = 0.0f; s.ponding2 = 0.0f;
  {
    int newnode = 0;
    if (s.isnow == 0 && s.qsnow > 0.f) {
      s.snowh = s.snowh + snowhin * dt;
      s.sneqv = s.sneqv + s.qsnow * dt;
    }
    if (s.isnow == 0 && s.qsnow > 0.f && s.snowh >= 0.025f) {
      s.isnow = -1;
      newnode = 1;
      y.dzsnso[L(0)] = s.snowh;
      s.snowh = 0.f;
      y.stc[L(0)] = nmp_min(273.16f, s.sfctmp);
      y.snice[L(0)] = s.sneqv;
      y.snliq[L(0)] = 0.f;
    }
    if (s.isnow < 0 && newnode == 0 && s.qsnow > 0.f) {
      y.snice[L(s.isnow + 1)] = y.snice[L(s.isnow + 1)] + s.qsnow * dt;
      y.dzsnso[L(s.isnow + 1)] = y.dzsnso[L(s.isnow + 1)] + snowhin * dt;
    }
  }
  if (s.isnow < 0) compact(c, s, y);
  if (s.isnow < 0) combine<false>(s, y);
  if (s.isnow < 0) divide<false>(s, y);
  snowh2o<false>(c, s, y, qsnfro, qsnsub, qrain);
  NMP_ASSUME(s.isnow >= -NSNOW && s.isnow <= 0);          // COMBINE only removes layers, DIVIDE returns 1..3, SNOWFALL creates the first
#pragma unroll
  for (int iz = -2; iz <= 0; iz++) {
    if (iz <= s.isnow) {
      y.snice[L(iz)] = 0.f; y.snliq[L(iz)] = 0.f; y.stc[L(iz)] = 0.f; y.dzsnso[L(iz)] = 0.f;
      y.zsnso[L(iz)] = 0.f;
    }
  }
  if (s.sneqv > 2000.f) {                              // lsm:6959-6965
    float bdsnow = y.snice[L(0)] / y.dzsnso[L(0)];
    snoflow = (s.sneqv - 2000.f);
    y.snice[L(0)] = y.snice[L(0)] - snoflow;
    y.dzsnso[L(0)] = y.dzsnso[L(0)] - snoflow / bdsnow;
    snoflow = div_rc(snoflow, c.u.dt);
  }
  if (s.isnow < 0) {
    float sw = 0.f;
#pragma unroll
    for (int iz = -2; iz <= 0; iz++)
      if (iz > s.isnow) sw = sw + y.snice[L(iz)] + y.snliq[L(iz)];
    s.sneqv = sw;
  }
  rebuild_layers(c, s, y);
}

// WDFCND1 lsm:8329-8362 / WDFCND2 lsm:8364-8400
NMP_DEV void wdfcnd1(const Parm& P, double r_smcmax, float& wdf, float& wcnd, float smc, float fcr) {
  float factr = nmp_max(0.01f, div_rc(smc, r_smcmax));
  wdf = P.dwsat * nmp_powf(factr, P.bexp + 2.0f);
  wdf = wdf * (1.0f - fcr);
  wcnd = P.dksat * nmp_powf(factr, 2.0f * P.bexp + 3.0f);
  wcnd = wcnd * (1.0f - fcr);
}
NMP_DEV void wdfcnd2(const Parm& P, double r_smcmax, float& wdf, float& wcnd, float smc, float sice) {
  float factr = nmp_max(0.01f, div_rc(smc, r_smcmax));
  float expon = P.bexp + 2.0f;
  wdf = P.dwsat * nmp_powf(factr, expon);
  if (sice > 0.0f) {
    float x = 500.f * sice;
    float vkwgt = 1.f / (1.f + nmp_powf(x, 3.f));
    wdf = vkwgt * wdf + (1.f - vkwgt) * P.dwsat * nmp_powf(div_rc(0.2f, r_smcmax), expon);
  }
  wcnd = P.dksat * nmp_powf(factr, 2.0f * P.bexp + 3.0f);
}

// SOILWATER lsm:7680-7936 with ZWTEQ (7938-7989), INFIL (7992-8087), SRT (8089-8217), SSTEP (8220-8327)
template <class A>
NMP_DEV void soilwater(const Ctx& c, const Parm& P, Col& s, const Lay<A>& y, float qinsur, float qseva,
                       const float* etrani, float& qdrain, float* wcnd, float& fcrmax) {
  const float dt = c.dt;
  const double r_smcmax = rc64(P.smcmax);      // SMCMAX divides ~20 times below (FICE, WDFCND x 4 layers x NITER)
  float sh2o[NL], smc[NL], sice[NL], dz[NL], fcr[NL];
  float pddum = 0.0f, rsat = 0.0f, sicemax = 0.0f;
  s.runsrf = 0.0f;
  fcrmax = 0.0f;
#pragma unroll
  for (int k = 1; k <= NSOIL; k++) {
    sh2o[L(k)] = y.sh2o[L(k)]; smc[L(k)] = y.smc[L(k)]; sice[L(k)] = y.sice[L(k)];
    dz[L(k)] = dz_soil(c, k);       // = DZSNSO(k) as SNOWWATER's rebuild_layers left it: uniform, so are its reciprocals (Urc)
  }
#pragma unroll
  for (int k = 1; k <= NSOIL; k++) {
    float epore = nmp_max(1.E-4f, (P.smcmax - sice[L(k)]));
    rsat = rsat + nmp_max(0.f, sh2o[L(k)] - epore) * dz[L(k)];
    sh2o[L(k)] = nmp_min(epore, sh2o[L(k)]);
  }
  const float ea4 = nmp_expf_const(-4.0f);
  float fcr_arg[NSOIL], fcr_exp[NSOIL];
#pragma unroll
  for (int k = 1; k <= NSOIL; k++) fcr_arg[k - 1] = -4.0f * (1.f - nmp_min(1.0f, div_rc(sice[L(k)], r_smcmax)));
  nmp_expfN<NSOIL>(fcr_arg, fcr_exp);          // the four layers' EXP as one batch
#pragma unroll
  for (int k = 1; k <= NSOIL; k++) {
    fcr[L(k)] = div_rc(nmp_max(0.0f, fcr_exp[k - 1] - ea4), c.u.one_m_ea4);
    if (sice[L(k)] > sicemax) sicemax = sice[L(k)];
    if (fcr[L(k)] > fcrmax) fcrmax = fcr[L(k)];
  }
  if (c.O.run == 2) {                                     // ZWTEQ
    float wd1 = 0.f, wd2 = 0.f;
#pragma unroll
    for (int k = 1; k <= NSOIL; k++) wd1 = wd1 + (P.smcmax - sh2o[L(k)]) * dz[L(k)];
    float dzfine = 3.0f * (-c.zsoil[L(NSOIL)]) / 100;
    s.zwt = -3.f * c.zsoil[L(NSOIL)] - 0.001f;
    const float zwt0 = s.zwt;
#pragma unroll 1
    for (int k = 1; k <= 100; k++) {
      float zfine = (float)k * dzfine;
      float temp = 1.f + (zwt0 - zfine) / P.psisat;
      wd2 = wd2 + P.smcmax * (1.f - nmp_powf(temp, -1.f / P.bexp)) * dzfine;
      if (fabsf(wd2 - wd1) <= 0.01f) { s.zwt = zfine; break; }
    }
    s.runsub = (1.0f - fcrmax) * 4.0f * nmp_expf_const(-TIMEAN) * nmp_expf(-2.0f * s.zwt);
  }
  if (s.vegtyp == c.isurban) fcr[L(1)] = 0.95f;
  if (c.O.run == 1 || c.O.run == 2 || c.O.run == 4 || c.O.run == 5) {
    float fsat;
    if (c.O.run == 1) fsat = FSATMX * nmp_expf(-0.5f * 6.0f * (s.zwt - 2.0f));
    else if (c.O.run == 5) fsat = FSATMX * nmp_expf(-0.5f * 6.0f * nmp_max(-2.0f - s.zwt, 0.f));
    else if (c.O.run == 2) fsat = FSATMX * nmp_expf(-0.5f * 2.0f * s.zwt);
    else {
      float smctot = 0.f, dztot = 0.f;
      bool done = false;
#pragma unroll
      for (int k = 1; k <= NSOIL; k++) {
        if (!done) {
          dztot = dztot + dz[L(k)];
          smctot = smctot + smc[L(k)] * dz[L(k)];
          if (dztot >= 2.0f) done = true;
        }
      }
      smctot = smctot / dztot;
      fsat = nmp_powf(nmp_max(0.01f, smctot / P.smcmax), 4.f);
    }
    if (qinsur > 0.f) {
      s.runsrf = qinsur * ((1.0f - fcr[L(1)]) * fsat + fcr[L(1)]);
      pddum = qinsur - s.runsrf;
    }
  }
  if (c.O.run == 3) {                                     // INFIL
    if (qinsur > 0.0f) {
      float dt1 = dt / 86400.f;
      float smcav = P.smcmax - P.smcwlt;
      float dmax = -c.zsoil[L(1)] * smcav;
      float dice = -c.zsoil[L(1)] * sice[L(1)];
      dmax = dmax * (1.0f - (sh2o[L(1)] + sice[L(1)] - P.smcwlt) / smcav);
      float dd = dmax;
#pragma unroll
      for (int k = 2; k <= NSOIL; k++) {
        float th = (c.zsoil[L(k - 1)] - c.zsoil[L(k)]);
        dice = dice + th * sice[L(k)];
        dmax = th * smcav;
        dmax = dmax * (1.0f - (sh2o[L(k)] + sice[L(k)] - P.smcwlt) / smcav);
        dd = dd + dmax;
      }
      float val = (1.f - nmp_expf(-P.kdt * dt1));
      float ddt = dd * val;
      float px = nmp_max(0.f, qinsur * dt);
      float infmax = (px * (ddt / (px + ddt))) / dt;
      float fcr_ = 1.f;
      if (dice > 1.E-2f) {
        float acrt = 3 * P.frzx / dice;                   // CVFRZ = 3
        float sum = 1.f;
        sum = sum + (acrt * acrt) / 2.f;                  // J=1: ACRT**2 / (2)
        sum = sum + acrt / 1.f;                           // J=2: ACRT**1 / 1
        fcr_ = 1.f - nmp_expf(-acrt) * sum;
      }
      infmax = infmax * fcr_;
      float wdf_, wcnd_;
      wdfcnd2(P, r_smcmax, wdf_, wcnd_, sh2o[L(1)], sicemax);
      infmax = nmp_max(infmax, wcnd_);
      infmax = nmp_min(infmax, px);
      s.runsrf = nmp_max(0.f, qinsur - infmax);
      pddum = qinsur - s.runsrf;
    }
  }
  int niter = 1;
  if (c.O.inf == 1) {
    niter = 3;
    if (pddum * dt > dz[L(1)] * P.smcmax) niter = niter * 2;
  }
  const float dtf = dt / niter;
  float qdrain_save = 0.0f;
  qdrain = 0.f;
#pragma unroll 1
  for (int iter = 1; iter <= niter; iter++) {
    // ---- SRT
    float wdf[NL], smx[NL], ddz[NL], dsmdz[NL], ai[NL], bi[NL], ci[NL], rhstt[NL];
    float smxwtd = 0.f;
    if (c.O.inf == 1) {        // WDFCND1 of the four layers as one batch: 4 log2 look-ups, then 8 exp2 look-ups (nmp_powf_pairN)
      float factr[NSOIL], pw1[NSOIL], pw2[NSOIL];
#pragma unroll
      for (int k = 1; k <= NSOIL; k++) factr[k - 1] = nmp_max(0.01f, div_rc(smc[L(k)], r_smcmax));
      nmp_powf_pairN<NSOIL>(factr, P.bexp + 2.0f, 2.0f * P.bexp + 3.0f, pw1, pw2);
#pragma unroll
      for (int k = 1; k <= NSOIL; k++) {
        wdf[L(k)] = P.dwsat * pw1[k - 1];
        wdf[L(k)] = wdf[L(k)] * (1.0f - fcr[L(k)]);
        wcnd[L(k)] = P.dksat * pw2[k - 1];
        wcnd[L(k)] = wcnd[L(k)] * (1.0f - fcr[L(k)]);
        smx[L(k)] = smc[L(k)];
      }
    } else {
#pragma unroll
      for (int k = 1; k <= NSOIL; k++) { wdfcnd2(P, r_smcmax, wdf[L(k)], wcnd[L(k)], sh2o[L(k)], sicemax); smx[L(k)] = sh2o[L(k)]; }
    }
    if (c.O.run == 5) smxwtd = (c.O.inf == 1) ? s.smcwtd : s.smcwtd * sh2o[L(NSOIL)] / smc[L(NSOIL)];
#pragma unroll
    for (int k = 1; k <= NSOIL; k++) {
      float denom, wflux;
      if (k == 1) {
        denom = -c.zsoil[L(k)];
        ddz[L(k)] = div_rc(2.0f, c.u.dz2[L(k)]);                       // TEMP1 = -ZSOIL(2)
        dsmdz[L(k)] = div_rc(2.0f * (smx[L(k)] - smx[L(k + 1)]), c.u.dz2[L(k)]);
        wflux = wdf[L(k)] * dsmdz[L(k)] + wcnd[L(k)] - pddum + etrani[L(k)] + qseva;
      } else if (k < NSOIL) {
        denom = (c.zsoil[L(k - 1)] - c.zsoil[L(k)]);
        ddz[L(k)] = div_rc(2.0f, c.u.dz2[L(k)]);                       // TEMP1 = ZSOIL(k-1) - ZSOIL(k+1)
        dsmdz[L(k)] = div_rc(2.0f * (smx[L(k)] - smx[L(k + 1)]), c.u.dz2[L(k)]);
        wflux = wdf[L(k)] * dsmdz[L(k)] + wcnd[L(k)] - wdf[L(k - 1)] * dsmdz[L(k - 1)] - wcnd[L(k - 1)] +
                etrani[L(k)];
      } else {
        denom = (c.zsoil[L(k - 1)] - c.zsoil[L(k)]);
        if (c.O.run == 1 || c.O.run == 2) qdrain = 0.f;
        if (c.O.run == 3) qdrain = P.slope * wcnd[L(k)];
        if (c.O.run == 4) qdrain = (1.0f - fcrmax) * wcnd[L(k)];
        if (c.O.run == 5) {
          float smxbot;
          if (s.zwt < c.zsoil[L(NSOIL)] - denom)
            smxbot = smx[L(k)] - (smx[L(k)] - smxwtd) * denom * 2.f / (denom + c.zsoil[L(k)] - s.zwt);
          else
            smxbot = smxwtd;
          dsmdz[L(k)] = div_rc(2.0f * (smx[L(k)] - smxbot), 0.5 * c.u.dz[L(k)]);      // TEMP1 = 2 DENOM
          qdrain = wdf[L(k)] * dsmdz[L(k)] + wcnd[L(k)];
        }
        wflux = -(wdf[L(k - 1)] * dsmdz[L(k - 1)]) - wcnd[L(k - 1)] + etrani[L(k)] + qdrain;
      }
      if (k == 1) {
        ai[L(k)] = 0.0f;
        bi[L(k)] = div_rc(wdf[L(k)] * ddz[L(k)], c.u.dz[L(k)]);
        ci[L(k)] = -bi[L(k)];
      } else if (k < NSOIL) {
        ai[L(k)] = div_rc(-wdf[L(k - 1)] * ddz[L(k - 1)], c.u.dz[L(k)]);
        ci[L(k)] = div_rc(-wdf[L(k)] * ddz[L(k)], c.u.dz[L(k)]);
        bi[L(k)] = -(ai[L(k)] + ci[L(k)]);
      } else {
        ai[L(k)] = div_rc(-wdf[L(k - 1)] * ddz[L(k - 1)], c.u.dz[L(k)]);
        ci[L(k)] = 0.0f;
        bi[L(k)] = -(ai[L(k)] + ci[L(k)]);
      }
      rhstt[L(k)] = div_rc(wflux, -c.u.dz[L(k)]);
    }
    // ---- SSTEP
    float wplus = 0.0f;
#pragma unroll
    for (int k = 1; k <= NSOIL; k++) {
      rhstt[L(k)] = rhstt[L(k)] * dtf;
      ai[L(k)] = ai[L(k)] * dtf;
      bi[L(k)] = 1.f + bi[L(k)] * dtf;
      ci[L(k)] = ci[L(k)] * dtf;
    }
    {                                                     // ROSR12 rows 1..4
      float p[NL], dl[NL];
      ci[L(NSOIL)] = 0.0f;
      p[L(1)] = -ci[L(1)] / bi[L(1)];
      dl[L(1)] = rhstt[L(1)] / bi[L(1)];
#pragma unroll
      for (int k = 2; k <= NSOIL; k++) {
        float inv = 1.0f / (bi[L(k)] + ai[L(k)] * p[L(k - 1)]);
        p[L(k)] = -ci[L(k)] * inv;
        dl[L(k)] = (rhstt[L(k)] - ai[L(k)] * dl[L(k - 1)]) * inv;
      }
      p[L(NSOIL)] = dl[L(NSOIL)];
#pragma unroll
      for (int kk = NSOIL - 1; kk >= 1; kk--) p[L(kk)] = p[L(kk)] * p[L(kk + 1)] + dl[L(kk)];
#pragma unroll
      for (int k = 1; k <= NSOIL; k++) sh2o[L(k)] = sh2o[L(k)] + p[L(k)];
    }
    if (c.O.run == 5) {
      if (s.zwt < c.zsoil[L(NSOIL)] - dz[L(NSOIL)]) {
        s.deeprech = s.deeprech + dtf * qdrain;
      } else {
        s.smcwtd = s.smcwtd + div_rc(dtf * qdrain, c.u.dz[L(NSOIL)]);
        wplus = nmp_max((s.smcwtd - P.smcmax), 0.0f) * dz[L(NSOIL)];
        float wminus = nmp_max((1.E-4f - s.smcwtd), 0.0f) * dz[L(NSOIL)];
        s.smcwtd = nmp_max(nmp_min(s.smcwtd, P.smcmax), 1.E-4f);
        sh2o[L(NSOIL)] = sh2o[L(NSOIL)] + div_rc(wplus, c.u.dz[L(NSOIL)]);
        qdrain = qdrain - wplus / dtf;
        s.deeprech = s.deeprech - wminus;
      }
    }
#pragma unroll
    for (int k = NSOIL; k >= 2; k--) {
      float epore = nmp_max(1.E-4f, (P.smcmax - sice[L(k)]));
      wplus = nmp_max((sh2o[L(k)] - epore), 0.0f) * dz[L(k)];
      sh2o[L(k)] = nmp_min(epore, sh2o[L(k)]);
      sh2o[L(k - 1)] = sh2o[L(k - 1)] + div_rc(wplus, c.u.dz[L(k - 1)]);
    }
    {
      float epore = nmp_max(1.E-4f, (P.smcmax - sice[L(1)]));
      wplus = nmp_max((sh2o[L(1)] - epore), 0.0f) * dz[L(1)];
      sh2o[L(1)] = nmp_min(epore, sh2o[L(1)]);
    }
#pragma unroll
    for (int k = 1; k <= NSOIL; k++) smc[L(k)] = sh2o[L(k)] + sice[L(k)];
    rsat = rsat + wplus;
    qdrain_save = qdrain_save + qdrain;
  }
  qdrain = (niter == 3) ? div_rc(qdrain_save, NMP_RCC(3.f)) : (niter == 6) ? div_rc(qdrain_save, NMP_RCC(6.f)) : qdrain_save;
  s.runsrf = s.runsrf * 1000.f + div_rc(rsat * 1000.f, c.u.dt);
  qdrain = qdrain * 1000.f;
  if (c.O.run == 2) {
    float wtsub = 0.f;
#pragma unroll
    for (int k = 1; k <= NSOIL; k++) wtsub = wtsub + wcnd[L(k)] * dz[L(k)];
#pragma unroll
    for (int k = 1; k <= NSOIL; k++) {
      float mh2o = s.runsub * dt * (wcnd[L(k)] * dz[L(k)]) / wtsub;
      sh2o[L(k)] = sh2o[L(k)] - div_rc(mh2o, c.u.dzmm[L(k)]);
    }
  }
  if (c.O.run != 1) {
    float mliq[NL], xs;
    const float watmin = 0.01f;
#pragma unroll
    for (int iz = 1; iz <= NSOIL; iz++) mliq[L(iz)] = sh2o[L(iz)] * dz[L(iz)] * 1000.f;
#pragma unroll
    for (int iz = 1; iz <= NSOIL - 1; iz++) {
      xs = (mliq[L(iz)] < 0.f) ? (watmin - mliq[L(iz)]) : 0.f;
      mliq[L(iz)] = mliq[L(iz)] + xs;
      mliq[L(iz + 1)] = mliq[L(iz + 1)] - xs;
    }
    xs = (mliq[L(NSOIL)] < watmin) ? (watmin - mliq[L(NSOIL)]) : 0.f;
    mliq[L(NSOIL)] = mliq[L(NSOIL)] + xs;
    s.runsub = s.runsub - div_rc(xs, c.u.dt);
    if (c.O.run == 5) s.deeprech = s.deeprech - xs * 1.E-3f;
#pragma unroll
    for (int iz = 1; iz <= NSOIL; iz++) sh2o[L(iz)] = div_rc(mliq[L(iz)], c.u.dzmm[L(iz)]);
  }
#pragma unroll
  for (int k = 1; k <= NSOIL; k++) { y.sh2o[L(k)] = sh2o[L(k)]; y.smc[L(k)] = smc[L(k)]; }
}

// GROUNDWATER lsm:8403-8585 (SIMGM).  S_NODE**(-BEXP) is evaluated in float64 like the reference.
template <class A>
NMP_DEV void groundwater(const Ctx& c, const Parm& P, Col& s, const Lay<A>& y, const float* wcnd,
                         float fcrmax, float& qdis) {
  const float ROUS = 0.2f, CMIC = 0.20f, dt = c.dt;
  float dzmm[NL], znode[NL], mliq[NL], epore[NL], hk[NL], smc[NL];
  dzmm[L(1)] = -c.zsoil[L(1)] * 1.E3f;
  znode[L(1)] = -c.zsoil[L(1)] / 2.f;
#pragma unroll
  for (int iz = 2; iz <= NSOIL; iz++) {
    dzmm[L(iz)] = 1.E3f * (c.zsoil[L(iz - 1)] - c.zsoil[L(iz)]);
    znode[L(iz)] = -c.zsoil[L(iz - 1)] + 0.5f * (c.zsoil[L(iz - 1)] - c.zsoil[L(iz)]);
  }
#pragma unroll
  for (int iz = 1; iz <= NSOIL; iz++) {
    float sh = y.sh2o[L(iz)], si = y.sice[L(iz)];
    smc[L(iz)] = sh + si;
    mliq[L(iz)] = sh * dzmm[L(iz)];
    epore[L(iz)] = nmp_max(0.01f, P.smcmax - si);
    hk[L(iz)] = 1.E3f * wcnd[L(iz)];
  }
  int iwt = NSOIL;
  {
    bool found = false;
#pragma unroll
    for (int iz = 2; iz <= NSOIL; iz++)
      if (!found && s.zwt <= -c.zsoil[L(iz)]) { iwt = iz - 1; found = true; }
  }
  qdis = (1.0f - fcrmax) * 5.0f * nmp_expf_const(-TIMEAN) * nmp_expf(-6.0f * (s.zwt - 2.0f));
  float smc_iwt = pick_soil(smc, iwt), hk_iwt = pick_soil(hk, iwt), zn_iwt = pick_soil(znode, iwt);
  double s_node = nmp_min(1.0f, smc_iwt / P.smcmax);
  s_node = (s_node > (double)0.01f) ? s_node : (double)0.01f;
  float smpfz = (float)(-((double)(P.psisat * 1000.f) * pow(s_node, (double)(-P.bexp))));
  smpfz = nmp_max(-120000.0f, CMIC * smpfz);
  float wh_zwt = -s.zwt * 1.E3f;
  float wh = smpfz - zn_iwt * 1.E3f;
  float qin = -hk_iwt * (wh_zwt - wh) / ((s.zwt - zn_iwt) * 1.E3f);
  qin = nmp_max(div_rc(-10.0f, c.u.dt), nmp_min(div_rc(10.f, c.u.dt), qin));
  s.wt = s.wt + (qin - qdis) * dt;
  if (iwt == NSOIL) {
    s.wa = s.wa + (qin - qdis) * dt;
    s.wt = s.wa;
    s.zwt = (-c.zsoil[L(NSOIL)] + 25.f) - div_rc(div_rc(s.wa, NMP_RCC(1000.f)), NMP_RCC(ROUS));
    mliq[L(NSOIL)] = mliq[L(NSOIL)] - qin * dt;
    mliq[L(NSOIL)] = mliq[L(NSOIL)] + nmp_max(0.f, (s.wa - 5000.f));
    s.wa = nmp_min(s.wa, 5000.f);
  } else {
    if (iwt == NSOIL - 1) {
      s.zwt = -c.zsoil[L(NSOIL)] - div_rc((s.wt - ROUS * 1000 * 25.f) / (epore[L(NSOIL)]), NMP_RCC(1000.f));
    } else {
      float ws = 0.f;
#pragma unroll
      for (int iz = 3; iz <= NSOIL; iz++)
        if (iz >= iwt + 2) ws = ws + epore[L(iz)] * dzmm[L(iz)];
      const float ep = pick_soil_below(epore, iwt), zs = pick_soil_below(c.zsoil, iwt);       // layer IWT+1, IWT <= NSOIL-2 here
      s.zwt = -zs - div_rc((s.wt - ROUS * 1000.f * 25.f - ws) / (ep), NMP_RCC(1000.f));
    }
    float wtsub = 0.f;
#pragma unroll
    for (int iz = 1; iz <= NSOIL; iz++) wtsub = wtsub + hk[L(iz)] * dzmm[L(iz)];
    const double r_wtsub = rc64(wtsub);
#pragma unroll
    for (int iz = 1; iz <= NSOIL; iz++) mliq[L(iz)] = mliq[L(iz)] - div_rc(qdis * dt * hk[L(iz)] * dzmm[L(iz)], r_wtsub);
  }
  s.zwt = nmp_max(1.5f, s.zwt);
  float xs;
  const float watmin = 0.01f;
#pragma unroll
  for (int iz = 1; iz <= NSOIL - 1; iz++) {
    xs = (mliq[L(iz)] < 0.f) ? (watmin - mliq[L(iz)]) : 0.f;
    mliq[L(iz)] = mliq[L(iz)] + xs;
    mliq[L(iz + 1)] = mliq[L(iz + 1)] - xs;
  }
  xs = (mliq[L(NSOIL)] < watmin) ? (watmin - mliq[L(NSOIL)]) : 0.f;
  mliq[L(NSOIL)] = mliq[L(NSOIL)] + xs;
  s.wa = s.wa - xs;
  s.wt = s.wt - xs;
#pragma unroll
  for (int iz = 1; iz <= NSOIL; iz++) y.sh2o[L(iz)] = div_rc(mliq[L(iz)], c.u.dzmm[L(iz)]);
}

// SHALLOWWATERTABLE lsm:8588-8718 (OPT_RUN = 5)
template <class A>
NMP_DEV void shallowwatertable(const Ctx& c, const Parm& P, Col& s, const Lay<A>& y) {
  float zsoil0[NL], dzs[NL], smc[NL], smceq[NL];
#pragma unroll
  for (int k = 1; k <= NSOIL; k++) {
    zsoil0[L(k)] = c.zsoil[L(k)]; dzs[L(k)] = y.dzsnso[L(k)]; smc[L(k)] = y.smc[L(k)];
    smceq[L(k)] = y.smceq[L(k)];
  }
  zsoil0[L(0)] = 0.f; dzs[L(0)] = 0.f; smc[L(0)] = 0.f; smceq[L(0)] = 0.f;
  float& wtd = s.zwt;
  int iz;
#pragma unroll 1
  for (iz = NSOIL; iz >= 1; iz--)
    if (wtd + 1.E-6f < zsoil0[L(iz)]) break;
  int iwtd = iz, kwtd = iwtd + 1;
  float wtdold;
  const float dzn = dzs[L(NSOIL)];
  if (kwtd <= NSOIL) {
    wtdold = wtd;
    if (smc[L(kwtd)] > smceq[L(kwtd)]) {
      if (smc[L(kwtd)] == P.smcmax) {
        wtd = zsoil0[L(iwtd)];
        s.rech = -(wtdold - wtd) * (P.smcmax - smceq[L(kwtd)]);
        iwtd = iwtd - 1;
        kwtd = kwtd - 1;
        if (kwtd >= 1) {
          if (smc[L(kwtd)] > smceq[L(kwtd)]) {
            wtdold = wtd;
            wtd = nmp_min((smc[L(kwtd)] * dzs[L(kwtd)] - smceq[L(kwtd)] * zsoil0[L(iwtd)] +
                         P.smcmax * zsoil0[L(kwtd)]) / (P.smcmax - smceq[L(kwtd)]), zsoil0[L(iwtd)]);
            s.rech = s.rech - (wtdold - wtd) * (P.smcmax - smceq[L(kwtd)]);
          }
        }
      } else {
        wtd = nmp_min((smc[L(kwtd)] * dzs[L(kwtd)] - smceq[L(kwtd)] * zsoil0[L(iwtd)] +
                     P.smcmax * zsoil0[L(kwtd)]) / (P.smcmax - smceq[L(kwtd)]), zsoil0[L(iwtd)]);
        s.rech = -(wtdold - wtd) * (P.smcmax - smceq[L(kwtd)]);
      }
    } else {
      wtd = zsoil0[L(kwtd)];
      s.rech = -(wtdold - wtd) * (P.smcmax - smceq[L(kwtd)]);
      kwtd = kwtd + 1;
      iwtd = iwtd + 1;
      if (kwtd <= NSOIL) {
        wtdold = wtd;
        if (smc[L(kwtd)] > smceq[L(kwtd)])
          wtd = nmp_min((smc[L(kwtd)] * dzs[L(kwtd)] - smceq[L(kwtd)] * zsoil0[L(iwtd)] +
                       P.smcmax * zsoil0[L(kwtd)]) / (P.smcmax - smceq[L(kwtd)]), zsoil0[L(iwtd)]);
        else
          wtd = zsoil0[L(kwtd)];
        s.rech = s.rech - (wtdold - wtd) * (P.smcmax - smceq[L(kwtd)]);
      } else {
        wtdold = wtd;
        float smceqdeep = P.smcmax * nmp_powf(-P.psisat / (-P.psisat - dzn), 1.f / P.bexp);
        wtd = nmp_min((s.smcwtd * dzn - smceqdeep * zsoil0[L(NSOIL)] + P.smcmax * (zsoil0[L(NSOIL)] - dzn)) /
                    (P.smcmax - smceqdeep), zsoil0[L(NSOIL)]);
        s.rech = s.rech - (wtdold - wtd) * (P.smcmax - smceqdeep);
      }
    }
  } else if (wtd >= zsoil0[L(NSOIL)] - dzn) {
    wtdold = wtd;
    float smceqdeep = P.smcmax * nmp_powf(-P.psisat / (-P.psisat - dzn), 1.f / P.bexp);
    if (s.smcwtd > smceqdeep) {
      wtd = nmp_min((s.smcwtd * dzn - smceqdeep * zsoil0[L(NSOIL)] + P.smcmax * (zsoil0[L(NSOIL)] - dzn)) /
                  (P.smcmax - smceqdeep), zsoil0[L(NSOIL)]);
      s.rech = -(wtdold - wtd) * (P.smcmax - smceqdeep);
    } else {
      s.rech = -(wtdold - (zsoil0[L(NSOIL)] - dzn)) * (P.smcmax - smceqdeep);
      wtdold = zsoil0[L(NSOIL)] - dzn;
      float dzup = (smceqdeep - s.smcwtd) * dzn / (P.smcmax - smceqdeep);
      wtd = wtdold - dzup;
      s.rech = s.rech - (P.smcmax - smceqdeep) * dzup;
      s.smcwtd = smceqdeep;
    }
  }
  if (iwtd < NSOIL) s.smcwtd = P.smcmax;
}

// WATER lsm:6382-6613
template <class A>
NMP_DEV void water(const Ctx& c, const Parm& P, Col& s, const Lay<A>& y, float qvap, float qdew) {
  const float dt = c.dt;
  float etrani[NL], wcnd[NL];
  float snoflow = 0.f, qrain, snowhin, qdrain = 0.f, fcrmax = 0.f;
  s.runsub = 0.f;
  canwater(c, P, s, qrain, snowhin);
  NMP_TIC(22);   // canwater
  float qsnsub = 0.f;
  if (s.sneqv > 0.f) qsnsub = nmp_min(qvap, div_rc(s.sneqv, c.u.dt));
  float qseva = qvap - qsnsub;
  float qsnfro = 0.f;
  if (s.sneqv > 0.f) qsnfro = qdew;
  float qsdew = qdew - qsnfro;
  snowwater(c, s, y, snowhin, qsnfro, qsnsub, qrain, snoflow);
  NMP_TIC(23);   // snowwater
  if (s.frozen_ground) {
    float si = y.sice[L(1)] + div_rc((qsdew - qseva) * dt, c.u.dzmm[L(1)]);     // DZSNSO(1) as rebuild_layers left it
    qsdew = 0.0f;
    qseva = 0.0f;
    if (si < 0.f) { y.sh2o[L(1)] = y.sh2o[L(1)] + si; si = 0.f; }
    y.sice[L(1)] = si;
  }
  float qinsur = div_rc(s.ponding + s.ponding1 + s.ponding2, c.u.dt) * 0.001f;
  if (s.isnow == 0) qinsur = qinsur + (s.qsnbot + qsdew + qrain) * 0.001f;
  else qinsur = qinsur + (s.qsnbot + qsdew) * 0.001f;
  qseva = qseva * 0.001f;
#pragma unroll
  for (int iz = 1; iz <= NSOIL; iz++) {
    etrani[L(iz)] = 0.f; wcnd[L(iz)] = 0.f;
    if (iz <= P.nroot) etrani[L(iz)] = s.etran * y.btrani[L(iz)] * 0.001f;
  }
  soilwater(c, P, s, y, qinsur, qseva, etrani, qdrain, wcnd, fcrmax);
  NMP_TIC(24);   // soilwater
  if (c.O.run == 1) {
    float qdis;
    groundwater(c, P, s, y, wcnd, fcrmax, qdis);
    s.runsub = qdis;
  }
  if (c.O.run == 3 || c.O.run == 4) s.runsub = s.runsub + qdrain;
#pragma unroll
  for (int iz = 1; iz <= NSOIL; iz++) y.smc[L(iz)] = y.sh2o[L(iz)] + y.sice[L(iz)];
  if (c.O.run == 5) {
    shallowwatertable(c, P, s, y);
    y.sh2o[L(NSOIL)] = y.smc[L(NSOIL)] - y.sice[L(NSOIL)];
    s.runsub = s.runsub + qdrain;
    s.wa = 0.f;
  }
  s.runsub = s.runsub + snoflow;
}

}  // namespace nmp

/* TEST INFRASTRUCTURE (oracle): water half of the column physics.  See noahmp_oracle.h.
 * Reference: phys/module_sf_noahmplsm.F90 ("lsm"), subroutine WATER and everything below it. */
#include <math.h>
#include <stdlib.h>
#include "nmp_internal.h"

/* CANWATER lsm:6615-6865 */
static void canwater(const nmp_ctx* c, real dt, real sfctmp, real uu, real vv, real fcev, real fctr,
                     real qprecc, real qprecl, real elai, real esai, int ist, real tg, real fveg,
                     int frozen_canopy, real* canliq, real* canice, real* tv, real* ecan, real* etran,
                     real* qrain, real* qsnow, real* snowhin, real* fwet, real* fpice) {
  const noahmp_tables* T = c->T;
  int v = c->vegtyp - 1;
  real fp = 0.0f, rain, snow, qintr = 0.f, qdripr = 0.f, qthror = 0.f, qints = 0.f, qdrips = 0.0f,
       qthros = 0.f;
  real qevac, qdewc, qsubc, qfroc;
  *fpice = 0.f;
  if (c->O.opt_snf == 1) {
    if (sfctmp > TFRZ + 2.5f) *fpice = 0.f;
    else {
      if (sfctmp <= TFRZ + 0.5f) *fpice = 1.0f;
      else if (sfctmp <= TFRZ + 2.f) *fpice = 1.f - (-54.632f + 0.2f * sfctmp);
      else *fpice = 0.6f;
    }
  }
  if (c->O.opt_snf == 2) { if (sfctmp >= TFRZ + 2.2f) *fpice = 0.f; else *fpice = 1.0f; }
  if (c->O.opt_snf == 3) { if (sfctmp >= TFRZ) *fpice = 0.f; else *fpice = 1.0f; }
  real bdfall = MINF(120.f, 67.92f + 51.25f * expf((sfctmp - TFRZ) / 2.59f));
  rain = (qprecc + qprecl) * (1.f - *fpice);
  snow = (qprecc + qprecl) * *fpice;
  if (qprecc + qprecl > 0.f) fp = (qprecc + qprecl) / (10.f * qprecc + qprecl);
  real maxliq = T->ch2op[v] * (elai + esai);
  if ((elai + esai) > 0.f) {
    qintr = fveg * rain * fp;
    qintr = MINF(qintr, (maxliq - *canliq) / dt * (1.f - expf(-rain * dt / maxliq)));
    qintr = MAXF(qintr, 0.f);
    qdripr = fveg * rain - qintr;
    qthror = (1.f - fveg) * rain;
  } else {
    qintr = 0.f; qdripr = 0.f; qthror = rain;
  }
  if (!frozen_canopy) {
    *etran = MAXF(fctr / HVAP, 0.f);
    qevac = MAXF(fcev / HVAP, 0.f);
    qdewc = fabsf(MINF(fcev / HVAP, 0.f));
    qsubc = 0.f; qfroc = 0.f;
  } else {
    *etran = MAXF(fctr / HSUB, 0.f);
    qevac = 0.f; qdewc = 0.f;
    qsubc = MAXF(fcev / HSUB, 0.f);
    qfroc = fabsf(MINF(fcev / HSUB, 0.f));
  }
  qevac = MINF(*canliq / dt, qevac);
  *canliq = MAXF(0.f, *canliq + (qintr + qdewc - qevac) * dt);
  if (*canliq <= 1.E-06f) *canliq = 0.0f;
  real maxsno = 6.6f * (0.27f + 46.f / bdfall) * (elai + esai);
  if ((elai + esai) > 0.f) {
    qints = fveg * snow * fp;
    qints = MINF(qints, (maxsno - *canice) / dt * (1.f - expf(-snow * dt / maxsno)));
    qints = MAXF(qints, 0.f);
    real ft = MAXF(0.0f, (*tv - 270.15f) / 1.87E5f);
    real fv = sqrtf(uu * uu + vv * vv) / 1.56E5f;
    qdrips = MAXF(0.f, *canice) * (fv + ft);
    qthros = (1.0f - fveg) * snow + (fveg * snow - qints);
  } else {
    qints = 0.f; qdrips = 0.f; qthros = snow;
  }
  qsubc = MINF(*canice / dt, qsubc);
  *canice = MAXF(0.f, *canice + (qints - qdrips) * dt + (qfroc - qsubc) * dt);
  if (*canice <= 1.E-6f) *canice = 0.f;
  if (*canice > 0.f) *fwet = MAXF(0.f, *canice) / MAXF(maxsno, 1.E-06f);
  else *fwet = MAXF(0.f, *canliq) / MAXF(maxliq, 1.E-06f);
  *fwet = powf(MINF(*fwet, 1.f), 0.667f);
  if (*canice > 1.E-6f && *tv > TFRZ) {
    real qmeltc = MINF(*canice / dt, (*tv - TFRZ) * CICE * *canice / DENICE / (dt * HFUS));
    *canice = MAXF(0.f, *canice - qmeltc * dt);
    *canliq = MAXF(0.f, *canliq + qmeltc * dt);
    *tv = *fwet * TFRZ + (1.f - *fwet) * *tv;
  }
  if (*canliq > 1.E-6f && *tv < TFRZ) {
    real qfrzc = MINF(*canliq / dt, (TFRZ - *tv) * CWAT * *canliq / DENH2O / (dt * HFUS));
    *canliq = MAXF(0.f, *canliq - qfrzc * dt);
    *canice = MAXF(0.f, *canice + qfrzc * dt);
    *tv = *fwet * TFRZ + (1.f - *fwet) * *tv;
  }
  *ecan = qevac + qsubc - qdewc - qfroc;
  *qrain = qdripr + qthror;
  *qsnow = qdrips + qthros;
  *snowhin = *qsnow / bdfall;
  if (ist == 2 && tg > TFRZ) { *qsnow = 0.f; *snowhin = 0.f; }
}

/* COMBO lsm:7375-7424: merge layer 2 into layer 1, conserving enthalpy */
static void combo(real* dz, real* wliq, real* wice, real* t, real dz2, real wliq2, real wice2, real t2) {
  real dzc = *dz + dz2;
  real wicec = (*wice + wice2);
  real wliqc = (*wliq + wliq2);
  real h = (CICE * *wice + CWAT * *wliq) * (*t - TFRZ) + HFUS * *wliq;
  real h2 = (CICE * wice2 + CWAT * wliq2) * (t2 - TFRZ) + HFUS * wliq2;
  real hc = h + h2, tc;
  if (hc < 0.f) tc = TFRZ + hc / (CICE * wicec + CWAT * wliqc);
  else if (hc <= HFUS * wliqc) tc = TFRZ;
  else tc = TFRZ + (hc - HFUS * wliqc) / (CICE * wicec + CWAT * wliqc);
  *dz = dzc; *wice = wicec; *wliq = wliqc; *t = tc;
}

/* COMBINE lsm:7065-7246 */
/* glacier != 0 selects COMBINE_GLACIER (gla:2403-2571): DZMIN /0.045,0.05,0.2/, collapse below
 * 0.05 m, PONDING1/2 accumulate, no negative-ice branch */
void nmp_combine(int glacier, int* isnow, real* sh2o, real* stc, real* snice, real* snliq, real* dzsnso,
                 real* sice, real* snowh, real* sneqv, real* ponding1, real* ponding2) {
  const real DZMIN_L[3] = {0.025f, 0.025f, 0.1f}, DZMIN_G[3] = {0.045f, 0.05f, 0.2f};
  const real* DZMIN = glacier ? DZMIN_G : DZMIN_L;
  const real hmin = glacier ? 0.05f : 0.025f;
  int isnow_old = *isnow;
  for (int j = isnow_old + 1; j <= 0; j++) {
    if (snice[L(j)] <= .1f) {
      if (j != 0) {
        snliq[L(j + 1)] = snliq[L(j + 1)] + snliq[L(j)];
        snice[L(j + 1)] = snice[L(j + 1)] + snice[L(j)];
      } else {
        if (isnow_old < -1) {
          snliq[L(j - 1)] = snliq[L(j - 1)] + snliq[L(j)];
          snice[L(j - 1)] = snice[L(j - 1)] + snice[L(j)];
        } else {
          if (glacier) {
            *ponding1 = *ponding1 + snliq[L(j)];
            *sneqv = snice[L(j)];
            *snowh = dzsnso[L(j)];
          } else if (snice[L(j)] >= 0.f) {
            *ponding1 = snliq[L(j)];
            *sneqv = snice[L(j)];
            *snowh = dzsnso[L(j)];
          } else {
            *ponding1 = snliq[L(j)] + snice[L(j)];
            if (*ponding1 < 0.f) {
              sice[L(1)] = MAXF(0.0f, sice[L(1)] + *ponding1 / (dzsnso[L(1)] * 1000.f));
              *ponding1 = 0.0f;
            }
            *sneqv = 0.0f;
            *snowh = 0.0f;
          }
          snliq[L(j)] = 0.0f; snice[L(j)] = 0.0f; dzsnso[L(j)] = 0.0f;
        }
      }
      if (j > *isnow + 1 && *isnow < -1) {
        for (int i = j; i >= *isnow + 2; i--) {
          stc[L(i)] = stc[L(i - 1)];
          snliq[L(i)] = snliq[L(i - 1)];
          snice[L(i)] = snice[L(i - 1)];
          dzsnso[L(i)] = dzsnso[L(i - 1)];
        }
      }
      *isnow = *isnow + 1;
    }
  }
  if (sice[L(1)] < 0.f) { sh2o[L(1)] = sh2o[L(1)] + sice[L(1)]; sice[L(1)] = 0.f; }
  if (*isnow == 0) return;
  *sneqv = 0.f; *snowh = 0.f;
  real zwice = 0.f, zwliq = 0.f;
  for (int j = *isnow + 1; j <= 0; j++) {
    *sneqv = *sneqv + snice[L(j)] + snliq[L(j)];
    *snowh = *snowh + dzsnso[L(j)];
    zwice = zwice + snice[L(j)];
    zwliq = zwliq + snliq[L(j)];
  }
  if (*snowh < hmin && *isnow < 0) {
    *isnow = 0;
    *sneqv = zwice;
    *ponding2 = glacier ? (*ponding2 + zwliq) : zwliq;
    if (*sneqv <= 0.f) *snowh = 0.f;
  }
  if (*isnow < -1) {
    isnow_old = *isnow;
    int mssi = 1;
    for (int i = isnow_old + 1; i <= 0; i++) {
      if (dzsnso[L(i)] < DZMIN[mssi - 1]) {
        int neibor, j, l;
        if (i == *isnow + 1) neibor = i + 1;
        else if (i == 0) neibor = i - 1;
        else {
          neibor = i + 1;
          if ((dzsnso[L(i - 1)] + dzsnso[L(i)]) < (dzsnso[L(i + 1)] + dzsnso[L(i)])) neibor = i - 1;
        }
        if (neibor > i) { j = neibor; l = i; } else { j = i; l = neibor; }
        combo(&dzsnso[L(j)], &snliq[L(j)], &snice[L(j)], &stc[L(j)], dzsnso[L(l)], snliq[L(l)],
              snice[L(l)], stc[L(l)]);
        if (j - 1 > *isnow + 1) {
          for (int k = j - 1; k >= *isnow + 2; k--) {
            stc[L(k)] = stc[L(k - 1)];
            snice[L(k)] = snice[L(k - 1)];
            snliq[L(k)] = snliq[L(k - 1)];
            dzsnso[L(k)] = dzsnso[L(k - 1)];
          }
        }
        *isnow = *isnow + 1;
        if (*isnow >= -1) break;
      } else {
        mssi = mssi + 1;
      }
    }
  }
}

/* DIVIDE lsm:7248-7371 */
/* dz2max: third-layer trigger, 0.20 m on land (lsm:7321), 0.10 m on glaciers (gla DIVIDE_GLACIER) */
void nmp_divide(int nsnow, real dz2max, int* isnow, real* stc, real* snice, real* snliq, real* dzsnso) {
  real dz[4] = {0, 0, 0, 0}, swice[4] = {0, 0, 0, 0}, swliq[4] = {0, 0, 0, 0}, tsno[4] = {0, 0, 0, 0};
  for (int j = 1; j <= nsnow; j++) {
    if (j <= abs(*isnow)) {
      dz[j] = dzsnso[L(j + *isnow)];
      swice[j] = snice[L(j + *isnow)];
      swliq[j] = snliq[L(j + *isnow)];
      tsno[j] = stc[L(j + *isnow)];
    }
  }
  int msno = abs(*isnow);
  if (msno == 1) {
    if (dz[1] > 0.05f) {
      msno = 2;
      dz[1] = dz[1] / 2.f; swice[1] = swice[1] / 2.f; swliq[1] = swliq[1] / 2.f;
      dz[2] = dz[1]; swice[2] = swice[1]; swliq[2] = swliq[1]; tsno[2] = tsno[1];
    }
  }
  if (msno > 1) {
    if (dz[1] > 0.05f) {
      real drr = dz[1] - 0.05f;
      real propor = drr / dz[1];
      real zwice = propor * swice[1];
      real zwliq = propor * swliq[1];
      propor = 0.05f / dz[1];
      swice[1] = propor * swice[1];
      swliq[1] = propor * swliq[1];
      dz[1] = 0.05f;
      combo(&dz[2], &swliq[2], &swice[2], &tsno[2], drr, zwliq, zwice, tsno[1]);
      if (msno <= 2 && dz[2] > dz2max) {
        msno = 3;
        real dtdz = (tsno[1] - tsno[2]) / ((dz[1] + dz[2]) / 2.f);
        dz[2] = dz[2] / 2.f; swice[2] = swice[2] / 2.f; swliq[2] = swliq[2] / 2.f;
        dz[3] = dz[2]; swice[3] = swice[2]; swliq[3] = swliq[2];
        tsno[3] = tsno[2] - dtdz * dz[2] / 2.f;
        if (tsno[3] >= TFRZ) tsno[3] = tsno[2];
        else tsno[2] = tsno[2] + dtdz * dz[2] / 2.f;
      }
    }
  }
  if (msno > 2) {
    if (dz[2] > 0.2f) {
      real drr = dz[2] - 0.2f;
      real propor = drr / dz[2];
      real zwice = propor * swice[2];
      real zwliq = propor * swliq[2];
      propor = 0.2f / dz[2];
      swice[2] = propor * swice[2];
      swliq[2] = propor * swliq[2];
      dz[2] = 0.2f;
      combo(&dz[3], &swliq[3], &swice[3], &tsno[3], drr, zwliq, zwice, tsno[2]);
    }
  }
  *isnow = -msno;
  for (int j = *isnow + 1; j <= 0; j++) {
    dzsnso[L(j)] = dz[j - *isnow];
    snice[L(j)] = swice[j - *isnow];
    snliq[L(j)] = swliq[j - *isnow];
    stc[L(j)] = tsno[j - *isnow];
  }
}

/* COMPACT lsm:7427-7528 */
void nmp_compact(real dt, const real* stc, const real* snice, const real* snliq, const int* imelt,
                    const real* ficeold, int isnow, real* dzsnso) {
  const real C2 = 21.e-3f, C3 = 2.5e-6f, C4 = 0.04f, C5 = 2.0f, DM = 100.0f, ETA0 = 0.8e+6f;
  real burden = 0.0f;
  for (int j = isnow + 1; j <= 0; j++) {
    real wx = snice[L(j)] + snliq[L(j)];
    real fice = snice[L(j)] / wx;
    real void_ = 1.f - (snice[L(j)] / DENICE + snliq[L(j)] / DENH2O) / dzsnso[L(j)];
    if (void_ > 0.001f && snice[L(j)] > 0.1f) {
      real bi = snice[L(j)] / dzsnso[L(j)];
      real td = MAXF(0.f, TFRZ - stc[L(j)]);
      real dexpf = expf(-C4 * td);
      real ddz1 = -C3 * dexpf, ddz3;
      if (bi > DM) ddz1 = ddz1 * expf(-46.0E-3f * (bi - DM));
      if (snliq[L(j)] > 0.01f * dzsnso[L(j)]) ddz1 = ddz1 * C5;
      real ddz2 = -(burden + 0.5f * wx) * expf(-0.08f * td - C2 * bi) / ETA0;
      if (imelt[L(j)] == 1) {
        ddz3 = MAXF(0.f, (ficeold[L(j)] - fice) / MAXF(1.E-6f, ficeold[L(j)]));
        ddz3 = -ddz3 / dt;
      } else {
        ddz3 = 0.f;
      }
      real pdzdtc = (ddz1 + ddz2 + ddz3) * dt;
      pdzdtc = MAXF(-0.5f, pdzdtc);
      dzsnso[L(j)] = dzsnso[L(j)] * (1.f + pdzdtc);
    }
    burden = burden + wx;
  }
}

/* SNOWH2O lsm:7530-7678 */
/* glacier != 0 selects SNOWH2O_GLACIER (gla:2751-2895): no SICE<0 repair in the bare-ground branch */
void nmp_snowh2o(const nmp_ctx* c, int glacier, real dt, real qsnfro, real qsnsub, real qrain, int* isnow,
                    real* dzsnso, real* snowh, real* sneqv, real* snice, real* snliq, real* sh2o,
                    real* sice, real* stc, real* qsnbot, real* ponding1, real* ponding2) {
  real vol_liq[NL], vol_ice[NL], epore[NL];
  if (*sneqv == 0.f) {
    sice[L(1)] = sice[L(1)] + (qsnfro - qsnsub) * dt / (dzsnso[L(1)] * 1000.f);
    if (!glacier && sice[L(1)] < 0.f) { sh2o[L(1)] = sh2o[L(1)] + sice[L(1)]; sice[L(1)] = 0.f; }
  }
  if (*isnow == 0 && *sneqv > 0.f) {
    real temp = *sneqv;
    *sneqv = *sneqv - qsnsub * dt + qsnfro * dt;
    real propor = *sneqv / temp;
    *snowh = MAXF(0.f, propor * *snowh);
    if (*sneqv < 0.f) {
      sice[L(1)] = sice[L(1)] + *sneqv / (dzsnso[L(1)] * 1000.f);
      *sneqv = 0.f; *snowh = 0.f;
    }
    if (sice[L(1)] < 0.f) { sh2o[L(1)] = sh2o[L(1)] + sice[L(1)]; sice[L(1)] = 0.f; }
  }
  if (*snowh <= 1.E-8f || *sneqv <= 1.E-6f) { *snowh = 0.0f; *sneqv = 0.0f; }
  if (*isnow < 0) {
    real wgdif = snice[L(*isnow + 1)] - qsnsub * dt + qsnfro * dt;
    snice[L(*isnow + 1)] = wgdif;
    if (wgdif < 1.e-6f && *isnow < 0)
      nmp_combine(glacier, isnow, sh2o, stc, snice, snliq, dzsnso, sice, snowh, sneqv, ponding1, ponding2);
    if (*isnow < 0) {
      snliq[L(*isnow + 1)] = snliq[L(*isnow + 1)] + qrain * dt;
      snliq[L(*isnow + 1)] = MAXF(0.f, snliq[L(*isnow + 1)]);
    }
  }
  for (int j = -c->nsnow + 1; j <= 0; j++) {
    if (j >= *isnow + 1) {
      vol_ice[L(j)] = MINF(1.f, snice[L(j)] / (dzsnso[L(j)] * DENICE));
      epore[L(j)] = 1.f - vol_ice[L(j)];
      vol_liq[L(j)] = MINF(epore[L(j)], snliq[L(j)] / (dzsnso[L(j)] * DENH2O));
    }
  }
  real qin = 0.f, qout = 0.f;
  for (int j = -c->nsnow + 1; j <= 0; j++) {
    if (j >= *isnow + 1) {
      snliq[L(j)] = snliq[L(j)] + qin;
      if (j <= -1) {
        if (epore[L(j)] < 0.05f || epore[L(j + 1)] < 0.05f) {
          qout = 0.f;
        } else {
          qout = MAXF(0.f, (vol_liq[L(j)] - SSI * epore[L(j)]) * dzsnso[L(j)]);
          qout = MINF(qout, (1.f - vol_ice[L(j + 1)] - vol_liq[L(j + 1)]) * dzsnso[L(j + 1)]);
        }
      } else {
        qout = MAXF(0.f, (vol_liq[L(j)] - SSI * epore[L(j)]) * dzsnso[L(j)]);
      }
      qout = qout * 1000.f;
      snliq[L(j)] = snliq[L(j)] - qout;
      qin = qout;
    }
  }
  *qsnbot = qout / dt;
}

/* SNOWWATER lsm:6868-6996 (with SNOWFALL lsm:6998-7063 inlined at the top) */
static void snowwater(const nmp_ctx* c, const int* imelt, real dt, real sfctmp, real snowhin,
                      real qsnow, real qsnfro, real qsnsub, real qrain, const real* ficeold,
                      int* isnow, real* snowh, real* sneqv, real* snice, real* snliq, real* sh2o,
                      real* sice, real* stc, real* zsnso, real* dzsnso, real* qsnbot, real* snoflow,
                      real* ponding1, real* ponding2) {
  int ns = c->nsoil;
  *snoflow = 0.0f; *ponding1 = 0.0f; *ponding2 = 0.0f;
  {                                                            /* SNOWFALL */
    int newnode = 0;
    if (*isnow == 0 && qsnow > 0.f) {
      *snowh = *snowh + snowhin * dt;
      *sneqv = *sneqv + qsnow * dt;
    }
    if (*isnow == 0 && qsnow > 0.f && *snowh >= 0.025f) {
      *isnow = -1;
      newnode = 1;
      dzsnso[L(0)] = *snowh;
      *snowh = 0.f;
      stc[L(0)] = MINF(273.16f, sfctmp);
      snice[L(0)] = *sneqv;
      snliq[L(0)] = 0.f;
    }
    if (*isnow < 0 && newnode == 0 && qsnow > 0.f) {
      snice[L(*isnow + 1)] = snice[L(*isnow + 1)] + qsnow * dt;
      dzsnso[L(*isnow + 1)] = dzsnso[L(*isnow + 1)] + snowhin * dt;
    }
  }
  if (*isnow < 0) nmp_compact(dt, stc, snice, snliq, imelt, ficeold, *isnow, dzsnso);
  if (*isnow < 0) nmp_combine(0, isnow, sh2o, stc, snice, snliq, dzsnso, sice, snowh, sneqv, ponding1, ponding2);
  if (*isnow < 0) nmp_divide(c->nsnow, 0.20f, isnow, stc, snice, snliq, dzsnso);
  nmp_snowh2o(c, 0, dt, qsnfro, qsnsub, qrain, isnow, dzsnso, snowh, sneqv, snice, snliq, sh2o, sice, stc,
              qsnbot, ponding1, ponding2);
  for (int iz = -c->nsnow + 1; iz <= *isnow; iz++) {
    snice[L(iz)] = 0.f; snliq[L(iz)] = 0.f; stc[L(iz)] = 0.f; dzsnso[L(iz)] = 0.f; zsnso[L(iz)] = 0.f;
  }
  if (*sneqv > 2000.f) {                                       /* glacier flow cap lsm:6959-6965 */
    real bdsnow = snice[L(0)] / dzsnso[L(0)];
    *snoflow = (*sneqv - 2000.f);
    snice[L(0)] = snice[L(0)] - *snoflow;
    dzsnso[L(0)] = dzsnso[L(0)] - *snoflow / bdsnow;
    *snoflow = *snoflow / dt;
  }
  if (*isnow < 0) {
    *sneqv = 0.f;
    for (int iz = *isnow + 1; iz <= 0; iz++) *sneqv = *sneqv + snice[L(iz)] + snliq[L(iz)];
  }
  for (int iz = *isnow + 1; iz <= 0; iz++) dzsnso[L(iz)] = -dzsnso[L(iz)];
  dzsnso[L(1)] = c->zsoil[L(1)];
  for (int iz = 2; iz <= ns; iz++) dzsnso[L(iz)] = (c->zsoil[L(iz)] - c->zsoil[L(iz - 1)]);
  zsnso[L(*isnow + 1)] = dzsnso[L(*isnow + 1)];
  for (int iz = *isnow + 2; iz <= ns; iz++) zsnso[L(iz)] = zsnso[L(iz - 1)] + dzsnso[L(iz)];
  for (int iz = *isnow + 1; iz <= ns; iz++) dzsnso[L(iz)] = -dzsnso[L(iz)];
}

/* WDFCND1 lsm:8329-8362, WDFCND2 lsm:8364-8400 */
static void wdfcnd1(const nmp_parm* P, real* wdf, real* wcnd, real smc, real fcr) {
  real factr = MAXF(0.01f, smc / P->smcmax);
  real expon = P->bexp + 2.0f;
  *wdf = P->dwsat * powf(factr, expon);
  *wdf = *wdf * (1.0f - fcr);
  expon = 2.0f * P->bexp + 3.0f;
  *wcnd = P->dksat * powf(factr, expon);
  *wcnd = *wcnd * (1.0f - fcr);
}
static void wdfcnd2(const nmp_parm* P, real* wdf, real* wcnd, real smc, real sice) {
  real factr = MAXF(0.01f, smc / P->smcmax);
  real expon = P->bexp + 2.0f;
  *wdf = P->dwsat * powf(factr, expon);
  if (sice > 0.0f) {
    real vkwgt = 1.f / (1.f + powf(500.f * sice, 3.f));
    *wdf = vkwgt * *wdf + (1.f - vkwgt) * P->dwsat * powf(0.2f / P->smcmax, expon);
  }
  expon = 2.0f * P->bexp + 3.0f;
  *wcnd = P->dksat * powf(factr, expon);
}

/* ZWTEQ lsm:7938-7989 */
static void zwteq(const nmp_ctx* c, const real* dzsnso, const real* sh2o, real* zwt) {
  const nmp_parm* P = &c->P;
  enum { NFINE = 100 };
  int ns = c->nsoil;
  real wd1 = 0.f, wd2, zfine[NFINE + 1];
  for (int k = 1; k <= ns; k++) wd1 = wd1 + (P->smcmax - sh2o[L(k)]) * dzsnso[L(k)];
  real dzfine = 3.0f * (-c->zsoil[L(ns)]) / NFINE;
  for (int k = 1; k <= NFINE; k++) zfine[k] = (real)k * dzfine;
  *zwt = -3.f * c->zsoil[L(ns)] - 0.001f;
  wd2 = 0.f;
  for (int k = 1; k <= NFINE; k++) {
    real temp = 1.f + (*zwt - zfine[k]) / P->psisat;
    wd2 = wd2 + P->smcmax * (1.f - powf(temp, -1.f / P->bexp)) * dzfine;
    if (fabsf(wd2 - wd1) <= 0.01f) { *zwt = zfine[k]; break; }
  }
}

/* INFIL lsm:7992-8087 */
static void infil(const nmp_ctx* c, real dt, const real* sh2o, const real* sice, real sicemax,
                  real qinsur, real* pddum, real* runsrf) {
  const nmp_parm* P = &c->P;
  const int CVFRZ = 3;
  int ns = c->nsoil;
  if (qinsur > 0.0f) {
    real dmax[NL];
    real dt1 = dt / 86400.f;
    real smcav = P->smcmax - P->smcwlt;
    dmax[L(1)] = -c->zsoil[L(1)] * smcav;
    real dice = -c->zsoil[L(1)] * sice[L(1)];
    dmax[L(1)] = dmax[L(1)] * (1.0f - (sh2o[L(1)] + sice[L(1)] - P->smcwlt) / smcav);
    real dd = dmax[L(1)];
    for (int k = 2; k <= ns; k++) {
      dice = dice + (c->zsoil[L(k - 1)] - c->zsoil[L(k)]) * sice[L(k)];
      dmax[L(k)] = (c->zsoil[L(k - 1)] - c->zsoil[L(k)]) * smcav;
      dmax[L(k)] = dmax[L(k)] * (1.0f - (sh2o[L(k)] + sice[L(k)] - P->smcwlt) / smcav);
      dd = dd + dmax[L(k)];
    }
    real val = (1.f - expf(-P->kdt * dt1));
    real ddt = dd * val;
    real px = MAXF(0.f, qinsur * dt);
    real infmax = (px * (ddt / (px + ddt))) / dt;
    real fcr = 1.f;
    if (dice > 1.E-2f) {
      real acrt = CVFRZ * P->frzx / dice;
      real sum = 1.f;
      int ialp1 = CVFRZ - 1;
      for (int j = 1; j <= ialp1; j++) {
        int k = 1;
        for (int jj = j + 1; jj <= ialp1; jj++) k = k * jj;
        sum = sum + powi(acrt, CVFRZ - j) / (real)k;
      }
      fcr = 1.f - expf(-acrt) * sum;
    }
    infmax = infmax * fcr;
    real wdf, wcnd;
    wdfcnd2(P, &wdf, &wcnd, sh2o[L(1)], sicemax);
    infmax = MAXF(infmax, wcnd);
    infmax = MINF(infmax, px);
    *runsrf = MAXF(0.f, qinsur - infmax);
    *pddum = qinsur - *runsrf;
  }
}

/* SRT lsm:8089-8217 */
static void srt(const nmp_ctx* c, real pddum, const real* etrani, real qseva, const real* sh2o,
                const real* smc, real zwt, const real* fcr, real sicemax, real fcrmax, real smcwtd,
                real* rhstt, real* ai, real* bi, real* ci, real* qdrain, real* wcnd) {
  const nmp_parm* P = &c->P;
  int ns = c->nsoil;
  real ddz[NL], denom[NL], dsmdz[NL], wflux[NL], wdf[NL], smx[NL], temp1, smxwtd = 0.f, smxbot;
  if (c->O.opt_inf == 1) {
    for (int k = 1; k <= ns; k++) {
      wdfcnd1(P, &wdf[L(k)], &wcnd[L(k)], smc[L(k)], fcr[L(k)]);
      smx[L(k)] = smc[L(k)];
    }
    if (c->O.opt_run == 5) smxwtd = smcwtd;
  }
  if (c->O.opt_inf == 2) {
    for (int k = 1; k <= ns; k++) {
      wdfcnd2(P, &wdf[L(k)], &wcnd[L(k)], sh2o[L(k)], sicemax);
      smx[L(k)] = sh2o[L(k)];
    }
    if (c->O.opt_run == 5) smxwtd = smcwtd * sh2o[L(ns)] / smc[L(ns)];
  }
  for (int k = 1; k <= ns; k++) {
    if (k == 1) {
      denom[L(k)] = -c->zsoil[L(k)];
      temp1 = -c->zsoil[L(k + 1)];
      ddz[L(k)] = 2.0f / temp1;
      dsmdz[L(k)] = 2.0f * (smx[L(k)] - smx[L(k + 1)]) / temp1;
      wflux[L(k)] = wdf[L(k)] * dsmdz[L(k)] + wcnd[L(k)] - pddum + etrani[L(k)] + qseva;
    } else if (k < ns) {
      denom[L(k)] = (c->zsoil[L(k - 1)] - c->zsoil[L(k)]);
      temp1 = (c->zsoil[L(k - 1)] - c->zsoil[L(k + 1)]);
      ddz[L(k)] = 2.0f / temp1;
      dsmdz[L(k)] = 2.0f * (smx[L(k)] - smx[L(k + 1)]) / temp1;
      wflux[L(k)] = wdf[L(k)] * dsmdz[L(k)] + wcnd[L(k)] - wdf[L(k - 1)] * dsmdz[L(k - 1)] -
                    wcnd[L(k - 1)] + etrani[L(k)];
    } else {
      denom[L(k)] = (c->zsoil[L(k - 1)] - c->zsoil[L(k)]);
      if (c->O.opt_run == 1 || c->O.opt_run == 2) *qdrain = 0.f;
      if (c->O.opt_run == 3) *qdrain = P->slope * wcnd[L(k)];
      if (c->O.opt_run == 4) *qdrain = (1.0f - fcrmax) * wcnd[L(k)];
      if (c->O.opt_run == 5) {
        temp1 = 2.0f * denom[L(k)];
        if (zwt < c->zsoil[L(ns)] - denom[L(ns)])
          smxbot = smx[L(k)] - (smx[L(k)] - smxwtd) * denom[L(k)] * 2.f /
                                   (denom[L(k)] + c->zsoil[L(k)] - zwt);
        else
          smxbot = smxwtd;
        dsmdz[L(k)] = 2.0f * (smx[L(k)] - smxbot) / temp1;
        *qdrain = wdf[L(k)] * dsmdz[L(k)] + wcnd[L(k)];
      }
      wflux[L(k)] = -(wdf[L(k - 1)] * dsmdz[L(k - 1)]) - wcnd[L(k - 1)] + etrani[L(k)] + *qdrain;
    }
  }
  for (int k = 1; k <= ns; k++) {
    if (k == 1) {
      ai[L(k)] = 0.0f;
      bi[L(k)] = wdf[L(k)] * ddz[L(k)] / denom[L(k)];
      ci[L(k)] = -bi[L(k)];
    } else if (k < ns) {
      ai[L(k)] = -wdf[L(k - 1)] * ddz[L(k - 1)] / denom[L(k)];
      ci[L(k)] = -wdf[L(k)] * ddz[L(k)] / denom[L(k)];
      bi[L(k)] = -(ai[L(k)] + ci[L(k)]);
    } else {
      ai[L(k)] = -wdf[L(k - 1)] * ddz[L(k - 1)] / denom[L(k)];
      ci[L(k)] = 0.0f;
      bi[L(k)] = -(ai[L(k)] + ci[L(k)]);
    }
    rhstt[L(k)] = wflux[L(k)] / (-denom[L(k)]);
  }
}

/* SSTEP lsm:8220-8327 */
static void sstep(const nmp_ctx* c, real dt, const real* dzsnso, const real* sice, real zwt,
                  real* sh2o, real* smc, real* ai, real* bi, real* ci, real* rhstt, real* smcwtd,
                  real* qdrain, real* deeprech, real* wplus) {
  const nmp_parm* P = &c->P;
  int ns = c->nsoil;
  real rhsttin[NL], ciin[NL], epore;
  *wplus = 0.0f;
  for (int k = 1; k <= ns; k++) {
    rhstt[L(k)] = rhstt[L(k)] * dt;
    ai[L(k)] = ai[L(k)] * dt;
    bi[L(k)] = 1.f + bi[L(k)] * dt;
    ci[L(k)] = ci[L(k)] * dt;
  }
  for (int k = 1; k <= ns; k++) { rhsttin[L(k)] = rhstt[L(k)]; ciin[L(k)] = ci[L(k)]; }
  nmp_rosr12(ci, ai, bi, ciin, rhsttin, rhstt, 1, ns);
  for (int k = 1; k <= ns; k++) sh2o[L(k)] = sh2o[L(k)] + ci[L(k)];
  if (c->O.opt_run == 5) {
    if (zwt < c->zsoil[L(ns)] - dzsnso[L(ns)]) {
      *deeprech = *deeprech + dt * *qdrain;
    } else {
      *smcwtd = *smcwtd + dt * *qdrain / dzsnso[L(ns)];
      *wplus = MAXF((*smcwtd - P->smcmax), 0.0f) * dzsnso[L(ns)];
      real wminus = MAXF((1.E-4f - *smcwtd), 0.0f) * dzsnso[L(ns)];
      *smcwtd = MAXF(MINF(*smcwtd, P->smcmax), 1.E-4f);
      sh2o[L(ns)] = sh2o[L(ns)] + *wplus / dzsnso[L(ns)];
      *qdrain = *qdrain - *wplus / dt;
      *deeprech = *deeprech - wminus;
    }
  }
  for (int k = ns; k >= 2; k--) {
    epore = MAXF(1.E-4f, (P->smcmax - sice[L(k)]));
    *wplus = MAXF((sh2o[L(k)] - epore), 0.0f) * dzsnso[L(k)];
    sh2o[L(k)] = MINF(epore, sh2o[L(k)]);
    sh2o[L(k - 1)] = sh2o[L(k - 1)] + *wplus / dzsnso[L(k - 1)];
  }
  epore = MAXF(1.E-4f, (P->smcmax - sice[L(1)]));
  *wplus = MAXF((sh2o[L(1)] - epore), 0.0f) * dzsnso[L(1)];
  sh2o[L(1)] = MINF(epore, sh2o[L(1)]);
  for (int k = 1; k <= ns; k++) smc[L(k)] = sh2o[L(k)] + sice[L(k)];
}

/* SOILWATER lsm:7680-7936 */
static void soilwater(const nmp_ctx* c, real dt, const real* dzsnso, real qinsur, real qseva,
                      const real* etrani, const real* sice, real* sh2o, real* smc, real* zwt,
                      real* smcwtd, real* deeprech, real* runsrf, real* qdrain, real* runsub,
                      real* wcnd, real* fcrmax) {
  const nmp_parm* P = &c->P;
  const real A = 4.0f;
  int ns = c->nsoil;
  real rhstt[NL], ai[NL], bi[NL], ci[NL], fcr[NL], mliq[NL];
  real fff, rsbmx, fsat, pddum = 0.0f, rsat = 0.0f, wplus, sicemax, sh2omin;
  *runsrf = 0.0f;
  for (int k = 1; k <= ns; k++) {
    real epore = MAXF(1.E-4f, (P->smcmax - sice[L(k)]));
    rsat = rsat + MAXF(0.f, sh2o[L(k)] - epore) * dzsnso[L(k)];
    sh2o[L(k)] = MINF(epore, sh2o[L(k)]);
  }
  for (int k = 1; k <= ns; k++) {
    real fice = MINF(1.0f, sice[L(k)] / P->smcmax);
    fcr[L(k)] = MAXF(0.0f, expf(-A * (1.f - fice)) - expf(-A)) / (1.0f - expf(-A));
  }
  sicemax = 0.0f; *fcrmax = 0.0f; sh2omin = P->smcmax;
  for (int k = 1; k <= ns; k++) {
    if (sice[L(k)] > sicemax) sicemax = sice[L(k)];
    if (fcr[L(k)] > *fcrmax) *fcrmax = fcr[L(k)];
    if (sh2o[L(k)] < sh2omin) sh2omin = sh2o[L(k)];
  }
  if (c->O.opt_run == 2) {
    fff = 2.0f; rsbmx = 4.0f;
    zwteq(c, dzsnso, sh2o, zwt);
    *runsub = (1.0f - *fcrmax) * rsbmx * expf(-TIMEAN) * expf(-fff * *zwt);
  }
  if (c->vegtyp == c->isurban) fcr[L(1)] = 0.95f;
  if (c->O.opt_run == 1) {
    fff = 6.0f;
    fsat = FSATMX * expf(-0.5f * fff * (*zwt - 2.0f));
    if (qinsur > 0.f) {
      *runsrf = qinsur * ((1.0f - fcr[L(1)]) * fsat + fcr[L(1)]);
      pddum = qinsur - *runsrf;
    }
  }
  if (c->O.opt_run == 5) {
    fff = 6.0f;
    fsat = FSATMX * expf(-0.5f * fff * MAXF(-2.0f - *zwt, 0.f));
    if (qinsur > 0.f) {
      *runsrf = qinsur * ((1.0f - fcr[L(1)]) * fsat + fcr[L(1)]);
      pddum = qinsur - *runsrf;
    }
  }
  if (c->O.opt_run == 2) {
    fff = 2.0f;
    fsat = FSATMX * expf(-0.5f * fff * *zwt);
    if (qinsur > 0.f) {
      *runsrf = qinsur * ((1.0f - fcr[L(1)]) * fsat + fcr[L(1)]);
      pddum = qinsur - *runsrf;
    }
  }
  if (c->O.opt_run == 3) infil(c, dt, sh2o, sice, sicemax, qinsur, &pddum, runsrf);
  if (c->O.opt_run == 4) {
    real smctot = 0.f, dztot = 0.f;
    for (int k = 1; k <= ns; k++) {
      dztot = dztot + dzsnso[L(k)];
      smctot = smctot + smc[L(k)] * dzsnso[L(k)];
      if (dztot >= 2.0f) break;
    }
    smctot = smctot / dztot;
    fsat = powf(MAXF(0.01f, smctot / P->smcmax), 4.f);
    if (qinsur > 0.f) {
      *runsrf = qinsur * ((1.0f - fcr[L(1)]) * fsat + fcr[L(1)]);
      pddum = qinsur - *runsrf;
    }
  }
  int niter = 1;
  if (c->O.opt_inf == 1) {
    niter = 3;
    if (pddum * dt > dzsnso[L(1)] * P->smcmax) niter = niter * 2;
  }
  real dtfine = dt / niter;
  real qdrain_save = 0.0f;
  for (int iter = 1; iter <= niter; iter++) {
    srt(c, pddum, etrani, qseva, sh2o, smc, *zwt, fcr, sicemax, *fcrmax, *smcwtd, rhstt, ai, bi, ci,
        qdrain, wcnd);
    sstep(c, dtfine, dzsnso, sice, *zwt, sh2o, smc, ai, bi, ci, rhstt, smcwtd, qdrain, deeprech,
          &wplus);
    rsat = rsat + wplus;
    qdrain_save = qdrain_save + *qdrain;
  }
  *qdrain = qdrain_save / niter;
  *runsrf = *runsrf * 1000.f + rsat * 1000.f / dt;
  *qdrain = *qdrain * 1000.f;
  if (c->O.opt_run == 2) {
    real wtsub = 0.f;
    for (int k = 1; k <= ns; k++) wtsub = wtsub + wcnd[L(k)] * dzsnso[L(k)];
    for (int k = 1; k <= ns; k++) {
      real mh2o = *runsub * dt * (wcnd[L(k)] * dzsnso[L(k)]) / wtsub;
      sh2o[L(k)] = sh2o[L(k)] - mh2o / (dzsnso[L(k)] * 1000.f);
    }
  }
  if (c->O.opt_run != 1) {
    real xs, watmin = 0.01f;
    for (int iz = 1; iz <= ns; iz++) mliq[L(iz)] = sh2o[L(iz)] * dzsnso[L(iz)] * 1000.f;
    for (int iz = 1; iz <= ns - 1; iz++) {
      if (mliq[L(iz)] < 0.f) xs = watmin - mliq[L(iz)]; else xs = 0.f;
      mliq[L(iz)] = mliq[L(iz)] + xs;
      mliq[L(iz + 1)] = mliq[L(iz + 1)] - xs;
    }
    int iz = ns;
    if (mliq[L(iz)] < watmin) xs = watmin - mliq[L(iz)]; else xs = 0.f;
    mliq[L(iz)] = mliq[L(iz)] + xs;
    *runsub = *runsub - xs / dt;
    if (c->O.opt_run == 5) *deeprech = *deeprech - xs * 1.E-3f;
    for (iz = 1; iz <= ns; iz++) sh2o[L(iz)] = mliq[L(iz)] / (dzsnso[L(iz)] * 1000.f);
  }
  (void)sh2omin;
}

/* GROUNDWATER lsm:8403-8585 (SIMGM); S_NODE is float64 in the reference (lsm:8443) */
static void groundwater(const nmp_ctx* c, real dt, const real* sice, const real* wcnd, real fcrmax,
                        real* sh2o, real* zwt, real* wa, real* wt, real* qin, real* qdis) {
  const nmp_parm* P = &c->P;
  const real ROUS = 0.2f, CMIC = 0.20f;
  int ns = c->nsoil, iwt;
  real dzmm[NL], znode[NL], mliq[NL], epore[NL], hk[NL], smc[NL];
  *qdis = 0.0f; *qin = 0.0f;
  dzmm[L(1)] = -c->zsoil[L(1)] * 1.E3f;
  for (int iz = 2; iz <= ns; iz++) dzmm[L(iz)] = 1.E3f * (c->zsoil[L(iz - 1)] - c->zsoil[L(iz)]);
  znode[L(1)] = -c->zsoil[L(1)] / 2.f;
  for (int iz = 2; iz <= ns; iz++)
    znode[L(iz)] = -c->zsoil[L(iz - 1)] + 0.5f * (c->zsoil[L(iz - 1)] - c->zsoil[L(iz)]);
  for (int iz = 1; iz <= ns; iz++) {
    smc[L(iz)] = sh2o[L(iz)] + sice[L(iz)];
    mliq[L(iz)] = sh2o[L(iz)] * dzmm[L(iz)];
    epore[L(iz)] = MAXF(0.01f, P->smcmax - sice[L(iz)]);
    hk[L(iz)] = 1.E3f * wcnd[L(iz)];
  }
  iwt = ns;
  for (int iz = 2; iz <= ns; iz++) {
    if (*zwt <= -c->zsoil[L(iz)]) { iwt = iz - 1; break; }
  }
  real fff = 6.0f, rsbmx = 5.0f;
  *qdis = (1.0f - fcrmax) * rsbmx * expf(-TIMEAN) * expf(-fff * (*zwt - 2.0f));
  double s_node = MINF(1.0f, smc[L(iwt)] / P->smcmax);
  s_node = (s_node > (double)0.01f) ? s_node : (double)0.01f;
  real smpfz = (real)(-((double)(P->psisat * 1000.f) * pow(s_node, (double)(-P->bexp))));
  smpfz = MAXF(-120000.0f, CMIC * smpfz);
  real ka = hk[L(iwt)];
  real wh_zwt = -*zwt * 1.E3f;
  real wh = smpfz - znode[L(iwt)] * 1.E3f;
  *qin = -ka * (wh_zwt - wh) / ((*zwt - znode[L(iwt)]) * 1.E3f);
  *qin = MAXF(-10.0f / dt, MINF(10.f / dt, *qin));
  *wt = *wt + (*qin - *qdis) * dt;
  if (iwt == ns) {
    *wa = *wa + (*qin - *qdis) * dt;
    *wt = *wa;
    *zwt = (-c->zsoil[L(ns)] + 25.f) - *wa / 1000.f / ROUS;
    mliq[L(ns)] = mliq[L(ns)] - *qin * dt;
    mliq[L(ns)] = mliq[L(ns)] + MAXF(0.f, (*wa - 5000.f));
    *wa = MINF(*wa, 5000.f);
  } else {
    if (iwt == ns - 1) {
      *zwt = -c->zsoil[L(ns)] - (*wt - ROUS * 1000 * 25.f) / (epore[L(ns)]) / 1000.f;
    } else {
      real ws = 0.f;
      for (int iz = iwt + 2; iz <= ns; iz++) ws = ws + epore[L(iz)] * dzmm[L(iz)];
      *zwt = -c->zsoil[L(iwt + 1)] - (*wt - ROUS * 1000.f * 25.f - ws) / (epore[L(iwt + 1)]) / 1000.f;
    }
    real wtsub = 0.f;
    for (int iz = 1; iz <= ns; iz++) wtsub = wtsub + hk[L(iz)] * dzmm[L(iz)];
    for (int iz = 1; iz <= ns; iz++)
      mliq[L(iz)] = mliq[L(iz)] - *qdis * dt * hk[L(iz)] * dzmm[L(iz)] / wtsub;
  }
  *zwt = MAXF(1.5f, *zwt);
  real xs, watmin = 0.01f;
  for (int iz = 1; iz <= ns - 1; iz++) {
    if (mliq[L(iz)] < 0.f) xs = watmin - mliq[L(iz)]; else xs = 0.f;
    mliq[L(iz)] = mliq[L(iz)] + xs;
    mliq[L(iz + 1)] = mliq[L(iz + 1)] - xs;
  }
  int iz = ns;
  if (mliq[L(iz)] < watmin) xs = watmin - mliq[L(iz)]; else xs = 0.f;
  mliq[L(iz)] = mliq[L(iz)] + xs;
  *wa = *wa - xs;
  *wt = *wt - xs;
  for (iz = 1; iz <= ns; iz++) sh2o[L(iz)] = mliq[L(iz)] / dzmm[L(iz)];
}

/* SHALLOWWATERTABLE lsm:8588-8718 (MMF in-column water-table diagnosis, OPT_RUN=5) */
static void shallowwatertable(const nmp_ctx* c, const real* dzsnso, const real* smceq, const real* smc,
                              real* wtd, real* smcwtd, real* rech) {
  const nmp_parm* P = &c->P;
  int ns = c->nsoil, iz, iwtd, kwtd;
  real wtdold = 0.f, dzup, smceqdeep;
  /* ZSOIL0(0:NSOIL): index 0 == L(0) slot */
  real zsoil0[NL];
  for (int k = 1; k <= ns; k++) zsoil0[L(k)] = c->zsoil[L(k)];
  zsoil0[L(0)] = 0.f;
  for (iz = ns; iz >= 1; iz--)
    if (*wtd + 1.E-6f < zsoil0[L(iz)]) break;
  iwtd = iz;
  kwtd = iwtd + 1;
  if (kwtd <= ns) {
    wtdold = *wtd;
    if (smc[L(kwtd)] > smceq[L(kwtd)]) {
      if (smc[L(kwtd)] == P->smcmax) {
        *wtd = zsoil0[L(iwtd)];
        *rech = -(wtdold - *wtd) * (P->smcmax - smceq[L(kwtd)]);
        iwtd = iwtd - 1;
        kwtd = kwtd - 1;
        if (kwtd >= 1) {
          if (smc[L(kwtd)] > smceq[L(kwtd)]) {
            wtdold = *wtd;
            *wtd = MINF((smc[L(kwtd)] * dzsnso[L(kwtd)] - smceq[L(kwtd)] * zsoil0[L(iwtd)] +
                         P->smcmax * zsoil0[L(kwtd)]) / (P->smcmax - smceq[L(kwtd)]),
                        zsoil0[L(iwtd)]);
            *rech = *rech - (wtdold - *wtd) * (P->smcmax - smceq[L(kwtd)]);
          }
        }
      } else {
        *wtd = MINF((smc[L(kwtd)] * dzsnso[L(kwtd)] - smceq[L(kwtd)] * zsoil0[L(iwtd)] +
                     P->smcmax * zsoil0[L(kwtd)]) / (P->smcmax - smceq[L(kwtd)]),
                    zsoil0[L(iwtd)]);
        *rech = -(wtdold - *wtd) * (P->smcmax - smceq[L(kwtd)]);
      }
    } else {
      *wtd = zsoil0[L(kwtd)];
      *rech = -(wtdold - *wtd) * (P->smcmax - smceq[L(kwtd)]);
      kwtd = kwtd + 1;
      iwtd = iwtd + 1;
      if (kwtd <= ns) {
        wtdold = *wtd;
        if (smc[L(kwtd)] > smceq[L(kwtd)])
          *wtd = MINF((smc[L(kwtd)] * dzsnso[L(kwtd)] - smceq[L(kwtd)] * zsoil0[L(iwtd)] +
                       P->smcmax * zsoil0[L(kwtd)]) / (P->smcmax - smceq[L(kwtd)]),
                      zsoil0[L(iwtd)]);
        else
          *wtd = zsoil0[L(kwtd)];
        *rech = *rech - (wtdold - *wtd) * (P->smcmax - smceq[L(kwtd)]);
      } else {
        wtdold = *wtd;
        smceqdeep = P->smcmax * powf(-P->psisat / (-P->psisat - dzsnso[L(ns)]), 1.f / P->bexp);
        *wtd = MINF((*smcwtd * dzsnso[L(ns)] - smceqdeep * zsoil0[L(ns)] +
                     P->smcmax * (zsoil0[L(ns)] - dzsnso[L(ns)])) / (P->smcmax - smceqdeep),
                    zsoil0[L(ns)]);
        *rech = *rech - (wtdold - *wtd) * (P->smcmax - smceqdeep);
      }
    }
  } else if (*wtd >= zsoil0[L(ns)] - dzsnso[L(ns)]) {
    wtdold = *wtd;
    smceqdeep = P->smcmax * powf(-P->psisat / (-P->psisat - dzsnso[L(ns)]), 1.f / P->bexp);
    if (*smcwtd > smceqdeep) {
      *wtd = MINF((*smcwtd * dzsnso[L(ns)] - smceqdeep * zsoil0[L(ns)] +
                   P->smcmax * (zsoil0[L(ns)] - dzsnso[L(ns)])) / (P->smcmax - smceqdeep),
                  zsoil0[L(ns)]);
      *rech = -(wtdold - *wtd) * (P->smcmax - smceqdeep);
    } else {
      *rech = -(wtdold - (zsoil0[L(ns)] - dzsnso[L(ns)])) * (P->smcmax - smceqdeep);
      wtdold = zsoil0[L(ns)] - dzsnso[L(ns)];
      dzup = (smceqdeep - *smcwtd) * dzsnso[L(ns)] / (P->smcmax - smceqdeep);
      *wtd = wtdold - dzup;
      *rech = *rech - (P->smcmax - smceqdeep) * dzup;
      *smcwtd = smceqdeep;
    }
  }
  if (iwtd < ns) *smcwtd = P->smcmax;
}

/* WATER lsm:6382-6613 */
void nmp_water(nmp_ctx* c, nmp_column* s, nmp_work* w, real qvap, real qdew) {
  const nmp_parm* P = &c->P;
  const real WSLMAX = 5000.f;
  int ns = c->nsoil;
  real etrani[NL], wcnd[NL];
  real snoflow = 0.f, qinsur = 0.f, qrain, snowhin, qsnsub, qseva, qsnfro, qsdew, qdrain = 0.f,
       fcrmax = 0.f, qin, qdis;
  real dt = c->dt;
  for (int iz = 1; iz <= ns; iz++) etrani[L(iz)] = 0.f;
  s->runsub = 0.f;
  canwater(c, dt, s->sfctmp, s->uu, s->vv, s->fcev, s->fctr, w->qprecc, w->qprecl, w->elai, w->esai,
           s->ist, s->tg, s->fveg, w->frozen_canopy, &s->canliq, &s->canice, &s->tv, &s->ecan,
           &s->etran, &qrain, &s->qsnow, &snowhin, &s->fwet, &s->fpice);
  qsnsub = 0.f;
  if (s->sneqv > 0.f) qsnsub = MINF(qvap, s->sneqv / dt);
  qseva = qvap - qsnsub;
  qsnfro = 0.f;
  if (s->sneqv > 0.f) qsnfro = qdew;
  qsdew = qdew - qsnfro;
  snowwater(c, w->imelt, dt, s->sfctmp, snowhin, s->qsnow, qsnfro, qsnsub, qrain, s->ficeold,
            &s->isnow, &s->snowh, &s->sneqv, s->snice, s->snliq, s->sh2o, w->sice, s->stc, s->zsnso,
            w->dzsnso, &s->qsnbot, &snoflow, &s->ponding1, &s->ponding2);
  if (w->frozen_ground) {
    w->sice[L(1)] = w->sice[L(1)] + (qsdew - qseva) * dt / (w->dzsnso[L(1)] * 1000.f);
    qsdew = 0.0f;
    qseva = 0.0f;
    if (w->sice[L(1)] < 0.f) {
      s->sh2o[L(1)] = s->sh2o[L(1)] + w->sice[L(1)];
      w->sice[L(1)] = 0.f;
    }
  }
  qinsur = (s->ponding + s->ponding1 + s->ponding2) / dt * 0.001f;
  if (s->isnow == 0) qinsur = qinsur + (s->qsnbot + qsdew + qrain) * 0.001f;
  else qinsur = qinsur + (s->qsnbot + qsdew) * 0.001f;
  qseva = qseva * 0.001f;
  for (int iz = 1; iz <= P->nroot; iz++) etrani[L(iz)] = s->etran * w->btrani[L(iz)] * 0.001f;
  if (s->ist == 2) {
    s->runsrf = 0.f;
    if (s->wslake >= WSLMAX) s->runsrf = qinsur * 1000.f;
    s->wslake = s->wslake + (qinsur - qseva) * 1000.f * dt - s->runsrf * dt;
  } else {
    soilwater(c, dt, w->dzsnso, qinsur, qseva, etrani, w->sice, s->sh2o, s->smc, &s->zwt, &s->smcwtd,
              &s->deeprech, &s->runsrf, &qdrain, &s->runsub, wcnd, &fcrmax);
    if (c->O.opt_run == 1) {
      groundwater(c, dt, w->sice, wcnd, fcrmax, s->sh2o, &s->zwt, &s->wa, &s->wt, &qin, &qdis);
      s->runsub = qdis;
    }
    if (c->O.opt_run == 3 || c->O.opt_run == 4) s->runsub = s->runsub + qdrain;
    for (int iz = 1; iz <= ns; iz++) s->smc[L(iz)] = s->sh2o[L(iz)] + w->sice[L(iz)];
    if (c->O.opt_run == 5) {
      shallowwatertable(c, w->dzsnso, s->smceq, s->smc, &s->zwt, &s->smcwtd, &s->rech);
      s->sh2o[L(ns)] = s->smc[L(ns)] - w->sice[L(ns)];
      s->runsub = s->runsub + qdrain;
      s->wa = 0.f;
    }
  }
  s->runsub = s->runsub + snoflow;
}

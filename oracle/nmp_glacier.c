/* TEST INFRASTRUCTURE (oracle): land-ice column, NOAHMP_GLACIER.
 * Reference: phys/module_sf_noahmp_glacier.F90 ("gla").  Routines that are textually identical to
 * their land twins (SFCDIF1, ESAT, HRT/HSTEP/ROSR12, COMPACT, COMBO) or differ only by a constant
 * (COMBINE, DIVIDE, SNOWH2O) are shared with nmp_energy.c / nmp_water.c through a `glacier` flag;
 * the differences are listed in SURVEY.md 8(a.2). */
#include <math.h>
#include <string.h>
#include "nmp_internal.h"

/* THERMOPROP_GLACIER gla:575-645 (+CSNOW_GLACIER gla:648-702) */
static void thermoprop_glacier(const nmp_ctx* c, int isnow, const real* dzsnso, real dt, real snowh,
                               const real* snice, const real* snliq, real* df, real* hcpct, real* fact) {
  int ns = c->nsoil;
  for (int iz = isnow + 1; iz <= 0; iz++) {
    real snicev = MINF(1.f, snice[L(iz)] / (dzsnso[L(iz)] * DENICE));
    real epore = 1.f - snicev;
    real snliqv = MINF(epore, snliq[L(iz)] / (dzsnso[L(iz)] * DENH2O));
    real bdsnoi = (snice[L(iz)] + snliq[L(iz)]) / dzsnso[L(iz)];
    hcpct[L(iz)] = CICE * snicev + CWAT * snliqv;
    df[L(iz)] = 3.2217E-6f * powf(bdsnoi, 2.f);
  }
  for (int iz = 1; iz <= ns; iz++) {                 /* ice properties from depth, gla:621-628 */
    real zmid = 0.5f * (dzsnso[L(iz)]);
    for (int iz2 = 1; iz2 <= iz - 1; iz2++) zmid = zmid + dzsnso[L(iz2)];
    hcpct[L(iz)] = 1.E6f * (0.8194f + 0.1309f * zmid);
    df[L(iz)] = 0.32333f + (0.10073f * zmid);
  }
  for (int iz = isnow + 1; iz <= ns; iz++) fact[L(iz)] = dt / (hcpct[L(iz)] * dzsnso[L(iz)]);
  if (isnow == 0)
    df[L(1)] = (df[L(1)] * dzsnso[L(1)] + 0.35f * snowh) / (snowh + dzsnso[L(1)]);
  else
    df[L(1)] = (df[L(1)] * dzsnso[L(1)] + df[L(0)] * dzsnso[L(0)]) / (dzsnso[L(0)] + dzsnso[L(1)]);
}

/* RADIATION_GLACIER gla:704-792: snow ages even at night (no COSZ gate), FSNO = 1 iff SNEQV > 0 */
static void radiation_glacier(const nmp_ctx* c, real dt, real tg, real sneqvo, real sneqv, real cosz,
                              real qsnow, const real* solad, const real* solai, real* albold,
                              real* tauss, real* sag, real* fsr, real* fsa) {
  real albsnd[2] = {0.f, 0.f}, albsni[2] = {0.f, 0.f};
  const real albice[2] = {0.80f, 0.55f};
  real fage;
  nmp_snow_age(dt, tg, sneqvo, sneqv, tauss, &fage);
  if (c->O.opt_alb == 1) {
    const real C1 = 0.2f, C2 = 0.5f;
    real sl = 2.0f, sl1 = 1.f / sl, sl2 = 2.f * sl;
    real cf1 = ((1.f + sl1) / (1.f + sl2 * cosz) - sl1);
    real fzen = MAXF(cf1, 0.f);
    albsni[0] = 0.95f * (1.f - C1 * fage);
    albsni[1] = 0.65f * (1.f - C2 * fage);
    albsnd[0] = albsni[0] + 0.4f * fzen * (1.f - albsni[0]);
    albsnd[1] = albsni[1] + 0.4f * fzen * (1.f - albsni[1]);
  }
  if (c->O.opt_alb == 2) {
    real alb = 0.55f + (*albold - 0.55f) * expf(-0.01f * dt / 3600.f);
    if (qsnow > 0.f) alb = alb + MINF(qsnow * dt, SWEMX) * (0.84f - alb) / (SWEMX);
    albsni[0] = albsni[1] = albsnd[0] = albsnd[1] = alb;
    *albold = alb;
  }
  *sag = 0.f; *fsa = 0.f; *fsr = 0.f;
  real fsno = 0.0f;
  if (sneqv > 0.0f) fsno = 1.0f;
  for (int ib = 0; ib < 2; ib++) {
    albsnd[ib] = albice[ib] * (1.f - fsno) + albsnd[ib] * fsno;
    albsni[ib] = albice[ib] * (1.f - fsno) + albsni[ib] * fsno;
    real abs_ = solad[ib] * (1.f - albsnd[ib]) + solai[ib] * (1.f - albsni[ib]);
    *sag = *sag + abs_;
    *fsa = *fsa + abs_;
    real ref = solad[ib] * albsnd[ib] + solai[ib] * albsni[ib];
    *fsr = *fsr + ref;
  }
}

/* GLACIER_FLUX gla:942-1148 */
static void glacier_flux(nmp_ctx* c, real emg, int isnow, const real* df, const real* dzsnso, real z0m,
                         real zlvl, real zpd, real qair, real sfctmp, real rhoair, real sfcprs, real ur,
                         real gamma, real rsurf, real lwdn, real rhsur, const real* smc, real eair,
                         const real* stc, real sag, real snowh, real lathea, const real* sh2o, real* cm,
                         real* ch, real* tgb, real* qsfc, real* irb, real* shb, real* evb, real* ghb,
                         real* t2mb, real* q2b, real* ehb2) {
  const real MPE = 1E-6f;
  mo_state mo = {0.f, 0.f, 0.f, 0.f, 0.f, 0.1f, 0};
  real h = 0.f, z0h = z0m, ch2, t, esatw, esati, dsatw, dsati, estg = 0.f, destg, csh = 0.f, cev = 0.f,
       rahb = 1.f;
  real cir = emg * SB;
  real cgh = 2.f * df[L(isnow + 1)] / dzsnso[L(isnow + 1)];
  for (int iter = 1; iter <= 5; iter++) {
    z0h = z0m;
    nmp_sfcdif1(c, iter, sfctmp, rhoair, h, qair, zlvl, zpd, z0m, z0h, ur, MPE, &mo, cm, ch, &ch2);
    if (c->err) return;
    rahb = MAXF(1.f, 1.f / (*ch * ur));
    real rawb = rahb;
    t = nmp_tdc(*tgb);
    nmp_esat(t, &esatw, &esati, &dsatw, &dsati);
    if (t > 0.f) { estg = esatw; destg = dsatw; } else { estg = esati; destg = dsati; }
    csh = rhoair * CPAIR / rahb;
    cev = rhoair * CPAIR / gamma / (rsurf + rawb);
    *irb = cir * powi(*tgb, 4) - emg * lwdn;
    *shb = csh * (*tgb - sfctmp);
    *evb = cev * (estg * rhsur - eair);
    *ghb = cgh * (*tgb - stc[L(isnow + 1)]);
    real b = sag - *irb - *shb - *evb - *ghb;
    real a = 4.f * cir * powi(*tgb, 3) + csh + cev * destg + cgh;
    real dtg = b / a;
    *irb = *irb + 4.f * cir * powi(*tgb, 3) * dtg;
    *shb = *shb + csh * dtg;
    *evb = *evb + cev * destg * dtg;
    *ghb = *ghb + cgh * dtg;
    *tgb = *tgb + dtg;
    h = csh * (*tgb - sfctmp);
    t = nmp_tdc(*tgb);
    nmp_esat(t, &esatw, &esati, &dsatw, &dsati);
    estg = (t > 0.f) ? esatw : esati;
    *qsfc = 0.622f * (estg * rhsur) / (sfcprs - 0.378f * (estg * rhsur));
  }
  real sicemax = -1.e30f;                             /* MAXVAL(SMC - SH2O), gla:1123-1125 */
  for (int k = 1; k <= c->nsoil; k++) sicemax = MAXF(sicemax, smc[L(k)] - sh2o[L(k)]);
  if (c->O.opt_stc == 1) {
    if ((sicemax > 0.0f || snowh > 0.0f) && *tgb > TFRZ) {
      *tgb = TFRZ;
      *irb = cir * powi(*tgb, 4) - emg * lwdn;
      *shb = csh * (*tgb - sfctmp);
      *evb = cev * (estg * rhsur - eair);
      *ghb = sag - (*irb + *shb + *evb);
    }
  }
  *ehb2 = mo.fv * VKC / (logf((2.f + z0h) / z0h) - mo.fh2);
  real cq2b = *ehb2;
  if (*ehb2 < 1.E-5f) {
    *t2mb = *tgb;
    *q2b = *qsfc;
  } else {
    *t2mb = *tgb - *shb / (rhoair * CPAIR) * 1.f / *ehb2;
    *q2b = *qsfc - *evb / (lathea * rhoair) * (1.f / cq2b + rsurf);
  }
  *ch = 1.f / rahb;
}

/* PHASECHANGE_GLACIER gla:1635-1922 */
static void phasechange_glacier(const nmp_ctx* c, int isnow, real dt, const real* fact,
                                const real* dzsnso, real* stc, real* snice, real* snliq, real* sneqv,
                                real* snowh, real* smc, real* sh2o, real* qmelt, int* imelt,
                                real* ponding) {
  int ns = c->nsoil;
  real hm[NL], xm[NL], wmass0[NL], wice0[NL], mice[NL], mliq[NL], heatr[NL];
  real xmf = 0.f;
  *qmelt = 0.f; *ponding = 0.f;
  memset(heatr, 0, sizeof(heatr)); memset(xm, 0, sizeof(xm)); memset(hm, 0, sizeof(hm));
  for (int j = isnow + 1; j <= 0; j++) { mice[L(j)] = snice[L(j)]; mliq[L(j)] = snliq[L(j)]; }
  for (int j = 1; j <= ns; j++) {
    mliq[L(j)] = sh2o[L(j)] * dzsnso[L(j)] * 1000.f;
    mice[L(j)] = (smc[L(j)] - sh2o[L(j)]) * dzsnso[L(j)] * 1000.f;
  }
  for (int j = isnow + 1; j <= ns; j++) {
    imelt[L(j)] = 0; hm[L(j)] = 0.f; xm[L(j)] = 0.f;
    wice0[L(j)] = mice[L(j)]; wmass0[L(j)] = mice[L(j)] + mliq[L(j)];
  }
  for (int j = isnow + 1; j <= ns; j++) {
    if (mice[L(j)] > 0.f && stc[L(j)] >= TFRZ) imelt[L(j)] = 1;
    if (mliq[L(j)] > 0.f && stc[L(j)] < TFRZ) imelt[L(j)] = 2;
    if (isnow == 0 && *sneqv > 0.f && j == 1) {
      if (stc[L(j)] >= TFRZ) imelt[L(j)] = 1;
    }
  }
  for (int j = isnow + 1; j <= ns; j++) {
    if (imelt[L(j)] > 0) { hm[L(j)] = (stc[L(j)] - TFRZ) / fact[L(j)]; stc[L(j)] = TFRZ; }
    if (imelt[L(j)] == 1 && hm[L(j)] < 0.f) { hm[L(j)] = 0.f; imelt[L(j)] = 0; }
    if (imelt[L(j)] == 2 && hm[L(j)] > 0.f) { hm[L(j)] = 0.f; imelt[L(j)] = 0; }
    xm[L(j)] = hm[L(j)] * dt / HFUS;
  }
  if (isnow == 0 && *sneqv > 0.f && xm[L(1)] > 0.f) {
    real temp1 = *sneqv;
    *sneqv = MAXF(0.f, temp1 - xm[L(1)]);
    real propor = *sneqv / temp1;
    *snowh = MAXF(0.f, propor * *snowh);
    heatr[L(1)] = hm[L(1)] - HFUS * (temp1 - *sneqv) / dt;
    if (heatr[L(1)] > 0.f) { xm[L(1)] = heatr[L(1)] * dt / HFUS; hm[L(1)] = heatr[L(1)]; imelt[L(1)] = 1; }
    else { xm[L(1)] = 0.f; hm[L(1)] = 0.f; imelt[L(1)] = 0; }
    *qmelt = MAXF(0.f, (temp1 - *sneqv)) / dt;
    xmf = HFUS * *qmelt;
    *ponding = temp1 - *sneqv;
  }
  for (int j = isnow + 1; j <= ns; j++) {
    if (imelt[L(j)] > 0 && fabsf(hm[L(j)]) > 0.f) {
      heatr[L(j)] = 0.f;
      if (xm[L(j)] > 0.f) {
        mice[L(j)] = MAXF(0.f, wice0[L(j)] - xm[L(j)]);
        heatr[L(j)] = hm[L(j)] - HFUS * (wice0[L(j)] - mice[L(j)]) / dt;
      } else if (xm[L(j)] < 0.f) {
        mice[L(j)] = MINF(wmass0[L(j)], wice0[L(j)] - xm[L(j)]);
        heatr[L(j)] = hm[L(j)] - HFUS * (wice0[L(j)] - mice[L(j)]) / dt;
      }
      mliq[L(j)] = MAXF(0.f, wmass0[L(j)] - mice[L(j)]);
      if (fabsf(heatr[L(j)]) > 0.f) {
        stc[L(j)] = stc[L(j)] + fact[L(j)] * heatr[L(j)];
        if (j <= 0) { if (mliq[L(j)] * mice[L(j)] > 0.f) stc[L(j)] = TFRZ; }
      }
      if (j > 0) xmf = xmf + HFUS * (wice0[L(j)] - mice[L(j)]) / dt;
      if (j < 1) *qmelt = *qmelt + MAXF(0.f, (wice0[L(j)] - mice[L(j)])) / dt;
    }
  }
  memset(heatr, 0, sizeof(heatr));
  memset(xm, 0, sizeof(xm));
  /* four residual-redistribution passes over the (hard-coded) 4 ice layers, gla:1804-1908 */
#define ANY_GT(a) (a[L(1)] > TFRZ || a[L(2)] > TFRZ || a[L(3)] > TFRZ || a[L(4)] > TFRZ)
#define ANY_LT(a) (a[L(1)] < TFRZ || a[L(2)] < TFRZ || a[L(3)] < TFRZ || a[L(4)] < TFRZ)
#define ANY_POS(a) (a[L(1)] > 0.f || a[L(2)] > 0.f || a[L(3)] > 0.f || a[L(4)] > 0.f)
  if (ANY_GT(stc) && ANY_LT(stc)) {
    for (int j = 1; j <= ns; j++) {
      if (stc[L(j)] > TFRZ) {
        heatr[L(j)] = (stc[L(j)] - TFRZ) / fact[L(j)];
        for (int k = 1; k <= ns; k++) {
          if (j != k && stc[L(k)] < TFRZ && heatr[L(j)] > 0.1f) {
            heatr[L(k)] = (stc[L(k)] - TFRZ) / fact[L(k)];
            if (fabsf(heatr[L(k)]) > heatr[L(j)]) {
              heatr[L(k)] = heatr[L(k)] + heatr[L(j)];
              stc[L(k)] = TFRZ + heatr[L(k)] * fact[L(k)];
              heatr[L(j)] = 0.0f;
            } else {
              heatr[L(j)] = heatr[L(j)] + heatr[L(k)];
              heatr[L(k)] = 0.0f;
              stc[L(k)] = TFRZ;
            }
          }
        }
        stc[L(j)] = TFRZ + heatr[L(j)] * fact[L(j)];
      }
    }
  }
  if (ANY_GT(stc) && ANY_LT(stc)) {
    for (int j = 1; j <= ns; j++) {
      if (stc[L(j)] < TFRZ) {
        heatr[L(j)] = (stc[L(j)] - TFRZ) / fact[L(j)];
        for (int k = 1; k <= ns; k++) {
          if (j != k && stc[L(k)] > TFRZ && heatr[L(j)] < -0.1f) {
            heatr[L(k)] = (stc[L(k)] - TFRZ) / fact[L(k)];
            if (heatr[L(k)] > fabsf(heatr[L(j)])) {
              heatr[L(k)] = heatr[L(k)] + heatr[L(j)];
              stc[L(k)] = TFRZ + heatr[L(k)] * fact[L(k)];
              heatr[L(j)] = 0.0f;
            } else {
              heatr[L(j)] = heatr[L(j)] + heatr[L(k)];
              heatr[L(k)] = 0.0f;
              stc[L(k)] = TFRZ;
            }
          }
        }
        stc[L(j)] = TFRZ + heatr[L(j)] * fact[L(j)];
      }
    }
  }
  if (ANY_GT(stc) && ANY_POS(mice)) {
    for (int j = 1; j <= ns; j++) {
      if (stc[L(j)] > TFRZ) {
        heatr[L(j)] = (stc[L(j)] - TFRZ) / fact[L(j)];
        xm[L(j)] = heatr[L(j)] * dt / HFUS;
        for (int k = 1; k <= ns; k++) {
          if (j != k && mice[L(k)] > 0.f && xm[L(j)] > 0.1f) {
            if (mice[L(k)] > xm[L(j)]) {
              mice[L(k)] = mice[L(k)] - xm[L(j)];
              xmf = xmf + HFUS * xm[L(j)] / dt;
              stc[L(k)] = TFRZ;
              xm[L(j)] = 0.0f;
            } else {
              xm[L(j)] = xm[L(j)] - mice[L(k)];
              xmf = xmf + HFUS * mice[L(k)] / dt;
              mice[L(k)] = 0.0f;
              stc[L(k)] = TFRZ;
            }
            mliq[L(k)] = MAXF(0.f, wmass0[L(k)] - mice[L(k)]);
          }
        }
        heatr[L(j)] = xm[L(j)] * HFUS / dt;
        stc[L(j)] = TFRZ + heatr[L(j)] * fact[L(j)];
      }
    }
  }
  if (ANY_LT(stc) && ANY_POS(mliq)) {
    for (int j = 1; j <= ns; j++) {
      if (stc[L(j)] < TFRZ) {
        heatr[L(j)] = (stc[L(j)] - TFRZ) / fact[L(j)];
        xm[L(j)] = heatr[L(j)] * dt / HFUS;
        for (int k = 1; k <= ns; k++) {
          if (j != k && mliq[L(k)] > 0.f && xm[L(j)] < -0.1f) {
            if (mliq[L(k)] > fabsf(xm[L(j)])) {
              mice[L(k)] = mice[L(k)] - xm[L(j)];
              xmf = xmf + HFUS * xm[L(j)] / dt;
              stc[L(k)] = TFRZ;
              xm[L(j)] = 0.0f;
            } else {
              xm[L(j)] = xm[L(j)] + mliq[L(k)];
              xmf = xmf - HFUS * mliq[L(k)] / dt;
              mice[L(k)] = wmass0[L(k)];
              stc[L(k)] = TFRZ;
            }
            mliq[L(k)] = MAXF(0.f, wmass0[L(k)] - mice[L(k)]);
          }
        }
        heatr[L(j)] = xm[L(j)] * HFUS / dt;
        stc[L(j)] = TFRZ + heatr[L(j)] * fact[L(j)];
      }
    }
  }
  (void)xmf;
  for (int j = isnow + 1; j <= 0; j++) { snliq[L(j)] = mliq[L(j)]; snice[L(j)] = mice[L(j)]; }
  for (int j = 1; j <= ns; j++) {
    sh2o[L(j)] = mliq[L(j)] / (1000.f * dzsnso[L(j)]);
    sh2o[L(j)] = MAXF(0.0f, MINF(1.0f, sh2o[L(j)]));
    smc[L(j)] = 1.0f;
  }
}

/* NOAHMP_GLACIER gla:150-338, with ENERGY_GLACIER (393-573), WATER_GLACIER (1924-2110),
 * SNOWWATER_GLACIER (2113-2237) and ERROR_GLACIER (2898-2972) inlined */
void nmp_glacier_column(nmp_ctx* c, nmp_column* s, real* fsr_out) {
  const real ZBOT = -8.0f;                               /* gla:260 */
  int ns = c->nsoil;
  real dt = c->dt;
  real dzsnso[NL], df[NL], hcpct[NL], fact[NL], sice[NL], sice_save[NL], sh2o_save[NL];
  int imelt[NL];
  memset(dzsnso, 0, sizeof(dzsnso)); memset(imelt, 0, sizeof(imelt));
  /* ATM_GLACIER */
  real qair = s->q2;
  real eair = qair * s->sfcprs / (0.622f + 0.378f * qair);
  real rhoair = (s->sfcprs - 0.378f * eair) / (RAIR * s->sfctmp);
  real swdown = (s->cosz <= 0.f) ? 0.f : s->soldn;
  real solad[2] = {swdown * 0.7f * 0.5f, swdown * 0.7f * 0.5f};
  real solai[2] = {swdown * 0.3f * 0.5f, swdown * 0.3f * 0.5f};
  real beg_wb = s->sneqv;
  for (int iz = s->isnow + 1; iz <= ns; iz++) {
    if (iz == s->isnow + 1) dzsnso[L(iz)] = -s->zsnso[L(iz)];
    else dzsnso[L(iz)] = s->zsnso[L(iz - 1)] - s->zsnso[L(iz)];
  }
  /* ENERGY_GLACIER */
  real ur = MAXF(sqrtf(powf(s->uu, 2.f) + powf(s->vv, 2.f)), 1.f);
  real z0mg = Z0SNO;
  real zpd = s->snowh;
  real zlvl = zpd + s->zlvl;
  thermoprop_glacier(c, s->isnow, dzsnso, dt, s->snowh, s->snice, s->snliq, df, hcpct, fact);
  radiation_glacier(c, dt, s->tg, s->sneqvo, s->sneqv, s->cosz, s->qsnow, solad, solai, &s->albold,
                    &s->tauss, &s->sag, &s->fsr, &s->fsa);
  real emg = 0.98f, rhsur = 1.0f, rsurf = 1.0f;
  real lathea = HSUB;
  real gamma = CPAIR * s->sfcprs / (0.622f * lathea);
  glacier_flux(c, emg, s->isnow, df, dzsnso, z0mg, zlvl, zpd, qair, s->sfctmp, rhoair, s->sfcprs, ur,
               gamma, rsurf, s->lwdn, rhsur, s->smc, eair, s->stc, s->sag, s->snowh, lathea, s->sh2o,
               &s->cm, &s->ch, &s->tg, &s->qsfc, &s->fira, &s->fsh, &s->fgev, &s->ssoil, &s->t2mb,
               &s->q2b, &s->chb2);
  if (c->err) return;
  real fire = s->lwdn + s->fira;
  if (fire <= 0.f) { if (!c->err) c->err = NOAHMP_ERR_GLACIER_FIRE_NONPOSITIVE; return; }
  s->emissi = emg;
  s->trad = powf((fire - (1 - s->emissi) * s->lwdn) / (s->emissi * SB), 0.25f);
  nmp_tsnosoi(c, s->isnow, s->tbot, s->zsnso, s->ssoil, df, hcpct, ZBOT, dt, s->snowh, s->stc);
  if (c->O.opt_stc == 2) {
    if (s->snowh > 0.05f && s->tg > TFRZ) s->tg = TFRZ;
  }
  real qmelt;
  phasechange_glacier(c, s->isnow, dt, fact, dzsnso, s->stc, s->snice, s->snliq, &s->sneqv, &s->snowh,
                      s->smc, s->sh2o, &qmelt, imelt, &s->ponding);
  /* back in NOAHMP_GLACIER gla:295-300 */
  for (int k = 1; k <= ns; k++) sice[L(k)] = MAXF(0.0f, s->smc[L(k)] - s->sh2o[L(k)]);
  s->sneqvo = s->sneqv;
  real qvap = MAXF(s->fgev / lathea, 0.f);
  real qdew = fabsf(MINF(s->fgev / lathea, 0.f));
  s->edir = qvap - qdew;
  /* WATER_GLACIER */
  real snoflow = 0.f;
  s->runsub = 0.f; s->runsrf = 0.f;
  for (int k = 1; k <= ns; k++) { sice_save[L(k)] = sice[L(k)]; sh2o_save[L(k)] = s->sh2o[L(k)]; }
  s->fpice = 0.f;
  if (c->O.opt_snf == 1) {
    if (s->sfctmp > TFRZ + 2.5f) s->fpice = 0.f;
    else {
      if (s->sfctmp <= TFRZ + 0.5f) s->fpice = 1.0f;
      else if (s->sfctmp <= TFRZ + 2.f) s->fpice = 1.f - (-54.632f + 0.2f * s->sfctmp);
      else s->fpice = 0.6f;
    }
  }
  if (c->O.opt_snf == 2) { if (s->sfctmp >= TFRZ + 2.2f) s->fpice = 0.f; else s->fpice = 1.0f; }
  if (c->O.opt_snf == 3) { if (s->sfctmp >= TFRZ) s->fpice = 0.f; else s->fpice = 1.0f; }
  real bdfall = MINF(120.f, 67.92f + 51.25f * expf((s->sfctmp - TFRZ) / 2.59f));
  real qrain = s->prcp * (1.f - s->fpice);
  s->qsnow = s->prcp * s->fpice;
  real snowhin = s->qsnow / bdfall;
  real qsnsub = qvap, qsnfro = qdew;
  /* SNOWWATER_GLACIER */
  s->ponding1 = 0.0f; s->ponding2 = 0.0f;
  {
    int newnode = 0;                                     /* SNOWFALL_GLACIER: new layer at 0.05 m */
    if (s->isnow == 0 && s->qsnow > 0.f) {
      s->snowh = s->snowh + snowhin * dt;
      s->sneqv = s->sneqv + s->qsnow * dt;
    }
    if (s->isnow == 0 && s->qsnow > 0.f && s->snowh >= 0.05f) {
      s->isnow = -1;
      newnode = 1;
      dzsnso[L(0)] = s->snowh;
      s->snowh = 0.f;
      s->stc[L(0)] = MINF(273.16f, s->sfctmp);
      s->snice[L(0)] = s->sneqv;
      s->snliq[L(0)] = 0.f;
    }
    if (s->isnow < 0 && newnode == 0 && s->qsnow > 0.f) {
      s->snice[L(s->isnow + 1)] = s->snice[L(s->isnow + 1)] + s->qsnow * dt;
      dzsnso[L(s->isnow + 1)] = dzsnso[L(s->isnow + 1)] + snowhin * dt;
    }
  }
  if (s->isnow < 0) {
    nmp_compact(dt, s->stc, s->snice, s->snliq, imelt, s->ficeold, s->isnow, dzsnso);
    nmp_combine(1, &s->isnow, s->sh2o, s->stc, s->snice, s->snliq, dzsnso, sice, &s->snowh, &s->sneqv,
                &s->ponding1, &s->ponding2);
    nmp_divide(c->nsnow, 0.10f, &s->isnow, s->stc, s->snice, s->snliq, dzsnso);
  }
  for (int iz = -c->nsnow + 1; iz <= s->isnow; iz++) {
    s->snice[L(iz)] = 0.f; s->snliq[L(iz)] = 0.f; s->stc[L(iz)] = 0.f; dzsnso[L(iz)] = 0.f;
    s->zsnso[L(iz)] = 0.f;
  }
  nmp_snowh2o(c, 1, dt, qsnfro, qsnsub, qrain, &s->isnow, dzsnso, &s->snowh, &s->sneqv, s->snice,
              s->snliq, s->sh2o, sice, s->stc, &s->qsnbot, &s->ponding1, &s->ponding2);
  if (s->sneqv > 2000.f) {
    real bdsnow = s->snice[L(0)] / dzsnso[L(0)];
    snoflow = (s->sneqv - 2000.f);
    s->snice[L(0)] = s->snice[L(0)] - snoflow;
    dzsnso[L(0)] = dzsnso[L(0)] - snoflow / bdsnow;
    snoflow = snoflow / dt;
  }
  if (s->isnow != 0) {
    s->sneqv = 0.f;
    for (int iz = s->isnow + 1; iz <= 0; iz++) s->sneqv = s->sneqv + s->snice[L(iz)] + s->snliq[L(iz)];
  }
  for (int iz = s->isnow + 1; iz <= 0; iz++) dzsnso[L(iz)] = -dzsnso[L(iz)];
  dzsnso[L(1)] = c->zsoil[L(1)];
  for (int iz = 2; iz <= ns; iz++) dzsnso[L(iz)] = (c->zsoil[L(iz)] - c->zsoil[L(iz - 1)]);
  s->zsnso[L(s->isnow + 1)] = dzsnso[L(s->isnow + 1)];
  for (int iz = s->isnow + 2; iz <= ns; iz++) s->zsnso[L(iz)] = s->zsnso[L(iz - 1)] + dzsnso[L(iz)];
  for (int iz = s->isnow + 1; iz <= ns; iz++) dzsnso[L(iz)] = -dzsnso[L(iz)];
  /* rest of WATER_GLACIER gla:2083-2108 */
  s->runsrf = (s->ponding + s->ponding1 + s->ponding2) / dt;
  if (s->isnow == 0) s->runsrf = s->runsrf + s->qsnbot + qrain;
  else s->runsrf = s->runsrf + s->qsnbot;
  real replace = 0.0f;
  for (int k = 1; k <= ns; k++)
    replace = replace + dzsnso[L(k)] * (sice[L(k)] - sice_save[L(k)] + s->sh2o[L(k)] - sh2o_save[L(k)]);
  replace = replace * 1000.0f / dt;
  for (int k = 1; k <= ns; k++) { sice[L(k)] = MINF(1.0f, sice_save[L(k)]); s->sh2o[L(k)] = 1.0f - sice[L(k)]; }
  s->runsub = snoflow + replace;
  /* ERROR_GLACIER: SW and energy checks are one-sided (no ABS), gla:2933,2943 */
  {
    real errsw = swdown - (s->fsa + s->fsr);
    if (errsw > 0.01f) { if (!c->err) c->err = NOAHMP_ERR_GLACIER_SW_BALANCE; return; }
    real erreng = s->sag - (s->fira + s->fsh + s->fgev + s->ssoil);
    if (erreng > 0.01f) { if (!c->err) c->err = NOAHMP_ERR_GLACIER_ENERGY_BALANCE; return; }
    real end_wb = s->sneqv;
    real errwat = end_wb - beg_wb - (s->prcp - s->edir - s->runsrf - s->runsub) * dt;
    if (fabsf(errwat) > 0.1f) { if (!c->err) c->err = NOAHMP_ERR_GLACIER_WATER_BALANCE; return; }
  }
  if (s->snowh <= 1.E-6f || s->sneqv <= 1.E-3f) { s->snowh = 0.0f; s->sneqv = 0.0f; }
  if (swdown != 0.f) s->albedo = s->fsr / swdown;
  else s->albedo = -999.9f;
  *fsr_out = s->fsr;
}

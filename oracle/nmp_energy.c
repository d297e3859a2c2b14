/* TEST INFRASTRUCTURE (oracle): energy half of the column physics.  See noahmp_oracle.h.
 * Reference: phys/module_sf_noahmplsm.F90 ("lsm"), subroutine ENERGY and everything below it. */
#include <math.h>
#include "nmp_internal.h"

/* ESAT, lsm:5272-5321: degree-6 polynomials, T in deg C */
void nmp_esat(real t, real* esw, real* esi, real* desw, real* desi) {
  const real A0 = 6.107799961f, A1 = 4.436518521E-01f, A2 = 1.428945805E-02f, A3 = 2.650648471E-04f,
             A4 = 3.031240396E-06f, A5 = 2.034080948E-08f, A6 = 6.136820929E-11f;
  const real B0 = 6.109177956f, B1 = 5.034698970E-01f, B2 = 1.886013408E-02f, B3 = 4.176223716E-04f,
             B4 = 5.824720280E-06f, B5 = 4.838803174E-08f, B6 = 1.838826904E-10f;
  const real C0 = 4.438099984E-01f, C1 = 2.857002636E-02f, C2 = 7.938054040E-04f, C3 = 1.215215065E-05f,
             C4 = 1.036561403E-07f, C5 = 3.532421810e-10f, C6 = -7.090244804E-13f;
  const real D0 = 5.030305237E-01f, D1 = 3.773255020E-02f, D2 = 1.267995369E-03f, D3 = 2.477563108E-05f,
             D4 = 3.005693132E-07f, D5 = 2.158542548E-09f, D6 = 7.131097725E-12f;
  *esw = 100.f * (A0 + t * (A1 + t * (A2 + t * (A3 + t * (A4 + t * (A5 + t * A6))))));
  *esi = 100.f * (B0 + t * (B1 + t * (B2 + t * (B3 + t * (B4 + t * (B5 + t * B6))))));
  *desw = 100.f * (C0 + t * (C1 + t * (C2 + t * (C3 + t * (C4 + t * (C5 + t * C6))))));
  *desi = 100.f * (D0 + t * (D1 + t * (D2 + t * (D3 + t * (D4 + t * (D5 + t * D6))))));
}

/* statement function TDC, lsm:3247 / 3752 */
real nmp_tdc(real t) { return MINF(50.f, MAXF(-50.f, (t - TFRZ))); }

/* TDFCND, lsm:2014-2118 (Peters-Lidard soil thermal conductivity) */
static real tdfcnd(const nmp_ctx* c, real smc, real sh2o) {
  const nmp_parm* P = &c->P;
  real satratio = smc / P->smcmax;
  real thkw = 0.57f, thko = 2.0f, thkqtz = 7.7f;
  real thks = powf(thkqtz, P->quartz) * powf(thko, 1.f - P->quartz);
  real xunfroz = sh2o / smc;                 /* no zero guard in the reference (lsm:2079) */
  real xu = xunfroz * P->smcmax;
  real thksat = powf(thks, 1.f - P->smcmax) * powf(TKICE, P->smcmax - xu) * powf(thkw, xu);
  real gammd = (1.f - P->smcmax) * 2700.f;
  real thkdry = (0.135f * gammd + 64.7f) / (2700.f - 0.947f * gammd);
  real ake;
  if ((sh2o + 0.0005f) < smc) {
    ake = satratio;
  } else {
    if (satratio > 0.1f) ake = log10f(satratio) + 1.0f;
    else ake = 0.0f;
  }
  return ake * (thksat - thkdry) + thkdry;
}

/* THERMOPROP lsm:1845-1954 + CSNOW lsm:1957-2011 */
static void thermoprop(const nmp_ctx* c, int isnow, int ist, const real* dzsnso, real dt, real snowh,
                       const real* snice, const real* snliq, const real* smc, const real* sh2o,
                       const real* stc, real* df, real* hcpct, real* snicev, real* snliqv,
                       real* epore, real* fact) {
  const nmp_parm* P = &c->P;
  int ns = c->nsoil;
  for (int iz = isnow + 1; iz <= 0; iz++) {          /* CSNOW */
    snicev[L(iz)] = MINF(1.f, snice[L(iz)] / (dzsnso[L(iz)] * DENICE));
    epore[L(iz)] = 1.f - snicev[L(iz)];
    snliqv[L(iz)] = MINF(epore[L(iz)], snliq[L(iz)] / (dzsnso[L(iz)] * DENH2O));
  }
  for (int iz = isnow + 1; iz <= 0; iz++) {
    real bdsnoi = (snice[L(iz)] + snliq[L(iz)]) / dzsnso[L(iz)];
    hcpct[L(iz)] = CICE * snicev[L(iz)] + CWAT * snliqv[L(iz)];
    df[L(iz)] = 3.2217E-6f * powf(bdsnoi, 2.f);     /* lsm:2004, real exponent */
  }
  for (int iz = 1; iz <= ns; iz++) {
    real sice = smc[L(iz)] - sh2o[L(iz)];
    hcpct[L(iz)] = sh2o[L(iz)] * CWAT + (1.0f - P->smcmax) * P->csoil +
                   (P->smcmax - smc[L(iz)]) * CPAIR + sice * CICE;
    df[L(iz)] = tdfcnd(c, smc[L(iz)], sh2o[L(iz)]);
  }
  if (c->vegtyp == c->isurban)
    for (int iz = 1; iz <= ns; iz++) df[L(iz)] = 3.24f;
  if (ist == 2) {                                     /* lake, dead in HRLDAS (IST=1, drv:526) */
    for (int iz = 1; iz <= ns; iz++) {
      if (stc[L(iz)] > TFRZ) { hcpct[L(iz)] = CWAT; df[L(iz)] = TKWAT; }
      else { hcpct[L(iz)] = CICE; df[L(iz)] = TKICE; }
    }
  }
  for (int iz = isnow + 1; iz <= ns; iz++) fact[L(iz)] = dt / (hcpct[L(iz)] * dzsnso[L(iz)]);
  if (isnow == 0)
    df[L(1)] = (df[L(1)] * dzsnso[L(1)] + 0.35f * snowh) / (snowh + dzsnso[L(1)]);
  else
    df[L(1)] = (df[L(1)] * dzsnso[L(1)] + df[L(0)] * dzsnso[L(0)]) / (dzsnso[L(0)] + dzsnso[L(1)]);
}

/* SNOW_AGE lsm:2547-2596 */
void nmp_snow_age(real dt, real tg, real sneqvo, real sneqv, real* tauss, real* fage) {
  if (sneqv <= 0.0f) *tauss = 0.f;
  else if (sneqv > 800.f) *tauss = 0.f;
  else {
    real dela0 = 1.E-6f * dt;
    real arg = 5.E3f * (1.f / TFRZ - 1.f / tg);
    real age1 = expf(arg);
    real age2 = expf(MINF(0.f, 10.f * arg));
    real age3 = 0.3f;
    real tage = age1 + age2 + age3;
    real dela = dela0 * tage;
    real dels = MAXF(0.0f, sneqv - sneqvo) / SWEMX;
    real sge = (*tauss + dela) * (1.0f - dels);
    *tauss = MAXF(0.f, sge);
  }
  *fage = *tauss / (*tauss + 1.f);
}

/* TWOSTREAM lsm:2768-3016.  ib: 0/1 band; ic: 0 direct, 1 diffuse */
static void twostream(const nmp_ctx* c, int ib, int ic, real cosz, real vai, real fwet, real t,
                      const real* albgrd, const real* albgri, const real* rho, const real* tau,
                      real fveg, real* fab, real* fre, real* ftd, real* fti, real* gdir, real* frev,
                      real* freg, real* bgap, real* wgap) {
  const noahmp_tables* T = c->T;
  int v = c->vegtyp - 1;
  const real PAI = 3.14159265f;
  real gap, kopen;
  if (vai == 0.0f) {
    gap = 1.0f; kopen = 1.0f;
  } else {
    gap = 0.f; kopen = 0.f;
    if (c->O.opt_rad == 1) {
      real rc = T->rc[v];
      real denfveg = -logf(MAXF(1.0f - fveg, 0.01f)) / (PAI * powi(rc, 2));
      real hd = T->hvt[v] - T->hvb[v];
      real bb = 0.5f * hd;
      real thetap = atanf(bb / rc * tanf(acosf(MAXF(0.01f, cosz))));
      *bgap = expf(-denfveg * PAI * powi(rc, 2) / cosf(thetap));
      real fa = vai / (1.33f * PAI * powf(rc, 3.0f) * (bb / rc) * denfveg);
      real newvai = hd * fa;
      *wgap = (1.0f - *bgap) * expf(-0.5f * newvai / cosz);
      gap = MINF(1.0f - fveg, *bgap + *wgap);
      kopen = 0.05f;
    }
    if (c->O.opt_rad == 2) { gap = 0.0f; kopen = 0.0f; }
    if (c->O.opt_rad == 3) { gap = 1.0f - fveg; kopen = 1.0f - fveg; }
  }
  real coszi = MAXF(0.001f, cosz);
  real chil = MINF(MAXF(T->xl[v], -0.4f), 0.6f);
  if (fabsf(chil) <= 0.01f) chil = 0.01f;
  real phi1 = 0.5f - 0.633f * chil - 0.330f * chil * chil;
  real phi2 = 0.877f * (1.f - 2.f * phi1);
  *gdir = phi1 + phi2 * coszi;
  real ext = *gdir / coszi;
  real avmu = (1.f - phi1 / phi2 * logf((phi1 + phi2) / phi1)) / phi2;
  real omegal = rho[ib] + tau[ib];
  real tmp0 = *gdir + phi2 * coszi;
  real tmp1 = phi1 * coszi;
  real asu = 0.5f * omegal * *gdir / tmp0 * (1.f - tmp1 / tmp0 * logf((tmp1 + tmp0) / tmp1));
  real betadl = (1.f + avmu * ext) / (omegal * avmu * ext) * asu;
  real betail = 0.5f * (rho[ib] + tau[ib] + (rho[ib] - tau[ib]) * powi((1.f + chil) / 2.f, 2)) / omegal;
  real tmp2;
  if (t > TFRZ) {
    tmp0 = omegal; tmp1 = betadl; tmp2 = betail;
  } else {
    tmp0 = (1.f - fwet) * omegal + fwet * T->omegas[ib];
    tmp1 = ((1.f - fwet) * omegal * betadl + fwet * T->omegas[ib] * T->betads) / tmp0;
    tmp2 = ((1.f - fwet) * omegal * betail + fwet * T->omegas[ib] * T->betais) / tmp0;
  }
  real omega = tmp0, betad = tmp1, betai = tmp2;
  real b = 1.f - omega + omega * betai;
  real cc = omega * betai;
  tmp0 = avmu * ext;
  real d = tmp0 * omega * betad;
  real f = tmp0 * omega * (1.f - betad);
  tmp1 = b * b - cc * cc;
  real h = sqrtf(tmp1) / avmu;
  real sigma = tmp0 * tmp0 - tmp1;
  if (fabsf(sigma) < 1.e-6f) sigma = copysignf(1.e-6f, sigma);
  real p1 = b + avmu * h, p2 = b - avmu * h, p3 = b + tmp0, p4 = b - tmp0;
  real s1 = expf(-h * vai), s2 = expf(-ext * vai);
  real u1, u2, u3;
  if (ic == 0) {
    u1 = b - cc / albgrd[ib]; u2 = b - cc * albgrd[ib]; u3 = f + cc * albgrd[ib];
  } else {
    u1 = b - cc / albgri[ib]; u2 = b - cc * albgri[ib]; u3 = f + cc * albgri[ib];
  }
  tmp2 = u1 - avmu * h;
  real tmp3 = u1 + avmu * h;
  real d1 = p1 * tmp2 / s1 - p2 * tmp3 * s1;
  real tmp4 = u2 + avmu * h;
  real tmp5 = u2 - avmu * h;
  real d2 = tmp4 / s1 - tmp5 * s1;
  real h1 = -d * p4 - cc * f;
  real tmp6 = d - h1 * p3 / sigma;
  real tmp7 = (d - cc - h1 / sigma * (u1 + tmp0)) * s2;
  real h2 = (tmp6 * tmp2 / s1 - p2 * tmp7) / d1;
  real h3 = -(tmp6 * tmp3 * s1 - p1 * tmp7) / d1;
  real h4 = -f * p3 - cc * d;
  real tmp8 = h4 / sigma;
  real tmp9 = (u3 - tmp8 * (u2 - tmp0)) * s2;
  real h5 = -(tmp8 * tmp4 / s1 + tmp9) / d2;
  real h6 = (tmp8 * tmp5 * s1 + tmp9) / d2;
  real h7 = (cc * tmp2) / (d1 * s1);
  real h8 = (-cc * tmp3 * s1) / d1;
  real h9 = tmp4 / (d2 * s1);
  real h10 = (-tmp5 * s1) / d2;
  real ftds, ftis, fres, freveg, frebar;
  if (ic == 0) {
    ftds = s2 * (1.0f - gap) + gap;
    ftis = (h4 * s2 / sigma + h5 * s1 + h6 / s1) * (1.0f - gap);
  } else {
    ftds = 0.f;
    ftis = (h9 * s1 + h10 / s1) * (1.0f - kopen) + kopen;
  }
  ftd[ib] = ftds; fti[ib] = ftis;
  if (ic == 0) {
    fres = (h1 / sigma + h2 + h3) * (1.0f - gap) + albgrd[ib] * gap;
    freveg = (h1 / sigma + h2 + h3) * (1.0f - gap);
    frebar = albgrd[ib] * gap;
  } else {
    fres = (h7 + h8) * (1.0f - kopen) + albgri[ib] * kopen;
    freveg = (h7 + h8) * (1.0f - kopen) + albgri[ib] * kopen;
    frebar = 0.f;
  }
  fre[ib] = fres; frev[ib] = freveg; freg[ib] = frebar;
  fab[ib] = 1.f - fre[ib] - (1.f - albgrd[ib]) * ftd[ib] - (1.f - albgri[ib]) * fti[ib];
}

/* RADIATION lsm:2120-2240 = ALBEDO (2243-2423) + SURRAD (2426-2544) and their callees */
static void radiation(const nmp_ctx* c, int ist, int isc, int ice, real sneqvo, real sneqv, real dt,
                      real cosz, real snowh, real tg, real tv, real fsno, real qsnow, real fwet,
                      real elai, real esai, const real* smc, const real* solad, const real* solai,
                      real fveg, real* albold, real* tauss, real* fsun, real* laisun, real* laisha,
                      real* parsun, real* parsha, real* sav, real* sag, real* fsr, real* fsa,
                      real* fsrv, real* fsrg, real* bgap, real* wgap) {
  const noahmp_tables* T = c->T;
  int v = c->vegtyp - 1;
  const real MPE = 1.E-6f;
  real albd[2] = {0, 0}, albi[2] = {0, 0}, albgrd[2] = {0, 0}, albgri[2] = {0, 0}, fabd[2] = {0, 0},
       fabi[2] = {0, 0}, ftdd[2] = {0, 0}, ftid[2] = {0, 0}, ftii[2] = {0, 0}, ftdi[2];
  /* FREV*/ /* not initialised by the reference at night (lsm:2356 GOTO 100); they multiply
     SOLAD=SOLAI=0 there, so zero is the value-preserving choice */
  real frevd[2] = {0, 0}, frevi[2] = {0, 0}, fregd[2] = {0, 0}, fregi[2] = {0, 0};
  real rho[2], tau[2], albsnd[2], albsni[2];
  real fage, vai, wl, ws, gdir = 0.f, ext;
  *bgap = 0.f; *wgap = 0.f; *fsun = 0.f;
  (void)ice;
  if (!(cosz <= 0.f)) {                          /* lsm:2356 IF(COSZ <= 0) GOTO 100: a NaN COSZ is not skipped */
    for (int ib = 0; ib < 2; ib++) {
      vai = elai + esai;
      wl = elai / MAXF(vai, MPE);
      ws = esai / MAXF(vai, MPE);
      rho[ib] = MAXF(T->rhol[ib][v] * wl + T->rhos[ib][v] * ws, MPE);
      tau[ib] = MAXF(T->taul[ib][v] * wl + T->taus[ib][v] * ws, MPE);
    }
    nmp_snow_age(dt, tg, sneqvo, sneqv, tauss, &fage);
    if (c->O.opt_alb == 1) {                                   /* SNOWALB_BATS lsm:2599-2649 */
      const real C1 = 0.2f, C2 = 0.5f;
      real sl = 2.0f, sl1 = 1.f / sl, sl2 = 2.f * sl;
      real cf1 = ((1.f + sl1) / (1.f + sl2 * cosz) - sl1);
      real fzen = MAXF(cf1, 0.f);
      albsni[0] = 0.95f * (1.f - C1 * fage);
      albsni[1] = 0.65f * (1.f - C2 * fage);
      albsnd[0] = albsni[0] + 0.4f * fzen * (1.f - albsni[0]);
      albsnd[1] = albsni[1] + 0.4f * fzen * (1.f - albsni[1]);
    }
    if (c->O.opt_alb == 2) {                                   /* SNOWALB_CLASS lsm:2652-2700 */
      real alb = 0.55f + (*albold - 0.55f) * expf(-0.01f * dt / 3600.f);
      if (qsnow > 0.f) alb = alb + MINF(qsnow * dt, SWEMX) * (0.84f - alb) / (SWEMX);
      albsni[0] = albsni[1] = albsnd[0] = albsnd[1] = alb;
      *albold = alb;
    }
    for (int ib = 0; ib < 2; ib++) {                           /* GROUNDALB lsm:2703-2765 */
      real inc = MAXF(0.11f - 0.40f * smc[L(1)], 0.f);
      real albsod, albsoi;
      if (ist == 1) {
        albsod = MINF(T->albsat[ib][isc - 1] + inc, T->albdry[ib][isc - 1]);
        albsoi = albsod;
      } else if (tg > TFRZ) {
        albsod = 0.06f / (powf(MAXF(0.01f, cosz), 1.7f) + 0.15f);
        albsoi = 0.06f;
      } else {
        albsod = T->alblak[ib];
        albsoi = albsod;
      }
      if (ist == 1 && isc == 9) { albsod += 0.10f; albsoi += 0.10f; }
      albgrd[ib] = albsod * (1.f - fsno) + albsnd[ib] * fsno;
      albgri[ib] = albsoi * (1.f - fsno) + albsni[ib] * fsno;
    }
    for (int ib = 0; ib < 2; ib++) {
      twostream(c, ib, 0, cosz, vai, fwet, tv, albgrd, albgri, rho, tau, fveg, fabd, albd, ftdd,
                ftid, &gdir, frevd, fregd, bgap, wgap);
      twostream(c, ib, 1, cosz, vai, fwet, tv, albgrd, albgri, rho, tau, fveg, fabi, albi, ftdi,
                ftii, &gdir, frevi, fregi, bgap, wgap);
    }
    ext = gdir / cosz * sqrtf(1.f - rho[0] - tau[0]);
    *fsun = (1.f - expf(-ext * vai)) / MAXF(ext * vai, MPE);
    ext = *fsun;
    if (ext < 0.01f) wl = 0.f; else wl = ext;
    *fsun = wl;
  }
  /* RADIATION body after ALBEDO, lsm:2221-2238 */
  real fsha = 1.f - *fsun;
  *laisun = elai * *fsun;
  *laisha = elai * fsha;
  vai = elai + esai;
  /* SURRAD lsm:2426-2544 */
  real cad[2], cai[2];
  *sag = 0.f; *sav = 0.f; *fsa = 0.f;
  for (int ib = 0; ib < 2; ib++) {
    cad[ib] = solad[ib] * fabd[ib];
    cai[ib] = solai[ib] * fabi[ib];
    *sav = *sav + cad[ib] + cai[ib];
    *fsa = *fsa + cad[ib] + cai[ib];
    real trd = solad[ib] * ftdd[ib];
    real tri = solad[ib] * ftid[ib] + solai[ib] * ftii[ib];
    real abs_ = trd * (1.f - albgrd[ib]) + tri * (1.f - albgri[ib]);
    *sag = *sag + abs_;
    *fsa = *fsa + abs_;
  }
  real laifra = elai / MAXF(vai, MPE);
  if (*fsun > 0.f) {
    *parsun = (cad[0] + *fsun * cai[0]) * laifra / MAXF(*laisun, MPE);
    *parsha = (fsha * cai[0]) * laifra / MAXF(*laisha, MPE);
  } else {
    *parsun = 0.f;
    *parsha = (cad[0] + cai[0]) * laifra / MAXF(*laisha, MPE);
  }
  real rvis = albd[0] * solad[0] + albi[0] * solai[0];
  real rnir = albd[1] * solad[1] + albi[1] * solai[1];
  *fsr = rvis + rnir;
  *fsrv = frevd[0] * solad[0] + frevi[0] * solai[0] + frevd[1] * solad[1] + frevi[1] * solai[1];
  *fsrg = fregd[0] * solad[0] + fregi[0] * solai[0] + fregd[1] * solad[1] + fregi[1] * solai[1];
}

/* SFCDIF1 lsm:4061-4220 (Monin-Obukhov) */

void nmp_sfcdif1(nmp_ctx* c, int iter, real sfctmp, real rhoair, real h, real qair, real zlvl,
                    real zpd, real z0m, real z0h, real ur, real mpe, mo_state* s, real* cm, real* ch,
                    real* ch2) {
  real mozold = s->moz;
  real mol, moz2, fmnew, fhnew, fm2new, fh2new;
  if (zlvl <= zpd) { if (!c->err) c->err = NOAHMP_ERR_STABILITY_STOP; return; }
  real tmpcm = logf((zlvl - zpd) / z0m);
  real tmpch = logf((zlvl - zpd) / z0h);
  real tmpcm2 = logf((2.0f + z0m) / z0m);
  real tmpch2 = logf((2.0f + z0h) / z0h);
  if (iter == 1) {
    s->fv = 0.0f; s->moz = 0.0f; mol = 0.0f; moz2 = 0.0f;
  } else {
    real tvir = (1.f + 0.61f * qair) * sfctmp;
    real tmp1 = VKC * (GRAV / tvir) * h / (rhoair * CPAIR);
    if (fabsf(tmp1) <= mpe) tmp1 = mpe;
    mol = -1.f * powi(s->fv, 3) / tmp1;
    s->moz = MINF((zlvl - zpd) / mol, 1.f);
    moz2 = MINF((2.0f + z0h) / mol, 1.f);
  }
  if (mozold * s->moz < 0.f) s->mozsgn = s->mozsgn + 1;
  if (s->mozsgn >= 2) {
    s->moz = 0.f; s->fm = 0.f; s->fh = 0.f; moz2 = 0.f; s->fm2 = 0.f; s->fh2 = 0.f;
  }
  if (s->moz < 0.f) {
    real tmp1 = powf(1.f - 16.f * s->moz, 0.25f);
    real tmp2 = logf((1.f + tmp1 * tmp1) / 2.f);
    real tmp3 = logf((1.f + tmp1) / 2.f);
    fmnew = 2.f * tmp3 + tmp2 - 2.f * atanf(tmp1) + 1.5707963f;
    fhnew = 2 * tmp2;
    real tmp12 = powf(1.f - 16.f * moz2, 0.25f);
    real tmp22 = logf((1.f + tmp12 * tmp12) / 2.f);
    real tmp32 = logf((1.f + tmp12) / 2.f);
    fm2new = 2.f * tmp32 + tmp22 - 2.f * atanf(tmp12) + 1.5707963f;
    fh2new = 2 * tmp22;
  } else {
    fmnew = -5.f * s->moz; fhnew = fmnew;
    fm2new = -5.f * moz2; fh2new = fm2new;
  }
  if (iter == 1) {
    s->fm = fmnew; s->fh = fhnew; s->fm2 = fm2new; s->fh2 = fh2new;
  } else {
    s->fm = 0.5f * (s->fm + fmnew);
    s->fh = 0.5f * (s->fh + fhnew);
    s->fm2 = 0.5f * (s->fm2 + fm2new);
    s->fh2 = 0.5f * (s->fh2 + fh2new);
  }
  s->fh = MINF(s->fh, 0.9f * tmpch);
  s->fm = MINF(s->fm, 0.9f * tmpcm);
  s->fh2 = MINF(s->fh2, 0.9f * tmpch2);
  s->fm2 = MINF(s->fm2, 0.9f * tmpcm2);
  real cmfm = tmpcm - s->fm, chfh = tmpch - s->fh, cm2fm2 = tmpcm2 - s->fm2, ch2fh2 = tmpch2 - s->fh2;
  if (fabsf(cmfm) <= mpe) cmfm = mpe;
  if (fabsf(chfh) <= mpe) chfh = mpe;
  if (fabsf(cm2fm2) <= mpe) cm2fm2 = mpe;
  if (fabsf(ch2fh2) <= mpe) ch2fh2 = mpe;
  *cm = VKC * VKC / (cmfm * cmfm);
  *ch = VKC * VKC / (cmfm * chfh);
  s->fv = ur * sqrtf(*cm);
  *ch2 = VKC * s->fv / ch2fh2;
}

/* SFCDIF2 lsm:4224-4422 (Chen et al. 1997) */
static real pslmu(real zz) { return -0.96f * logf(1.0f - 4.5f * zz); }
static real pspmu(real xx) {
  const real PIHF = 3.14159265f / 2.f;
  return -2.f * logf((xx + 1.f) * 0.5f) - logf((xx * xx + 1.f) * 0.5f) + 2.f * atanf(xx) - PIHF;
}
static real psphu(real xx) { return -2.f * logf((xx * xx + 1.f) * 0.5f); }

static void sfcdif2(int iter, real z0, real thz0, real thlm, real sfcspd, real czil, real zlm,
                    real* akms, real* akhs, real* rlmo, real* wstar2, real* ustar) {
  const real WWST = 1.2f, WWST2 = WWST * WWST, VKRM = 0.40f, EXCM = 0.001f, BETA = 1.0f / 270.0f,
             BTG = BETA * GRAV, ELFC = VKRM * BTG, WOLD = 0.15f, WNEW = 1.0f - WOLD, EPSU2 = 1.E-4f,
             EPSUST = 0.07f, ZTMIN = -5.0f, ZTMAX = 1.0f, HPBL = 1000.0f, SQVISC = 258.2f,
             RIC = 0.183f, RRIC = 1.0f / RIC, FHNEU = 0.8f, RFC = 0.191f,
             RFAC = RIC / (FHNEU * RFC * RFC);
  int ilech = 0;
  real zilfc = -czil * VKRM * SQVISC;
  real zu = z0;
  real rdz = 1.f / zlm;
  real cxch = EXCM * rdz;
  real dthv = thlm - thz0;
  real du2 = MAXF(sfcspd * sfcspd, EPSU2);
  real btgh = BTG * HPBL;
  if (iter == 1) {
    if (btgh * *akhs * dthv != 0.0f)
      *wstar2 = WWST2 * powf(fabsf(btgh * *akhs * dthv), 2.f / 3.f);
    else
      *wstar2 = 0.0f;
    *ustar = MAXF(sqrtf(*akms * sqrtf(du2 + *wstar2)), EPSUST);
    *rlmo = ELFC * *akhs * dthv / powi(*ustar, 3);
  }
  real zt = MAXF(1.E-6f, expf(zilfc * sqrtf(*ustar * z0)) * z0);
  real zslu = zlm + zu;
  real zslt = zlm + zt;
  real rlogu = logf(zslu / zu);
  real rlogt = logf(zslt / zt);
  real zetalt = MAXF(zslt * *rlmo, ZTMIN);
  *rlmo = zetalt / zslt;
  real zetalu = zslu * *rlmo;
  real zetau = zu * *rlmo;
  real zetat = zt * *rlmo;
  real psmz, simm, pshz, simh;
  if (ilech == 0) {
    if (*rlmo < 0.f) {
      real xlu4 = 1.f - 16.f * zetalu, xlt4 = 1.f - 16.f * zetalt, xu4 = 1.f - 16.f * zetau,
           xt4 = 1.f - 16.f * zetat;
      real xlu = sqrtf(sqrtf(xlu4)), xlt = sqrtf(sqrtf(xlt4)), xu = sqrtf(sqrtf(xu4)),
           xt = sqrtf(sqrtf(xt4));
      psmz = pspmu(xu);
      simm = pspmu(xlu) - psmz + rlogu;
      pshz = psphu(xt);
      simh = psphu(xlt) - pshz + rlogt;
    } else {
      zetalu = MINF(zetalu, ZTMAX);
      zetalt = MINF(zetalt, ZTMAX);
      psmz = 5.f * zetau;
      simm = 5.f * zetalu - psmz + rlogu;
      pshz = 5.f * zetat;
      simh = 5.f * zetalt - pshz + rlogt;
    }
  } else {                                   /* ILECH is always 0 (lsm:4306); kept for fidelity */
    if (*rlmo < 0.f) {
      psmz = pslmu(zetau);
      simm = pslmu(zetalu) - psmz + rlogu;
      pshz = pslmu(zetat);
      simh = pslmu(zetalt) - pshz + rlogt;
    } else {
      zetalu = MINF(zetalu, ZTMAX);
      zetalt = MINF(zetalt, ZTMAX);
      psmz = zetau * RRIC - 2.076f * (1.f - 1.f / (zetau + 1.f));
      simm = zetalu * RRIC - 2.076f * (1.f - 1.f / (zetalu + 1.f)) - psmz + rlogu;
      pshz = zetat * RFAC - 2.076f * (1.f - 1.f / (zetat + 1.f));
      simh = zetalt * RFAC - 2.076f * (1.f - 1.f / (zetalt + 1.f)) - pshz + rlogt;
    }
  }
  *ustar = MAXF(sqrtf(*akms * sqrtf(du2 + *wstar2)), EPSUST);
  zt = MAXF(1.E-6f, expf(zilfc * sqrtf(*ustar * z0)) * z0);
  zslt = zlm + zt;
  rlogt = logf(zslt / zt);
  real ustark = *ustar * VKRM;
  *akms = MAXF(ustark / simm, cxch);
  *akhs = MAXF(ustark / simh, cxch);
  if (btgh * *akhs * dthv != 0.0f)
    *wstar2 = WWST2 * powf(fabsf(btgh * *akhs * dthv), 2.f / 3.f);
  else
    *wstar2 = 0.0f;
  real rlmn = ELFC * *akhs * dthv / powi(*ustar, 3);
  real rlma = *rlmo * WOLD + rlmn * WNEW;
  *rlmo = rlma;
  (void)rlogt;
}

/* RAGRB lsm:3960-4057 */
static void ragrb(const nmp_ctx* c, int iter, real vai, real rhoair, real hg, real tah, real zpd,
                  real z0mg, real z0hg, real hcan, real uc, real z0h, real fv, real cwp, real mpe,
                  real* fhg, real* rahg, real* rawg, real* rb) {
  real mozg = 0.f, molg, fhgnew;
  if (iter > 1) {
    real tmp1 = VKC * (GRAV / tah) * hg / (rhoair * CPAIR);
    if (fabsf(tmp1) <= mpe) tmp1 = mpe;
    molg = -1.f * powi(fv, 3) / tmp1;
    mozg = MINF((zpd - z0mg) / molg, 1.f);
  }
  if (mozg < 0.f) fhgnew = powf(1.f - 15.f * mozg, -0.25f);
  else fhgnew = 1.f + 4.7f * mozg;
  if (iter == 1) *fhg = fhgnew;
  else *fhg = 0.5f * (*fhg + fhgnew);
  real cwpc = powf(cwp * vai * hcan * *fhg, 0.5f);
  real tmp1 = expf(-cwpc * z0hg / hcan);
  real tmp2 = expf(-cwpc * (z0h + zpd) / hcan);
  real tmprah2 = hcan * expf(cwpc) / cwpc * (tmp1 - tmp2);
  real kh = MAXF(VKC * fv * (hcan - zpd), mpe);
  *rahg = tmprah2 / kh;
  *rawg = *rahg;
  real tmprb = cwpc * 50.f / (1.f - expf(-cwpc / 2.f));
  *rb = tmprb * sqrtf(c->T->dleaf[c->vegtyp - 1] / uc);
}

/* STOMATA + CI2CI lsm:5323-5464 (bisection variant of this fork) */
typedef struct {
  real cp, j, vcmx, awc, rlb, co2, sfcprs, ea, ei, igs, mpe, c3, mpv, bpv;
} ci_env;

static void ci2ci(const ci_env* e, real ci, real* fci, real* rs, real* psn) {
  real wj = MAXF(ci - e->cp, 0.0f) * e->j / (ci + 2.0f * e->cp) * e->c3 + e->j * (1.f - e->c3);
  real wc = MAXF(ci - e->cp, 0.0f) * e->vcmx / (ci + e->awc) * e->c3 + e->vcmx * (1.f - e->c3);
  real we = 0.5f * e->vcmx * e->c3 + 4000.0f * e->vcmx * ci / e->sfcprs * (1.f - e->c3);
  *psn = MINF(MINF(wj, wc), we) * e->igs;
  real cs = MAXF(e->co2 - 1.37f * e->rlb * e->sfcprs * *psn, e->mpe);
  real a = e->mpv * *psn * e->sfcprs * e->ea / (cs * e->ei) + e->bpv;
  real b = (e->mpv * *psn * e->sfcprs / cs + e->bpv) * e->rlb - 1.f;
  real cq = -e->rlb;
  real q;
  if (b >= 0.0f) q = -0.5f * (b + sqrtf(b * b - 4.0f * a * cq));
  else q = -0.5f * (b - sqrtf(b * b - 4.0f * a * cq));
  real r1 = q / a, r2 = cq / q;
  *rs = MAXF(r1, r2);
  *fci = MAXF(cs - *psn * e->sfcprs * 1.65f * *rs, 0.0f);
}

static void stomata(const nmp_ctx* c, real mpe, real apar, real foln, real tv, real ei, real ea,
                    real sfctmp, real sfcprs, real o2, real co2, real igs, real btran, real rb,
                    real* rs, real* psn) {
  const noahmp_tables* T = c->T;
  int v = c->vegtyp - 1;
  const real CIERR = 5e-2f;
  real cf = sfcprs / (8.314f * sfctmp) * 1.0e06f;
  *rs = 1.0f / T->bp[v] * cf;
  *psn = 0.0f;
  if (apar <= 0.0f) return;
  ci_env e;
  real fnf = MINF(foln / MAXF(mpe, T->folnmx[v]), 1.0f);
  real tc = tv - TFRZ;
  real ppf = 4.6f * apar;
  e.j = ppf * T->qe25[v];
  real kc = T->kc25[v] * powf(T->akc[v], (tc - 25.0f) / 10.0f);
  real ko = T->ko25[v] * powf(T->ako[v], (tc - 25.0f) / 10.0f);
  e.awc = kc * (1.0f + o2 / ko);
  e.cp = 0.5f * kc / ko * o2 * 0.21f;
  e.vcmx = T->vcmx25[v] / (1.0f + expf((-2.2E05f + 710.0f * (tc + TFRZ)) / (8.314f * (tc + TFRZ)))) *
           fnf * btran * powf(T->avcmx[v], (tc - 25.0f) / 10.0f);
  e.rlb = rb / cf;
  e.co2 = co2; e.sfcprs = sfcprs; e.ea = ea; e.ei = ei; e.igs = igs; e.mpe = mpe;
  e.c3 = T->c3psn[v]; e.mpv = T->mp[v]; e.bpv = T->bp[v];
  real cihi = 1.5f * co2, cilow = 0.0f, ci, fci;
  for (int iter = 1; iter <= 20; iter++) {
    ci = 0.5f * (cihi + cilow);
    ci2ci(&e, ci, &fci, rs, psn);
    if (((cihi - cilow) <= CIERR) || fabsf(fci - ci) <= mpe) break;
    else if (fci > ci) cilow = ci;
    else cihi = ci;
  }
  *rs = *rs * cf;
}

/* CANRES lsm:5598-5677 + CALHUM lsm:5679-5705 (Jarvis) */
static void canres(const nmp_ctx* c, real par, real sfctmp, real rcsoil, real eah, real sfcprs,
                   real* rc, real* psn) {
  const nmp_parm* P = &c->P;
  real q2 = 0.622f * eah / (sfcprs - 0.378f * eah);
  q2 = q2 / (1.0f + q2);
  const real A3 = 273.15f, ELWV = 2.501E6f, E0 = 0.611f, RV = 461.0f, EPSILON = 0.622f;
  real es = E0 * expf(ELWV / RV * (1.f / A3 - 1.f / sfctmp));
  real sfcprsx = sfcprs * 1.E-3f;
  real q2sat = EPSILON * es / (sfcprsx - es);
  q2sat = q2sat * 1.E3f;
  q2sat = q2sat / 1.E3f;
  real ff = 2.0f * par / P->rgl;
  real rcs = (ff + P->rsmin / P->rsmax) / (1.0f + ff);
  rcs = MAXF(rcs, 0.0001f);
  real rct = 1.0f - 0.0016f * powf(P->topt - sfctmp, 2.0f);
  rct = MAXF(rct, 0.0001f);
  real rcq = 1.0f / (1.0f + P->hs * MAXF(0.f, q2sat - q2));
  rcq = MAXF(rcq, 0.01f);
  *rc = P->rsmin / (rcs * rct * rcq * rcsoil);
  *psn = -999.99f;
}

/* VEGE_FLUX lsm:3018-3589 */
static void vege_flux(nmp_ctx* c, int isnow, real dt, real sav, real sag, real lwdn, real ur, real uu,
                      real vv, real sfctmp, real thair, real qair, real eair, real rhoair, real snowh,
                      real vai, real gammav, real gammag, real fwet, real laisun, real laisha,
                      real cwp, const real* dzsnso, real htop, real zlvl, real zpd, real z0m,
                      real fveg, real z0mg, real emv, real emg, real canliq, real canice,
                      const real* stc, const real* df, real* rssun, real* rssha, real rsurf,
                      real latheav, real latheag, real parsun, real parsha, real igs, real foln,
                      real co2air, real o2air, real btran, real sfcprs, real rhsur, real* eah,
                      real* tah, real* tv, real* tg, real* cm, real* ch, real* tauxv, real* tauyv,
                      real* irg, real* irc, real* shg, real* shc, real* evg, real* evc, real* tr,
                      real* gh, real* t2mv, real* psnsun, real* psnsha, real* qsfc, real psfc,
                      real* q2v, real* cah2, real* chleaf, real* chuc) {
  const real MPE = 1E-6f;
  int liter = 0;
  mo_state mo = {0.f, 0.f, 0.f, 0.f, 0.f, 0.1f, 0};
  real dtv = 0.f, hg = 0.f, h = 0.f;
  real t, esatw, esati, dsatw, dsati, estg, destg, estv = 0.f, destv = 0.f;
  real rahc = 0.f, rawc, rahg = 0.f, rawg = 0.f, rb = 0.f, fhg = 0.f;
  real cah = 0.f, cvh = 0.f, cgh, cond, ata, bta, csh, caw, cew, ctw, cgw, aea, bea, cev, ctr;
  real a, b, z0h = z0m, z0hg = z0mg, ch2, wstar = 0.f;
  real vaie = MINF(6.f, vai / fveg);
  real laisune = MINF(6.f, laisun / fveg);
  real laishae = MINF(6.f, laisha / fveg);
  t = nmp_tdc(*tg);
  nmp_esat(t, &esatw, &esati, &dsatw, &dsati);
  estg = (t > 0.f) ? esatw : esati;
  *qsfc = 0.622f * eair / (psfc - 0.378f * eair);
  real hcan = htop;
  real uc = ur * logf(hcan / z0m) / logf(zlvl / z0m);
  if ((hcan - zpd) <= 0.f) { if (!c->err) c->err = NOAHMP_ERR_HCAN_LE_ZPD; return; }
  real air = -emv * (1.f + (1.f - emv) * (1.f - emg)) * lwdn - emv * emg * SB * powi(*tg, 4);
  real cir = (2.f - emv * (1.f - emg)) * emv * SB;
  for (int iter = 1; iter <= 20; iter++) {           /* loop1, NITERC=20 (lsm:3234) */
    z0h = z0m; z0hg = z0mg;
    if (c->O.opt_sfc == 1) {
      nmp_sfcdif1(c, iter, sfctmp, rhoair, h, qair, zlvl, zpd, z0m, z0h, ur, MPE, &mo, cm, ch, &ch2);
      if (c->err) return;
    }
    if (c->O.opt_sfc == 2) {
      sfcdif2(iter, z0m, *tah, thair, ur, c->P.czil, zlvl, cm, ch, &mo.moz, &wstar, &mo.fv);
      *ch = *ch / ur;
      *cm = *cm / ur;
    }
    real ramc = MAXF(1.f, 1.f / (*cm * ur));
    (void)ramc;
    rahc = MAXF(1.f, 1.f / (*ch * ur));
    rawc = rahc;
    ragrb(c, iter, vaie, rhoair, hg, *tah, zpd, z0mg, z0hg, hcan, uc, z0h, mo.fv, cwp, MPE, &fhg,
          &rahg, &rawg, &rb);
    t = nmp_tdc(*tv);
    nmp_esat(t, &esatw, &esati, &dsatw, &dsati);
    if (t > 0.f) { estv = esatw; destv = dsatw; } else { estv = esati; destv = dsati; }
    if (iter == 1) {
      if (c->O.opt_crs == 1) {
        stomata(c, MPE, parsun, foln, *tv, estv, *eah, sfctmp, sfcprs, o2air, co2air, igs, btran, rb,
                rssun, psnsun);
        stomata(c, MPE, parsha, foln, *tv, estv, *eah, sfctmp, sfcprs, o2air, co2air, igs, btran, rb,
                rssha, psnsha);
      }
      if (c->O.opt_crs == 2) {
        canres(c, parsun, *tv, btran, *eah, sfcprs, rssun, psnsun);
        canres(c, parsha, *tv, btran, *eah, sfcprs, rssha, psnsha);
      }
    }
    cah = 1.f / rahc;
    cvh = 2.f * vaie / rb;
    cgh = 1.f / rahg;
    cond = cah + cvh + cgh;
    ata = (sfctmp * cah + *tg * cgh) / cond;
    bta = cvh / cond;
    csh = (1.f - bta) * rhoair * CPAIR * cvh;
    caw = 1.f / rawc;
    cew = fwet * vaie / rb;
    ctw = (1.f - fwet) * (laisune / (rb + *rssun) + laishae / (rb + *rssha));
    cgw = 1.f / (rawg + rsurf);
    cond = caw + cew + ctw + cgw;
    aea = (eair * caw + estg * cgw) / cond;
    bea = (cew + ctw) / cond;
    cev = (1.f - bea) * cew * rhoair * CPAIR / gammav;
    ctr = (1.f - bea) * ctw * rhoair * CPAIR / gammav;
    *tah = ata + bta * *tv;
    *eah = aea + bea * estv;
    *irc = fveg * (air + cir * powi(*tv, 4));
    *shc = fveg * rhoair * CPAIR * cvh * (*tv - *tah);
    *evc = fveg * rhoair * CPAIR * cew * (estv - *eah) / gammav;
    *tr = fveg * rhoair * CPAIR * ctw * (estv - *eah) / gammav;
    if (*tv > TFRZ) *evc = MINF(canliq * latheav / dt, *evc);
    else *evc = MINF(canice * latheav / dt, *evc);
    b = sav - *irc - *shc - *evc - *tr;
    a = fveg * (4.f * cir * powi(*tv, 3) + csh + (cev + ctr) * destv);
    dtv = b / a;
    *irc = *irc + fveg * 4.f * cir * powi(*tv, 3) * dtv;
    *shc = *shc + fveg * csh * dtv;
    *evc = *evc + fveg * cev * destv * dtv;
    *tr = *tr + fveg * ctr * destv * dtv;
    *tv = *tv + dtv;
    h = rhoair * CPAIR * (*tah - sfctmp) / rahc;
    hg = rhoair * CPAIR * (*tg - *tah) / rahg;
    *qsfc = (0.622f * *eah) / (sfcprs - 0.378f * *eah);
    if (liter == 1) break;
    if (iter >= 5 && fabsf(dtv) <= 0.01f && liter == 0) liter = 1;
  }
  /* under-canopy ground fluxes, lsm:3495-3528 */
  air = -emg * (1.f - emv) * lwdn - emg * emv * SB * powi(*tv, 4);
  cir = emg * SB;
  csh = rhoair * CPAIR / rahg;
  cev = rhoair * CPAIR / (gammag * (rawg + rsurf));
  cgh = 2.f * df[L(isnow + 1)] / dzsnso[L(isnow + 1)];
  for (int iter = 1; iter <= 5; iter++) {            /* loop2, NITERG=5 */
    t = nmp_tdc(*tg);
    nmp_esat(t, &esatw, &esati, &dsatw, &dsati);
    if (t > 0.f) { estg = esatw; destg = dsatw; } else { estg = esati; destg = dsati; }
    *irg = cir * powi(*tg, 4) + air;
    *shg = csh * (*tg - *tah);
    *evg = cev * (estg * rhsur - *eah);
    *gh = cgh * (*tg - stc[L(isnow + 1)]);
    b = sag - *irg - *shg - *evg - *gh;
    a = 4.f * cir * powi(*tg, 3) + csh + cev * destg + cgh;
    real dtg = b / a;
    *irg = *irg + 4.f * cir * powi(*tg, 3) * dtg;
    *shg = *shg + csh * dtg;
    *evg = *evg + cev * destg * dtg;
    *gh = *gh + cgh * dtg;
    *tg = *tg + dtg;
  }
  if (c->O.opt_stc == 1) {
    if (snowh > 0.05f && *tg > TFRZ) {
      *tg = TFRZ;
      *irg = cir * powi(*tg, 4) - emg * (1.f - emv) * lwdn - emg * emv * SB * powi(*tv, 4);
      *shg = csh * (*tg - *tah);
      *evg = cev * (estg * rhsur - *eah);
      *gh = sag - (*irg + *shg + *evg);
    }
  }
  *tauxv = -rhoair * *cm * ur * uu;
  *tauyv = -rhoair * *cm * ur * vv;
  /* 2-m diagnostics lsm:3557-3571.  OPT_SFC=2 leaves FH2 undefined in the reference
     (SFCDIF2 never sets it); mo.fh2 is 0 there. */
  *cah2 = mo.fv * VKC / (logf((2.f + z0h) / z0h) - mo.fh2);
  real cq2v = *cah2;
  if (*cah2 < 1.E-5f) {
    *t2mv = *tah;
    *q2v = *qsfc;
  } else {
    *t2mv = *tah - (*shg + *shc / fveg) / (rhoair * CPAIR) * 1.f / *cah2;
    *q2v = *qsfc - ((*evc + *tr) / fveg + *evg) / (latheav * rhoair) * 1.f / cq2v;
  }
  *ch = cah;
  *chleaf = cvh;
  *chuc = 1.f / rahg;
  (void)latheag; (void)thair;
}

/* BARE_FLUX lsm:3591-3958 */
static void bare_flux(nmp_ctx* c, int isnow, real sag, real lwdn, real ur, real uu, real vv,
                      real sfctmp, real thair, real qair, real eair, real rhoair, real snowh,
                      const real* dzsnso, real zlvl, real zpd, real z0m, real emg, const real* stc,
                      const real* df, real rsurf, real lathea, real gamma, real rhsur, real* tgb,
                      real* cm, real* ch, real* tauxb, real* tauyb, real* irb, real* shb, real* evb,
                      real* ghb, real* t2mb, real* qsfc, real psfc, real* q2b, real* ehb2) {
  const real MPE = 1E-6f;
  mo_state mo = {0.f, 0.f, 0.f, 0.f, 0.f, 0.1f, 0};
  real h = 0.f, z0h = z0m, ch2, wstar = 0.f;
  real t, esatw, esati, dsatw, dsati, estg = 0.f, destg, csh = 0.f, cev = 0.f, ehb = 0.f;
  real cir = emg * SB;
  real cgh = 2.f * df[L(isnow + 1)] / dzsnso[L(isnow + 1)];
  for (int iter = 1; iter <= 5; iter++) {            /* loop3, NITERB=5 (lsm:3749) */
    z0h = z0m;
    if (c->O.opt_sfc == 1) {
      nmp_sfcdif1(c, iter, sfctmp, rhoair, h, qair, zlvl, zpd, z0m, z0h, ur, MPE, &mo, cm, ch, &ch2);
      if (c->err) return;
    }
    if (c->O.opt_sfc == 2) {
      sfcdif2(iter, z0m, *tgb, thair, ur, c->P.czil, zlvl, cm, ch, &mo.moz, &wstar, &mo.fv);
      *ch = *ch / ur;
      *cm = *cm / ur;
      if (snowh > 0.f) { *cm = MINF(0.01f, *cm); *ch = MINF(0.01f, *ch); }
    }
    real rahb = MAXF(1.f, 1.f / (*ch * ur));
    real rawb = rahb;
    ehb = 1.f / rahb;
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
    *qsfc = 0.622f * (estg * rhsur) / (psfc - 0.378f * (estg * rhsur));
  }
  if (c->O.opt_stc == 1) {
    if (snowh > 0.05f && *tgb > TFRZ) {
      *tgb = TFRZ;
      *irb = cir * powi(*tgb, 4) - emg * lwdn;
      *shb = csh * (*tgb - sfctmp);
      *evb = cev * (estg * rhsur - eair);
      *ghb = sag - (*irb + *shb + *evb);
    }
  }
  *tauxb = -rhoair * *cm * ur * uu;
  *tauyb = -rhoair * *cm * ur * vv;
  *ehb2 = mo.fv * VKC / (logf((2.f + z0h) / z0h) - mo.fh2);
  real cq2b = *ehb2;
  if (*ehb2 < 1.E-5f) {
    *t2mb = *tgb;
    *q2b = *qsfc;
  } else {
    *t2mb = *tgb - *shb / (rhoair * CPAIR) * 1.f / *ehb2;
    *q2b = *qsfc - *evb / (lathea * rhoair) * (1.f / cq2b + rsurf);
  }
  if (c->vegtyp == c->isurban) *q2b = *qsfc;
  *ch = ehb;
}

/* ROSR12 lsm:5979-6036: Thomas algorithm, rows ntop..nsoil; result returned in p[] */
void nmp_rosr12(real* p, const real* a, const real* b, real* cc, const real* d, real* delta, int ntop,
                int nsoil) {
  cc[L(nsoil)] = 0.0f;
  p[L(ntop)] = -cc[L(ntop)] / b[L(ntop)];
  delta[L(ntop)] = d[L(ntop)] / b[L(ntop)];
  for (int k = ntop + 1; k <= nsoil; k++) {
    p[L(k)] = -cc[L(k)] * (1.0f / (b[L(k)] + a[L(k)] * p[L(k - 1)]));
    delta[L(k)] = (d[L(k)] - a[L(k)] * delta[L(k - 1)]) * (1.0f / (b[L(k)] + a[L(k)] * p[L(k - 1)]));
  }
  p[L(nsoil)] = delta[L(nsoil)];
  for (int k = ntop + 1; k <= nsoil; k++) {
    int kk = nsoil - k + (ntop - 1) + 1;
    p[L(kk)] = p[L(kk)] * p[L(kk + 1)] + delta[L(kk)];
  }
}

/* TSNOSOI lsm:5707-5822 = HRT (5825-5922) + HSTEP (5925-5977); returns before its energy check */
void nmp_tsnosoi(const nmp_ctx* c, int isnow, real tbot, const real* zsnso, real ssoil,
                    const real* df, const real* hcpct, real zbot, real dt, real snowh, real* stc) {
  int ns = c->nsoil;
  real zbotsno = zbot - snowh;
  real denom[NL], ddz[NL], dtsdz[NL], eflux[NL], ai[NL], bi[NL], ci[NL], rhsts[NL];
  real botflx = 0.f;
  for (int k = isnow + 1; k <= ns; k++) {
    real temp1;
    if (k == isnow + 1) {
      denom[L(k)] = -zsnso[L(k)] * hcpct[L(k)];
      temp1 = -zsnso[L(k + 1)];
      ddz[L(k)] = 2.0f / temp1;
      dtsdz[L(k)] = 2.0f * (stc[L(k)] - stc[L(k + 1)]) / temp1;
      eflux[L(k)] = df[L(k)] * dtsdz[L(k)] - ssoil - 0.f;
    } else if (k < ns) {
      denom[L(k)] = (zsnso[L(k - 1)] - zsnso[L(k)]) * hcpct[L(k)];
      temp1 = zsnso[L(k - 1)] - zsnso[L(k + 1)];
      ddz[L(k)] = 2.0f / temp1;
      dtsdz[L(k)] = 2.0f * (stc[L(k)] - stc[L(k + 1)]) / temp1;
      eflux[L(k)] = (df[L(k)] * dtsdz[L(k)] - df[L(k - 1)] * dtsdz[L(k - 1)]) - 0.f;
    } else {
      denom[L(k)] = (zsnso[L(k - 1)] - zsnso[L(k)]) * hcpct[L(k)];
      if (c->O.opt_tbot == 1) botflx = 0.f;
      if (c->O.opt_tbot == 2) {
        dtsdz[L(k)] = (stc[L(k)] - tbot) / (0.5f * (zsnso[L(k - 1)] + zsnso[L(k)]) - zbotsno);
        botflx = -df[L(k)] * dtsdz[L(k)];
      }
      eflux[L(k)] = (-botflx - df[L(k - 1)] * dtsdz[L(k - 1)]) - 0.f;
    }
  }
  for (int k = isnow + 1; k <= ns; k++) {
    if (k == isnow + 1) {
      ai[L(k)] = 0.0f;
      ci[L(k)] = -df[L(k)] * ddz[L(k)] / denom[L(k)];
      if (c->O.opt_stc == 1) bi[L(k)] = -ci[L(k)];
      if (c->O.opt_stc == 2)
        bi[L(k)] = -ci[L(k)] + df[L(k)] / (0.5f * zsnso[L(k)] * zsnso[L(k)] * hcpct[L(k)]);
    } else if (k < ns) {
      ai[L(k)] = -df[L(k - 1)] * ddz[L(k - 1)] / denom[L(k)];
      ci[L(k)] = -df[L(k)] * ddz[L(k)] / denom[L(k)];
      bi[L(k)] = -(ai[L(k)] + ci[L(k)]);
    } else {
      ai[L(k)] = -df[L(k - 1)] * ddz[L(k - 1)] / denom[L(k)];
      ci[L(k)] = 0.0f;
      bi[L(k)] = -(ai[L(k)] + ci[L(k)]);
    }
    rhsts[L(k)] = eflux[L(k)] / (-denom[L(k)]);
  }
  /* HSTEP */
  real rhstsin[NL], ciin[NL];
  for (int k = isnow + 1; k <= ns; k++) {
    rhsts[L(k)] = rhsts[L(k)] * dt;
    ai[L(k)] = ai[L(k)] * dt;
    bi[L(k)] = 1.f + bi[L(k)] * dt;
    ci[L(k)] = ci[L(k)] * dt;
  }
  for (int k = isnow + 1; k <= ns; k++) { rhstsin[L(k)] = rhsts[L(k)]; ciin[L(k)] = ci[L(k)]; }
  nmp_rosr12(ci, ai, bi, ciin, rhstsin, rhsts, isnow + 1, ns);
  for (int k = isnow + 1; k <= ns; k++) stc[L(k)] = stc[L(k)] + ci[L(k)];
}

/* FRH2O lsm:6247-6377 (Koren et al. 1999 supercooled water) */
static real frh2o(const nmp_ctx* c, real tkelv, real smc, real sh2o) {
  const nmp_parm* P = &c->P;
  const real CK = 8.0f, BLIM = 5.5f, ERROR = 0.005f;
  real bx = P->bexp, free_ = 0.f;
  if (P->bexp > BLIM) bx = BLIM;
  int nlog = 0, kcount = 0;
  if (tkelv > (TFRZ - 1.E-3f)) {
    free_ = smc;
  } else {
    real swl = smc - sh2o;
    if (swl > (smc - 0.02f)) swl = smc - 0.02f;
    if (swl < 0.f) swl = 0.f;
    while ((nlog < 10) && (kcount == 0)) {
      nlog = nlog + 1;
      real df = logf((P->psisat * GRAV / HFUS) * powf(1.f + CK * swl, 2.f) *
                     powf(P->smcmax / (smc - swl), bx)) -
                logf(-(tkelv - TFRZ) / tkelv);
      real denom = 2.f * CK / (1.f + CK * swl) + bx / (smc - swl);
      real swlk = swl - df / denom;
      if (swlk > (smc - 0.02f)) swlk = smc - 0.02f;
      if (swlk < 0.f) swlk = 0.f;
      real dswl = fabsf(swlk - swl);
      swl = swlk;
      if (dswl <= ERROR) kcount = kcount + 1;
    }
    free_ = smc - swl;
    if (kcount == 0) {
      real fk = powf((HFUS / (GRAV * (-P->psisat))) * ((tkelv - TFRZ) / tkelv), -1 / bx) * P->smcmax;
      if (fk < 0.02f) fk = 0.02f;
      free_ = MINF(fk, smc);
    }
  }
  return free_;
}

/* PHASECHANGE lsm:6039-6245 */
static void phasechange(const nmp_ctx* c, int isnow, real dt, const real* fact, const real* dzsnso,
                        int ist, real* stc, real* snice, real* snliq, real* sneqv, real* snowh,
                        real* smc, real* sh2o, real* qmelt, int* imelt, real* ponding) {
  const nmp_parm* P = &c->P;
  int ns = c->nsoil;
  real hm[NL], xm[NL], wmass0[NL], wice0[NL], wliq0[NL], mice[NL], mliq[NL], supercool[NL];
  real xmf = 0.f, heatr;
  *qmelt = 0.f; *ponding = 0.f;
  for (int j = -c->nsnow + 1; j <= ns; j++) supercool[L(j)] = 0.0f;
  for (int j = isnow + 1; j <= 0; j++) { mice[L(j)] = snice[L(j)]; mliq[L(j)] = snliq[L(j)]; }
  for (int j = 1; j <= ns; j++) {
    mliq[L(j)] = sh2o[L(j)] * dzsnso[L(j)] * 1000.f;
    mice[L(j)] = (smc[L(j)] - sh2o[L(j)]) * dzsnso[L(j)] * 1000.f;
  }
  for (int j = isnow + 1; j <= ns; j++) {
    imelt[L(j)] = 0; hm[L(j)] = 0.f; xm[L(j)] = 0.f;
    wice0[L(j)] = mice[L(j)]; wliq0[L(j)] = mliq[L(j)]; wmass0[L(j)] = mice[L(j)] + mliq[L(j)];
  }
  (void)wliq0;
  if (ist == 1) {
    for (int j = 1; j <= ns; j++) {
      if (c->O.opt_frz == 1) {
        if (stc[L(j)] < TFRZ) {
          real smp = HFUS * (TFRZ - stc[L(j)]) / (GRAV * stc[L(j)]);
          supercool[L(j)] = P->smcmax * powf(smp / P->psisat, -1.f / P->bexp);
          supercool[L(j)] = supercool[L(j)] * dzsnso[L(j)] * 1000.f;
        }
      }
      if (c->O.opt_frz == 2) {
        supercool[L(j)] = frh2o(c, stc[L(j)], smc[L(j)], sh2o[L(j)]);
        supercool[L(j)] = supercool[L(j)] * dzsnso[L(j)] * 1000.f;
      }
    }
  }
  for (int j = isnow + 1; j <= ns; j++) {
    if (mice[L(j)] > 0.f && stc[L(j)] >= TFRZ) imelt[L(j)] = 1;
    if (mliq[L(j)] > supercool[L(j)] && stc[L(j)] < TFRZ) imelt[L(j)] = 2;
    if (isnow == 0 && *sneqv > 0.f && j == 1) {
      if (stc[L(j)] >= TFRZ) imelt[L(j)] = 1;
    }
  }
  for (int j = isnow + 1; j <= ns; j++) {
    if (imelt[L(j)] > 0) {
      hm[L(j)] = (stc[L(j)] - TFRZ) / fact[L(j)];
      stc[L(j)] = TFRZ;
    }
    if (imelt[L(j)] == 1 && hm[L(j)] < 0.f) { hm[L(j)] = 0.f; imelt[L(j)] = 0; }
    if (imelt[L(j)] == 2 && hm[L(j)] > 0.f) { hm[L(j)] = 0.f; imelt[L(j)] = 0; }
    xm[L(j)] = hm[L(j)] * dt / HFUS;
  }
  if (isnow == 0 && *sneqv > 0.f && xm[L(1)] > 0.f) {
    real temp1 = *sneqv;
    *sneqv = MAXF(0.f, temp1 - xm[L(1)]);
    real propor = *sneqv / temp1;
    *snowh = MAXF(0.f, propor * *snowh);
    heatr = hm[L(1)] - HFUS * (temp1 - *sneqv) / dt;
    if (heatr > 0.f) { xm[L(1)] = heatr * dt / HFUS; hm[L(1)] = heatr; }
    else { xm[L(1)] = 0.f; hm[L(1)] = 0.f; }
    *qmelt = MAXF(0.f, (temp1 - *sneqv)) / dt;
    xmf = HFUS * *qmelt;
    *ponding = temp1 - *sneqv;
  }
  for (int j = isnow + 1; j <= ns; j++) {
    if (imelt[L(j)] > 0 && fabsf(hm[L(j)]) > 0.f) {
      heatr = 0.f;
      if (xm[L(j)] > 0.f) {
        mice[L(j)] = MAXF(0.f, wice0[L(j)] - xm[L(j)]);
        heatr = hm[L(j)] - HFUS * (wice0[L(j)] - mice[L(j)]) / dt;
      } else if (xm[L(j)] < 0.f) {
        if (j <= 0) {
          mice[L(j)] = MINF(wmass0[L(j)], wice0[L(j)] - xm[L(j)]);
        } else {
          if (wmass0[L(j)] < supercool[L(j)]) {
            mice[L(j)] = 0.f;
          } else {
            mice[L(j)] = MINF(wmass0[L(j)] - supercool[L(j)], wice0[L(j)] - xm[L(j)]);
            mice[L(j)] = MAXF(mice[L(j)], 0.0f);
          }
        }
        heatr = hm[L(j)] - HFUS * (wice0[L(j)] - mice[L(j)]) / dt;
      }
      mliq[L(j)] = MAXF(0.f, wmass0[L(j)] - mice[L(j)]);
      if (fabsf(heatr) > 0.f) {
        stc[L(j)] = stc[L(j)] + fact[L(j)] * heatr;
        if (j <= 0) {
          if (mliq[L(j)] * mice[L(j)] > 0.f) stc[L(j)] = TFRZ;
        }
      }
      xmf = xmf + HFUS * (wice0[L(j)] - mice[L(j)]) / dt;
      if (j < 1) *qmelt = *qmelt + MAXF(0.f, (wice0[L(j)] - mice[L(j)])) / dt;
    }
  }
  (void)xmf;
  for (int j = isnow + 1; j <= 0; j++) { snliq[L(j)] = mliq[L(j)]; snice[L(j)] = mice[L(j)]; }
  for (int j = 1; j <= ns; j++) {
    sh2o[L(j)] = mliq[L(j)] / (1000.f * dzsnso[L(j)]);
    smc[L(j)] = (mliq[L(j)] + mice[L(j)]) / (1000.f * dzsnso[L(j)]);
  }
}

/* ENERGY lsm:1231-1843 */
void nmp_energy(nmp_ctx* c, nmp_column* s, nmp_work* w) {
  const noahmp_tables* T = c->T;
  const nmp_parm* P = &c->P;
  int v = c->vegtyp - 1, ns = c->nsoil;
  const real MPE = 1.E-6f, PSIWLT = -150.f, Z0 = 0.01f;
  real tauxv = 0.f, tauyv = 0.f, tauxb, tauyb;
  real psnsun = 0.f, psnsha = 0.f;
  s->irc = 0.f; s->shc = 0.f; s->irg = 0.f; s->shg = 0.f; s->evg = 0.f; s->evc = 0.f; s->tr = 0.f;
  s->ghv = 0.f; s->t2mv = 0.f; s->q2v = 0.f; s->chv = 0.f; s->chleaf = 0.f; s->chuc = 0.f;
  s->chv2 = 0.f;
  real ur = MAXF(sqrtf(powf(s->uu, 2.f) + powf(s->vv, 2.f)), 1.f);     /* lsm:1536 UU**2. */
  real vai = w->elai + w->esai;
  int veg = (vai > 0.f);
  s->fsno = 0.f;
  if (s->snowh > 0.f) {
    real bdsno = s->sneqv / s->snowh;
    real fmelt = powf(bdsno / 100.f, M_MELT);
    s->fsno = tanhf(s->snowh / (2.5f * Z0 * fmelt));
  }
  real z0mg;
  if (s->ist == 2) {
    if (s->tg <= TFRZ) z0mg = 0.01f * (1.0f - s->fsno) + s->fsno * Z0SNO;
    else z0mg = 0.01f;
  } else {
    z0mg = Z0 * (1.0f - s->fsno) + s->fsno * Z0SNO;
  }
  real zpdg = s->snowh, z0m, zpd;
  if (veg) {
    z0m = T->z0mvt[v];
    zpd = 0.65f * w->htop;
    if (s->snowh > zpd) zpd = s->snowh;
  } else {
    z0m = z0mg;
    zpd = zpdg;
  }
  real zlvl = MAXF(zpd, w->htop) + s->zlvl;
  if (zpdg >= zlvl) zlvl = zpdg + s->zlvl;
  real cwp = T->cwpvt[v];
  real df[NL], hcpct[NL], fact[NL];
  thermoprop(c, s->isnow, s->ist, w->dzsnso, c->dt, s->snowh, s->snice, s->snliq, s->smc, s->sh2o,
             s->stc, df, hcpct, w->snicev, w->snliqv, w->epore, fact);
  real fsun, laisun, laisha, parsun, parsha;
  radiation(c, s->ist, s->isc, s->ice, s->sneqvo, s->sneqv, c->dt, s->cosz, s->snowh, s->tg, s->tv,
            s->fsno, s->qsnow, s->fwet, w->elai, w->esai, s->smc, w->solad, w->solai, s->fveg,
            &s->albold, &s->tauss, &fsun, &laisun, &laisha, &parsun, &parsha, &s->sav, &s->sag,
            &s->fsr, &s->fsa, &w->fsrv, &w->fsrg, &s->bgap, &s->wgap);
  real emv = 1.f - expf(-(w->elai + w->esai) / 1.0f);
  real emg;
  if (s->ice == 1) emg = 0.98f * (1.f - s->fsno) + 1.0f * s->fsno;
  else emg = T->eg[s->ist - 1] * (1.f - s->fsno) + 1.0f * s->fsno;
  /* soil moisture factor BTRAN lsm:1617-1640 */
  w->btran = 0.f;
  if (s->ist == 1) {
    for (int iz = 1; iz <= P->nroot; iz++) {
      real gx = 0.f, psi;
      if (c->O.opt_btr == 1) gx = (s->sh2o[L(iz)] - P->smcwlt) / (P->smcref - P->smcwlt);
      if (c->O.opt_btr == 2) {
        psi = MAXF(PSIWLT, -P->psisat * powf(MAXF(0.01f, s->sh2o[L(iz)]) / P->smcmax, -P->bexp));
        gx = (1.f - psi / PSIWLT) / (1.f + P->psisat / PSIWLT);
      }
      if (c->O.opt_btr == 3) {
        psi = MAXF(PSIWLT, -P->psisat * powf(MAXF(0.01f, s->sh2o[L(iz)]) / P->smcmax, -P->bexp));
        gx = 1.f - expf(-5.8f * (logf(PSIWLT / psi)));
      }
      gx = MINF(1.f, MAXF(0.f, gx));
      w->btrani[L(iz)] = MAXF(MPE, w->dzsnso[L(iz)] / (-c->zsoil[L(P->nroot)]) * gx);
      w->btran = w->btran + w->btrani[L(iz)];
    }
    w->btran = MAXF(MPE, w->btran);
    for (int iz = 1; iz <= P->nroot; iz++) w->btrani[L(iz)] = w->btrani[L(iz)] / w->btran;
  }
  /* soil surface resistance lsm:1644-1669 */
  real rsurf, rhsur;
  if (s->ist == 2) {
    rsurf = 1.f; rhsur = 1.0f;
  } else {
    real l_rsurf = (-c->zsoil[L(1)]) *
                   (expf(powi(1.0f - MINF(1.0f, s->sh2o[L(1)] / P->smcmax), 5)) - 1.0f) /
                   (2.71828f - 1.0f);
    real d_rsurf = 2.2E-5f * P->smcmax * P->smcmax *
                   powf(1.0f - P->smcwlt / P->smcmax, 2.0f + 3.0f / P->bexp);
    rsurf = l_rsurf / d_rsurf;
    if (s->sh2o[L(1)] < 0.01f && s->snowh == 0.f) rsurf = 1.E6f;
    real psi = -P->psisat * powf(MAXF(0.01f, s->sh2o[L(1)]) / P->smcmax, -P->bexp);
    rhsur = s->fsno + (1.f - s->fsno) * expf(psi * GRAV / (RW * s->tg));
  }
  if (c->vegtyp == c->isurban && s->snowh == 0.f) rsurf = 1.E6f;
  /* psychrometric constants lsm:1673-1689 */
  if (s->tv > TFRZ) { w->latheav = HVAP; w->frozen_canopy = 0; }
  else { w->latheav = HSUB; w->frozen_canopy = 1; }
  real gammav = CPAIR * s->sfcprs / (0.622f * w->latheav);
  if (s->tg > TFRZ) { w->latheag = HVAP; w->frozen_ground = 0; }
  else { w->latheag = HSUB; w->frozen_ground = 1; }
  real gammag = CPAIR * s->sfcprs / (0.622f * w->latheag);

  real cmv = 0.f, cmb;
  if (veg && s->fveg > 0) {
    s->tgv = s->tg;
    cmv = s->cm;
    s->chv = s->ch;
    vege_flux(c, s->isnow, c->dt, s->sav, s->sag, s->lwdn, ur, s->uu, s->vv, s->sfctmp, w->thair,
              w->qair, w->eair, w->rhoair, s->snowh, vai, gammav, gammag, s->fwet, laisun, laisha,
              cwp, w->dzsnso, w->htop, zlvl, zpd, z0m, s->fveg, z0mg, emv, emg, s->canliq, s->canice,
              s->stc, df, &s->rssun, &s->rssha, rsurf, w->latheav, w->latheag, parsun, parsha, w->igs,
              s->foln, s->co2air, s->o2air, w->btran, s->sfcprs, rhsur, &s->eah, &s->tah, &s->tv,
              &s->tgv, &cmv, &s->chv, &tauxv, &tauyv, &s->irg, &s->irc, &s->shg, &s->shc, &s->evg,
              &s->evc, &s->tr, &s->ghv, &s->t2mv, &psnsun, &psnsha, &s->qsfc, s->psfc, &s->q2v,
              &s->chv2, &s->chleaf, &s->chuc);
    if (c->err) return;
  }
  s->tgb = s->tg;
  cmb = s->cm;
  s->chb = s->ch;
  bare_flux(c, s->isnow, s->sag, s->lwdn, ur, s->uu, s->vv, s->sfctmp, w->thair, w->qair, w->eair,
            w->rhoair, s->snowh, w->dzsnso, zlvl, zpdg, z0mg, emg, s->stc, df, rsurf, w->latheag,
            gammag, rhsur, &s->tgb, &cmb, &s->chb, &tauxb, &tauyb, &s->irb, &s->shb, &s->evb, &s->ghb,
            &s->t2mb, &s->qsfc, s->psfc, &s->q2b, &s->chb2);
  if (c->err) return;
  /* tile blend lsm:1747-1783 */
  if (veg && s->fveg > 0) {
    s->fira = s->fveg * s->irg + (1.0f - s->fveg) * s->irb + s->irc;
    s->fsh = s->fveg * s->shg + (1.0f - s->fveg) * s->shb + s->shc;
    s->fgev = s->fveg * s->evg + (1.0f - s->fveg) * s->evb;
    s->ssoil = s->fveg * s->ghv + (1.0f - s->fveg) * s->ghb;
    s->fcev = s->evc;
    s->fctr = s->tr;
    s->tg = s->fveg * s->tgv + (1.0f - s->fveg) * s->tgb;
    s->cm = s->fveg * cmv + (1.0f - s->fveg) * cmb;
    s->ch = s->fveg * s->chv + (1.0f - s->fveg) * s->chb;
  } else {
    s->fira = s->irb; s->fsh = s->shb; s->fgev = s->evb; s->ssoil = s->ghb; s->tg = s->tgb;
    s->fcev = 0.f; s->fctr = 0.f;
    s->cm = cmb; s->ch = s->chb;
    s->rssun = 0.0f; s->rssha = 0.0f;
    s->tgv = s->tgb; s->chv = s->chb;
  }
  real fire = s->lwdn + s->fira;
  if (fire <= 0.f) { if (!c->err) c->err = NOAHMP_ERR_FIRE_NONPOSITIVE; return; }
  s->emissi = s->fveg * (emg * (1 - emv) + emv + emv * (1 - emv) * (1 - emg)) + (1 - s->fveg) * emg;
  s->trad = powf((fire - (1 - s->emissi) * s->lwdn) / (s->emissi * SB), 0.25f);
  s->apar = parsun * laisun + parsha * laisha;
  s->psn = psnsun * laisun + psnsha * laisha;
  nmp_tsnosoi(c, s->isnow, s->tbot, s->zsnso, s->ssoil, df, hcpct, P->zbot, c->dt, s->snowh, s->stc);
  if (c->O.opt_stc == 2) {
    if (s->snowh > 0.05f && s->tg > TFRZ) {
      s->tgv = TFRZ; s->tgb = TFRZ;
      if (veg && s->fveg > 0) s->tg = s->fveg * s->tgv + (1.0f - s->fveg) * s->tgb;
      else s->tg = s->tgb;
    }
  }
  phasechange(c, s->isnow, c->dt, fact, w->dzsnso, s->ist, s->stc, s->snice, s->snliq, &s->sneqv,
              &s->snowh, s->smc, s->sh2o, &w->qmelt, w->imelt, &s->ponding);
  (void)ns; (void)tauxv; (void)tauyv; (void)tauxb; (void)tauyb;
}

/* TEST INFRASTRUCTURE (oracle): NOAHMP_SFLX orchestration + the noahmplsm column loop.
 * Reference: lsm:518-1228 (NOAHMP_SFLX, ATM, PHENOLOGY, ERROR), lsm:8723-9104 (CARBON),
 * lsm:9202-9349 (REDPRM), drv:376-840 (noahmplsm gather/scatter). */
#include <math.h>
#include <string.h>
#include "nmp_internal.h"

void nmp_glacier_column(nmp_ctx* c, nmp_column* s, real* fsr_out);   /* nmp_glacier.c */

/* REDPRM lsm:9202-9349 */
void nmp_redprm(nmp_ctx* c, int vegtyp, int soiltyp, int slopetyp) {
  const noahmp_tables* T = c->T;
  nmp_parm* P = &c->P;
  if (soiltyp > T->slcats) { if (!c->err) c->err = NOAHMP_ERR_SOILTYP_RANGE; return; }
  if (vegtyp > T->lucats) { if (!c->err) c->err = NOAHMP_ERR_VEGTYP_RANGE; return; }
  int st = soiltyp - 1, vt = vegtyp - 1;
  P->csoil = T->csoil_data;
  P->bexp = T->bb[st];
  P->dksat = T->satdk[st];
  P->dwsat = T->satdw[st];
  P->f1 = T->f11[st];
  P->psisat = T->satpsi[st];
  P->quartz = T->qtz[st];
  P->smcdry = T->drysmc[st];
  P->smcmax = T->maxsmc[st];
  P->smcref = T->refsmc[st];
  P->smcwlt = T->wltsmc[st];
  if (vegtyp == c->isurban) {
    P->smcmax = 0.45f; P->smcref = 0.42f; P->smcwlt = 0.40f; P->smcdry = 0.40f; P->csoil = 3.E6f;
  }
  P->zbot = T->zbot_data;
  P->czil = T->czil_data;
  real frzk = T->frzk_data, refdk = T->refdk_data, refkdt = T->refkdt_data;
  P->kdt = refkdt * P->dksat / refdk;
  P->slope = T->slope_data[slopetyp - 1];
  /* the reference leaves FRZX stale for soil 14 (lsm:9316); computed unconditionally here */
  {
    real frzfact = (P->smcmax / P->smcref) * (0.412f / 0.468f);
    P->frzx = frzk * frzfact;
  }
  P->topt = T->topt_data;
  P->rgl = T->rgltbl[vt];
  P->rsmax = T->rsmax_data;
  P->rsmin = T->rstbl[vt];
  P->hs = T->hstbl[vt];
  P->nroot = T->nrotbl[vt];
  if (vegtyp == c->isurban) P->rsmin = 400.0f;
  if (P->nroot > c->nsoil) { if (!c->err) c->err = NOAHMP_ERR_NROOT_GT_NSOIL; }
}

/* PHENOLOGY lsm:1010-1104 */
static void phenology(const nmp_ctx* c, nmp_column* s, nmp_work* w) {
  const noahmp_tables* T = c->T;
  int v = c->vegtyp - 1;
  if (c->O.dveg == 1 || c->O.dveg == 3 || c->O.dveg == 4) {
    real day;
    if (s->lat >= 0.f) day = s->julian;
    else day = fmodf(s->julian + (0.5f * s->yearlen), (real)s->yearlen);
    real t = 12.f * day / (real)s->yearlen;
    int it1 = (int)(t + 0.5f);                      /* REAL->INTEGER truncation, lsm:1063 */
    int it2 = it1 + 1;
    real wt1 = (it1 + 0.5f) - t;
    real wt2 = 1.f - wt1;
    if (it1 < 1) it1 = 12;
    if (it2 > 12) it2 = 1;
    s->lai = wt1 * T->laim[it1 - 1][v] + wt2 * T->laim[it2 - 1][v];
    s->sai = wt1 * T->saim[it1 - 1][v] + wt2 * T->saim[it2 - 1][v];
  }
  if (s->sai < 0.01f) s->sai = 0.0f;
  if (s->lai < 0.05f || s->sai == 0.0f) s->lai = 0.0f;
  if ((c->vegtyp == T->iswater) || (c->vegtyp == T->isbarren) || (c->vegtyp == T->issnow) ||
      (c->vegtyp == c->isurban)) {
    s->lai = 0.f; s->sai = 0.f;
  }
  real db = MINF(MAXF(s->snowh - T->hvb[v], 0.f), T->hvt[v] - T->hvb[v]);
  real fb = db / MAXF(1.E-06f, T->hvt[v] - T->hvb[v]);
  if (T->hvt[v] > 0.f && T->hvt[v] <= 1.0f) {
    real snowhc = T->hvt[v] * expf(-s->snowh / 0.2f);
    fb = MINF(s->snowh, snowhc) / snowhc;
  }
  w->elai = s->lai * (1.f - fb);
  w->esai = s->sai * (1.f - fb);
  if (w->esai < 0.01f) w->esai = 0.0f;
  if (w->elai < 0.05f || w->esai == 0.0f) w->elai = 0.0f;
  if (s->tv > T->tmin[v]) w->igs = 1.f; else w->igs = 0.f;
  w->htop = T->hvt[v];
}

/* CARBON lsm:8723-8835 + CO2FLUX lsm:8837-9104 (only DVEG 2/5) */
void nmp_carbon(nmp_ctx* c, nmp_column* s, nmp_work* w) {
  const noahmp_tables* T = c->T;
  const nmp_parm* P = &c->P;
  int v = c->vegtyp - 1;
  real dt = c->dt;
  if ((c->vegtyp == T->iswater) || (c->vegtyp == T->isbarren) || (c->vegtyp == T->issnow) ||
      (c->vegtyp == c->isurban)) {
    s->lai = 0.f; s->sai = 0.f; s->gpp = 0.f; s->npp = 0.f; s->nee = 0.f;
    s->lfmass = 0.f; s->rtmass = 0.f; s->stmass = 0.f; s->wood = 0.f; s->stblcp = 0.f; s->fastcp = 0.f;
    return;
  }
  real lapm = T->sla[v] / 1000.f;
  real wstres = 1.f - w->btran;
  real wroot = 0.f;
  for (int j = 1; j <= P->nroot; j++)
    wroot = wroot + s->smc[L(j)] / P->smcmax * w->dzsnso[L(j)] / (-c->zsoil[L(P->nroot)]);
  /* CO2FLUX */
  real rtovrc = 2.0E-8f, rswoodc = 3.0E-10f, bf = 0.90f, wstrc = 100.0f, laimin = 0.05f,
       xsamin = 0.01f;
  real sapm = 3.f * 0.001f;
  real lfmsmn = laimin / lapm, stmsmn = xsamin / sapm;
  real rf = (w->igs == 0.f) ? 0.5f : 1.0f;
  real tv = s->tv;
  real fnf = MINF(s->foln / MAXF(1.E-06f, T->folnmx[v]), 1.0f);
  real tf = powf(T->arm[v], (tv - 298.16f) / 10.f);
  real resp = T->rmf25[v] * tf * fnf * s->lai * rf * (1.f - wstres);
  real rsleaf = MINF(s->lfmass / dt, resp * 12.e-6f);
  real rsroot = T->rmr25[v] * (s->rtmass * 1E-3f) * tf * rf * 12.e-6f;
  real rsstem = T->rms25[v] * (s->stmass * 1E-3f) * tf * rf * 12.e-6f;
  real rswood = rswoodc * expf(0.08f * (tv - 298.16f)) * s->wood * T->wdpool[v];
  real carbfx = s->psn * 12.e-6f;
  real leafpt = expf(0.01f * (1.f - expf(0.75f * s->lai)) * s->lai);
  if (c->vegtyp == T->eblforest) leafpt = expf(0.01f * (1.f - expf(0.50f * s->lai)) * s->lai);
  real nonlef = 1.0f - leafpt;
  real stempt = s->lai / 10.0f;
  leafpt = leafpt - stempt;
  real woodf;
  if (s->wood > 0) woodf = (1.f - expf(-bf * (T->wrrat[v] * s->rtmass / s->wood)) / bf) * T->wdpool[v];
  else woodf = 0.f;
  real rootpt = nonlef * (1.f - woodf);
  real woodpt = nonlef * woodf;
  real lftovr = T->ltovrc[v] * 1.E-6f * s->lfmass;
  real sttovr = T->ltovrc[v] * 1.E-6f * s->stmass;
  real rttovr = rtovrc * s->rtmass;
  real wdtovr = 9.5E-10f * s->wood;
  real sc = expf(-0.3f * MAXF(0.f, tv - T->tdlef[v])) * (s->lfmass / 120.f);
  real sd = expf((wstres - 1.f) * wstrc);
  real dielf = s->lfmass * 1.E-6f * (T->dilefw[v] * sd + T->dilefc[v] * sc);
  real diest = s->stmass * 1.E-6f * (T->dilefw[v] * sd + T->dilefc[v] * sc);
  real grleaf = MAXF(0.0f, T->fragr[v] * (leafpt * carbfx - rsleaf));
  real grstem = MAXF(0.0f, T->fragr[v] * (stempt * carbfx - rsstem));
  real grroot = MAXF(0.0f, T->fragr[v] * (rootpt * carbfx - rsroot));
  real grwood = MAXF(0.0f, T->fragr[v] * (woodpt * carbfx - rswood));
  real addnpplf = MAXF(0.f, leafpt * carbfx - grleaf - rsleaf);
  real addnppst = MAXF(0.f, stempt * carbfx - grstem - rsstem);
  if (tv < T->tmin[v]) addnpplf = 0.f;
  if (tv < T->tmin[v]) addnppst = 0.f;
  real lfdel = (s->lfmass - lfmsmn) / dt;
  real stdel = (s->stmass - stmsmn) / dt;
  dielf = MINF(dielf, lfdel + addnpplf - lftovr);
  diest = MINF(diest, stdel + addnppst - sttovr);
  real nppl = MAXF(addnpplf, -lfdel);
  real npps = MAXF(addnppst, -stdel);
  real nppr = rootpt * carbfx - rsroot - grroot;
  real nppw = woodpt * carbfx - rswood - grwood;
  s->lfmass = s->lfmass + (nppl - lftovr - dielf) * dt;
  s->stmass = s->stmass + (npps - sttovr - diest) * dt;
  s->rtmass = s->rtmass + (nppr - rttovr) * dt;
  if (s->rtmass < 0.0f) { rttovr = nppr; s->rtmass = 0.0f; }
  s->wood = (s->wood + (nppw - wdtovr) * dt) * T->wdpool[v];
  s->fastcp = s->fastcp + (rttovr + lftovr + sttovr + wdtovr + dielf) * dt;
  real fst = powf(2.0f, (s->stc[L(1)] - 283.16f) / 10.f);
  real fsw = wroot / (0.20f + wroot) * 0.23f / (0.23f + wroot);
  real rssoil = fsw * fst * T->mrp[v] * MAXF(0.f, s->fastcp * 1.E-3f) * 12.E-6f;
  real stablc = 0.1f * rssoil;
  s->fastcp = s->fastcp - (rssoil + stablc) * dt;
  s->stblcp = s->stblcp + stablc * dt;
  s->gpp = carbfx;
  s->npp = nppl + nppw + nppr;
  real autors = rsroot + rswood + rsleaf + grleaf + grroot + grwood;
  real heters = rssoil;
  s->nee = (autors + heters - s->gpp) * 44.f / 12.f;
  s->lai = MAXF(s->lfmass * lapm, laimin);
  s->sai = MAXF(s->stmass * sapm, xsamin);
  (void)npps; (void)rsstem; (void)grstem;
}

/* NOAHMP_SFLX lsm:518-947 (+ ATM lsm:949-1007, ERROR lsm:1106-1228) */
void nmp_sflx(nmp_ctx* c, nmp_column* s) {
  const noahmp_tables* T = c->T;
  const nmp_parm* P = &c->P;
  nmp_work w;
  int ns = c->nsoil;
  memset(&w, 0, sizeof(w));
  s->nee = 0.f; s->npp = 0.f; s->gpp = 0.f;
  /* ATM */
  {
    real pair = s->sfcprs;
    w.thair = s->sfctmp * powf(s->sfcprs / pair, RAIR / CPAIR);
    w.qair = s->q2;
    w.eair = w.qair * s->sfcprs / (0.622f + 0.378f * w.qair);
    w.rhoair = (s->sfcprs - 0.378f * w.eair) / (RAIR * s->sfctmp);
    w.qprecc = 0.10f * s->prcp;
    w.qprecl = 0.90f * s->prcp;
    if (s->cosz <= 0.f) w.swdown = 0.f; else w.swdown = s->soldn;
    w.solad[0] = w.swdown * 0.7f * 0.5f;
    w.solad[1] = w.swdown * 0.7f * 0.5f;
    w.solai[0] = w.swdown * 0.3f * 0.5f;
    w.solai[1] = w.swdown * 0.3f * 0.5f;
  }
  for (int iz = s->isnow + 1; iz <= ns; iz++) {
    if (iz == s->isnow + 1) w.dzsnso[L(iz)] = -s->zsnso[L(iz)];
    else w.dzsnso[L(iz)] = s->zsnso[L(iz - 1)] - s->zsnso[L(iz)];
  }
  w.troot = 0.f;
  for (int iz = 1; iz <= P->nroot; iz++)
    w.troot = w.troot + s->stc[L(iz)] * w.dzsnso[L(iz)] / (-c->zsoil[L(P->nroot)]);
  real beg_wb = 0.f;
  if (s->ist == 1) {
    beg_wb = s->canliq + s->canice + s->sneqv + s->wa;
    for (int iz = 1; iz <= ns; iz++) beg_wb = beg_wb + s->smc[L(iz)] * w.dzsnso[L(iz)] * 1000.f;
  }
  phenology(c, s, &w);
  if (c->O.dveg == 1) {
    s->fveg = s->shdfac;
    if (s->fveg <= 0.01f) s->fveg = 0.01f;
  } else if (c->O.dveg == 2 || c->O.dveg == 3) {
    s->fveg = 1.f - expf(-0.52f * (s->lai + s->sai));
    if (s->fveg <= 0.01f) s->fveg = 0.01f;
  } else if (c->O.dveg == 4 || c->O.dveg == 5) {
    s->fveg = s->shdmax;
    if (s->fveg <= 0.01f) s->fveg = 0.01f;
  } else {
    if (!c->err) c->err = NOAHMP_ERR_DVEG_UNKNOWN;
    return;
  }
  if (c->vegtyp == c->isurban || c->vegtyp == T->isbarren) s->fveg = 0.0f;
  if (w.elai + w.esai == 0.0f) s->fveg = 0.0f;

  nmp_energy(c, s, &w);
  if (c->err) return;

  for (int iz = 1; iz <= ns; iz++) w.sice[L(iz)] = MAXF(0.0f, s->smc[L(iz)] - s->sh2o[L(iz)]);
  s->sneqvo = s->sneqv;
  real qvap = MAXF(s->fgev / w.latheag, 0.f);
  real qdew = fabsf(MINF(s->fgev / w.latheag, 0.f));
  s->edir = qvap - qdew;

  nmp_water(c, s, &w, qvap, qdew);

  if (c->O.dveg == 2 || c->O.dveg == 5) nmp_carbon(c, s, &w);

  /* ERROR lsm:1106-1228 */
  {
    real errsw = w.swdown - (s->fsa + s->fsr);
    if (fabsf(errsw) > 0.01f) { if (!c->err) c->err = NOAHMP_ERR_SW_BALANCE; return; }
    real erreng = s->sav + s->sag - (s->fira + s->fsh + s->fcev + s->fgev + s->fctr + s->ssoil);
    if (fabsf(erreng) > 0.01f) { if (!c->err) c->err = NOAHMP_ERR_ENERGY_BALANCE; return; }
    if (s->ist == 1) {
      real end_wb = s->canliq + s->canice + s->sneqv + s->wa;
      for (int iz = 1; iz <= ns; iz++) end_wb = end_wb + s->smc[L(iz)] * w.dzsnso[L(iz)] * 1000.f;
      real errwat = end_wb - beg_wb -
                    (s->prcp - s->ecan - s->etran - s->edir - s->runsrf - s->runsub) * c->dt;
      if (fabsf(errwat) > 0.1f) { if (!c->err) c->err = NOAHMP_ERR_WATER_BALANCE; return; }
    }
  }
  real qfx = s->etran + s->ecan + s->edir;
  if (c->vegtyp == c->isurban) {
    s->qsfc = (qfx / w.rhoair * s->ch) + w.qair;
    s->q2b = s->qsfc;
  }
  if (s->snowh <= 1.E-6f || s->sneqv <= 1.E-3f) { s->snowh = 0.0f; s->sneqv = 0.0f; }
  if (w.swdown != 0.f) s->albedo = s->fsr / w.swdown;
  else s->albedo = -999.9f;
}

/* ---------------------------------------------------------------------------------------------
 * noahmplsm column loop, drv:376-840.  Arrays are host pointers in Fortran (i,k,j) layout. */
static noahmp_tables g_tables;
static int g_have_tables = 0;

const noahmp_tables* nmp_oracle_tables(void) { return g_have_tables ? &g_tables : 0; }

int nmp_oracle_set_tables(const noahmp_tables* t) {
  g_tables = *t;
  g_have_tables = 1;
  return 0;
}

#define A2(f) a->f[ij]
#define A3(f, k, nk) a->f[((size_t)jj * (nk) + (k)) * ni + ii]

int nmp_oracle_step(const noahmp_step_args* a, noahmp_status* st) {
  const real UNDEF = -1.E36f, UNDEF2 = 0.0f;
  int ni = a->ime - a->ims + 1;
  int ns = a->nsoil, nka = a->kme - a->kms + 1, k1 = 1 - a->kms;   /* level 1 of (kms:kme) */
  memset(st, 0, sizeof(*st));
  if (!g_have_tables) return -1;
  if (ns != NOAHMP_NSOIL) { st->code = NOAHMP_ERR_NSOIL_UNSUPPORTED; return st->code; }
  if (a->iopt_sfc == 3 || a->iopt_sfc == 4) { st->code = NOAHMP_ERR_OPT_SFC_UNSUPPORTED; return st->code; }
  nmp_ctx c;
  memset(&c, 0, sizeof(c));
  c.T = &g_tables;
  c.O.dveg = a->idveg; c.O.opt_crs = a->iopt_crs; c.O.opt_btr = a->iopt_btr; c.O.opt_run = a->iopt_run;
  c.O.opt_sfc = a->iopt_sfc; c.O.opt_frz = a->iopt_frz; c.O.opt_inf = a->iopt_inf;
  c.O.opt_rad = a->iopt_rad; c.O.opt_alb = a->iopt_alb; c.O.opt_snf = a->iopt_snf;
  c.O.opt_tbot = a->iopt_tbot; c.O.opt_stc = a->iopt_stc;
  c.dt = a->dt; c.nsoil = ns; c.nsnow = NOAHMP_NSNOW; c.isurban = a->isurban;
  int yearlen = 365;                                           /* drv:381-390 */
  if (a->yr % 4 == 0) { yearlen = 366; if (a->yr % 100 == 0) { yearlen = 365; if (a->yr % 400 == 0) yearlen = 366; } }
  c.zsoil[L(1)] = -a->dzs[0];
  for (int k = 2; k <= ns; k++) c.zsoil[L(k)] = -a->dzs[k - 1] + c.zsoil[L(k - 1)];

  for (int j = a->jts; j <= a->jte; j++) {
    int jj = j - a->jms;
    if (a->itimestep == 1) {                                   /* drv:399-419 */
      for (int i = a->its; i <= a->ite; i++) {
        int ii = i - a->ims; size_t ij = (size_t)jj * ni + ii;
        if ((A2(xland) - 1.5f) >= 0.f) {
          A2(smstav) = 1.0f; A2(smstot) = 1.0f;
          for (int k = 0; k < ns; k++) { A3(smois, k, ns) = 1.0f; A3(tslb, k, ns) = 273.16f; }
        } else if (A2(xice) == 1.f) {
          A2(smstav) = 1.0f; A2(smstot) = 1.0f;
          for (int k = 0; k < ns; k++) A3(smois, k, ns) = 1.0f;
        }
      }
    }
    for (int i = a->its; i <= a->ite; i++) {
      int ii = i - a->ims; size_t ij = (size_t)jj * ni + ii;
      int ice;
      if (A2(xice) >= a->xice_thres) ice = 1;
      else if (A2(ivgtyp) == a->isice) ice = -1;
      else ice = 0;
      if ((A2(xland) - 1.5f) >= 0.f) { st->n_skipped++; continue; }
      if (ice == 1) {
        for (int k = 0; k < ns; k++) A3(sh2o, k, ns) = 1.0f;
        A2(xlaixy) = 0.01f;
        st->n_skipped++;
        continue;
      }
      nmp_column s;
      memset(&s, 0, sizeof(s));
      s.cosz = A2(coszin); s.lat = A2(xlatin);
      s.zlvl = 0.5f * A3(dz8w, k1, nka);
      int vegtyp = A2(ivgtyp), soiltyp = A2(isltyp);
      s.shdfac = A2(vegfra) / 100.f;
      s.shdmax = A2(vegmax) / 100.f;
      s.tbot = A2(tmn);
      s.sfctmp = A3(t3d, k1, nka);
      s.q2 = A3(qv3d, k1, nka) / (1.0f + A3(qv3d, k1, nka));
      s.uu = A3(u_phy, k1, nka); s.vv = A3(v_phy, k1, nka);
      s.soldn = A2(swdown); s.lwdn = A2(glw);
      s.sfcprs = (A3(p8w3d, a->kts + 1 - a->kms, nka) + A3(p8w3d, a->kts - a->kms, nka)) * 0.5f;
      s.psfc = A3(p8w3d, k1, nka);
      s.prcp = A2(rainbl) / a->dt;
      s.isnow = A2(isnowxy);
      for (int k = 1; k <= ns; k++) {
        s.smc[L(k)] = A3(smois, k - 1, ns); s.sh2o[L(k)] = A3(sh2o, k - 1, ns);
        s.stc[L(k)] = A3(tslb, k - 1, ns); s.smceq[L(k)] = A3(smoiseq, k - 1, ns);
      }
      for (int k = -2; k <= 0; k++) {
        s.stc[L(k)] = A3(tsnoxy, k + 2, 3); s.snice[L(k)] = A3(snicexy, k + 2, 3);
        s.snliq[L(k)] = A3(snliqxy, k + 2, 3);
      }
      for (int k = -2; k <= ns; k++) s.zsnso[L(k)] = A3(zsnsoxy, k + 2, ns + 3);
      s.sneqv = A2(snow); s.snowh = A2(snowh); s.qsfc = A2(qsfc);
      s.tv = A2(tvxy); s.tg = A2(tgxy); s.canliq = A2(canliqxy); s.canice = A2(canicexy);
      s.eah = A2(eahxy); s.tah = A2(tahxy); s.cm = A2(cmxy); s.ch = A2(chxy); s.fwet = A2(fwetxy);
      s.sneqvo = A2(sneqvoxy); s.albold = A2(alboldxy); s.qsnow = A2(qsnowxy);
      s.wslake = A2(wslakexy); s.zwt = A2(zwtxy); s.wa = A2(waxy); s.wt = A2(wtxy);
      s.lfmass = A2(lfmassxy); s.rtmass = A2(rtmassxy); s.stmass = A2(stmassxy); s.wood = A2(woodxy);
      s.stblcp = A2(stblcpxy); s.fastcp = A2(fastcpxy); s.lai = A2(xlaixy); s.sai = A2(xsaixy);
      s.tauss = A2(taussxy); s.smcwtd = A2(smcwtdxy);
      s.rech = 0.f; s.deeprech = 0.f;
      for (int k = -2; k <= 0; k++) s.ficeold[L(k)] = 0.f;     /* drv:516-518, no zero guard */
      for (int k = s.isnow + 1; k <= 0; k++)
        s.ficeold[L(k)] = s.snice[L(k)] / (s.snice[L(k)] + s.snliq[L(k)]);
      s.co2air = 395.e-06f * s.sfcprs;
      s.o2air = 0.209f * s.sfcprs;
      s.foln = 1.0f;
      s.dz8w = A3(dz8w, k1, nka);
      s.ist = 1; s.isc = 4; s.ice = ice;
      s.yearlen = yearlen; s.julian = a->julian; s.dx = a->dx;
      if (soiltyp == 14 && A2(xice) == 0.f) soiltyp = 7;
      if (A2(ivgtyp) == a->isurban || A2(ivgtyp) == 31 || A2(ivgtyp) == 32 || A2(ivgtyp) == 33)
        vegtyp = a->isurban;
      if (vegtyp == 25) { s.shdfac = 0.0f; s.lai = 0.0f; }
      if (vegtyp == 26) { s.shdfac = 0.0f; s.lai = 0.0f; }
      if (vegtyp == 27) { s.shdfac = 0.0f; s.lai = 0.0f; }
      c.err = 0;
      c.vegtyp = vegtyp;
      nmp_redprm(&c, vegtyp, soiltyp, 1);
      real fsr = 0.f;
      if (!c.err) {
        if (ice == -1) {
          s.tbot = MINF(s.tbot, 263.15f);
          nmp_glacier_column(&c, &s, &fsr);
          st->n_glacier++;
          if (!c.err) {
            s.fsno = 1.0f; s.tv = UNDEF; s.tgb = s.tg; s.canice = UNDEF2; s.canliq = UNDEF2;
            s.eah = UNDEF; s.tah = UNDEF; s.fwet = UNDEF2; s.wslake = UNDEF2; s.zwt = UNDEF;
            s.wa = UNDEF; s.wt = UNDEF; s.lfmass = UNDEF2; s.rtmass = UNDEF2; s.stmass = UNDEF2;
            s.wood = UNDEF2; s.stblcp = UNDEF; s.fastcp = UNDEF; s.lai = UNDEF2; s.sai = UNDEF2;
            s.t2mv = UNDEF; s.q2v = UNDEF; s.nee = UNDEF2; s.gpp = UNDEF2; s.npp = UNDEF2;
            s.fveg = 0.0f; s.ecan = UNDEF2; s.etran = UNDEF2; s.apar = UNDEF2; s.psn = UNDEF2;
            s.sav = UNDEF2; s.rssun = UNDEF; s.rssha = UNDEF; s.bgap = UNDEF; s.wgap = UNDEF;
            s.tgv = UNDEF; s.chv = UNDEF; s.chb = s.ch; s.irc = UNDEF; s.irg = UNDEF; s.shc = UNDEF;
            s.shg = UNDEF; s.evg = UNDEF; s.ghv = UNDEF; s.irb = s.fira; s.shb = s.fsh;
            s.evb = s.fgev; s.ghb = s.ssoil; s.tr = UNDEF2; s.evc = UNDEF2; s.chleaf = UNDEF;
            s.chuc = UNDEF; s.chv2 = UNDEF; s.fcev = UNDEF2; s.fctr = UNDEF2;
            A2(qfx) = s.edir;
            A2(lh) = s.fgev;
          }
        } else {
          nmp_sflx(&c, &s);
          st->n_land++;
          if (!c.err) {
            A2(qfx) = s.ecan + s.edir + s.etran;
            A2(lh) = s.fcev + s.fgev + s.fctr;
          }
        }
      }
      if (c.err) {
        if (!st->code) { st->code = c.err; st->i = i; st->j = j; }
        continue;                                               /* the reference STOPs here */
      }
      /* scatter, drv:728-835 */
      A2(tsk) = s.trad; A2(hfx) = s.fsh; A2(grdflx) = s.ssoil;
      A2(smstav) = 0.0f; A2(smstot) = 0.0f;
      A2(sfcrunoff) = A2(sfcrunoff) + s.runsrf * a->dt;
      A2(udrunoff) = A2(udrunoff) + s.runsub * a->dt;
      if (s.albedo > -999) A2(albedo) = s.albedo;
      A2(snowc) = s.fsno;
      for (int k = 1; k <= ns; k++) {
        A3(smois, k - 1, ns) = s.smc[L(k)]; A3(sh2o, k - 1, ns) = s.sh2o[L(k)];
        A3(tslb, k - 1, ns) = s.stc[L(k)];
      }
      A2(snow) = s.sneqv; A2(snowh) = s.snowh;
      A2(canwat) = s.canliq + s.canice;
      A2(acsnow) = A2(acsnow) + s.prcp * s.fpice;               /* no *DT in the reference, drv:751 */
      A2(acsnom) = A2(acsnom) + s.qsnbot * a->dt + s.ponding + s.ponding1 + s.ponding2;
      A2(emiss) = s.emissi; A2(qsfc) = s.qsfc;
      A2(isnowxy) = s.isnow; A2(tvxy) = s.tv; A2(tgxy) = s.tg; A2(canliqxy) = s.canliq;
      A2(canicexy) = s.canice; A2(eahxy) = s.eah; A2(tahxy) = s.tah; A2(cmxy) = s.cm; A2(chxy) = s.ch;
      A2(fwetxy) = s.fwet; A2(sneqvoxy) = s.sneqvo; A2(alboldxy) = s.albold; A2(qsnowxy) = s.qsnow;
      A2(wslakexy) = s.wslake; A2(zwtxy) = s.zwt; A2(waxy) = s.wa; A2(wtxy) = s.wt;
      for (int k = -2; k <= 0; k++) {
        A3(tsnoxy, k + 2, 3) = s.stc[L(k)]; A3(snicexy, k + 2, 3) = s.snice[L(k)];
        A3(snliqxy, k + 2, 3) = s.snliq[L(k)];
      }
      for (int k = -2; k <= ns; k++) A3(zsnsoxy, k + 2, ns + 3) = s.zsnso[L(k)];
      A2(lfmassxy) = s.lfmass; A2(rtmassxy) = s.rtmass; A2(stmassxy) = s.stmass; A2(woodxy) = s.wood;
      A2(stblcpxy) = s.stblcp; A2(fastcpxy) = s.fastcp; A2(xlaixy) = s.lai; A2(xsaixy) = s.sai;
      A2(taussxy) = s.tauss;
      A2(t2mvxy) = s.t2mv; A2(t2mbxy) = s.t2mb;
      A2(q2mvxy) = s.q2v / (1.0f - s.q2v); A2(q2mbxy) = s.q2b / (1.0f - s.q2b);
      A2(tradxy) = s.trad; A2(neexy) = s.nee; A2(gppxy) = s.gpp; A2(nppxy) = s.npp;
      A2(fvegxy) = s.fveg; A2(runsfxy) = s.runsrf; A2(runsbxy) = s.runsub; A2(ecanxy) = s.ecan;
      A2(edirxy) = s.edir; A2(etranxy) = s.etran; A2(fsaxy) = s.fsa; A2(firaxy) = s.fira;
      A2(aparxy) = s.apar; A2(psnxy) = s.psn; A2(savxy) = s.sav; A2(sagxy) = s.sag;
      A2(rssunxy) = s.rssun; A2(rsshaxy) = s.rssha; A2(bgapxy) = s.bgap; A2(wgapxy) = s.wgap;
      A2(tgvxy) = s.tgv; A2(tgbxy) = s.tgb; A2(chvxy) = s.chv; A2(chbxy) = s.chb;
      A2(ircxy) = s.irc; A2(irgxy) = s.irg; A2(shcxy) = s.shc; A2(shgxy) = s.shg; A2(evgxy) = s.evg;
      A2(ghvxy) = s.ghv; A2(irbxy) = s.irb; A2(shbxy) = s.shb; A2(evbxy) = s.evb; A2(ghbxy) = s.ghb;
      A2(trxy) = s.tr; A2(evcxy) = s.evc; A2(chleafxy) = s.chleaf; A2(chucxy) = s.chuc;
      A2(chv2xy) = s.chv2; A2(chb2xy) = s.chb2;
      A2(rechxy) = A2(rechxy) + s.rech * 1.E3f;
      A2(deeprechxy) = A2(deeprechxy) + s.deeprech;
      A2(smcwtdxy) = s.smcwtd;
    }
  }
  return st->code;
}

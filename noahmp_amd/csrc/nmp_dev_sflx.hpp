// Noah-MP column engine for MI355X -- NOAHMP_SFLX orchestration, REDPRM, PHENOLOGY, CARBON, ERROR.
// Reference: lsm:518-1228, lsm:8723-9104, lsm:9202-9349.
#pragma once
#include "nmp_dev_energy.hpp"
#include "nmp_dev_water.hpp"

namespace nmp {

// the soil half of REDPRM (lsm:9282-9300) for soil type st+1 with or without the urban override
NMP_DEV void redprm_soil(const noahmp_tables* T, float csoil_data, int st, bool urban, Parm& P) {
  P.st = st; P.u = urban ? 1 : 0;
  P.csoil = csoil_data;
  P.bexp = T->bb[st];
  P.psisat = T->satpsi[st];
  P.quartz = T->qtz[st];
  P.smcmax = T->maxsmc[st];
  P.smcref = T->refsmc[st];
  P.smcwlt = T->wltsmc[st];
  if (urban) { P.smcmax = 0.45f; P.smcref = 0.42f; P.smcwlt = 0.40f; P.csoil = 3.E6f; }
}
// KDT and FRZX (lsm:9316-9322)
NMP_DEV float redprm_kdt(const noahmp_tables* T, float dksat) { return T->refkdt_data * dksat / T->refdk_data; }
NMP_DEV float redprm_frzx(const noahmp_tables* T, const Parm& P) { return T->frzk_data * ((P.smcmax / P.smcref) * (0.412f / 0.468f)); }

// REDPRM lsm:9202-9349: table gather into per-thread registers (the reference writes module globals)
NMP_DEV void redprm(const Ctx& c, Col& s, Parm& P, int vegtyp, int soiltyp) {
  const noahmp_tables* T = c.T;
  if (soiltyp > c.ts.slcats || soiltyp < 1) { raise(s, NOAHMP_ERR_SOILTYP_RANGE); soiltyp = 1; }
  if (vegtyp > c.ts.lucats || vegtyp < 1) { raise(s, NOAHMP_ERR_VEGTYP_RANGE); vegtyp = 1; }
  const int st = soiltyp - 1, vt = vegtyp - 1;
  // PHENOLOGY's month interpolation (lsm:1054-1071) first: its indices need no table, and the rows they select then join the ONE batch
  // of gathers below (evaluated between the gathers, the division and the FMOD split the batch into dependent memory round trips)
  int it1, it2;
  {
    float day;
    if (s.lat >= 0.f) day = s.julian;
    else day = fmodf(s.julian + (0.5f * s.yearlen), (float)s.yearlen);
    float t = 12.f * day / (float)s.yearlen;
    it1 = (int)(t + 0.5f);                                  // REAL -> INTEGER truncation (lsm:1063)
    it2 = it1 + 1;
    P.ph_wt1 = (it1 + 0.5f) - t;
    if (it1 < 1) it1 = 12;
    if (it2 > 12) it2 = 1;
  }
  // ---- the gathers: table rows of the vegetation / soil type, per-type derived constants, the two months' LAI / SAI
  redprm_soil(T, c.ts.csoil, st, vegtyp == c.isurban, P);
  float rsmin = T->rstbl[vt];
  P.rgl = T->rgltbl[vt];
  P.hs = T->hstbl[vt];
  int nroot = T->nrotbl[vt];
  const Derived* D = c.D;
  P.thks_pow = D->thks_pow[P.u][st]; P.thkdry = D->thkdry[P.u][st]; P.d_rsurf = D->d_rsurf[P.u][st];
  P.chil = D->chil[vt]; P.phi1 = D->phi1[vt]; P.phi2 = D->phi2[vt]; P.avmu = D->avmu[vt];
  P.lai1 = T->laim[it1 - 1][vt]; P.lai2 = T->laim[it2 - 1][vt];
  P.sai1 = T->saim[it1 - 1][vt]; P.sai2 = T->saim[it2 - 1][vt];
  P.hvt = T->hvt[vt]; P.hvb = T->hvb[vt]; P.tmin = T->tmin[vt];
  P.z0mvt = T->z0mvt[vt]; P.cwpvt = T->cwpvt[vt]; P.dleaf = T->dleaf[vt];
  // ---- what depends on gathered values comes last
  P.zbot = c.ts.zbot;
  P.czil = c.ts.czil;
  P.topt = c.ts.topt;
  P.rsmax = c.ts.rsmax;
  if (vegtyp == c.isurban) rsmin = 400.0f;
  P.rsmin = rsmin;
  if (nroot > NSOIL) { raise(s, NOAHMP_ERR_NROOT_GT_NSOIL); nroot = NSOIL; }
  P.nroot = nroot;
}

// The REDPRM outputs only the WATER phase reads (DKSAT, DWSAT, KDT, SLOPE, FRZX; lsm:9286-9287, 9316-9322): looked up when
// that phase starts, so that they do not sit in registers through ENERGY.  `soiltyp` as REDPRM validated it.
NMP_DEV void redprm_water(const Ctx& c, Parm& P, int soiltyp, int vegtyp) {
  const noahmp_tables* T = c.T;
  P.ch2op = T->ch2op[vegtyp - 1];
  if (soiltyp > c.ts.slcats || soiltyp < 1) soiltyp = 1;
  const int st = soiltyp - 1;
  P.dksat = T->satdk[st];
  P.dwsat = T->satdw[st];
  P.kdt = c.D->kdt[st];                               // = redprm_kdt(T, DKSAT), evaluated per soil type by derive_tables
  P.slope = c.ts.slope0;                              // SLOPE_DATA(SLOPETYP = 1) (drv:525)
  P.frzx = c.D->frzx[P.u][st];                        // = redprm_frzx(T, P)
}

#ifndef __HIPCC_RTC__
// HOST: the per-type constants of `Derived`, by the functions the device code ran per column before round 3
inline void derive_tables(const noahmp_tables& T, Derived& D) {
  memset(&D, 0, sizeof D);
  // only the rows the tables define (SLCATS / LUCATS of SOILPARM.TBL / MPTABLE.TBL): the rows behind them are zero, and evaluating
  // them would raise FE_DIVBYZERO / FE_INVALID in the caller's process (a Fortran host built with FP traps)
  const int nst = T.slcats < NSLT ? T.slcats : NSLT, nvt = T.lucats < NVEGT ? T.lucats : NVEGT;
  for (int st = 0; st < nst; st++) {
    for (int u = 0; u < 2; u++) {
      Parm P = {};
      redprm_soil(&T, T.csoil_data, st, u == 1, P);
      D.thks_pow[u][st] = tdfcnd_thks_pow(P);
      D.thkdry[u][st] = tdfcnd_thkdry(P);
      D.d_rsurf[u][st] = rsurf_dry_layer(P);
      D.frzx[u][st] = redprm_frzx(&T, P);
    }
    D.kdt[st] = redprm_kdt(&T, T.satdk[st]);
  }
  for (int v = 0; v < nvt; v++) leaf_orientation(T.xl[v], D.chil[v], D.phi1[v], D.phi2[v], D.avmu[v]);
}
#endif

// PHENOLOGY lsm:1010-1104
NMP_DEV void phenology(const Ctx& c, const Parm& P, Col& s) {
  const noahmp_tables* T = c.T;
  if (c.O.dveg == 1 || c.O.dveg == 3 || c.O.dveg == 4) {      // months, weight and table rows: REDPRM fetched them (same values)
    const float wt1 = P.ph_wt1;
    const float wt2 = 1.f - wt1;
    s.lai = wt1 * P.lai1 + wt2 * P.lai2;
    s.sai = wt1 * P.sai1 + wt2 * P.sai2;
  }
  if (s.sai < 0.01f) s.sai = 0.0f;
  if (s.lai < 0.05f || s.sai == 0.0f) s.lai = 0.0f;
  if ((s.vegtyp == c.ts.iswater) || (s.vegtyp == c.ts.isbarren) || (s.vegtyp == c.ts.issnow) ||
      (s.vegtyp == c.isurban)) {
    s.lai = 0.f; s.sai = 0.f;
  }
  const float hvt = P.hvt, hvb = P.hvb;
  float db = nmp_min(nmp_max(s.snowh - hvb, 0.f), hvt - hvb);
  float fb = db / nmp_max(1.E-06f, hvt - hvb);
  if (hvt > 0.f && hvt <= 1.0f) {
    float snowhc = hvt * nmp_expf(-s.snowh / 0.2f);
    fb = nmp_min(s.snowh, snowhc) / snowhc;
  }
  s.elai = s.lai * (1.f - fb);
  s.esai = s.sai * (1.f - fb);
  if (s.esai < 0.01f) s.esai = 0.0f;
  if (s.elai < 0.05f || s.esai == 0.0f) s.elai = 0.0f;
  s.igs = (s.tv > P.tmin) ? 1.f : 0.f;
  s.htop = hvt;
}

// CARBON lsm:8723-8835 + CO2FLUX lsm:8837-9104 (DVEG 2 / 5 only)
template <class A>
NMP_DEV void carbon_veg(const Ctx& c, const Parm& P, Col& s, const Lay<A>& y);

template <class A>
NMP_DEV void carbon(const Ctx& c, const Parm& P, Col& s, const Lay<A>& y) {
  const noahmp_tables* T = c.T;
  const bool novegc = (s.vegtyp == c.ts.iswater) || (s.vegtyp == c.ts.isbarren) || (s.vegtyp == c.ts.issnow) ||
                      (s.vegtyp == c.isurban);
  if (novegc) {                                       // lsm:8792-8810
    s.lai = 0.f; s.sai = 0.f; s.gpp = 0.f; s.npp = 0.f; s.nee = 0.f;
    s.lfmass = 0.f; s.rtmass = 0.f; s.stmass = 0.f; s.wood = 0.f; s.stblcp = 0.f; s.fastcp = 0.f;
  } else {
    carbon_veg(c, P, s, y);
  }
}

template <class A>
NMP_DEV void carbon_veg(const Ctx& c, const Parm& P, Col& s, const Lay<A>& y) {
  const noahmp_tables* T = c.T;
  const int v = s.vegtyp - 1;
  const float dt = c.dt;
  float lapm = T->sla[v] / 1000.f;
  float wstres = 1.f - s.btran;
  float wroot = 0.f;
  const float zroot = -pick_layer(c.zsoil, P.nroot);
#pragma unroll
  for (int j = 1; j <= NSOIL; j++)
    if (j <= P.nroot) wroot = wroot + y.smc[L(j)] / P.smcmax * y.dzsnso[L(j)] / zroot;
  const float rtovrc = 2.0E-8f, rswoodc = 3.0E-10f, bf = 0.90f, wstrc = 100.0f, laimin = 0.05f,
              xsamin = 0.01f;
  float sapm = 3.f * 0.001f;
  float lfmsmn = laimin / lapm, stmsmn = xsamin / sapm;
  float rf = (s.igs == 0.f) ? 0.5f : 1.0f;
  float tv = s.tv;
  float fnf = nmp_min(s.foln / nmp_max(1.E-06f, T->folnmx[v]), 1.0f);
  float tf = nmp_powf(T->arm[v], (tv - 298.16f) / 10.f);
  float resp = T->rmf25[v] * tf * fnf * s.lai * rf * (1.f - wstres);
  float rsleaf = nmp_min(s.lfmass / dt, resp * 12.e-6f);
  float rsroot = T->rmr25[v] * (s.rtmass * 1E-3f) * tf * rf * 12.e-6f;
  float rsstem = T->rms25[v] * (s.stmass * 1E-3f) * tf * rf * 12.e-6f;
  float rswood = rswoodc * nmp_expf(0.08f * (tv - 298.16f)) * s.wood * T->wdpool[v];
  float carbfx = s.psn * 12.e-6f;
  float leafpt = nmp_expf(0.01f * (1.f - nmp_expf(0.75f * s.lai)) * s.lai);
  if (s.vegtyp == c.ts.eblforest) leafpt = nmp_expf(0.01f * (1.f - nmp_expf(0.50f * s.lai)) * s.lai);
  float nonlef = 1.0f - leafpt;
  float stempt = s.lai / 10.0f;
  leafpt = leafpt - stempt;
  float woodf;
  if (s.wood > 0) woodf = (1.f - nmp_expf(-bf * (T->wrrat[v] * s.rtmass / s.wood)) / bf) * T->wdpool[v];
  else woodf = 0.f;
  float rootpt = nonlef * (1.f - woodf);
  float woodpt = nonlef * woodf;
  float lftovr = T->ltovrc[v] * 1.E-6f * s.lfmass;
  float sttovr = T->ltovrc[v] * 1.E-6f * s.stmass;
  float rttovr = rtovrc * s.rtmass;
  float wdtovr = 9.5E-10f * s.wood;
  float sc = nmp_expf(-0.3f * nmp_max(0.f, tv - T->tdlef[v])) * (s.lfmass / 120.f);
  float sd = nmp_expf((wstres - 1.f) * wstrc);
  float dielf = s.lfmass * 1.E-6f * (T->dilefw[v] * sd + T->dilefc[v] * sc);
  float diest = s.stmass * 1.E-6f * (T->dilefw[v] * sd + T->dilefc[v] * sc);
  float grleaf = nmp_max(0.0f, T->fragr[v] * (leafpt * carbfx - rsleaf));
  float grstem = nmp_max(0.0f, T->fragr[v] * (stempt * carbfx - rsstem));
  float grroot = nmp_max(0.0f, T->fragr[v] * (rootpt * carbfx - rsroot));
  float grwood = nmp_max(0.0f, T->fragr[v] * (woodpt * carbfx - rswood));
  float addnpplf = nmp_max(0.f, leafpt * carbfx - grleaf - rsleaf);
  float addnppst = nmp_max(0.f, stempt * carbfx - grstem - rsstem);
  if (tv < T->tmin[v]) { addnpplf = 0.f; addnppst = 0.f; }
  float lfdel = (s.lfmass - lfmsmn) / dt;
  float stdel = (s.stmass - stmsmn) / dt;
  dielf = nmp_min(dielf, lfdel + addnpplf - lftovr);
  diest = nmp_min(diest, stdel + addnppst - sttovr);
  float nppl = nmp_max(addnpplf, -lfdel);
  float npps = nmp_max(addnppst, -stdel);
  float nppr = rootpt * carbfx - rsroot - grroot;
  float nppw = woodpt * carbfx - rswood - grwood;
  s.lfmass = s.lfmass + (nppl - lftovr - dielf) * dt;
  s.stmass = s.stmass + (npps - sttovr - diest) * dt;
  s.rtmass = s.rtmass + (nppr - rttovr) * dt;
  if (s.rtmass < 0.0f) { rttovr = nppr; s.rtmass = 0.0f; }
  s.wood = (s.wood + (nppw - wdtovr) * dt) * T->wdpool[v];
  s.fastcp = s.fastcp + (rttovr + lftovr + sttovr + wdtovr + dielf) * dt;
  float fst = nmp_powf(2.0f, (y.stc[L(1)] - 283.16f) / 10.f);
  float fsw = wroot / (0.20f + wroot) * 0.23f / (0.23f + wroot);
  float rssoil = fsw * fst * T->mrp[v] * nmp_max(0.f, s.fastcp * 1.E-3f) * 12.E-6f;
  float stablc = 0.1f * rssoil;
  s.fastcp = s.fastcp - (rssoil + stablc) * dt;
  s.stblcp = s.stblcp + stablc * dt;
  s.gpp = carbfx;
  s.npp = nppl + nppw + nppr;
  float autors = rsroot + rswood + rsleaf + grleaf + grroot + grwood;
  s.nee = (autors + rssoil - s.gpp) * 44.f / 12.f;
  s.lai = nmp_max(s.lfmass * lapm, laimin);
  s.sai = nmp_max(s.stmass * sapm, xsamin);
}

// NOAHMP_SFLX lsm:518-947 (with ATM lsm:949-1007 and ERROR lsm:1106-1228)
// Split in two phases so that the caller can store the energy-phase outputs before the water phase.
// Called by ALL threads of the workgroup (`live` = this thread carries a land column): see energy().
template <class A, class Runner, class Hook>
NMP_DEV void sflx_energy(const Ctx& c, const Parm& P, Col& s, const Lay<A>& y, float& beg_wb_out, const bool live,
                         Runner& runner, Hook& before_soil_heat) {
  const noahmp_tables* T = c.T;
  float beg_wb = 0.f;
  if (live) {
  s.nee = 0.f; s.npp = 0.f; s.gpp = 0.f;
  // ATM
  s.thair = s.sfctmp;                                  // SFCTMP*(SFCPRS/PAIR)**(RAIR/CPAIR), PAIR==SFCPRS
  s.qair = s.q2;
  s.eair = s.qair * s.sfcprs / (0.622f + 0.378f * s.qair);
  s.rhoair = (s.sfcprs - 0.378f * s.eair) / (RAIR * s.sfctmp);
  s.qprecc = 0.10f * s.prcp;
  s.qprecl = 0.90f * s.prcp;
  s.swdown = (s.cosz <= 0.f) ? 0.f : s.soldn;
  s.solad0 = s.swdown * 0.7f * 0.5f; s.solad1 = s.swdown * 0.7f * 0.5f;
  s.solai0 = s.swdown * 0.3f * 0.5f; s.solai1 = s.swdown * 0.3f * 0.5f;
  // layer thicknesses (lsm:788-794)
  {
    float prev = 0.f;
#pragma unroll
    for (int iz = -2; iz <= NSOIL; iz++) {
      if (iz > s.isnow) {
        float z = y.zsnso[L(iz)];
        y.dzsnso[L(iz)] = (iz == s.isnow + 1) ? -z : (prev - z);
        prev = z;
      }
    }
  }
  beg_wb = s.canliq + s.canice + s.sneqv + s.wa;
#pragma unroll
  for (int iz = 1; iz <= NSOIL; iz++) beg_wb = beg_wb + y.smc[L(iz)] * y.dzsnso[L(iz)] * 1000.f;
  phenology(c, P, s);
  if (c.O.dveg == 1) {
    s.fveg = s.shdfac;
    if (s.fveg <= 0.01f) s.fveg = 0.01f;
  } else if (c.O.dveg == 2 || c.O.dveg == 3) {
    s.fveg = 1.f - nmp_expf(-0.52f * (s.lai + s.sai));
    if (s.fveg <= 0.01f) s.fveg = 0.01f;
  } else if (c.O.dveg == 4 || c.O.dveg == 5) {
    s.fveg = s.shdmax;
    if (s.fveg <= 0.01f) s.fveg = 0.01f;
  } else {
    raise(s, NOAHMP_ERR_DVEG_UNKNOWN);
    s.fveg = 0.01f;
  }
  if (s.vegtyp == c.isurban || s.vegtyp == c.ts.isbarren) s.fveg = 0.0f;
  if (s.elai + s.esai == 0.0f) s.fveg = 0.0f;
  }  // live

  energy(c, P, s, y, live, runner, before_soil_heat);
  if (!live) return;

  s.sneqvo = s.sneqv;
  beg_wb_out = beg_wb;
  // the SW and energy parts of ERROR (lsm:1164-1197) only involve ENERGY outputs: evaluate them here
  {
    float errsw = s.swdown - (s.fsa + s.fsr);
    if (fabsf(errsw) > 0.01f) raise(s, NOAHMP_ERR_SW_BALANCE);
    float erreng = s.sav + s.sag - (s.fira + s.fsh + s.fcev + s.fgev + s.fctr + s.ssoil);
    if (fabsf(erreng) > 0.01f) raise(s, NOAHMP_ERR_ENERGY_BALANCE);
  }
  s.albedo = (s.swdown != 0.f) ? (s.fsr / s.swdown) : -999.9f;                      // lsm:940-944
}

template <class A>
NMP_DEV void sflx_water(const Ctx& c, const Parm& P, Col& s, const Lay<A>& y, float beg_wb) {
#pragma unroll
  for (int iz = 1; iz <= NSOIL; iz++) y.sice[L(iz)] = nmp_max(0.0f, y.smc[L(iz)] - y.sh2o[L(iz)]);
  float qvap = nmp_max(s.fgev / s.latheag, 0.f);
  float qdew = fabsf(nmp_min(s.fgev / s.latheag, 0.f));
  s.edir = qvap - qdew;

  NMP_TIC(12);   // water preamble
  water(c, P, s, y, qvap, qdew);
  NMP_TIC(13);   // water

  if (c.O.dveg == 2 || c.O.dveg == 5) carbon(c, P, s, y);
  NMP_TIC(14);   // carbon

  // water part of ERROR (lsm:1199-1222): the reference STOPs; here the column raises its status word
  {
    float end_wb = s.canliq + s.canice + s.sneqv + s.wa;
#pragma unroll
    for (int iz = 1; iz <= NSOIL; iz++) end_wb = end_wb + y.smc[L(iz)] * y.dzsnso[L(iz)] * 1000.f;
    float errwat = end_wb - beg_wb - (s.prcp - s.ecan - s.etran - s.edir - s.runsrf - s.runsub) * c.dt;
    if (fabsf(errwat) > 0.1f) raise(s, NOAHMP_ERR_WATER_BALANCE);
  }
  float qfx = s.etran + s.ecan + s.edir;
  if (s.vegtyp == c.isurban) {
    s.qsfc = (qfx / s.rhoair * s.ch) + s.qair;
    s.q2b = s.qsfc;
  }
  if (s.snowh <= 1.E-6f || s.sneqv <= 1.E-3f) { s.snowh = 0.0f; s.sneqv = 0.0f; }
}

}  // namespace nmp

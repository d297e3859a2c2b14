// Noah-MP column engine for MI355X -- energy phase device code.
// Follows ENERGY and its callees in the reference (phys/module_sf_noahmplsm.F90, "lsm"),
// statement order preserved so results track the float32 reference.
// Layer loops are written as fully unrolled, predicated loops over the 7 fixed slots so that
// temporaries with compile-time indices stay in VGPRs; only the state arrays in `Lay` are
// addressed with run-time layer indices.
#pragma once
#include "nmp_dev_common.hpp"

namespace nmp {

// ESAT lsm:5272-5321
// The callers of ESAT (lsm:3246-3250 etc.) always pick the water pair for T > 0 C and the ice pair otherwise; evaluating
// only the pair that is picked gives the same bits with half the arithmetic (the branch is wave-uniform except in
// waves that straddle 0 C).
NMP_DEV void esat_sel(float t, float& es, float& des) {
  if (t > 0.f) {
    es = 100.f * (6.107799961f + t * (4.436518521E-01f + t * (1.428945805E-02f + t * (2.650648471E-04f +
         t * (3.031240396E-06f + t * (2.034080948E-08f + t * 6.136820929E-11f))))));
    des = 100.f * (4.438099984E-01f + t * (2.857002636E-02f + t * (7.938054040E-04f + t * (1.215215065E-05f +
          t * (1.036561403E-07f + t * (3.532421810e-10f + t * -7.090244804E-13f))))));
  } else {
    es = 100.f * (6.109177956f + t * (5.034698970E-01f + t * (1.886013408E-02f + t * (4.176223716E-04f +
         t * (5.824720280E-06f + t * (4.838803174E-08f + t * 1.838826904E-10f))))));
    des = 100.f * (5.030305237E-01f + t * (3.773255020E-02f + t * (1.267995369E-03f + t * (2.477563108E-05f +
          t * (3.005693132E-07f + t * (2.158542548E-09f + t * 7.131097725E-12f))))));
  }
}
NMP_DEV float tdc(float t) { return nmp_min(50.f, nmp_max(-50.f, (t - TFRZ))); }   // lsm:3247

// TDFCND lsm:2014-2118
// `thks_pow` = THKS**(1-SMCMAX) with THKS = 7.7**QUARTZ * 2**(1-QUARTZ): per-column constants, evaluated once by
// the caller instead of once per soil layer.
NMP_DEV float tdfcnd_thks_pow(const Parm& P) {
  float thks = nmp_powf(7.7f, P.quartz) * nmp_powf(2.0f, 1.f - P.quartz);
  return nmp_powf(thks, 1.f - P.smcmax);
}
NMP_DEV float tdfcnd_thkdry(const Parm& P) {           // lsm:2100-2101: depends on SMCMAX only
  float gammd = (1.f - P.smcmax) * 2700.f;
  return (0.135f * gammd + 64.7f) / (2700.f - 0.947f * gammd);
}
// pw_ice = TKICE ** (SMCMAX - XU), pw_liq = 0.57 ** XU with XU = (SH2O / SMC) SMCMAX: evaluated by the caller for its four layers at once
NMP_DEV float tdfcnd(const Parm& P, double r_smcmax, float thks_pow, float thkdry, float smc, float sh2o, float pw_ice, float pw_liq) {
  float satratio = div_rc(smc, r_smcmax);
  float thksat = thks_pow * pw_ice * pw_liq;
  float ake;
  if ((sh2o + 0.0005f) < smc) ake = satratio;
  else ake = (satratio > 0.1f) ? (nmp_log10f(satratio) + 1.0f) : 0.0f;
  return ake * (thksat - thkdry) + thkdry;
}

// THERMOPROP lsm:1845-1954 + CSNOW lsm:1957-2011
template <class A>
NMP_DEV void thermoprop(const Ctx& c, const Parm& P, const Col& s, const Lay<A>& y, float* df,
                        float* hcpct, float* fact) {
  const int isnow = s.isnow;
#pragma unroll
  for (int iz = -2; iz <= 0; iz++) {
    if (iz > isnow) {
      float dz = y.dzsnso[L(iz)];
      float snicev = nmp_min(1.f, y.snice[L(iz)] / (dz * DENICE));
      float epore = 1.f - snicev;
      float snliqv = nmp_min(epore, y.snliq[L(iz)] / (dz * DENH2O));
      float bdsnoi = (y.snice[L(iz)] + y.snliq[L(iz)]) / dz;
      hcpct[L(iz)] = CICE * snicev + CWAT * snliqv;
      df[L(iz)] = 3.2217E-6f * pow_two(bdsnoi);     // BDSNOI**2. (lsm:2004)
    }
  }
  const bool urban = (s.vegtyp == c.isurban);
  const float thks_pow = P.thks_pow, thkdry = P.thkdry;     // per soil type (Derived, fetched by REDPRM)
  const double r_smcmax = rc64(P.smcmax);
  // TDFCND of the four soil layers: TKICE ** (SMCMAX - XU) and 0.57 ** XU have compile-time bases (their log2 folds), the eight exp2
  // look-ups form one batch
  float pw_y[2 * NSOIL], pw[2 * NSOIL], pw_b[2 * NSOIL];
  double pw_l[2 * NSOIL];
#pragma unroll
  for (int iz = 1; iz <= NSOIL; iz++) {
    const float smc = y.smc[L(iz)], sh2o = y.sh2o[L(iz)];
    const float xu = (sh2o / smc) * P.smcmax;
    pw_b[2 * iz - 2] = TKICE; pw_l[2 * iz - 2] = NMP_LOG2K(TKICE); pw_y[2 * iz - 2] = P.smcmax - xu;
    pw_b[2 * iz - 1] = 0.57f; pw_l[2 * iz - 1] = NMP_LOG2K(0.57f); pw_y[2 * iz - 1] = xu;
  }
  if (!urban) nmp_powf_constbaseN<2 * NSOIL>(pw_b, pw_l, pw_y, pw);
#pragma unroll
  for (int iz = 1; iz <= NSOIL; iz++) {
    float smc = y.smc[L(iz)], sh2o = y.sh2o[L(iz)];
    float sice = smc - sh2o;
    hcpct[L(iz)] = sh2o * CWAT + (1.0f - P.smcmax) * P.csoil + (P.smcmax - smc) * CPAIR + sice * CICE;
    df[L(iz)] = urban ? 3.24f : tdfcnd(P, r_smcmax, thks_pow, thkdry, smc, sh2o, pw[2 * iz - 2], pw[2 * iz - 1]);
  }
#pragma unroll
  for (int iz = -2; iz <= NSOIL; iz++)
    if (iz > isnow) fact[L(iz)] = c.dt / (hcpct[L(iz)] * y.dzsnso[L(iz)]);
  if (isnow == 0)
    df[L(1)] = (df[L(1)] * y.dzsnso[L(1)] + 0.35f * s.snowh) / (s.snowh + y.dzsnso[L(1)]);
  else
    df[L(1)] = (df[L(1)] * y.dzsnso[L(1)] + df[L(0)] * y.dzsnso[L(0)]) / (y.dzsnso[L(0)] + y.dzsnso[L(1)]);
}

// SNOW_AGE lsm:2547-2596
NMP_DEV void snow_age(float dt, float tg, float sneqvo, float sneqv, float& tauss, float& fage) {
  if (sneqv <= 0.0f) tauss = 0.f;
  else if (sneqv > 800.f) tauss = 0.f;
  else {
    float dela0 = 1.E-6f * dt;
    float arg = 5.E3f * (1.f / TFRZ - 1.f / tg);
    float age1, age2;
    { const float aa[2] = {arg, nmp_min(0.f, 10.f * arg)}; float ae[2]; nmp_expfN<2>(aa, ae); age1 = ae[0]; age2 = ae[1]; }
    float tage = age1 + age2 + 0.3f;
    float dela = dela0 * tage;
    float dels = nmp_max(0.0f, sneqv - sneqvo) / SWEMX;
    float sge = (tauss + dela) * (1.0f - dels);
    tauss = nmp_max(0.f, sge);
  }
  fage = tauss / (tauss + 1.f);
}

struct TwoStreamOut { float fab, fre, ftd, fti, frev, freg; };

// the leaf-orientation part of TWOSTREAM (lsm:2891-2897): a function of the vegetation type's XL only
NMP_DEV void leaf_orientation(float xl, float& chil, float& phi1, float& phi2, float& avmu) {
  chil = nmp_min(nmp_max(xl, -0.4f), 0.6f);
  if (fabsf(chil) <= 0.01f) chil = 0.01f;
  phi1 = 0.5f - 0.633f * chil - 0.330f * chil * chil;
  phi2 = 0.877f * (1.f - 2.f * phi1);
  avmu = (1.f - phi1 / phi2 * nmp_logf((phi1 + phi2) / phi1)) / phi2;
}

// TWOSTREAM lsm:2768-3016 for one band (rho,tau,albgrd,albgri,omegas of that band), ic 0=direct 1=diffuse
NMP_DEV TwoStreamOut twostream(const Ctx& c, const Parm& P, int ic, int v, float cosz, float vai, float fwet, float t,
                               float albgrd, float albgri, float rho, float tau, float omegas,
                               float fveg, float& gdir, float& bgap, float& wgap) {
  const noahmp_tables* T = c.T;
  const float PAI = 3.14159265f;
  float gap, kopen;
  if (vai == 0.0f) {
    gap = 1.0f; kopen = 1.0f;
  } else {
    gap = 0.f; kopen = 0.f;
    if (c.O.rad == 1) {
      float rc = T->rc[v];
      float denfveg = -nmp_logf(nmp_max(1.0f - fveg, 0.01f)) / (PAI * powi2(rc));
      float hd = T->hvt[v] - T->hvb[v];
      float bb = 0.5f * hd;
      float thetap = nmp_atanf(bb / rc * nmp_tanf(nmp_acosf(nmp_max(0.01f, cosz))));
      bgap = nmp_expf(-denfveg * PAI * powi2(rc) / nmp_cosf(thetap));
      float fa = vai / (1.33f * PAI * nmp_powf(rc, 3.0f) * (bb / rc) * denfveg);
      float newvai = hd * fa;
      wgap = (1.0f - bgap) * nmp_expf(-0.5f * newvai / cosz);
      gap = nmp_min(1.0f - fveg, bgap + wgap);
      kopen = 0.05f;
    }
    if (c.O.rad == 2) { gap = 0.0f; kopen = 0.0f; }
    if (c.O.rad == 3) { gap = 1.0f - fveg; kopen = 1.0f - fveg; }
  }
  float coszi = nmp_max(0.001f, cosz);
  const float chil = P.chil, phi1 = P.phi1, phi2 = P.phi2, avmu = P.avmu;   // leaf_orientation(XL), per vegetation type (Derived, fetched by REDPRM)
  gdir = phi1 + phi2 * coszi;
  float ext = gdir / coszi;
  float omegal = rho + tau;
  float tmp0 = gdir + phi2 * coszi;
  float tmp1 = phi1 * coszi;
  float asu = 0.5f * omegal * gdir / tmp0 * (1.f - tmp1 / tmp0 * nmp_logf((tmp1 + tmp0) / tmp1));
  float betadl = (1.f + avmu * ext) / (omegal * avmu * ext) * asu;
  float betail = 0.5f * (rho + tau + (rho - tau) * powi2((1.f + chil) / 2.f)) / omegal;
  float tmp2;
  if (t > TFRZ) {
    tmp0 = omegal; tmp1 = betadl; tmp2 = betail;
  } else {
    tmp0 = (1.f - fwet) * omegal + fwet * omegas;
    tmp1 = ((1.f - fwet) * omegal * betadl + fwet * omegas * c.ts.betads) / tmp0;
    tmp2 = ((1.f - fwet) * omegal * betail + fwet * omegas * c.ts.betais) / tmp0;
  }
  float omega = tmp0, betad = tmp1, betai = tmp2;
  float b = 1.f - omega + omega * betai;
  float cc = omega * betai;
  tmp0 = avmu * ext;
  float d = tmp0 * omega * betad;
  float f = tmp0 * omega * (1.f - betad);
  tmp1 = b * b - cc * cc;
  float h = sqrtf(tmp1) / avmu;
  float sigma = tmp0 * tmp0 - tmp1;
  if (fabsf(sigma) < 1.e-6f) sigma = copysignf(1.e-6f, sigma);
  float p1 = b + avmu * h, p2 = b - avmu * h, p3 = b + tmp0, p4 = b - tmp0;
  float s1, s2;
  { const float sa[2] = {-h * vai, -ext * vai}; float se[2]; nmp_expfN<2>(sa, se); s1 = se[0]; s2 = se[1]; }
  float alb = (ic == 0) ? albgrd : albgri;
  float u1 = b - cc / alb, u2 = b - cc * alb, u3 = f + cc * alb;
  tmp2 = u1 - avmu * h;
  float tmp3 = u1 + avmu * h;
  // S1, SIGMA, D1, D2 each divide three to five times: one float64 reciprocal each (rc64 / div_rc, nmp_dev_common.hpp)
  const double r_s1 = rc64(s1), r_sigma = rc64(sigma);
  float d1 = div_rc(p1 * tmp2, r_s1) - p2 * tmp3 * s1;
  float tmp4 = u2 + avmu * h;
  float tmp5 = u2 - avmu * h;
  float d2 = div_rc(tmp4, r_s1) - tmp5 * s1;
  const double r_d1 = rc64(d1), r_d2 = rc64(d2);
  float h1 = -d * p4 - cc * f;
  const float h1_sigma = div_rc(h1, r_sigma);
  float tmp6 = d - div_rc(h1 * p3, r_sigma);
  float tmp7 = (d - cc - h1_sigma * (u1 + tmp0)) * s2;
  float h2 = div_rc(div_rc(tmp6 * tmp2, r_s1) - p2 * tmp7, r_d1);
  float h3 = -div_rc(tmp6 * tmp3 * s1 - p1 * tmp7, r_d1);
  float h4 = -f * p3 - cc * d;
  float tmp8 = div_rc(h4, r_sigma);
  float tmp9 = (u3 - tmp8 * (u2 - tmp0)) * s2;
  float h5 = -div_rc(div_rc(tmp8 * tmp4, r_s1) + tmp9, r_d2);
  float h6 = div_rc(tmp8 * tmp5 * s1 + tmp9, r_d2);
  float h7 = (cc * tmp2) / (d1 * s1);
  float h8 = div_rc(-cc * tmp3 * s1, r_d1);
  float h9 = tmp4 / (d2 * s1);
  float h10 = div_rc(-tmp5 * s1, r_d2);
  TwoStreamOut o;
  if (ic == 0) {
    o.ftd = s2 * (1.0f - gap) + gap;
    o.fti = (div_rc(h4 * s2, r_sigma) + h5 * s1 + div_rc(h6, r_s1)) * (1.0f - gap);
    o.fre = (h1_sigma + h2 + h3) * (1.0f - gap) + albgrd * gap;
    o.frev = (h1_sigma + h2 + h3) * (1.0f - gap);
    o.freg = albgrd * gap;
  } else {
    o.ftd = 0.f;
    o.fti = (h9 * s1 + div_rc(h10, r_s1)) * (1.0f - kopen) + kopen;
    o.fre = (h7 + h8) * (1.0f - kopen) + albgri * kopen;
    o.frev = (h7 + h8) * (1.0f - kopen) + albgri * kopen;
    o.freg = 0.f;
  }
  o.fab = 1.f - o.fre - (1.f - albgrd) * o.ftd - (1.f - albgri) * o.fti;
  return o;
}

struct RadOut { float fsun, laisun, laisha, parsun, parsha; };

// RADIATION lsm:2120-2240 (ALBEDO lsm:2243-2423 + SURRAD lsm:2426-2544)
// the vegetation type's leaf / stem optical rows (lsm:2361-2366), requested by ENERGY before THERMOPROP for sunlit columns
struct RadP { float rhol[2], rhos[2], taul[2], taus[2]; };
NMP_DEV RadP radiation_rows(const noahmp_tables* T, int v) {
  return RadP{{T->rhol[0][v], T->rhol[1][v]}, {T->rhos[0][v], T->rhos[1][v]}, {T->taul[0][v], T->taul[1][v]}, {T->taus[0][v], T->taus[1][v]}};
}
NMP_DEV RadOut radiation(const Ctx& c, const Parm& P, Col& s, float smc1, const RadP& rp) {
  const noahmp_tables* T = c.T;
  const int v = s.vegtyp - 1;
  const float MPE = 1.E-6f;
  float albd[2] = {0, 0}, albi[2] = {0, 0}, albgrd[2] = {0, 0}, albgri[2] = {0, 0}, fabd[2] = {0, 0},
        fabi[2] = {0, 0}, ftdd[2] = {0, 0}, ftid[2] = {0, 0}, ftii[2] = {0, 0};
  float frevd[2] = {0, 0}, frevi[2] = {0, 0}, fregd[2] = {0, 0}, fregi[2] = {0, 0};
  float fsun = 0.f;
  s.bgap = 0.f; s.wgap = 0.f;
  const float vai = s.elai + s.esai;
  if (!(s.cosz <= 0.f)) {                               // lsm:2356 IF(COSZ <= 0) GOTO 100: skipped at night -- a NaN COSZ is not
    float wl = s.elai / nmp_max(vai, MPE);
    float ws = s.esai / nmp_max(vai, MPE);
    float rho[2], tau[2], albsnd[2], albsni[2];
#pragma unroll
    for (int ib = 0; ib < 2; ib++) {
      rho[ib] = nmp_max(rp.rhol[ib] * wl + rp.rhos[ib] * ws, MPE);
      tau[ib] = nmp_max(rp.taul[ib] * wl + rp.taus[ib] * ws, MPE);
    }
    float fage;
    snow_age(c.dt, s.tg, s.sneqvo, s.sneqv, s.tauss, fage);
    if (c.O.alb == 1) {                                 // SNOWALB_BATS lsm:2599-2649
      float sl = 2.0f, sl1 = 1.f / sl, sl2 = 2.f * sl;
      float cf1 = ((1.f + sl1) / (1.f + sl2 * s.cosz) - sl1);
      float fzen = nmp_max(cf1, 0.f);
      albsni[0] = 0.95f * (1.f - 0.2f * fage);
      albsni[1] = 0.65f * (1.f - 0.5f * fage);
      albsnd[0] = albsni[0] + 0.4f * fzen * (1.f - albsni[0]);
      albsnd[1] = albsni[1] + 0.4f * fzen * (1.f - albsni[1]);
    } else {                                            // SNOWALB_CLASS lsm:2652-2700
      float alb = 0.55f + (s.albold - 0.55f) * nmp_expf(-0.01f * c.dt / 3600.f);
      if (s.qsnow > 0.f) alb = alb + nmp_min(s.qsnow * c.dt, SWEMX) * (0.84f - alb) / (SWEMX);
      albsni[0] = albsni[1] = albsnd[0] = albsnd[1] = alb;
      s.albold = alb;
    }
#pragma unroll
    for (int ib = 0; ib < 2; ib++) {                    // GROUNDALB lsm:2703-2765 (IST=1 branch)
      float inc = nmp_max(0.11f - 0.40f * smc1, 0.f);
      float albsod = nmp_min(c.ts.albsat4[ib] + inc, c.ts.albdry4[ib]);          // ALBSAT / ALBDRY(ISC = 4, ib): drv:526 fixes ISC
      float albsoi = albsod;
      if (s.isc == 9) { albsod += 0.10f; albsoi += 0.10f; }
      albgrd[ib] = albsod * (1.f - s.fsno) + albsnd[ib] * s.fsno;
      albgri[ib] = albsoi * (1.f - s.fsno) + albsni[ib] * s.fsno;
    }
    float gdir = 0.f;
#pragma unroll 1
    for (int ib = 0; ib < 2; ib++) {
      // the direct (ic = 0) and the diffuse (ic = 1) call of a band share everything up to the ground albedo: unrolled, the compiler
      // evaluates the common part once (same operations, same bits)
#pragma unroll
      for (int ic = 0; ic < 2; ic++) {
        TwoStreamOut o = twostream(c, P, ic, v, s.cosz, vai, s.fwet, s.tv, albgrd[ib], albgri[ib],
                                   rho[ib], tau[ib], c.ts.omegas[ib], s.fveg, gdir, s.bgap, s.wgap);
        if (ic == 0) { fabd[ib] = o.fab; albd[ib] = o.fre; ftdd[ib] = o.ftd; ftid[ib] = o.fti;
                       frevd[ib] = o.frev; fregd[ib] = o.freg; }
        else { fabi[ib] = o.fab; albi[ib] = o.fre; ftii[ib] = o.fti; frevi[ib] = o.frev;
               fregi[ib] = o.freg; }
      }
    }
    float ext = gdir / s.cosz * sqrtf(1.f - rho[0] - tau[0]);
    fsun = (1.f - nmp_expf(-ext * vai)) / nmp_max(ext * vai, MPE);
    if (fsun < 0.01f) fsun = 0.f;
  }
  RadOut r;
  r.fsun = fsun;
  float fsha = 1.f - fsun;
  r.laisun = s.elai * fsun;
  r.laisha = s.elai * fsha;
  const float solad[2] = {s.solad0, s.solad1}, solai[2] = {s.solai0, s.solai1};
  float cad[2], cai[2];
  s.sag = 0.f; s.sav = 0.f; s.fsa = 0.f;
#pragma unroll
  for (int ib = 0; ib < 2; ib++) {
    cad[ib] = solad[ib] * fabd[ib];
    cai[ib] = solai[ib] * fabi[ib];
    s.sav = s.sav + cad[ib] + cai[ib];
    s.fsa = s.fsa + cad[ib] + cai[ib];
    float trd = solad[ib] * ftdd[ib];
    float tri = solad[ib] * ftid[ib] + solai[ib] * ftii[ib];
    float abs_ = trd * (1.f - albgrd[ib]) + tri * (1.f - albgri[ib]);
    s.sag = s.sag + abs_;
    s.fsa = s.fsa + abs_;
  }
  float laifra = s.elai / nmp_max(vai, MPE);
  if (fsun > 0.f) {
    r.parsun = (cad[0] + fsun * cai[0]) * laifra / nmp_max(r.laisun, MPE);
    r.parsha = (fsha * cai[0]) * laifra / nmp_max(r.laisha, MPE);
  } else {
    r.parsun = 0.f;
    r.parsha = (cad[0] + cai[0]) * laifra / nmp_max(r.laisha, MPE);
  }
  float rvis = albd[0] * solad[0] + albi[0] * solai[0];
  float rnir = albd[1] * solad[1] + albi[1] * solai[1];
  s.fsr = rvis + rnir;
  s.fsrv = frevd[0] * solad[0] + frevi[0] * solai[0] + frevd[1] * solad[1] + frevi[1] * solai[1];
  s.fsrg = fregd[0] * solad[0] + fregi[0] * solai[0] + fregd[1] * solad[1] + fregi[1] * solai[1];
  return r;
}

struct MoState {
  float moz, fm, fh, fh2, fv; int mozsgn;      // FM2 (lsm:4165, 4178, 4189) feeds nothing: not evaluated
  // LOG((ZLVL-ZPD)/Z0M) etc. (lsm:4117-4120): the reference re-evaluates them in every iteration of the
  // flux loops with unchanged arguments; here they are evaluated when the arguments change (iteration 1)
  float tmpcm, tmpcm2;     // every caller passes Z0H = Z0M (lsm:3311-3314, 3772-3774; gla:1059), so TMPCH = TMPCM and TMPCH2 = TMPCM2
};

// SFCDIF1 lsm:4061-4220
// r_rhocp: float64 reciprocal of RHOAIR*CPAIR (the callers' loops divide by it once per iteration; div_rc, nmp_dev_common.hpp)
NMP_DEV void sfcdif1(int& err, int iter, float sfctmp, double r_rhocp, float h, float qair, float zlvl,
                     float zpd, float z0m, float ur, float mpe, MoState& m, float& cm,
                     float& ch) {
  const float z0h = z0m;
  float mozold = m.moz;
  float moz2, fmnew, fhnew, fh2new;
  if (zlvl <= zpd) { if (!err) err = NOAHMP_ERR_STABILITY_STOP; }
  if (iter == 1) {                     // zlvl, zpd, z0m are fixed over the caller's loop
    m.tmpcm = nmp_logf((zlvl - zpd) / z0m);
    m.tmpcm2 = nmp_logf((2.0f + z0m) / z0m);
  }
  const float tmpcm = m.tmpcm, tmpch = m.tmpcm, tmpch2 = m.tmpcm2;
  if (iter == 1) {
    m.fv = 0.0f; m.moz = 0.0f; moz2 = 0.0f;
  } else {
    float tvir = (1.f + 0.61f * qair) * sfctmp;
    float tmp1 = div_rc(VKC * (GRAV / tvir) * h, r_rhocp);
    if (fabsf(tmp1) <= mpe) tmp1 = mpe;
    float mol = -1.f * powi3(m.fv) / tmp1;
    const double r_mol = rc64(mol);
    m.moz = nmp_min(div_rc(zlvl - zpd, r_mol), 1.f);
    moz2 = nmp_min(div_rc(2.0f + z0h, r_mol), 1.f);
  }
  if (mozold * m.moz < 0.f) m.mozsgn = m.mozsgn + 1;
  if (m.mozsgn >= 2) { m.moz = 0.f; m.fm = 0.f; m.fh = 0.f; moz2 = 0.f; m.fh2 = 0.f; }
  if (m.moz < 0.f) {
    float tmp1, tmp12;                       // the two X = (1-16 MOZ)**0.25 are independent: evaluate them interleaved
    pow_quarter2(1.f - 16.f * m.moz, 1.f - 16.f * moz2, tmp1, tmp12);
    const float la[3] = {(1.f + tmp1 * tmp1) / 2.f, (1.f + tmp1) / 2.f, (1.f + tmp12 * tmp12) / 2.f};
    float lg[3];
    nmp_logfN<3>(la, lg);                    // the LOGs are independent: one batch
    const float tmp2 = lg[0], tmp3 = lg[1];
    fmnew = 2.f * tmp3 + tmp2 - 2.f * nmp_atanf_ge1(tmp1) + 1.5707963f;
    fhnew = 2 * tmp2;
    const float tmp22 = lg[2];
    fh2new = 2 * tmp22;
  } else {
    fmnew = -5.f * m.moz; fhnew = fmnew;
    fh2new = -5.f * moz2;
  }
  if (iter == 1) {
    m.fm = fmnew; m.fh = fhnew; m.fh2 = fh2new;
  } else {
    m.fm = 0.5f * (m.fm + fmnew);
    m.fh = 0.5f * (m.fh + fhnew);
    m.fh2 = 0.5f * (m.fh2 + fh2new);
  }
  m.fh = nmp_min(m.fh, 0.9f * tmpch);
  m.fm = nmp_min(m.fm, 0.9f * tmpcm);
  m.fh2 = nmp_min(m.fh2, 0.9f * tmpch2);
  float cmfm = tmpcm - m.fm, chfh = tmpch - m.fh;
  if (fabsf(cmfm) <= mpe) cmfm = mpe;
  if (fabsf(chfh) <= mpe) chfh = mpe;
  cm = VKC * VKC / (cmfm * cmfm);
  ch = VKC * VKC / (cmfm * chfh);
  m.fv = ur * sqrtf(cm);
}

// SFCDIF2 lsm:4224-4422
NMP_DEV float pspmu(float xx) {
  return -2.f * nmp_logf((xx + 1.f) * 0.5f) - nmp_logf((xx * xx + 1.f) * 0.5f) + 2.f * nmp_atanf(xx) -
         (3.14159265f / 2.f);
}
NMP_DEV float psphu(float xx) { return -2.f * nmp_logf((xx * xx + 1.f) * 0.5f); }

// rlogu_in = LOG((ZLM + Z0) / Z0) (RLOGU, lsm:4336): ZU = Z0 and ZLM are fixed over the caller's iteration loop, so the callers evaluate it once
// (sfcdif2_rlogu).  Nothing else of the routine is loop-invariant: ZT follows USTAR (lsm:4334, 4395) and XLU / XLT / XU / XT follow RLMO, which every
// call relaxes (lsm:4417-4420) -- the eight square roots, seven LOGs and two ATANs of the unstable branch are the scheme's own arithmetic.
NMP_DEV float sfcdif2_rlogu(float z0, float zlm) { return nmp_logf((zlm + z0) / z0); }
NMP_DEV void sfcdif2(int iter, float z0, float thz0, float thlm, float sfcspd, float czil, float zlm,
                     float& akms, float& akhs, float& rlmo, float& wstar2, float& ustar, const float rlogu_in) {
  const float WWST2 = 1.2f * 1.2f, VKRM = 0.40f, EXCM = 0.001f, BTG = (1.0f / 270.0f) * GRAV,
              ELFC = VKRM * BTG, WOLD = 0.15f, WNEW = 1.0f - WOLD, EPSU2 = 1.E-4f, EPSUST = 0.07f,
              ZTMIN = -5.0f, ZTMAX = 1.0f, HPBL = 1000.0f, SQVISC = 258.2f;
  float zilfc = -czil * VKRM * SQVISC;
  float zu = z0;
  float rdz = 1.f / zlm;
  float cxch = EXCM * rdz;
  float dthv = thlm - thz0;
  float du2 = nmp_max(sfcspd * sfcspd, EPSU2);
  float btgh = BTG * HPBL;
  if (iter == 1) {
    if (btgh * akhs * dthv != 0.0f) wstar2 = WWST2 * nmp_powf(fabsf(btgh * akhs * dthv), 2.f / 3.f);
    else wstar2 = 0.0f;
    ustar = nmp_max(sqrtf(akms * sqrtf(du2 + wstar2)), EPSUST);
    rlmo = ELFC * akhs * dthv / powi3(ustar);
  }
  float zt = nmp_max(1.E-6f, nmp_expf(zilfc * sqrtf(ustar * z0)) * z0);
  float zslu = zlm + zu;
  float zslt = zlm + zt;
  // RLOGT = LOG(ZSLT / ZT) (lsm:4337) feeds only SIMH at the end: it joins the batch of the branch that follows (seven independent LOGs in
  // the unstable branch: one round of table look-ups instead of two dependent ones)
  const float rlt_arg = zslt / zt;
  const float rlogu = rlogu_in;
  float rlogt;
  float zetalt = nmp_max(zslt * rlmo, ZTMIN);
  rlmo = zetalt / zslt;
  float zetalu = zslu * rlmo;
  float zetau = zu * rlmo;
  float zetat = zt * rlmo;
  float psmz, simm, pshz, simh;
  if (rlmo < 0.f) {
    float xlu = sqrtf(sqrtf(1.f - 16.f * zetalu)), xlt = sqrtf(sqrtf(1.f - 16.f * zetalt)),
          xu = sqrtf(sqrtf(1.f - 16.f * zetau)), xt = sqrtf(sqrtf(1.f - 16.f * zetat));
    // PSPMU(xu), PSPMU(xlu), PSPHU(xt), PSPHU(xlt) (lsm:4290-4299 statement functions): their six LOGs are independent -- one batch of
    // table look-ups instead of six dependent LDS round trips per iteration; the arithmetic of pspmu / psphu is unchanged
    const float la[7] = {(xu + 1.f) * 0.5f, (xu * xu + 1.f) * 0.5f, (xlu + 1.f) * 0.5f, (xlu * xlu + 1.f) * 0.5f,
                         (xt * xt + 1.f) * 0.5f, (xlt * xlt + 1.f) * 0.5f, rlt_arg};
    float lg[7];
    nmp_logfN<7>(la, lg);
    rlogt = lg[6];
    psmz = -2.f * lg[0] - lg[1] + 2.f * nmp_atanf(xu) - (3.14159265f / 2.f);
    const float pspmu_xlu = -2.f * lg[2] - lg[3] + 2.f * nmp_atanf(xlu) - (3.14159265f / 2.f);
    simm = pspmu_xlu - psmz + rlogu;
    pshz = -2.f * lg[4];
    simh = -2.f * lg[5] - pshz + rlogt;
  } else {
    rlogt = nmp_logf(rlt_arg);
    zetalu = nmp_min(zetalu, ZTMAX);
    zetalt = nmp_min(zetalt, ZTMAX);
    psmz = 5.f * zetau;
    simm = 5.f * zetalu - psmz + rlogu;
    pshz = 5.f * zetat;
    simh = 5.f * zetalt - pshz + rlogt;
  }
  ustar = nmp_max(sqrtf(akms * sqrtf(du2 + wstar2)), EPSUST);
  float ustark = ustar * VKRM;
  akms = nmp_max(ustark / simm, cxch);
  akhs = nmp_max(ustark / simh, cxch);
  if (btgh * akhs * dthv != 0.0f) wstar2 = WWST2 * nmp_powf(fabsf(btgh * akhs * dthv), 2.f / 3.f);
  else wstar2 = 0.0f;
  float rlmn = ELFC * akhs * dthv / powi3(ustar);
  rlmo = rlmo * WOLD + rlmn * WNEW;
}

// STOMATA + CI2CI lsm:5323-5464 (bisection on Ci, <= 20 iterations)
// The temperature-only part of STOMATA (lsm:5505-5516): the sunlit and the shaded call of an iteration get the
// same TV, so KC/KO/AWC/CP and the Arrhenius factors of VCMX are evaluated once for both (3 powf + 1 expf each).
struct StomataT { float awc, cp, vcmx_t; };
// the vegetation type's rows of the photosynthesis tables: requested when VEGE_FLUX starts, consumed in its first iteration
// (their memory round trip runs under the set-up, SFCDIF1 and RAGRB instead of in front of STOMATA)
struct StomataP { float bp, c3psn, mp, folnmx, qe25, vcmx25, kc25, akc, ko25, ako, avcmx; };
NMP_DEV StomataP stomata_rows(const noahmp_tables* T, int v) {
  return StomataP{T->bp[v], T->c3psn[v], T->mp[v], T->folnmx[v], T->qe25[v], T->vcmx25[v], T->kc25[v], T->akc[v], T->ko25[v], T->ako[v], T->avcmx[v]};
}
NMP_DEV StomataT stomata_temperature(const StomataP& T, float tv, float o2) {
  StomataT r;
  float tc = tv - TFRZ;
  const float ex = div_rc(tc - 25.0f, NMP_RCC(10.0f));
  float pk[2];
  { const float pb[2] = {T.akc, T.ako}, py[2] = {ex, ex}; nmp_powfN<2>(pb, py, pk); }    // AKC ** ex, AKO ** ex as one batch
  float kc = T.kc25 * pk[0];
  float ko = T.ko25 * pk[1];
  r.awc = kc * (1.0f + o2 / ko);
  r.cp = 0.5f * kc / ko * o2 * 0.21f;
  // VCMX = VCMX25 / F2(TC) * FNF * BTRAN * AVCMX**((TC-25)/10): the first quotient and the last factor are kept
  // apart because the products in between (FNF, BTRAN) must be applied in the reference's order
  r.vcmx_t = nmp_expf((-2.2E05f + 710.0f * (tc + TFRZ)) / (8.314f * (tc + TFRZ)));
  return r;
}

NMP_DEV void stomata(const StomataP& T, float mpe, float apar, float foln, float tv, float ei,
                     float ea, float sfctmp, float sfcprs, float o2, float co2, float igs,
                     float btran, float rb, const StomataT& st, float avcmx_pow, float& rs, float& psn, int& steps) {
  const float bpv = T.bp;
  float cf = sfcprs / (8.314f * sfctmp) * 1.0e06f;
  rs = 1.0f / bpv * cf;
  psn = 0.0f;
  if (apar <= 0.0f) return;
  const float c3 = T.c3psn, mpv = T.mp;
  float fnf = nmp_min(foln / nmp_max(mpe, T.folnmx), 1.0f);
  float ppf = 4.6f * apar;
  float j = ppf * T.qe25;
  const float awc = st.awc, cp = st.cp;
  float vcmx = T.vcmx25 / (1.0f + st.vcmx_t) * fnf * btran * avcmx_pow;
  float rlb = rb / cf;
  float cihi = 1.5f * co2, cilow = 0.0f;
  const double r_sfcprs = rc64(sfcprs);                 // divides once per bisection step
#pragma unroll 1
  for (int iter = 1; iter <= 20; iter++) {                              // ITER = 1, 20 (lsm:5413)
    float ci = 0.5f * (cihi + cilow);
    float wj = nmp_max(ci - cp, 0.0f) * j / (ci + 2.0f * cp) * c3 + j * (1.f - c3);
    float wc = nmp_max(ci - cp, 0.0f) * vcmx / (ci + awc) * c3 + vcmx * (1.f - c3);
    float we = 0.5f * vcmx * c3 + div_rc(4000.0f * vcmx * ci, r_sfcprs) * (1.f - c3);
    psn = nmp_min(nmp_min(wj, wc), we) * igs;
    float cs = nmp_max(co2 - 1.37f * rlb * sfcprs * psn, mpe);
    float a = mpv * psn * sfcprs * ea / (cs * ei) + bpv;
    float b = (mpv * psn * sfcprs / cs + bpv) * rlb - 1.f;
    float cq = -rlb;
    float q;
    if (b >= 0.0f) q = -0.5f * (b + sqrtf(b * b - 4.0f * a * cq));
    else q = -0.5f * (b - sqrtf(b * b - 4.0f * a * cq));
    float r1 = q / a, r2 = cq / q;
    rs = nmp_max(r1, r2);
    float fci = nmp_max(cs - psn * sfcprs * 1.65f * rs, 0.0f);
    steps++;
    if (((cihi - cilow) <= 5e-2f) || fabsf(fci - ci) <= mpe) break;
    else if (fci > ci) cilow = ci;
    else cihi = ci;
  }
  rs = rs * cf;
}

// CANRES lsm:5598-5677 (+ CALHUM lsm:5679-5705)
NMP_DEV void canres(const Parm& P, float par, float sfctmp, float rcsoil, float eah, float sfcprs,
                    float& rc, float& psn) {
  float q2 = 0.622f * eah / (sfcprs - 0.378f * eah);
  q2 = q2 / (1.0f + q2);
  float es = 0.611f * nmp_expf(2.501E6f / 461.0f * (1.f / 273.15f - 1.f / sfctmp));
  float sfcprsx = sfcprs * 1.E-3f;
  float q2sat = 0.622f * es / (sfcprsx - es);
  q2sat = q2sat * 1.E3f;
  q2sat = q2sat / 1.E3f;
  float ff = 2.0f * par / P.rgl;
  float rcs = (ff + P.rsmin / P.rsmax) / (1.0f + ff);
  rcs = nmp_max(rcs, 0.0001f);
  float dt_ = P.topt - sfctmp;
  float rct = 1.0f - 0.0016f * pow_two(dt_);                    // (TOPT - SFCTMP)**2.0 (lsm:5664)
  rct = nmp_max(rct, 0.0001f);
  float rcq = 1.0f / (1.0f + P.hs * nmp_max(0.f, q2sat - q2));
  rcq = nmp_max(rcq, 0.01f);
  rc = P.rsmin / (rcs * rct * rcq * rcsoil);
  psn = -999.99f;
}

// VEGE_FLUX lsm:3018-3589.  `top` quantities are those of layer ISNOW+1.
struct VegIn {
  float ur, vai, gammav, gammag, laisun, laisha, cwp, zlvl, zpd, z0m, z0mg, emv, emg, rsurf, rhsur,
        parsun, parsha, df_top, dz_top, stc_top;
  double r_rhocp, r_gammav, r_gammag;     // 1 / (RHOAIR*CPAIR), 1 / GAMMAV, 1 / GAMMAG (div_rc)
};

// ---- the canopy iteration (loop1 of VEGE_FLUX, lsm:3234-3459) as an explicit state machine ------------------
// Trip counts of this loop run from 6 to NITERC = 20 and differ from column to column: a 64-lane wavefront needs
// ~18 rounds for a mean of ~9.  Everything an iteration reads or carries is in VegLoop, so
// that a runner can decide where and when a column's iterations execute.
struct VegLoop {
  // fixed during the loop
  float sfctmp, rhoair, qair, zlvl, zpd, z0m, ur, z0mg, hcan, cwp, vaie, sqrt_dleaf_uc, fveg, tg, laisune,
        laishae, rssun, rssha, rsurf, eair, estg, gammav, air, cir, canliq, canice, latheav, sav, fwet, sfcprs,
        thair, czil, rlogu;
  double r_rhocp, r_hcan, r_gammav;     // 1 / (RHOAIR*CPAIR), 1 / HCAN, 1 / GAMMAV: divisors of every iteration (div_rc)
  double r_ur;                          // 1 / UR (OPT_SFC = 2 only)
  // carried from iteration to iteration / read after the loop
  MoState mo;
  float cm, ch, tv, tah, eah, h, hg, fhg, dtv, rahc, rahg, rb, cah, cvh, estv, destv, irc, shc, evc, tr, wstar,
        tv_in, csh, cev, ctr;        // the last iteration's TV on entry and conductances (flux corrections after the loop)
  int liter, err, iter, done;    // iter = the next iteration to run (2..21); done = loop1 has exited
};
constexpr int VEGLOOP_WORDS = sizeof(VegLoop) / 4;

struct VegFirst {   // what only iteration 1 needs (STOMATA / CANRES run there, lsm:3287-3320)
  const Parm* P; int v; float parsun, parsha, foln, o2air, co2air, igs, btran; float psnsun, psnsha;
  StomataP sp;
  int bisections;     // STOMATA's bisection steps, both leaves (record_cost)
};

// one pass of the loop body, lsm:3236-3456
template <bool FIRST>
NMP_DEV void vege_iter(const Ctx& c, VegLoop& L, const int iter, VegFirst* f) {
  const float MPE = 1E-6f;
  const float sfctmp = L.sfctmp, rhoair = L.rhoair, ur = L.ur, fveg = L.fveg, tg = L.tg;
  const float z0h = L.z0m, z0hg = L.z0mg, hcan = L.hcan;
  if (c.O.sfc == 1) {
    sfcdif1(L.err, iter, sfctmp, L.r_rhocp, L.h, L.qair, L.zlvl, L.zpd, L.z0m, ur, MPE, L.mo, L.cm, L.ch);
  } else {
    sfcdif2(iter, L.z0m, L.tah, L.thair, ur, L.czil, L.zlvl, L.cm, L.ch, L.mo.moz, L.wstar, L.mo.fv, L.rlogu);
    L.ch = div_rc(L.ch, L.r_ur);                      // CH / UR, CM / UR (lsm:3318-3319): UR is fixed over the loop
    L.cm = div_rc(L.cm, L.r_ur);
  }
  NMP_TIC(16);   // vege loop1: sfcdif
  L.rahc = nmp_max(1.f, 1.f / (L.ch * ur));
  const float rawc = L.rahc;
  {                                                   // RAGRB lsm:3960-4057
    float mozg = 0.f, fhgnew;
    if (!FIRST) {
      float tmp1 = div_rc(VKC * (GRAV / L.tah) * L.hg, L.r_rhocp);
      if (fabsf(tmp1) <= MPE) tmp1 = MPE;
      float molg = -1.f * powi3(L.mo.fv) / tmp1;
      mozg = nmp_min((L.zpd - L.z0mg) / molg, 1.f);
    }
    if (mozg < 0.f) fhgnew = pow_neg_quarter(1.f - 15.f * mozg);
    else fhgnew = 1.f + 4.7f * mozg;
    if (FIRST) L.fhg = fhgnew;
    else L.fhg = 0.5f * (L.fhg + fhgnew);
    float cwpc = pow_half(L.cwp * L.vaie * hcan * L.fhg);
    const float ea[4] = {div_rc(-cwpc * z0hg, L.r_hcan), div_rc(-cwpc * (z0h + L.zpd), L.r_hcan), cwpc, -cwpc / 2.f};
    float ex[4];
    nmp_expfN<4>(ea, ex);                    // the four EXPs of RAGRB are independent: one batch
    float tmp1 = ex[0];
    float tmp2 = ex[1];
    float tmprah2 = hcan * ex[2] / cwpc * (tmp1 - tmp2);
    float kh = nmp_max(VKC * L.mo.fv * (hcan - L.zpd), MPE);
    L.rahg = tmprah2 / kh;
    float tmprb = cwpc * 50.f / (1.f - ex[3]);
    L.rb = tmprb * L.sqrt_dleaf_uc;
  }
  const float rawg = L.rahg, rb = L.rb;
  NMP_TIC(17);   // vege loop1: ragrb
  float t = tdc(L.tv);
  esat_sel(t, L.estv, L.destv);
  const float estv = L.estv, destv = L.destv;
  NMP_TIC(18);   // vege loop1: esat
  if (FIRST) {
    StomataT st = {0.f, 0.f, 0.f};
    float avcmx_pow = 0.f;
    // STOMATA returns early for APAR <= 0 -- and only then: a NaN APAR (OPT_RAD = 1 on a type without crown geometry) goes on
    if (c.O.crs == 1 && (!(f->parsun <= 0.0f) || !(f->parsha <= 0.0f))) {
      st = stomata_temperature(f->sp, L.tv, f->o2air);
      avcmx_pow = nmp_powf(f->sp.avcmx, div_rc((L.tv - TFRZ) - 25.0f, NMP_RCC(10.0f)));
    }
#pragma unroll 1
    for (int leaf = 0; leaf < 2; leaf++) {            // sunlit, then shaded
      float par = leaf ? f->parsha : f->parsun, rs_, psn_;
      if (c.O.crs == 1)
        stomata(f->sp, MPE, par, f->foln, L.tv, estv, L.eah, sfctmp, L.sfcprs, f->o2air, f->co2air, f->igs,
                f->btran, rb, st, avcmx_pow, rs_, psn_, f->bisections);
      else
        canres(*f->P, par, L.tv, f->btran, L.eah, L.sfcprs, rs_, psn_);
      if (leaf) { L.rssha = rs_; f->psnsha = psn_; } else { L.rssun = rs_; f->psnsun = psn_; }
    }
  }
  NMP_TIC(19);   // vege loop1: stomata (first iteration only)
  // RAHC, RB, RAHG and the two conductance sums each divide two or three times: one float64 reciprocal each (rc64)
  const double r_rahc = rc64(L.rahc), r_rb = rc64(rb), r_rahg = rc64(L.rahg);
  L.cah = div_rc(1.f, r_rahc);
  L.cvh = div_rc(2.f * L.vaie, r_rb);
  const float cah = L.cah, cvh = L.cvh;
  float cgh = div_rc(1.f, r_rahg);
  float cond = cah + cvh + cgh;
  const double r_cond = rc64(cond);
  float ata = div_rc(sfctmp * cah + tg * cgh, r_cond);
  float bta = div_rc(cvh, r_cond);
  float csh = (1.f - bta) * rhoair * CPAIR * cvh;
  float caw = cah;                                    // 1 / RAWC, RAWC = RAHC (lsm:3325)
  float cew = div_rc(L.fwet * L.vaie, r_rb);
  float ctw = (1.f - L.fwet) * (L.laisune / (rb + L.rssun) + L.laishae / (rb + L.rssha));
  float cgw = 1.f / (rawg + L.rsurf);
  cond = caw + cew + ctw + cgw;
  const double r_cond2 = rc64(cond);
  float aea = div_rc(L.eair * caw + L.estg * cgw, r_cond2);
  float bea = div_rc(cew + ctw, r_cond2);
  float cev = div_rc((1.f - bea) * cew * rhoair * CPAIR, L.r_gammav);
  float ctr = div_rc((1.f - bea) * ctw * rhoair * CPAIR, L.r_gammav);
  L.tah = ata + bta * L.tv;
  L.eah = aea + bea * estv;
  L.irc = fveg * (L.air + L.cir * powi4(L.tv));
  L.shc = fveg * rhoair * CPAIR * cvh * (L.tv - L.tah);
  L.evc = div_rc(fveg * rhoair * CPAIR * cew * (estv - L.eah), L.r_gammav);
  L.tr = div_rc(fveg * rhoair * CPAIR * ctw * (estv - L.eah), L.r_gammav);
  if (L.tv > TFRZ) L.evc = nmp_min(div_rc(L.canliq * L.latheav, c.u.dt), L.evc);
  else L.evc = nmp_min(div_rc(L.canice * L.latheav, c.u.dt), L.evc);
  float b = L.sav - L.irc - L.shc - L.evc - L.tr;
  float a = fveg * (4.f * L.cir * powi3(L.tv) + csh + (cev + ctr) * destv);
  L.dtv = b / a;
  // The flux corrections by DTV (lsm:3426-3429) are overwritten by the next iteration before anything reads them: what they need of the
  // last iteration is carried out of the loop and they are applied once, after it (vege_flux)
  L.tv_in = L.tv; L.csh = csh; L.cev = cev; L.ctr = ctr;
  L.tv = L.tv + L.dtv;
  L.h = div_rc(rhoair * CPAIR * (L.tah - sfctmp), r_rahc);
  L.hg = div_rc(rhoair * CPAIR * (tg - L.tah), r_rahg);
  // QSFC (lsm:3447) is a function of EAH that nothing inside the loop reads: evaluated once, after the loop (vege_flux)
  NMP_TIC(20);   // vege loop1: flux solve
  NMP_CNT(7);    // (host-emulation instrumentation) loop1 iterations
  // loop control, lsm:3451-3456
  if (L.liter == 1) L.done = 1;
  else if (iter >= 5 && fabsf(L.dtv) <= 0.01f && L.liter == 0) L.liter = 1;
  L.iter = iter + 1;
  if (L.iter > 20) L.done = 1;                        // NITERC = 20 (lsm:3234)
}

// iterations 2.. of a column until it exits or `last` has been run
NMP_DEV void vege_run_until(const Ctx& c, VegLoop& L, int last) {
#pragma unroll 1
  while (!L.done && L.iter <= last) vege_iter<false>(c, L, L.iter, nullptr);
}

// The plain runner: every lane iterates its own column to the end.
struct SimpleLoop {
  NMP_DEV void run(const Ctx& c, VegLoop& L, bool active) const {
    if (active) vege_run_until(c, L, 20);
  }
};

// VEGE_FLUX lsm:3018-3589.  All threads call it (the runner may contain workgroup barriers); `canopy` says
// whether this thread has a vegetated column to work on.
template <class Runner>
NMP_DEV void vege_flux(const Ctx& c, const Parm& P, Col& s, const VegIn& q, float& cmv, float& psnsun,
                       float& psnsha, const bool canopy, Runner& runner) {
  const noahmp_tables* T = c.T;
  const float MPE = 1E-6f;
  VegLoop L = {};
  L.done = 1;
  if (canopy) {
    const int v = s.vegtyp - 1;
    VegFirst f = {};
    f.P = &P; f.v = v;
    if (c.O.crs == 1) f.sp = stomata_rows(T, v);          // requested first: eleven table rows
    const float fveg = s.fveg, ur = q.ur;
    L.sfctmp = s.sfctmp; L.rhoair = s.rhoair; L.qair = s.qair; L.zlvl = q.zlvl; L.zpd = q.zpd; L.z0m = q.z0m;
    L.ur = ur; L.z0mg = q.z0mg; L.cwp = q.cwp; L.fveg = fveg; L.rsurf = q.rsurf; L.eair = s.eair;
    L.gammav = q.gammav; L.canliq = s.canliq; L.canice = s.canice; L.latheav = s.latheav; L.sav = s.sav;
    L.fwet = s.fwet; L.sfcprs = s.sfcprs; L.thair = s.thair; L.czil = P.czil;
    L.mo = MoState{0.f, 0.f, 0.f, 0.f, 0.1f, 0, 0.f, 0.f};
    L.tv = s.tv; L.tg = s.tgv; L.tah = s.tah; L.eah = s.eah; L.ch = s.chv; L.cm = cmv;
    L.r_rhocp = q.r_rhocp; L.r_gammav = q.r_gammav;
    if (c.O.sfc != 1) { L.r_ur = rc64(ur); L.rlogu = sfcdif2_rlogu(L.z0m, L.zlvl); }
    const double r_fveg = rc64(fveg);
    L.vaie = nmp_min(6.f, div_rc(q.vai, r_fveg));
    L.laisune = nmp_min(6.f, div_rc(q.laisun, r_fveg));
    L.laishae = nmp_min(6.f, div_rc(q.laisha, r_fveg));
    float t = tdc(L.tg), destg_unused;
    esat_sel(t, L.estg, destg_unused);
    L.hcan = s.htop;
    L.r_hcan = rc64(L.hcan);
    float uc = ur * nmp_logf(L.hcan / q.z0m) / nmp_logf(q.zlvl / q.z0m);
    if ((L.hcan - q.zpd) <= 0.f) raise(s, NOAHMP_ERR_HCAN_LE_ZPD);
    L.air = -q.emv * (1.f + (1.f - q.emv) * (1.f - q.emg)) * s.lwdn - q.emv * q.emg * SB * powi4(L.tg);
    L.cir = (2.f - q.emv * (1.f - q.emg)) * q.emv * SB;
    L.sqrt_dleaf_uc = sqrtf(P.dleaf / uc);          // loop-invariant factor of RB (lsm:4054)
    L.irc = s.irc; L.shc = s.shc; L.evc = s.evc; L.tr = s.tr;
    L.done = 0; L.iter = 1;
    f.parsun = q.parsun; f.parsha = q.parsha; f.foln = s.foln; f.o2air = s.o2air; f.co2air = s.co2air; f.igs = s.igs; f.btran = s.btran;
    vege_iter<true>(c, L, 1, &f);                       // iteration 1 (with STOMATA / CANRES)
    psnsun = f.psnsun; psnsha = f.psnsha;
    record_cost(c, 1, f.bisections);
  } else {
    record_cost(c, 1, 0);
  }
  runner.run(c, L, canopy);                             // iterations 2..20
  record_cost(c, 0, canopy ? L.iter - 1 : 0);
  NMP_TIC(21);
  if (!canopy) return;
  if (L.err) raise(s, L.err);
  {                                                   // lsm:3426-3429 of the last iteration
    const float fveg = L.fveg, destv = L.destv;
    L.irc = L.irc + fveg * 4.f * L.cir * powi3(L.tv_in) * L.dtv;
    L.shc = L.shc + fveg * L.csh * L.dtv;
    L.evc = L.evc + fveg * L.cev * destv * L.dtv;
    L.tr = L.tr + fveg * L.ctr * destv * L.dtv;
  }
  float& tv = s.tv; float& tg = s.tgv; float& tah = s.tah; float& eah = s.eah;
  const float rhoair = s.rhoair;
  tv = L.tv; tah = L.tah; eah = L.eah; cmv = L.cm;
  s.rssun = L.rssun; s.rssha = L.rssha;
  s.irc = L.irc; s.shc = L.shc; s.evc = L.evc; s.tr = L.tr;
  s.qsfc = (0.622f * L.eah) / (L.sfcprs - 0.378f * L.eah);        // lsm:3447, with the loop's last EAH
  const float rahg = L.rahg, rawg = L.rahg, cah = L.cah, cvh = L.cvh, z0h = q.z0m, fveg = s.fveg;
  float t, estg = L.estg, destg = 0.f;
  // under-canopy ground, lsm:3495-3542
  float air = -q.emg * (1.f - q.emv) * s.lwdn - q.emg * q.emv * SB * powi4(tv);
  float cir = q.emg * SB;
  float csh = rhoair * CPAIR / rahg;
  float cev = rhoair * CPAIR / (q.gammag * (rawg + q.rsurf));
  float cgh = 2.f * q.df_top / q.dz_top;
#pragma unroll 1
  for (int iter = 1; iter <= 5; iter++) {               // loop2, NITERG = 5
    t = tdc(tg);
    esat_sel(t, estg, destg);
    s.irg = cir * powi4(tg) + air;
    s.shg = csh * (tg - tah);
    s.evg = cev * (estg * q.rhsur - eah);
    s.ghv = cgh * (tg - q.stc_top);
    float b = s.sag - s.irg - s.shg - s.evg - s.ghv;
    float a = 4.f * cir * powi3(tg) + csh + cev * destg + cgh;
    float dtg = b / a;
    s.irg = s.irg + 4.f * cir * powi3(tg) * dtg;
    s.shg = s.shg + csh * dtg;
    s.evg = s.evg + cev * destg * dtg;
    s.ghv = s.ghv + cgh * dtg;
    tg = tg + dtg;
  }
  if (c.O.stc == 1) {
    if (s.snowh > 0.05f && tg > TFRZ) {
      tg = TFRZ;
      s.irg = cir * powi4(tg) - q.emg * (1.f - q.emv) * s.lwdn - q.emg * q.emv * SB * powi4(tv);
      s.shg = csh * (tg - tah);
      s.evg = cev * (estg * q.rhsur - eah);
      s.ghv = s.sag - (s.irg + s.shg + s.evg);
    }
  }
  // 2-m diagnostics lsm:3557-3571 (OPT_SFC 1/2; FH2 is 0 under OPT_SFC=2)
  // LOG((2 + Z0H) / Z0H) with Z0H = Z0M: SFCDIF1 evaluated the same expression in its first iteration (tmpcm2)
  float cah2 = L.mo.fv * VKC / ((c.O.sfc == 1 ? L.mo.tmpcm2 : nmp_logf((2.f + z0h) / z0h)) - L.mo.fh2);
  s.chv2 = cah2;
  if (cah2 < 1.E-5f) {
    s.t2mv = tah;
    s.q2v = s.qsfc;
  } else {
    const double r_fveg = rc64(fveg), r_cah2 = rc64(cah2);
    s.t2mv = tah - div_rc(div_rc(s.shg + div_rc(s.shc, r_fveg), q.r_rhocp) * 1.f, r_cah2);
    s.q2v = s.qsfc - div_rc(((div_rc(s.evc + s.tr, r_fveg) + s.evg) / (s.latheav * rhoair)) * 1.f, r_cah2);
  }
  s.chv = cah;
  s.chleaf = cvh;
  s.chuc = 1.f / rahg;
}

// BARE_FLUX lsm:3591-3958
NMP_DEV void bare_flux(const Ctx& c, const Parm& P, Col& s, const VegIn& q, float zpdg, float& cmb) {
  const float MPE = 1E-6f;
  const float rhoair = s.rhoair, sfctmp = s.sfctmp, ur = q.ur, z0m = q.z0mg;
  MoState mo = {0.f, 0.f, 0.f, 0.f, 0.1f, 0, 0.f, 0.f};
  float h = 0.f, wstar = 0.f;
  float t, estg = 0.f, destg, csh = 0.f, cev = 0.f, ehb = 0.f;
  float& tgb = s.tgb; float& ch = s.chb; float& cm = cmb;
  const float z0h = z0m;
  const float cir = q.emg * SB;
  const float cgh = 2.f * q.df_top / q.dz_top;
  const float gamma = q.gammag, lathea = s.latheag;
  // ESAT of the ground temperature (lsm:3789, 3836): the call that follows TGB's update in one iteration has the argument of the call
  // that opens the next one, so each TGB is evaluated once; the flux corrections by DTG and QSFC (lsm:3821-3824, 3842) are
  // overwritten by the next iteration before anything reads them, so only the fifth evaluates them
  t = tdc(tgb);
  esat_sel(t, estg, destg);
  const double r_ur = (c.O.sfc != 1) ? rc64(ur) : 0.0;
  const float rlogu2 = (c.O.sfc != 1) ? sfcdif2_rlogu(z0m, q.zlvl) : 0.f;       // SFCDIF2's RLOGU: Z0M and ZLVL are fixed over loop3
#pragma unroll 1
  for (int iter = 1; iter <= 5; iter++) {               // loop3, NITERB = 5 (lsm:3749)
    if (c.O.sfc == 1) {
      sfcdif1(s.err, iter, sfctmp, q.r_rhocp, h, s.qair, q.zlvl, zpdg, z0m, ur, MPE, mo, cm, ch);
    } else {
      sfcdif2(iter, z0m, tgb, s.thair, ur, P.czil, q.zlvl, cm, ch, mo.moz, wstar, mo.fv, rlogu2);
      ch = div_rc(ch, r_ur);                         // CH / UR, CM / UR (lsm:3776-3777)
      cm = div_rc(cm, r_ur);
      if (s.snowh > 0.f) { cm = nmp_min(0.01f, cm); ch = nmp_min(0.01f, ch); }
    }
    float rahb = nmp_max(1.f, 1.f / (ch * ur));
    float rawb = rahb;
    const double r_rahb = rc64(rahb);
    ehb = div_rc(1.f, r_rahb);
    csh = div_rc(rhoair * CPAIR, r_rahb);
    cev = div_rc(rhoair * CPAIR, q.r_gammag) / (q.rsurf + rawb);
    s.irb = cir * powi4(tgb) - q.emg * s.lwdn;
    s.shb = csh * (tgb - sfctmp);
    s.evb = cev * (estg * q.rhsur - s.eair);
    s.ghb = cgh * (tgb - q.stc_top);
    float b = s.sag - s.irb - s.shb - s.evb - s.ghb;
    float a = 4.f * cir * powi3(tgb) + csh + cev * destg + cgh;
    float dtg = b / a;
    if (iter == 5) {
      s.irb = s.irb + 4.f * cir * powi3(tgb) * dtg;
      s.shb = s.shb + csh * dtg;
      s.evb = s.evb + cev * destg * dtg;
      s.ghb = s.ghb + cgh * dtg;
    }
    tgb = tgb + dtg;
    h = csh * (tgb - sfctmp);
    t = tdc(tgb);
    esat_sel(t, estg, destg);
    if (iter == 5) s.qsfc = 0.622f * (estg * q.rhsur) / (s.psfc - 0.378f * (estg * q.rhsur));
  }
  if (c.O.stc == 1) {
    if (s.snowh > 0.05f && tgb > TFRZ) {
      tgb = TFRZ;
      s.irb = cir * powi4(tgb) - q.emg * s.lwdn;
      s.shb = csh * (tgb - sfctmp);
      s.evb = cev * (estg * q.rhsur - s.eair);
      s.ghb = s.sag - (s.irb + s.shb + s.evb);
    }
  }
  float ehb2 = mo.fv * VKC / ((c.O.sfc == 1 ? mo.tmpcm2 : nmp_logf((2.f + z0h) / z0h)) - mo.fh2);   // as in VEGE_FLUX
  s.chb2 = ehb2;
  if (ehb2 < 1.E-5f) {
    s.t2mb = tgb;
    s.q2b = s.qsfc;
  } else {
    s.t2mb = tgb - div_rc(s.shb, q.r_rhocp) * 1.f / ehb2;
    s.q2b = s.qsfc - s.evb / (lathea * rhoair) * (1.f / ehb2 + q.rsurf);
  }
  if (s.vegtyp == c.isurban) s.q2b = s.qsfc;
  ch = ehb;
}

// TSNOSOI lsm:5707-5822 = HRT (5825-5922) + HSTEP (5925-5977) + ROSR12 (5979-6036).
// Rows ISNOW+1..4 of the 7 fixed slots; inactive slots are predicated off.
template <class A>
NMP_DEV void tsnosoi(const Ctx& c, const Parm& P, const Col& s, const Lay<A>& y, const float* df,
                     const float* hcpct) {
  const int isnow = s.isnow, ntop = isnow + 1;
  const float dt = c.dt;
  const float zbotsno = P.zbot - s.snowh;
  float ai[NL], bi[NL], ci[NL], rhsts[NL], ddz[NL], dtsdz[NL], zs[NL], tt[NL];
#pragma unroll
  for (int k = -2; k <= NSOIL; k++) { zs[L(k)] = y.zsnso[L(k)]; tt[L(k)] = y.stc[L(k)]; }
#pragma unroll
  for (int k = -2; k <= NSOIL; k++) {
    const int km = (k > -2) ? k - 1 : -2;          // clamped neighbours keep every unrolled index
    const int kp = (k < NSOIL) ? k + 1 : NSOIL;    // in range; clamped values are never consumed
    ai[L(k)] = 0.f; bi[L(k)] = 1.f; ci[L(k)] = 0.f; rhsts[L(k)] = 0.f; ddz[L(k)] = 0.f; dtsdz[L(k)] = 0.f;
    if (k >= ntop) {
      float denom, eflux;
      if (k == ntop) {
        denom = -zs[L(k)] * hcpct[L(k)];
        float temp1 = -zs[L(kp)];
        const double r_temp1 = rc64(temp1);
        ddz[L(k)] = div_rc(2.0f, r_temp1);
        dtsdz[L(k)] = div_rc(2.0f * (tt[L(k)] - tt[L(kp)]), r_temp1);
        eflux = df[L(k)] * dtsdz[L(k)] - s.ssoil - 0.f;
      } else if (k < NSOIL) {
        denom = (zs[L(km)] - zs[L(k)]) * hcpct[L(k)];
        float temp1 = zs[L(km)] - zs[L(kp)];
        const double r_temp1 = rc64(temp1);
        ddz[L(k)] = div_rc(2.0f, r_temp1);
        dtsdz[L(k)] = div_rc(2.0f * (tt[L(k)] - tt[L(kp)]), r_temp1);
        eflux = (df[L(k)] * dtsdz[L(k)] - df[L(km)] * dtsdz[L(km)]) - 0.f;
      } else {
        denom = (zs[L(km)] - zs[L(k)]) * hcpct[L(k)];
        float botflx = 0.f;
        if (c.O.tbot == 2) {
          dtsdz[L(k)] = (tt[L(k)] - s.tbot) / (0.5f * (zs[L(km)] + zs[L(k)]) - zbotsno);
          botflx = -df[L(k)] * dtsdz[L(k)];
        }
        eflux = (-botflx - df[L(km)] * dtsdz[L(km)]) - 0.f;
      }
      const double r_denom = rc64(denom);
      if (k == ntop) {
        ai[L(k)] = 0.0f;
        ci[L(k)] = div_rc(-df[L(k)] * ddz[L(k)], r_denom);
        if (c.O.stc == 1) bi[L(k)] = -ci[L(k)];
        else bi[L(k)] = -ci[L(k)] + df[L(k)] / (0.5f * zs[L(k)] * zs[L(k)] * hcpct[L(k)]);
      } else if (k < NSOIL) {
        ai[L(k)] = div_rc(-df[L(km)] * ddz[L(km)], r_denom);
        ci[L(k)] = div_rc(-df[L(k)] * ddz[L(k)], r_denom);
        bi[L(k)] = -(ai[L(k)] + ci[L(k)]);
      } else {
        ai[L(k)] = div_rc(-df[L(km)] * ddz[L(km)], r_denom);
        ci[L(k)] = 0.0f;
        bi[L(k)] = -(ai[L(k)] + ci[L(k)]);
      }
      rhsts[L(k)] = div_rc(eflux, -r_denom);
      // HSTEP scaling
      rhsts[L(k)] = rhsts[L(k)] * dt;
      ai[L(k)] = ai[L(k)] * dt;
      bi[L(k)] = 1.f + bi[L(k)] * dt;
      ci[L(k)] = ci[L(k)] * dt;
    }
  }
  // ROSR12: forward sweep then back substitution (P -> p[], DELTA -> dl[])
  float p[NL], dl[NL];
  ci[L(NSOIL)] = 0.0f;
#pragma unroll
  for (int k = -2; k <= NSOIL; k++) {
    p[L(k)] = 0.f; dl[L(k)] = 0.f;
    if (k == ntop) {
      p[L(k)] = -ci[L(k)] / bi[L(k)];
      dl[L(k)] = rhsts[L(k)] / bi[L(k)];
    } else if (k > ntop) {
      const int km = (k > -2) ? k - 1 : -2;
      float inv = 1.0f / (bi[L(k)] + ai[L(k)] * p[L(km)]);
      p[L(k)] = -ci[L(k)] * inv;
      dl[L(k)] = (rhsts[L(k)] - ai[L(k)] * dl[L(km)]) * inv;
    }
  }
  p[L(NSOIL)] = dl[L(NSOIL)];
#pragma unroll
  for (int kk = NSOIL - 1; kk >= -2; kk--)
    if (kk >= ntop) p[L(kk)] = p[L(kk)] * p[L(kk + 1)] + dl[L(kk)];
#pragma unroll
  for (int k = -2; k <= NSOIL; k++)
    if (k >= ntop) y.stc[L(k)] = tt[L(k)] + p[L(k)];
}

// FRH2O lsm:6247-6377
NMP_DEV float frh2o(const Parm& P, float tkelv, float smc, float sh2o) {
  const float CK = 8.0f, BLIM = 5.5f, ERROR = 0.005f;
  float bx = P.bexp, free_;
  if (P.bexp > BLIM) bx = BLIM;
  int nlog = 0, kcount = 0;
  if (tkelv > (TFRZ - 1.E-3f)) return smc;
  float swl = smc - sh2o;
  if (swl > (smc - 0.02f)) swl = smc - 0.02f;
  if (swl < 0.f) swl = 0.f;
#pragma unroll 1
  while ((nlog < 10) && (kcount == 0)) {
    nlog = nlog + 1;
    float t1 = 1.f + CK * swl;
    float df = nmp_logf((P.psisat * GRAV / HFUS) * pow_two(t1) * nmp_powf(P.smcmax / (smc - swl), bx)) -
               nmp_logf(-(tkelv - TFRZ) / tkelv);
    float denom = 2.f * CK / (1.f + CK * swl) + bx / (smc - swl);
    float swlk = swl - df / denom;
    if (swlk > (smc - 0.02f)) swlk = smc - 0.02f;
    if (swlk < 0.f) swlk = 0.f;
    float dswl = fabsf(swlk - swl);
    swl = swlk;
    if (dswl <= ERROR) kcount = kcount + 1;
  }
  free_ = smc - swl;
  if (kcount == 0) {
    float fk = nmp_powf((HFUS / (GRAV * (-P.psisat))) * ((tkelv - TFRZ) / tkelv), -1 / bx) * P.smcmax;
    if (fk < 0.02f) fk = 0.02f;
    free_ = nmp_min(fk, smc);
  }
  return free_;
}

// PHASECHANGE lsm:6039-6245
template <class A>
NMP_DEV void phasechange(const Ctx& c, const Parm& P, Col& s, const Lay<A>& y, const float* fact) {
  const int isnow = s.isnow;
  const float dt = c.dt;
  float qmelt = 0.f, ponding = 0.f;
  float hm1 = 0.f, xm1 = 0.f;
  // layer 1 first pass values are needed by the no-layer snow melt (lsm:6177); compute all layers
  float hm[NL], xm[NL], wmass0[NL], wice0[NL], mice[NL], mliq[NL], supercool[NL], stc[NL];
  int imelt[NL];
#pragma unroll
  for (int j = -2; j <= NSOIL; j++) {
    supercool[L(j)] = 0.f; imelt[L(j)] = 0; hm[L(j)] = 0.f; xm[L(j)] = 0.f; mice[L(j)] = 0.f;
    mliq[L(j)] = 0.f; wice0[L(j)] = 0.f; wmass0[L(j)] = 0.f; stc[L(j)] = y.stc[L(j)];
  }
#pragma unroll
  for (int j = -2; j <= 0; j++)
    if (j > isnow) { mice[L(j)] = y.snice[L(j)]; mliq[L(j)] = y.snliq[L(j)]; }
#pragma unroll
  for (int j = 1; j <= NSOIL; j++) {
    float dz = y.dzsnso[L(j)];
    mliq[L(j)] = y.sh2o[L(j)] * dz * 1000.f;
    mice[L(j)] = (y.smc[L(j)] - y.sh2o[L(j)]) * dz * 1000.f;
  }
#pragma unroll
  for (int j = -2; j <= NSOIL; j++)
    if (j > isnow) { wice0[L(j)] = mice[L(j)]; wmass0[L(j)] = mice[L(j)] + mliq[L(j)]; }
  if (c.O.frz == 1) {          // the four layers' (SMP / PSISAT) ** (-1 / BEXP) as one batch; layers at or above TFRZ get the benign base 1
    float sx[NSOIL], sy[NSOIL], sp[NSOIL];
    bool any = false;
    const double r_psisat = rc64(P.psisat);
    const float neg_inv_bexp = -1.f / P.bexp;
#pragma unroll
    for (int j = 1; j <= NSOIL; j++) {
      const bool frozen = stc[L(j)] < TFRZ;
      const float smp = HFUS * (TFRZ - stc[L(j)]) / (GRAV * stc[L(j)]);
      sx[j - 1] = frozen ? div_rc(smp, r_psisat) : 1.0f;
      sy[j - 1] = neg_inv_bexp;
      any = any || frozen;
    }
    if (any) {
      nmp_powfN<NSOIL>(sx, sy, sp);
#pragma unroll
      for (int j = 1; j <= NSOIL; j++)
        if (stc[L(j)] < TFRZ) {
          supercool[L(j)] = P.smcmax * sp[j - 1];
          supercool[L(j)] = supercool[L(j)] * y.dzsnso[L(j)] * 1000.f;
        }
    }
  }
#pragma unroll
  for (int j = 1; j <= NSOIL; j++) {
    if (c.O.frz == 1) {
    } else {
      supercool[L(j)] = frh2o(P, stc[L(j)], y.smc[L(j)], y.sh2o[L(j)]);
      supercool[L(j)] = supercool[L(j)] * y.dzsnso[L(j)] * 1000.f;
    }
  }
#pragma unroll
  for (int j = -2; j <= NSOIL; j++) {
    if (j > isnow) {
      if (mice[L(j)] > 0.f && stc[L(j)] >= TFRZ) imelt[L(j)] = 1;
      if (mliq[L(j)] > supercool[L(j)] && stc[L(j)] < TFRZ) imelt[L(j)] = 2;
      if (isnow == 0 && s.sneqv > 0.f && j == 1) {
        if (stc[L(j)] >= TFRZ) imelt[L(j)] = 1;
      }
      if (imelt[L(j)] > 0) {
        hm[L(j)] = (stc[L(j)] - TFRZ) / fact[L(j)];
        stc[L(j)] = TFRZ;
      }
      if (imelt[L(j)] == 1 && hm[L(j)] < 0.f) { hm[L(j)] = 0.f; imelt[L(j)] = 0; }
      if (imelt[L(j)] == 2 && hm[L(j)] > 0.f) { hm[L(j)] = 0.f; imelt[L(j)] = 0; }
      xm[L(j)] = div_rc(hm[L(j)] * dt, NMP_RCC(HFUS));
    }
  }
  (void)hm1; (void)xm1;
  if (isnow == 0 && s.sneqv > 0.f && xm[L(1)] > 0.f) {
    float temp1 = s.sneqv;
    s.sneqv = nmp_max(0.f, temp1 - xm[L(1)]);
    float propor = s.sneqv / temp1;
    s.snowh = nmp_max(0.f, propor * s.snowh);
    float heatr = hm[L(1)] - div_rc(HFUS * (temp1 - s.sneqv), c.u.dt);
    if (heatr > 0.f) { xm[L(1)] = div_rc(heatr * dt, NMP_RCC(HFUS)); hm[L(1)] = heatr; }
    else { xm[L(1)] = 0.f; hm[L(1)] = 0.f; }
    qmelt = div_rc(nmp_max(0.f, (temp1 - s.sneqv)), c.u.dt);
    ponding = temp1 - s.sneqv;
  }
#pragma unroll
  for (int j = -2; j <= NSOIL; j++) {
    if (j > isnow) {
      if (imelt[L(j)] > 0 && fabsf(hm[L(j)]) > 0.f) {
        float heatr = 0.f;
        if (xm[L(j)] > 0.f) {
          mice[L(j)] = nmp_max(0.f, wice0[L(j)] - xm[L(j)]);
          heatr = hm[L(j)] - div_rc(HFUS * (wice0[L(j)] - mice[L(j)]), c.u.dt);
        } else if (xm[L(j)] < 0.f) {
          if (j <= 0) {
            mice[L(j)] = nmp_min(wmass0[L(j)], wice0[L(j)] - xm[L(j)]);
          } else {
            if (wmass0[L(j)] < supercool[L(j)]) {
              mice[L(j)] = 0.f;
            } else {
              mice[L(j)] = nmp_min(wmass0[L(j)] - supercool[L(j)], wice0[L(j)] - xm[L(j)]);
              mice[L(j)] = nmp_max(mice[L(j)], 0.0f);
            }
          }
          heatr = hm[L(j)] - div_rc(HFUS * (wice0[L(j)] - mice[L(j)]), c.u.dt);
        }
        mliq[L(j)] = nmp_max(0.f, wmass0[L(j)] - mice[L(j)]);
        if (fabsf(heatr) > 0.f) {
          stc[L(j)] = stc[L(j)] + fact[L(j)] * heatr;
          if (j <= 0) {
            if (mliq[L(j)] * mice[L(j)] > 0.f) stc[L(j)] = TFRZ;
          }
        }
        if (j < 1) qmelt = qmelt + div_rc(nmp_max(0.f, (wice0[L(j)] - mice[L(j)])), c.u.dt);
      }
    }
  }
#pragma unroll
  for (int j = -2; j <= NSOIL; j++) {
    if (j > isnow) { y.stc[L(j)] = stc[L(j)]; }
    y.imelt[L(j)] = (float)imelt[L(j)];
  }
#pragma unroll
  for (int j = -2; j <= 0; j++)
    if (j > isnow) { y.snliq[L(j)] = mliq[L(j)]; y.snice[L(j)] = mice[L(j)]; }
#pragma unroll
  for (int j = 1; j <= NSOIL; j++) {
    const double r_dzmm = rc64(1000.f * y.dzsnso[L(j)]);
    y.sh2o[L(j)] = div_rc(mliq[L(j)], r_dzmm);
    y.smc[L(j)] = div_rc(mliq[L(j)] + mice[L(j)], r_dzmm);
  }
  s.qmelt = qmelt;
  s.ponding = ponding;
}

// the dry-layer factor of the Sakaguchi-Zeng soil surface resistance (lsm:1655): soil parameters only
NMP_DEV float rsurf_dry_layer(const Parm& P) {
  return 2.2E-5f * P.smcmax * P.smcmax * nmp_powf(1.0f - P.smcwlt / P.smcmax, 2.0f + 3.0f / P.bexp);
}

// dynamic-top accessor: value of a 7-slot array at layer ISNOW+1 without run-time indexing
NMP_DEV float at_top(const float* a, int isnow) {
  return (isnow == 0) ? a[L(1)] : (isnow == -1) ? a[L(0)] : (isnow == -2) ? a[L(-1)] : a[L(-2)];
}

// ENERGY lsm:1231-1843
// All threads of the workgroup call it: `live` says whether this thread carries a land column; the canopy
// iteration runner in the middle may synchronise the workgroup (CompactLoop).
// `before_soil_heat`: a hook of the caller, run once the surface fluxes are final and before TSNOSOI -- the column step issues the
// loads of the WATER phase there (its state words and table rows), so that their memory round trip runs under TSNOSOI / PHASECHANGE
// instead of being waited for at the start of WATER.
template <class A, class Runner, class Hook>
NMP_DEV void energy(const Ctx& c, const Parm& P, Col& s, const Lay<A>& y, const bool live, Runner& runner, Hook& before_soil_heat) {
  const noahmp_tables* T = c.T;
  const int v = s.vegtyp - 1;
  const float MPE = 1.E-6f, PSIWLT = -150.f, Z0 = 0.01f;
  float psnsun = 0.f, psnsha = 0.f;
  VegIn q = {};
  bool veg = false;
  float zpdg = 0.f;
  RadOut r = {};
  float df[NL], hcpct[NL], fact[NL];
#pragma unroll
  for (int k = 0; k < NL; k++) { df[k] = 0.f; hcpct[k] = 0.f; fact[k] = 0.f; }
  if (live) {
  s.irc = 0.f; s.shc = 0.f; s.irg = 0.f; s.shg = 0.f; s.evg = 0.f; s.evc = 0.f; s.tr = 0.f;
  s.ghv = 0.f; s.t2mv = 0.f; s.q2v = 0.f; s.chv = 0.f; s.chleaf = 0.f; s.chuc = 0.f; s.chv2 = 0.f;
  q.ur = nmp_max(sqrtf(pow_two(s.uu) + pow_two(s.vv)), 1.f);      // UU**2.+VV**2. (lsm:1536)
  q.vai = s.elai + s.esai;
  veg = (q.vai > 0.f);
  s.fsno = 0.f;
  if (s.snowh > 0.f) {
    float bdsno = s.sneqv / s.snowh;
    float fmelt = nmp_powf(div_rc(bdsno, NMP_RCC(100.f)), M_MELT);
    s.fsno = nmp_tanhf(s.snowh / (2.5f * Z0 * fmelt));
  }
  q.z0mg = Z0 * (1.0f - s.fsno) + s.fsno * Z0SNO;
  zpdg = s.snowh;
  if (veg) {
    q.z0m = P.z0mvt;
    q.zpd = 0.65f * s.htop;
    if (s.snowh > q.zpd) q.zpd = s.snowh;
  } else {
    q.z0m = q.z0mg;
    q.zpd = zpdg;
  }
  q.zlvl = nmp_max(q.zpd, s.htop) + s.zlvl;
  if (zpdg >= q.zlvl) q.zlvl = zpdg + s.zlvl;
  q.cwp = P.cwpvt;
  NMP_TIC(2);    // energy: preamble
  RadP rp = {};
  if (!(s.cosz <= 0.f)) rp = radiation_rows(T, v);      // their round trip runs under THERMOPROP
  thermoprop(c, P, s, y, df, hcpct, fact);
  NMP_TIC(3);    // thermoprop
  r = radiation(c, P, s, y.smc[L(1)], rp);
  NMP_TIC(4);    // radiation
  q.laisun = r.laisun; q.laisha = r.laisha; q.parsun = r.parsun; q.parsha = r.parsha;
  q.emv = 1.f - nmp_expf(-(s.elai + s.esai) / 1.0f);
  q.emg = c.ts.eg1 * (1.f - s.fsno) + 1.0f * s.fsno;            // EG(IST = 1) (drv:527); ICE is 0 on this path (drv:549)
  // BTRAN lsm:1617-1640
  s.btran = 0.f;
  const float zroot = -pick_layer(c.zsoil, P.nroot);
  const double r_zroot = pick_layer(c.u.zs, P.nroot), r_smcmax = rc64(P.smcmax);
  const double r_refwlt = (c.O.btr == 1) ? rc64(P.smcref - P.smcwlt) : 0.0;
#pragma unroll
  for (int iz = 1; iz <= NSOIL; iz++) {
    if (iz <= P.nroot) {
      float gx, sh = y.sh2o[L(iz)];
      if (c.O.btr == 1) {
        gx = div_rc(sh - P.smcwlt, r_refwlt);
      } else {
        float psi = nmp_max(PSIWLT, -P.psisat * nmp_powf(div_rc(nmp_max(0.01f, sh), r_smcmax), -P.bexp));
        if (c.O.btr == 2) gx = (1.f - psi / PSIWLT) / (1.f + P.psisat / PSIWLT);
        else gx = 1.f - nmp_expf(-5.8f * (nmp_logf(PSIWLT / psi)));
      }
      gx = nmp_min(1.f, nmp_max(0.f, gx));
      float bt = nmp_max(MPE, div_rc(y.dzsnso[L(iz)], r_zroot) * gx);
      y.btrani[L(iz)] = bt;
      s.btran = s.btran + bt;
    }
  }
  s.btran = nmp_max(MPE, s.btran);
  const double r_btran = rc64(s.btran);
#pragma unroll
  for (int iz = 1; iz <= NSOIL; iz++)
    if (iz <= P.nroot) y.btrani[L(iz)] = div_rc(y.btrani[L(iz)], r_btran);
  // soil surface resistance lsm:1644-1669
  {
    float sh1 = y.sh2o[L(1)];
    float l_rsurf = div_rc((-c.zsoil[L(1)]) * (nmp_expf(powi5(1.0f - nmp_min(1.0f, div_rc(sh1, r_smcmax)))) - 1.0f),
                           NMP_RCC(2.71828f - 1.0f));
    float d_rsurf = P.d_rsurf;                         // rsurf_dry_layer(P), per soil type (Derived, fetched by REDPRM)
    q.rsurf = l_rsurf / d_rsurf;
    if (sh1 < 0.01f && s.snowh == 0.f) q.rsurf = 1.E6f;
    float psi = -P.psisat * nmp_powf(div_rc(nmp_max(0.01f, sh1), r_smcmax), -P.bexp);
    q.rhsur = s.fsno + (1.f - s.fsno) * nmp_expf(psi * GRAV / (RW * s.tg));
  }
  if (s.vegtyp == c.isurban && s.snowh == 0.f) q.rsurf = 1.E6f;
  if (s.tv > TFRZ) { s.latheav = HVAP; s.frozen_canopy = 0; } else { s.latheav = HSUB; s.frozen_canopy = 1; }
  q.gammav = div_rc(CPAIR * s.sfcprs, s.frozen_canopy ? NMP_RCC(0.622f * HSUB) : NMP_RCC(0.622f * HVAP));
  if (s.tg > TFRZ) { s.latheag = HVAP; s.frozen_ground = 0; } else { s.latheag = HSUB; s.frozen_ground = 1; }
  q.gammag = div_rc(CPAIR * s.sfcprs, s.frozen_ground ? NMP_RCC(0.622f * HSUB) : NMP_RCC(0.622f * HVAP));
  q.r_rhocp = rc64(s.rhoair * CPAIR); q.r_gammav = rc64(q.gammav); q.r_gammag = rc64(q.gammag);
  q.df_top = at_top(df, s.isnow);
  q.dz_top = y.dzsnso[L(s.isnow + 1)];
  q.stc_top = y.stc[L(s.isnow + 1)];

  }  // live
  float cmv = 0.f, cmb = 0.f;
  const bool canopy = live && veg && s.fveg > 0;
  NMP_TIC(5);    // btran, rsurf, psychrometric constants
  if (canopy) {
    s.tgv = s.tg;
    cmv = s.cm;
    s.chv = s.ch;
  }
  vege_flux(c, P, s, q, cmv, psnsun, psnsha, canopy, runner);
  NMP_TIC(6);    // vege_flux
  if (!live) return;
  s.tgb = s.tg;
  cmb = s.cm;
  s.chb = s.ch;
  bare_flux(c, P, s, q, zpdg, cmb);
  NMP_TIC(7);    // bare_flux
  if (canopy) {
    s.fira = s.fveg * s.irg + (1.0f - s.fveg) * s.irb + s.irc;
    s.fsh = s.fveg * s.shg + (1.0f - s.fveg) * s.shb + s.shc;
    s.fgev = s.fveg * s.evg + (1.0f - s.fveg) * s.evb;
    s.ssoil = s.fveg * s.ghv + (1.0f - s.fveg) * s.ghb;
    s.fcev = s.evc;
    s.fctr = s.tr;
    s.tg = s.fveg * s.tgv + (1.0f - s.fveg) * s.tgb;
    s.cm = s.fveg * cmv + (1.0f - s.fveg) * cmb;
    s.ch = s.fveg * s.chv + (1.0f - s.fveg) * s.chb;
  } else {
    s.fira = s.irb; s.fsh = s.shb; s.fgev = s.evb; s.ssoil = s.ghb; s.tg = s.tgb;
    s.fcev = 0.f; s.fctr = 0.f;
    s.cm = cmb; s.ch = s.chb;
    s.rssun = 0.0f; s.rssha = 0.0f;
    s.tgv = s.tgb; s.chv = s.chb;
  }
  float fire = s.lwdn + s.fira;
  if (fire <= 0.f) raise(s, NOAHMP_ERR_FIRE_NONPOSITIVE);
  s.emissi = s.fveg * (q.emg * (1 - q.emv) + q.emv + q.emv * (1 - q.emv) * (1 - q.emg)) +
             (1 - s.fveg) * q.emg;
  s.trad = pow_quarter((fire - (1 - s.emissi) * s.lwdn) / (s.emissi * SB));
  s.apar = r.parsun * r.laisun + r.parsha * r.laisha;
  s.psn = psnsun * r.laisun + psnsha * r.laisha;
  NMP_TIC(8);    // flux merge, trad
  before_soil_heat();
  tsnosoi(c, P, s, y, df, hcpct);
  NMP_TIC(9);    // tsnosoi
  if (c.O.stc == 2) {
    if (s.snowh > 0.05f && s.tg > TFRZ) {
      s.tgv = TFRZ; s.tgb = TFRZ;
      if (canopy) s.tg = s.fveg * s.tgv + (1.0f - s.fveg) * s.tgb;
      else s.tg = s.tgb;
    }
  }
  phasechange(c, P, s, y, fact);
  NMP_TIC(10);   // phasechange
}

}  // namespace nmp

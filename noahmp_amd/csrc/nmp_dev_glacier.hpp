// Noah-MP column engine for MI355X -- land-ice column, NOAHMP_GLACIER
// (reference phys/module_sf_noahmp_glacier.F90:150-338, "gla").  No canopy, no soil water.
// Shares SFCDIF1 / ESAT / the heat-diffusion solve / COMPACT / COMBO with the land path and the
// GLAC-templated COMBINE / DIVIDE / SNOWH2O; the glacier-only pieces are below.
#pragma once
#include "nmp_dev_energy.hpp"
#include "nmp_dev_water.hpp"

namespace nmp {

// PHASECHANGE_GLACIER gla:1635-1922 (ice layers 1..4 are hard-coded there too)
template <class A>
NMP_DEV void phasechange_glacier(const Ctx& c, Col& s, const Lay<A>& y, const float* fact) {
  const int isnow = s.isnow;
  const float dt = c.dt;
  float hm[NL], xm[NL], wmass0[NL], wice0[NL], mice[NL], mliq[NL], heatr[NL], stc[NL];
  int imelt[NL];
  float qmelt = 0.f, ponding = 0.f;
#pragma unroll
  for (int j = -2; j <= NSOIL; j++) {
    hm[L(j)] = 0.f; xm[L(j)] = 0.f; heatr[L(j)] = 0.f; imelt[L(j)] = 0; mice[L(j)] = 0.f; mliq[L(j)] = 0.f;
    wice0[L(j)] = 0.f; wmass0[L(j)] = 0.f; stc[L(j)] = y.stc[L(j)];
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
  for (int j = -2; j <= NSOIL; j++) {
    if (j > isnow) {
      wice0[L(j)] = mice[L(j)]; wmass0[L(j)] = mice[L(j)] + mliq[L(j)];
      if (mice[L(j)] > 0.f && stc[L(j)] >= TFRZ) imelt[L(j)] = 1;
      if (mliq[L(j)] > 0.f && stc[L(j)] < TFRZ) imelt[L(j)] = 2;
      if (isnow == 0 && s.sneqv > 0.f && j == 1) {
        if (stc[L(j)] >= TFRZ) imelt[L(j)] = 1;
      }
      if (imelt[L(j)] > 0) { hm[L(j)] = (stc[L(j)] - TFRZ) / fact[L(j)]; stc[L(j)] = TFRZ; }
      if (imelt[L(j)] == 1 && hm[L(j)] < 0.f) { hm[L(j)] = 0.f; imelt[L(j)] = 0; }
      if (imelt[L(j)] == 2 && hm[L(j)] > 0.f) { hm[L(j)] = 0.f; imelt[L(j)] = 0; }
      xm[L(j)] = hm[L(j)] * dt / HFUS;
    }
  }
  if (isnow == 0 && s.sneqv > 0.f && xm[L(1)] > 0.f) {
    float temp1 = s.sneqv;
    s.sneqv = nmp_max(0.f, temp1 - xm[L(1)]);
    float propor = s.sneqv / temp1;
    s.snowh = nmp_max(0.f, propor * s.snowh);
    float h1 = hm[L(1)] - HFUS * (temp1 - s.sneqv) / dt;
    if (h1 > 0.f) { xm[L(1)] = h1 * dt / HFUS; hm[L(1)] = h1; imelt[L(1)] = 1; }
    else { xm[L(1)] = 0.f; hm[L(1)] = 0.f; imelt[L(1)] = 0; }
    qmelt = nmp_max(0.f, (temp1 - s.sneqv)) / dt;
    ponding = temp1 - s.sneqv;
  }
#pragma unroll
  for (int j = -2; j <= NSOIL; j++) {
    if (j > isnow) {
      if (imelt[L(j)] > 0 && fabsf(hm[L(j)]) > 0.f) {
        float hr = 0.f;
        if (xm[L(j)] > 0.f) {
          mice[L(j)] = nmp_max(0.f, wice0[L(j)] - xm[L(j)]);
          hr = hm[L(j)] - HFUS * (wice0[L(j)] - mice[L(j)]) / dt;
        } else if (xm[L(j)] < 0.f) {
          mice[L(j)] = nmp_min(wmass0[L(j)], wice0[L(j)] - xm[L(j)]);
          hr = hm[L(j)] - HFUS * (wice0[L(j)] - mice[L(j)]) / dt;
        }
        mliq[L(j)] = nmp_max(0.f, wmass0[L(j)] - mice[L(j)]);
        if (fabsf(hr) > 0.f) {
          stc[L(j)] = stc[L(j)] + fact[L(j)] * hr;
          if (j <= 0) { if (mliq[L(j)] * mice[L(j)] > 0.f) stc[L(j)] = TFRZ; }
        }
        if (j < 1) qmelt = qmelt + nmp_max(0.f, (wice0[L(j)] - mice[L(j)])) / dt;
      }
    }
  }
#pragma unroll
  for (int j = -2; j <= NSOIL; j++) { heatr[L(j)] = 0.f; xm[L(j)] = 0.f; }
  // four residual-redistribution passes between the ice layers, gla:1804-1908
  auto any_gt = [&]() { return any_of_layers_1_to_4(stc, [](float t) { return t > TFRZ; }); };
  auto any_lt = [&]() { return any_of_layers_1_to_4(stc, [](float t) { return t < TFRZ; }); };
  if (any_gt() && any_lt()) {
#pragma unroll
    for (int j = 1; j <= NSOIL; j++) {
      if (stc[L(j)] > TFRZ) {
        heatr[L(j)] = (stc[L(j)] - TFRZ) / fact[L(j)];
#pragma unroll
        for (int k = 1; k <= NSOIL; k++) {
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
  if (any_gt() && any_lt()) {
#pragma unroll
    for (int j = 1; j <= NSOIL; j++) {
      if (stc[L(j)] < TFRZ) {
        heatr[L(j)] = (stc[L(j)] - TFRZ) / fact[L(j)];
#pragma unroll
        for (int k = 1; k <= NSOIL; k++) {
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
  if (any_gt() && any_of_layers_1_to_4(mice, [](float m) { return m > 0.f; })) {
#pragma unroll
    for (int j = 1; j <= NSOIL; j++) {
      if (stc[L(j)] > TFRZ) {
        heatr[L(j)] = (stc[L(j)] - TFRZ) / fact[L(j)];
        xm[L(j)] = heatr[L(j)] * dt / HFUS;
#pragma unroll
        for (int k = 1; k <= NSOIL; k++) {
          if (j != k && mice[L(k)] > 0.f && xm[L(j)] > 0.1f) {
            if (mice[L(k)] > xm[L(j)]) {
              mice[L(k)] = mice[L(k)] - xm[L(j)];
              stc[L(k)] = TFRZ;
              xm[L(j)] = 0.0f;
            } else {
              xm[L(j)] = xm[L(j)] - mice[L(k)];
              mice[L(k)] = 0.0f;
              stc[L(k)] = TFRZ;
            }
            mliq[L(k)] = nmp_max(0.f, wmass0[L(k)] - mice[L(k)]);
          }
        }
        heatr[L(j)] = xm[L(j)] * HFUS / dt;
        stc[L(j)] = TFRZ + heatr[L(j)] * fact[L(j)];
      }
    }
  }
  if (any_lt() && any_of_layers_1_to_4(mliq, [](float m) { return m > 0.f; })) {
#pragma unroll
    for (int j = 1; j <= NSOIL; j++) {
      if (stc[L(j)] < TFRZ) {
        heatr[L(j)] = (stc[L(j)] - TFRZ) / fact[L(j)];
        xm[L(j)] = heatr[L(j)] * dt / HFUS;
#pragma unroll
        for (int k = 1; k <= NSOIL; k++) {
          if (j != k && mliq[L(k)] > 0.f && xm[L(j)] < -0.1f) {
            if (mliq[L(k)] > fabsf(xm[L(j)])) {
              mice[L(k)] = mice[L(k)] - xm[L(j)];
              stc[L(k)] = TFRZ;
              xm[L(j)] = 0.0f;
            } else {
              xm[L(j)] = xm[L(j)] + mliq[L(k)];
              mice[L(k)] = wmass0[L(k)];
              stc[L(k)] = TFRZ;
            }
            mliq[L(k)] = nmp_max(0.f, wmass0[L(k)] - mice[L(k)]);
          }
        }
        heatr[L(j)] = xm[L(j)] * HFUS / dt;
        stc[L(j)] = TFRZ + heatr[L(j)] * fact[L(j)];
      }
    }
  }
#pragma unroll
  for (int j = -2; j <= NSOIL; j++) {
    if (j > isnow) y.stc[L(j)] = stc[L(j)];
    y.imelt[L(j)] = (float)imelt[L(j)];
  }
#pragma unroll
  for (int j = -2; j <= 0; j++)
    if (j > isnow) { y.snliq[L(j)] = mliq[L(j)]; y.snice[L(j)] = mice[L(j)]; }
#pragma unroll
  for (int j = 1; j <= NSOIL; j++) {
    float sh = mliq[L(j)] / (1000.f * y.dzsnso[L(j)]);
    y.sh2o[L(j)] = nmp_max(0.0f, nmp_min(1.0f, sh));
    y.smc[L(j)] = 1.0f;
  }
  s.qmelt = qmelt;
  s.ponding = ponding;
}

// NOAHMP_GLACIER gla:150-338 with ENERGY_GLACIER (393-573), GLACIER_FLUX (942-1148),
// WATER_GLACIER (1924-2110), SNOWWATER_GLACIER (2113-2237), ERROR_GLACIER (2898-2972)
template <class A>
NMP_DEV void glacier(const Ctx& c, Col& s, const Lay<A>& y) {
  const float dt = c.dt;
  const float MPE = 1E-6f;
  // ATM_GLACIER
  s.qair = s.q2;
  s.eair = s.qair * s.sfcprs / (0.622f + 0.378f * s.qair);
  s.rhoair = (s.sfcprs - 0.378f * s.eair) / (RAIR * s.sfctmp);
  s.swdown = (s.cosz <= 0.f) ? 0.f : s.soldn;
  const float solad = s.swdown * 0.7f * 0.5f, solai = s.swdown * 0.3f * 0.5f;
  const float beg_wb = s.sneqv;
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
  // ---- ENERGY_GLACIER
  const float ur = nmp_max(sqrtf(pow_two(s.uu) + pow_two(s.vv)), 1.f);      // gla:490
  const float z0m = Z0SNO, zpd = s.snowh, zlvl = zpd + s.zlvl;
  float df[NL], hcpct[NL], fact[NL];
#pragma unroll
  for (int k = 0; k < NL; k++) { df[k] = 0.f; hcpct[k] = 0.f; fact[k] = 0.f; }
  {                                                       // THERMOPROP_GLACIER gla:575-645
#pragma unroll
    for (int iz = -2; iz <= 0; iz++) {
      if (iz > s.isnow) {
        float dz = y.dzsnso[L(iz)];
        float snicev = nmp_min(1.f, y.snice[L(iz)] / (dz * DENICE));
        float epore = 1.f - snicev;
        float snliqv = nmp_min(epore, y.snliq[L(iz)] / (dz * DENH2O));
        float bdsnoi = (y.snice[L(iz)] + y.snliq[L(iz)]) / dz;
        hcpct[L(iz)] = CICE * snicev + CWAT * snliqv;
        df[L(iz)] = 3.2217E-6f * pow_two(bdsnoi);     // gla:695
      }
    }
    float above = 0.f;
#pragma unroll
    for (int iz = 1; iz <= NSOIL; iz++) {
      float dz = y.dzsnso[L(iz)];
      float zmid = 0.5f * dz;
      // ZMID = 0.5*DZ(IZ) + DZ(1) + ... + DZ(IZ-1), summed in the reference's order (gla:622-625)
#pragma unroll
      for (int m = 1; m < NSOIL; m++)
        if (m < iz) zmid = zmid + y.dzsnso[L(m)];
      hcpct[L(iz)] = 1.E6f * (0.8194f + 0.1309f * zmid);
      df[L(iz)] = 0.32333f + (0.10073f * zmid);
    }
    (void)above;
#pragma unroll
    for (int iz = -2; iz <= NSOIL; iz++)
      if (iz > s.isnow) fact[L(iz)] = dt / (hcpct[L(iz)] * y.dzsnso[L(iz)]);
    if (s.isnow == 0)
      df[L(1)] = (df[L(1)] * y.dzsnso[L(1)] + 0.35f * s.snowh) / (s.snowh + y.dzsnso[L(1)]);
    else
      df[L(1)] = (df[L(1)] * y.dzsnso[L(1)] + df[L(0)] * y.dzsnso[L(0)]) / (y.dzsnso[L(0)] + y.dzsnso[L(1)]);
  }
  {                                                       // RADIATION_GLACIER gla:704-792
    float albsnd[2] = {0.f, 0.f}, albsni[2] = {0.f, 0.f};
    const float albice[2] = {0.80f, 0.55f};
    float fage;
    snow_age(dt, s.tg, s.sneqvo, s.sneqv, s.tauss, fage);   // ages at night too (no COSZ gate)
    if (c.O.alb == 1) {
      float sl = 2.0f, sl1 = 1.f / sl, sl2 = 2.f * sl;
      float cf1 = ((1.f + sl1) / (1.f + sl2 * s.cosz) - sl1);
      float fzen = nmp_max(cf1, 0.f);
      albsni[0] = 0.95f * (1.f - 0.2f * fage);
      albsni[1] = 0.65f * (1.f - 0.5f * fage);
      albsnd[0] = albsni[0] + 0.4f * fzen * (1.f - albsni[0]);
      albsnd[1] = albsni[1] + 0.4f * fzen * (1.f - albsni[1]);
    }
    if (c.O.alb == 2) {
      float alb = 0.55f + (s.albold - 0.55f) * nmp_expf(-0.01f * dt / 3600.f);
      if (s.qsnow > 0.f) alb = alb + nmp_min(s.qsnow * dt, SWEMX) * (0.84f - alb) / (SWEMX);
      albsni[0] = albsni[1] = albsnd[0] = albsnd[1] = alb;
      s.albold = alb;
    }
    s.sag = 0.f; s.fsa = 0.f; s.fsr = 0.f;
    const float fsno = (s.sneqv > 0.0f) ? 1.0f : 0.0f;
#pragma unroll
    for (int ib = 0; ib < 2; ib++) {
      albsnd[ib] = albice[ib] * (1.f - fsno) + albsnd[ib] * fsno;
      albsni[ib] = albice[ib] * (1.f - fsno) + albsni[ib] * fsno;
      float abs_ = solad * (1.f - albsnd[ib]) + solai * (1.f - albsni[ib]);
      s.sag = s.sag + abs_;
      s.fsa = s.fsa + abs_;
      float ref = solad * albsnd[ib] + solai * albsni[ib];
      s.fsr = s.fsr + ref;
    }
  }
  const float emg = 0.98f, rhsur = 1.0f, rsurf = 1.0f, lathea = HSUB;
  const float gamma = CPAIR * s.sfcprs / (0.622f * lathea);
  {                                                       // GLACIER_FLUX gla:942-1148
    MoState mo = {0.f, 0.f, 0.f, 0.f, 0.1f, 0};
    float h = 0.f, t, estg = 0.f, destg, csh = 0.f, cev = 0.f, rahb = 1.f;
    const float cir = emg * SB;
    const float df_top = at_top(df, s.isnow);
    const float cgh = 2.f * df_top / y.dzsnso[L(s.isnow + 1)];
    const float stc_top = y.stc[L(s.isnow + 1)];
    float& tgb = s.tg;
    const double r_rhocp = rc64(s.rhoair * CPAIR);
#pragma unroll 1
    for (int iter = 1; iter <= 5; iter++) {
      sfcdif1(s.err, iter, s.sfctmp, r_rhocp, h, s.qair, zlvl, zpd, z0m, ur, MPE, mo, s.cm, s.ch);
      rahb = nmp_max(1.f, 1.f / (s.ch * ur));
      float rawb = rahb;
      t = tdc(tgb);
      esat_sel(t, estg, destg);
      csh = s.rhoair * CPAIR / rahb;
      cev = s.rhoair * CPAIR / gamma / (rsurf + rawb);
      s.fira = cir * powi4(tgb) - emg * s.lwdn;
      s.fsh = csh * (tgb - s.sfctmp);
      s.fgev = cev * (estg * rhsur - s.eair);
      s.ssoil = cgh * (tgb - stc_top);
      float b = s.sag - s.fira - s.fsh - s.fgev - s.ssoil;
      float a = 4.f * cir * powi3(tgb) + csh + cev * destg + cgh;
      float dtg = b / a;
      s.fira = s.fira + 4.f * cir * powi3(tgb) * dtg;
      s.fsh = s.fsh + csh * dtg;
      s.fgev = s.fgev + cev * destg * dtg;
      s.ssoil = s.ssoil + cgh * dtg;
      tgb = tgb + dtg;
      h = csh * (tgb - s.sfctmp);
      t = tdc(tgb);
      { float dummy; esat_sel(t, estg, dummy); }
      s.qsfc = 0.622f * (estg * rhsur) / (s.sfcprs - 0.378f * (estg * rhsur));
    }
    float sicemax = -1.e30f;
#pragma unroll
    for (int k = 1; k <= NSOIL; k++) sicemax = nmp_max(sicemax, y.smc[L(k)] - y.sh2o[L(k)]);
    if (c.O.stc == 1) {
      if ((sicemax > 0.0f || s.snowh > 0.0f) && tgb > TFRZ) {
        tgb = TFRZ;
        s.fira = cir * powi4(tgb) - emg * s.lwdn;
        s.fsh = csh * (tgb - s.sfctmp);
        s.fgev = cev * (estg * rhsur - s.eair);
        s.ssoil = s.sag - (s.fira + s.fsh + s.fgev);
      }
    }
    float ehb2 = mo.fv * VKC / (nmp_logf((2.f + z0m) / z0m) - mo.fh2);
    s.chb2 = ehb2;
    if (ehb2 < 1.E-5f) {
      s.t2mb = tgb;
      s.q2b = s.qsfc;
    } else {
      s.t2mb = tgb - s.fsh / (s.rhoair * CPAIR) * 1.f / ehb2;
      s.q2b = s.qsfc - s.fgev / (lathea * s.rhoair) * (1.f / ehb2 + rsurf);
    }
    s.ch = 1.f / rahb;
  }
  float fire = s.lwdn + s.fira;
  if (fire <= 0.f) raise(s, NOAHMP_ERR_GLACIER_FIRE_NONPOSITIVE);
  s.emissi = emg;
  s.trad = pow_quarter((fire - (1 - s.emissi) * s.lwdn) / (s.emissi * SB));
  {
    Parm P = {};
    P.zbot = -8.0f;                                        // gla:260
    tsnosoi(c, P, s, y, df, hcpct);
  }
  if (c.O.stc == 2) {
    if (s.snowh > 0.05f && s.tg > TFRZ) s.tg = TFRZ;
  }
  phasechange_glacier(c, s, y, fact);
  // ---- back in NOAHMP_GLACIER gla:295-300
  float sice_save[NL], sh2o_save[NL];
#pragma unroll
  for (int k = 1; k <= NSOIL; k++) {
    float si = nmp_max(0.0f, y.smc[L(k)] - y.sh2o[L(k)]);
    y.sice[L(k)] = si; sice_save[L(k)] = si; sh2o_save[L(k)] = y.sh2o[L(k)];
  }
  s.sneqvo = s.sneqv;
  float qvap = nmp_max(s.fgev / lathea, 0.f);
  float qdew = fabsf(nmp_min(s.fgev / lathea, 0.f));
  s.edir = qvap - qdew;
  // ---- WATER_GLACIER gla:1924-2110
  float snoflow = 0.f;
  float fpice = 0.f;
  if (c.O.snf == 1) {
    if (s.sfctmp > TFRZ + 2.5f) fpice = 0.f;
    else if (s.sfctmp <= TFRZ + 0.5f) fpice = 1.0f;
    else if (s.sfctmp <= TFRZ + 2.f) fpice = 1.f - (-54.632f + 0.2f * s.sfctmp);
    else fpice = 0.6f;
  } else if (c.O.snf == 2) {
    fpice = (s.sfctmp >= TFRZ + 2.2f) ? 0.f : 1.0f;
  } else if (c.O.snf == 3) {
    fpice = (s.sfctmp >= TFRZ) ? 0.f : 1.0f;
  }
  s.fpice = fpice;
  float bdfall = nmp_min(120.f, 67.92f + 51.25f * nmp_expf((s.sfctmp - TFRZ) / 2.59f));
  float qrain = s.prcp * (1.f - fpice);
  s.qsnow = s.prcp * fpice;
  float snowhin = s.qsnow / bdfall;
  const float qsnsub = qvap, qsnfro = qdew;
  // SNOWWATER_GLACIER gla:2113-2237 (SNOWFALL_GLACIER: a new layer forms at 0.05 m)
  s.ponding1 = 0.0f; s.ponding2 = 0.0f;
  {
    int newnode = 0;
    if (s.isnow == 0 && s.qsnow > 0.f) {
      s.snowh = s.snowh + snowhin * dt;
      s.sneqv = s.sneqv + s.qsnow * dt;
    }
    if (s.isnow == 0 && s.qsnow > 0.f && s.snowh >= 0.05f) {
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
  if (s.isnow < 0) {
    compact(c, s, y);
    combine<true>(s, y);
    divide<true>(s, y);
  }
#pragma unroll
  for (int iz = -2; iz <= 0; iz++) {
    if (iz <= s.isnow) {
      y.snice[L(iz)] = 0.f; y.snliq[L(iz)] = 0.f; y.stc[L(iz)] = 0.f; y.dzsnso[L(iz)] = 0.f;
      y.zsnso[L(iz)] = 0.f;
    }
  }
  snowh2o<true>(c, s, y, qsnfro, qsnsub, qrain);
  NMP_ASSUME(s.isnow >= -NSNOW && s.isnow <= 0);
  if (s.sneqv > 2000.f) {
    float bdsnow = y.snice[L(0)] / y.dzsnso[L(0)];
    snoflow = (s.sneqv - 2000.f);
    y.snice[L(0)] = y.snice[L(0)] - snoflow;
    y.dzsnso[L(0)] = y.dzsnso[L(0)] - snoflow / bdsnow;
    snoflow = snoflow / dt;
  }
  if (s.isnow != 0) {
    float sw = 0.f;
#pragma unroll
    for (int iz = -2; iz <= 0; iz++)
      if (iz > s.isnow) sw = sw + y.snice[L(iz)] + y.snliq[L(iz)];
    s.sneqv = sw;
  }
  rebuild_layers(c, s, y);
  s.runsrf = (s.ponding + s.ponding1 + s.ponding2) / dt;
  if (s.isnow == 0) s.runsrf = s.runsrf + s.qsnbot + qrain;
  else s.runsrf = s.runsrf + s.qsnbot;
  float replace = 0.0f;
#pragma unroll
  for (int k = 1; k <= NSOIL; k++)
    replace = replace + y.dzsnso[L(k)] * (y.sice[L(k)] - sice_save[L(k)] + y.sh2o[L(k)] - sh2o_save[L(k)]);
  replace = replace * 1000.0f / dt;
#pragma unroll
  for (int k = 1; k <= NSOIL; k++) {
    float si = nmp_min(1.0f, sice_save[L(k)]);
    y.sice[L(k)] = si;
    y.sh2o[L(k)] = 1.0f - si;
  }
  s.runsub = snoflow + replace;
  // ERROR_GLACIER: the SW and energy checks are one-sided (no ABS), gla:2933,2943
  {
    float errsw = s.swdown - (s.fsa + s.fsr);
    if (errsw > 0.01f) raise(s, NOAHMP_ERR_GLACIER_SW_BALANCE);
    float erreng = s.sag - (s.fira + s.fsh + s.fgev + s.ssoil);
    if (erreng > 0.01f) raise(s, NOAHMP_ERR_GLACIER_ENERGY_BALANCE);
    float errwat = s.sneqv - beg_wb - (s.prcp - s.edir - s.runsrf - s.runsub) * dt;
    if (fabsf(errwat) > 0.1f) raise(s, NOAHMP_ERR_GLACIER_WATER_BALANCE);
  }
  if (s.snowh <= 1.E-6f || s.sneqv <= 1.E-3f) { s.snowh = 0.0f; s.sneqv = 0.0f; }
  s.albedo = (s.swdown != 0.f) ? (s.fsr / s.swdown) : -999.9f;
}

// sentinel fill of the Noah-MP outputs that have no meaning on land ice, drv:571-625
NMP_DEV void glacier_fill_undefined(Col& s) {
  const float U = -1.E36f, Z = 0.0f;
  s.fsno = 1.0f; s.tv = U; s.tgb = s.tg; s.canice = Z; s.canliq = Z; s.eah = U; s.tah = U; s.fwet = Z;
  s.wslake = Z; s.zwt = U; s.wa = U; s.wt = U; s.lfmass = Z; s.rtmass = Z; s.stmass = Z; s.wood = Z;
  s.stblcp = U; s.fastcp = U; s.lai = Z; s.sai = Z; s.t2mv = U; s.q2v = U; s.nee = Z; s.gpp = Z;
  s.npp = Z; s.fveg = 0.0f; s.ecan = Z; s.etran = Z; s.apar = Z; s.psn = Z; s.sav = Z; s.rssun = U;
  s.rssha = U; s.bgap = U; s.wgap = U; s.tgv = U; s.chv = U; s.chb = s.ch; s.irc = U; s.irg = U;
  s.shc = U; s.shg = U; s.evg = U; s.ghv = U; s.irb = s.fira; s.shb = s.fsh; s.evb = s.fgev;
  s.ghb = s.ssoil; s.tr = Z; s.evc = Z; s.chleaf = U; s.chuc = U; s.chv2 = U; s.fcev = Z; s.fctr = Z;
}

}  // namespace nmp

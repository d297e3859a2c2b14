/* TEST INFRASTRUCTURE (oracle) -- CPU restatement of the Miguez-Macho & Fan groundwater step,
 * reference phys/module_sf_noahmp_groundwater.F90 ("gw"): WTABLE_mmf_noahmp gw:14-198,
 * LATERALFLOW gw:201-295, UPDATEWTD gw:298-606.  float32, source operation order.
 * Pinned bit-exact against oracle/_ref (ref_wtable_mmf) by tests/test_groundwater.py.
 * Never linked into the product. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "noahmp_oracle.h"
#include "nmp_internal.h"

const noahmp_tables* nmp_oracle_tables(void);

/* gw:298-606.  zsoil[0..nsoil] (zsoil[0]=0), dzs/smc/sh2o/smceq 1-based views (index k-1). */
static void updatewtd(int nsoil, const real* dzs, const real* zsoil, const real* smceq, real smcmax,
                      real smcwlt, real psisat, real bexp, real* totwater_io, real* wtd_io, real* smc,
                      real* sh2o, real* smcwtd_io, real* qspring_out) {
#define DZS(k) dzs[(k) - 1]
#define SMC(k) smc[(k) - 1]
#define SMCEQ(k) smceq[(k) - 1]
  real totwater = *totwater_io, wtd = *wtd_io, smcwtd = *smcwtd_io, qspring = 0.f;
  real sice[NOAHMP_NSOIL], maxwatup, maxwatdw, wtdold, wgpmid, syielddw, dzup, smceqdeep;
  int k, k1, iwtd = 1, kwtd;
  (void)smcwlt;
  for (k = 1; k <= nsoil; k++) sice[k - 1] = smc[k - 1] - sh2o[k - 1];            /* gw:340 */

  if (totwater > 0.f) {                                                            /* gw:345 */
    if (wtd >= zsoil[nsoil]) {                                                     /* gw:348 */
      for (k = nsoil - 1; k >= 1; k--) if (wtd < zsoil[k]) break;
      iwtd = k; kwtd = iwtd + 1;
      maxwatup = DZS(kwtd) * (smcmax - SMC(kwtd));
      if (totwater <= maxwatup) {
        SMC(kwtd) = SMC(kwtd) + totwater / DZS(kwtd);
        SMC(kwtd) = MINF(SMC(kwtd), smcmax);
        if (SMC(kwtd) > SMCEQ(kwtd))
          wtd = MINF((SMC(kwtd) * DZS(kwtd) - SMCEQ(kwtd) * zsoil[iwtd] + smcmax * zsoil[kwtd]) /
                     (smcmax - SMCEQ(kwtd)), zsoil[iwtd]);
        totwater = 0.f;
      } else {                                                                     /* gw:366 */
        SMC(kwtd) = smcmax;
        totwater = totwater - maxwatup;
        k1 = iwtd;
        for (k = k1; k >= 0; k--) {
          wtd = zsoil[k];
          iwtd = k - 1;
          if (k == 0) break;
          maxwatup = DZS(k) * (smcmax - SMC(k));
          if (totwater <= maxwatup) {
            SMC(k) = SMC(k) + totwater / DZS(k);
            SMC(k) = MINF(SMC(k), smcmax);
            if (SMC(k) > SMCEQ(k))
              wtd = MINF((SMC(k) * DZS(k) - SMCEQ(k) * zsoil[iwtd] + smcmax * zsoil[k]) /
                         (smcmax - SMCEQ(k)), zsoil[iwtd]);
            totwater = 0.f;
            break;
          } else {
            SMC(k) = smcmax;
            totwater = totwater - maxwatup;
          }
        }
      }
    } else if (wtd >= zsoil[nsoil] - DZS(nsoil)) {                                 /* gw:392 */
      smceqdeep = smcmax * powf(psisat / (psisat - DZS(nsoil)), 1.f / bexp);
      smceqdeep = MAXF(smceqdeep, 1.E-4f);
      maxwatup = (smcmax - smcwtd) * DZS(nsoil);
      if (totwater <= maxwatup) {
        smcwtd = smcwtd + totwater / DZS(nsoil);
        smcwtd = MINF(smcwtd, smcmax);
        if (smcwtd > smceqdeep)
          wtd = MINF((smcwtd * DZS(nsoil) - smceqdeep * zsoil[nsoil] +
                      smcmax * (zsoil[nsoil] - DZS(nsoil))) / (smcmax - smceqdeep), zsoil[nsoil]);
        totwater = 0.f;
      } else {
        smcwtd = smcmax;
        totwater = totwater - maxwatup;
        for (k = nsoil; k >= 0; k--) {
          wtd = zsoil[k];
          iwtd = k - 1;
          if (k == 0) break;
          maxwatup = DZS(k) * (smcmax - SMC(k));
          if (totwater <= maxwatup) {
            SMC(k) = MINF(SMC(k) + totwater / DZS(k), smcmax);
            if (SMC(k) > SMCEQ(k))
              wtd = MINF((SMC(k) * DZS(k) - SMCEQ(k) * zsoil[iwtd] + smcmax * zsoil[k]) /
                         (smcmax - SMCEQ(k)), zsoil[iwtd]);
            totwater = 0.f;
            break;
          } else {
            SMC(k) = smcmax;
            totwater = totwater - maxwatup;
          }
        }
      }
    } else {                                                                       /* gw:432 deep */
      maxwatup = (smcmax - smcwtd) * (zsoil[nsoil] - DZS(nsoil) - wtd);
      if (totwater <= maxwatup) {
        wtd = wtd + totwater / (smcmax - smcwtd);
        totwater = 0.f;
      } else {
        totwater = totwater - maxwatup;
        wtd = zsoil[nsoil] - DZS(nsoil);
        maxwatup = (smcmax - smcwtd) * DZS(nsoil);
        if (totwater <= maxwatup) {
          smceqdeep = smcmax * powf(psisat / (psisat - DZS(nsoil)), 1.f / bexp);
          smceqdeep = MAXF(smceqdeep, 1.E-4f);
          smcwtd = smcwtd + totwater / DZS(nsoil);
          smcwtd = MINF(smcwtd, smcmax);
          wtd = (smcwtd * DZS(nsoil) - smceqdeep * zsoil[nsoil] +
                 smcmax * (zsoil[nsoil] - DZS(nsoil))) / (smcmax - smceqdeep);
          totwater = 0.f;
        } else {
          smcwtd = smcmax;
          totwater = totwater - maxwatup;
          for (k = nsoil; k >= 0; k--) {
            wtd = zsoil[k];
            iwtd = k - 1;
            if (k == 0) break;
            maxwatup = DZS(k) * (smcmax - SMC(k));
            if (totwater <= maxwatup) {
              SMC(k) = SMC(k) + totwater / DZS(k);
              SMC(k) = MINF(SMC(k), smcmax);
              if (SMC(k) > SMCEQ(k))
                wtd = (SMC(k) * DZS(k) - SMCEQ(k) * zsoil[iwtd] + smcmax * zsoil[k]) /
                      (smcmax - SMCEQ(k));
              totwater = 0.f;
              break;
            } else {
              SMC(k) = smcmax;
              totwater = totwater - maxwatup;
            }
          }
        }
      }
    }
    qspring = totwater;                                                            /* gw:483 */
  } else if (totwater < 0.f) {                                                     /* gw:486 */
    if (wtd >= zsoil[nsoil]) {
      for (k = nsoil - 1; k >= 1; k--) if (wtd < zsoil[k]) break;
      iwtd = k;
      k1 = iwtd + 1;
      for (kwtd = k1; kwtd <= nsoil; kwtd++) {
        maxwatdw = DZS(kwtd) * (SMC(kwtd) - MAXF(SMCEQ(kwtd), sice[kwtd - 1]));
        if (-totwater <= maxwatdw) {
          SMC(kwtd) = SMC(kwtd) + totwater / DZS(kwtd);
          if (SMC(kwtd) > SMCEQ(kwtd)) {
            wtd = (SMC(kwtd) * DZS(kwtd) - SMCEQ(kwtd) * zsoil[iwtd] + smcmax * zsoil[kwtd]) /
                  (smcmax - SMCEQ(kwtd));
          } else {
            wtd = zsoil[kwtd];
            iwtd = iwtd + 1;
          }
          totwater = 0.f;
          break;
        } else {
          wtd = zsoil[kwtd];
          iwtd = iwtd + 1;
          if (maxwatdw >= 0.f) {
            SMC(kwtd) = SMC(kwtd) + maxwatdw / DZS(kwtd);
            totwater = totwater + maxwatdw;
          }
        }
      }
      if (iwtd == nsoil && totwater < 0.f) {                                       /* gw:525 */
        smceqdeep = smcmax * powf(psisat / (psisat - DZS(nsoil)), 1.f / bexp);
        smceqdeep = MAXF(smceqdeep, 1.E-4f);
        maxwatdw = DZS(nsoil) * (smcwtd - smceqdeep);
        if (-totwater <= maxwatdw) {
          smcwtd = smcwtd + totwater / DZS(nsoil);
          wtd = MAXF((smcwtd * DZS(nsoil) - smceqdeep * zsoil[nsoil] +
                      smcmax * (zsoil[nsoil] - DZS(nsoil))) / (smcmax - smceqdeep),
                     zsoil[nsoil] - DZS(nsoil));
        } else {
          wtd = zsoil[nsoil] - DZS(nsoil);
          smcwtd = smcwtd + totwater / DZS(nsoil);
          dzup = (smceqdeep - smcwtd) * DZS(nsoil) / (smcmax - smceqdeep);
          wtd = wtd - dzup;
          smcwtd = smceqdeep;
        }
      }
    } else if (wtd >= zsoil[nsoil] - DZS(nsoil)) {                                 /* gw:556 */
      smceqdeep = smcmax * powf(psisat / (psisat - DZS(nsoil)), 1.f / bexp);
      smceqdeep = MAXF(smceqdeep, 1.E-4f);
      maxwatdw = DZS(nsoil) * (smcwtd - smceqdeep);
      if (-totwater <= maxwatdw) {
        smcwtd = smcwtd + totwater / DZS(nsoil);
        wtd = MAXF((smcwtd * DZS(nsoil) - smceqdeep * zsoil[nsoil] +
                    smcmax * (zsoil[nsoil] - DZS(nsoil))) / (smcmax - smceqdeep),
                   zsoil[nsoil] - DZS(nsoil));
      } else {
        wtd = zsoil[nsoil] - DZS(nsoil);
        smcwtd = smcwtd + totwater / DZS(nsoil);
        dzup = (smceqdeep - smcwtd) * DZS(nsoil) / (smcmax - smceqdeep);
        wtd = wtd - dzup;
        smcwtd = smceqdeep;
      }
    } else {                                                                       /* gw:585 */
      wgpmid = smcmax * powf(psisat / (psisat - (zsoil[nsoil] - wtd)), 1.f / bexp);
      wgpmid = MAXF(wgpmid, 1.E-4f);
      syielddw = smcmax - wgpmid;
      wtdold = wtd;
      wtd = wtdold + totwater / syielddw;
      smcwtd = (smcwtd * (zsoil[nsoil] - wtdold) + wgpmid * (wtdold - wtd)) / (zsoil[nsoil] - wtd);
    }
    qspring = 0.f;
  }
  for (k = 1; k <= nsoil; k++) sh2o[k - 1] = smc[k - 1] - sice[k - 1];             /* gw:603 */
  *totwater_io = totwater; *wtd_io = wtd; *smcwtd_io = smcwtd; *qspring_out = qspring;
#undef DZS
#undef SMC
#undef SMCEQ
}

/* gw:14-198 incl. LATERALFLOW gw:201-295 */
int nmp_oracle_wtable_mmf(const noahmp_wtable_args* a, noahmp_status* st) {
  static const real KLATFACTOR[19] = {2.f, 3.f, 4.f, 10.f, 10.f, 12.f, 14.f, 20.f, 24.f, 28.f, 40.f,
                                      48.f, 2.f, 0.f, 10.f, 0.f, 20.f, 2.f, 2.f};   /* gw:225 */
  const real FANGLE = 0.45508986056f;                                               /* gw:229 */
  const noahmp_tables* T = nmp_oracle_tables();
  int ni = a->ime - a->ims + 1, nj = a->jme - a->jms + 1, ns = a->nsoil;
  size_t n2 = (size_t)ni * nj;
  if (st) memset(st, 0, sizeof(*st));
  if (!T) return -1;
  if (ns != NOAHMP_NSOIL) { if (st) st->code = NOAHMP_ERR_NSOIL_UNSUPPORTED; return NOAHMP_ERR_NSOIL_UNSUPPORTED; }
  real* qlat = (real*)calloc(n2, sizeof(real));
  real* kcell = (real*)calloc(n2, sizeof(real));
  real* head = (real*)calloc(n2, sizeof(real));
  signed char* landmask = (signed char*)malloc(n2);
#define IX(i, j) ((size_t)((j) - a->jms) * ni + ((i) - a->ims))
#define IX3(i, k, j) (((size_t)((j) - a->jms) * ns + ((k) - 1)) * ni + ((i) - a->ims))
  real deltat = a->wtddt * 60.f;                                                    /* gw:89 */
  real zsoil[NOAHMP_NSOIL + 1];
  zsoil[0] = 0.f; zsoil[1] = -a->dzs[0];
  for (int k = 2; k <= ns; k++) zsoil[k] = -a->dzs[k - 1] + zsoil[k - 1];
  for (int j = a->jms; j <= a->jme; j++)                                            /* gw:97-101 */
    for (int i = a->ims; i <= a->ime; i++) {
      size_t x = IX(i, j);
      landmask[x] = (a->xland[x] - 1.5f < 0.f && a->xice[x] < a->xice_threshold &&
                     a->ivgtyp[x] != a->isice) ? 1 : -1;
    }
  /* LATERALFLOW gw:231-292 */
  int itsh = a->its - 1 > a->ids ? a->its - 1 : a->ids, iteh = a->ite + 1 < a->ide - 1 ? a->ite + 1 : a->ide - 1;
  int jtsh = a->jts - 1 > a->jds ? a->jts - 1 : a->jds, jteh = a->jte + 1 < a->jde - 1 ? a->jte + 1 : a->jde - 1;
  for (int j = jtsh; j <= jteh; j++)
    for (int i = itsh; i <= iteh; i++) {
      size_t x = IX(i, j);
      if (a->fdepth[x] > 0.f) {
        int st_ = a->isltyp[x];
        real klat = T->satdk[st_ - 1] * KLATFACTOR[st_ - 1];
        if (a->wtd[x] < -1.5f) kcell[x] = a->fdepth[x] * klat * expf((a->wtd[x] + 1.5f) / a->fdepth[x]);
        else kcell[x] = klat * (a->wtd[x] + 1.5f + a->fdepth[x]);
      } else kcell[x] = 0.f;
      head[x] = a->topo[x] + a->wtd[x];
    }
  itsh = a->its > a->ids + 1 ? a->its : a->ids + 1; iteh = a->ite < a->ide - 2 ? a->ite : a->ide - 2;
  jtsh = a->jts > a->jds + 1 ? a->jts : a->jds + 1; jteh = a->jte < a->jde - 2 ? a->jte : a->jde - 2;
  const real SQRT2 = sqrtf(2.f);
  for (int j = jtsh; j <= jteh; j++)
    for (int i = itsh; i <= iteh; i++) {
      size_t x = IX(i, j);
      if (landmask[x] > 0) {
        real q = 0.f, kc = kcell[x], hd = head[x];
        q = q + (kcell[IX(i - 1, j + 1)] + kc) * (head[IX(i - 1, j + 1)] - hd) / SQRT2;
        q = q + (kcell[IX(i - 1, j)] + kc) * (head[IX(i - 1, j)] - hd);
        q = q + (kcell[IX(i - 1, j - 1)] + kc) * (head[IX(i - 1, j - 1)] - hd) / SQRT2;
        q = q + (kcell[IX(i, j + 1)] + kc) * (head[IX(i, j + 1)] - hd);
        q = q + (kcell[IX(i, j - 1)] + kc) * (head[IX(i, j - 1)] - hd);
        q = q + (kcell[IX(i + 1, j + 1)] + kc) * (head[IX(i + 1, j + 1)] - hd) / SQRT2;
        q = q + (kcell[IX(i + 1, j)] + kc) * (head[IX(i + 1, j)] - hd);
        q = q + (kcell[IX(i + 1, j - 1)] + kc) * (head[IX(i + 1, j - 1)] - hd) / SQRT2;
        qlat[x] = FANGLE * q * deltat / a->area[x];
      }
    }
  /* river flux gw:114-129 */
  for (int j = a->jts; j <= a->jte; j++)
    for (int i = a->its; i <= a->ite; i++) {
      size_t x = IX(i, j);
      if (landmask[x] > 0) {
        real rcond;
        if (a->wtd[x] > a->riverbed[x] && a->eqwtd[x] > a->riverbed[x])
          rcond = a->rivercond[x] * expf(a->pexp[x] * (a->wtd[x] - a->eqwtd[x]));
        else rcond = a->rivercond[x];
        a->qrf[x] = rcond * (a->wtd[x] - a->riverbed[x]) * deltat / a->area[x];
        a->qrf[x] = MAXF(a->qrf[x], 0.f);
      } else a->qrf[x] = 0.f;
    }
  /* column update gw:132-182 */
  int nland = 0;
  for (int j = a->jts; j <= a->jte; j++)
    for (int i = a->its; i <= a->ite; i++) {
      size_t x = IX(i, j);
      if (landmask[x] <= 0) continue;
      nland++;
      int sl = a->isltyp[x];
      real bexp = T->bb[sl - 1], dksat = T->satdk[sl - 1], smcmax = T->maxsmc[sl - 1];
      real psisat = -T->satpsi[sl - 1], smcwlt = T->wltsmc[sl - 1];
      if (a->ivgtyp[x] == a->isurban) { smcmax = 0.45f; smcwlt = 0.40f; }
      if (a->wtd[x] < zsoil[ns] - a->dzs[ns - 1]) {                                 /* gw:147-161 */
        real ddz = zsoil[ns] - a->wtd[x];
        real smcwtdmid = 0.5f * (a->smcwtd[x] + smcmax);
        real psi = psisat * powf(smcmax / a->smcwtd[x], bexp);
        real wcnddeep = dksat * powf(smcwtdmid / smcmax, 2.0f * bexp + 3.0f);
        real wfluxdeep = -deltat * wcnddeep * ((psisat - psi) / ddz - 1.f);
        a->smcwtd[x] = a->smcwtd[x] + (a->deeprech[x] - wfluxdeep) / ddz;
        real wplus = MAXF((a->smcwtd[x] - smcmax), 0.0f) * ddz;
        real wminus = MAXF((1.E-4f - a->smcwtd[x]), 0.0f) * ddz;
        a->smcwtd[x] = MAXF(MINF(a->smcwtd[x], smcmax), 1.E-4f);
        wfluxdeep = wfluxdeep + wplus - wminus;
        a->deeprech[x] = wfluxdeep;
      }
      real totwater = qlat[x] - a->qrf[x] + a->deeprech[x];                         /* gw:165 */
      real smc[NOAHMP_NSOIL], sh2o[NOAHMP_NSOIL], smceq[NOAHMP_NSOIL];
      for (int k = 1; k <= ns; k++) {
        smc[k - 1] = a->smois[IX3(i, k, j)]; sh2o[k - 1] = a->sh2oxy[IX3(i, k, j)];
        smceq[k - 1] = a->smoiseq[IX3(i, k, j)];
      }
      updatewtd(ns, a->dzs, zsoil, smceq, smcmax, smcwlt, psisat, bexp, &totwater, &a->wtd[x], smc, sh2o,
                &a->smcwtd[x], &a->qspring[x]);
      for (int k = 1; k <= ns; k++) { a->smois[IX3(i, k, j)] = smc[k - 1]; a->sh2oxy[IX3(i, k, j)] = sh2o[k - 1]; }
    }
  for (int j = a->jts; j <= a->jte; j++)                                            /* gw:186-195 */
    for (int i = a->its; i <= a->ite; i++) {
      size_t x = IX(i, j);
      a->qslat[x] = a->qslat[x] + qlat[x] * 1.E3f;
      a->qrfs[x] = a->qrfs[x] + a->qrf[x] * 1.E3f;
      a->qsprings[x] = a->qsprings[x] + a->qspring[x] * 1.E3f;
      a->rech[x] = a->rech[x] + a->deeprech[x] * 1.E3f;
      a->deeprech[x] = 0.f;
    }
  if (st) { st->n_land = nland; st->n_skipped = (a->ite - a->its + 1) * (a->jte - a->jts + 1) - nland; }
  free(qlat); free(kcell); free(head); free(landmask);
  return 0;
}

/* ---------------------------------------------------------------------------------------------
 * GROUNDWATER_INIT + EQSMOISTURE, reference phys/module_sf_noahmpdrv.F90:1286-1522 ("drv").
 * Same argument block as the time step; SMOISEQ is written.  Caller passes ide+1 / jde+1 (hdrv:291). */
static void eqsmoisture(int nsoil, const real* zsoil /*1-based view: zsoil[k]*/, real smcmax, real dwsat, real dksat,
                        real bexp, real* smceq /*1-based*/) {
  for (int k = 1; k <= nsoil; k++) {                                               /* drv:1491-1519 */
    real ddz;
    if (k == 1) ddz = -zsoil[k + 1] * 0.5f;
    else if (k < nsoil) ddz = (zsoil[k - 1] - zsoil[k + 1]) * 0.5f;
    else ddz = zsoil[k - 1] - zsoil[k];
    real expon = bexp + 1.f;
    real aa = dwsat / ddz;
    real bb = dksat / powf(smcmax, expon);
    real smc = 0.5f * smcmax;
    for (int iter = 1; iter <= 100; iter++) {
      real func = (smc - smcmax) * aa + bb * powf(smc, expon);
      real dfunc = aa + bb * expon * powf(smc, bexp);
      real dx = func / dfunc;
      smc = smc - dx;
      if (fabsf(dx) < 1.E-6f) break;
    }
    smceq[k] = MINF(MAXF(smc, 1.E-4f), smcmax * 0.99f);
  }
}

int nmp_oracle_groundwater_init(const noahmp_wtable_args* a, int iswater, noahmp_status* st) {
  static const real KLATFACTOR[19] = {2.f, 3.f, 4.f, 10.f, 10.f, 12.f, 14.f, 20.f, 24.f, 28.f, 40.f,
                                      48.f, 2.f, 0.f, 10.f, 0.f, 20.f, 2.f, 2.f};
  const real FANGLE = 0.45508986056f;
  const noahmp_tables* T = nmp_oracle_tables();
  int ni = a->ime - a->ims + 1, nj = a->jme - a->jms + 1, ns = a->nsoil;
  size_t n2 = (size_t)ni * nj;
  if (st) memset(st, 0, sizeof(*st));
  if (!T) return -1;
  real* smoiseq = (real*)a->smoiseq;
  real* qlat = (real*)calloc(n2, sizeof(real));
  real* qrf = (real*)calloc(n2, sizeof(real));
  real* kcell = (real*)calloc(n2, sizeof(real));
  real* head = (real*)calloc(n2, sizeof(real));
  signed char* landmask = (signed char*)malloc(n2);
  int itf = a->ite < a->ide - 1 ? a->ite : a->ide - 1, jtf = a->jte < a->jde - 1 ? a->jte : a->jde - 1;   /* drv:1330 */
  real deltat = a->wtddt * 60.f;
  real zsoil[NOAHMP_NSOIL + 1];                                                     /* ZSOIL(1:NSOIL), argument of the reference */
  zsoil[0] = 0.f; zsoil[1] = -a->dzs[0];
  for (int k = 2; k <= ns; k++) zsoil[k] = zsoil[k - 1] - a->dzs[k - 1];            /* drv:1139-1142 */
  for (size_t x = 0; x < n2; x++) landmask[x] = (a->ivgtyp[x] != iswater && a->ivgtyp[x] != a->isice) ? 1 : -1;
  /* LATERALFLOW gw:201-295 with this land mask */
  int itsh = a->its - 1 > a->ids ? a->its - 1 : a->ids, iteh = a->ite + 1 < a->ide - 1 ? a->ite + 1 : a->ide - 1;
  int jtsh = a->jts - 1 > a->jds ? a->jts - 1 : a->jds, jteh = a->jte + 1 < a->jde - 1 ? a->jte + 1 : a->jde - 1;
  for (int j = jtsh; j <= jteh; j++)
    for (int i = itsh; i <= iteh; i++) {
      size_t x = IX(i, j);
      if (a->fdepth[x] > 0.f) {
        int s_ = a->isltyp[x];
        real klat = T->satdk[s_ - 1] * KLATFACTOR[s_ - 1];
        if (a->wtd[x] < -1.5f) kcell[x] = a->fdepth[x] * klat * expf((a->wtd[x] + 1.5f) / a->fdepth[x]);
        else kcell[x] = klat * (a->wtd[x] + 1.5f + a->fdepth[x]);
      } else kcell[x] = 0.f;
      head[x] = a->topo[x] + a->wtd[x];
    }
  itsh = a->its > a->ids + 1 ? a->its : a->ids + 1; iteh = a->ite < a->ide - 2 ? a->ite : a->ide - 2;
  jtsh = a->jts > a->jds + 1 ? a->jts : a->jds + 1; jteh = a->jte < a->jde - 2 ? a->jte : a->jde - 2;
  const real SQRT2 = sqrtf(2.f);
  for (int j = jtsh; j <= jteh; j++)
    for (int i = itsh; i <= iteh; i++) {
      size_t x = IX(i, j);
      if (landmask[x] > 0) {
        real q = 0.f, kc = kcell[x], hd = head[x];
        q = q + (kcell[IX(i - 1, j + 1)] + kc) * (head[IX(i - 1, j + 1)] - hd) / SQRT2;
        q = q + (kcell[IX(i - 1, j)] + kc) * (head[IX(i - 1, j)] - hd);
        q = q + (kcell[IX(i - 1, j - 1)] + kc) * (head[IX(i - 1, j - 1)] - hd) / SQRT2;
        q = q + (kcell[IX(i, j + 1)] + kc) * (head[IX(i, j + 1)] - hd);
        q = q + (kcell[IX(i, j - 1)] + kc) * (head[IX(i, j - 1)] - hd);
        q = q + (kcell[IX(i + 1, j + 1)] + kc) * (head[IX(i + 1, j + 1)] - hd) / SQRT2;
        q = q + (kcell[IX(i + 1, j)] + kc) * (head[IX(i + 1, j)] - hd);
        q = q + (kcell[IX(i + 1, j - 1)] + kc) * (head[IX(i + 1, j - 1)] - hd) / SQRT2;
        qlat[x] = FANGLE * q * deltat / a->area[x];
      }
    }
  for (int j = a->jts; j <= jtf; j++)                                               /* drv:1356-1370 */
    for (int i = a->its; i <= itf; i++) {
      size_t x = IX(i, j);
      if (landmask[x] > 0) {
        real rcond;
        if (a->wtd[x] > a->riverbed[x] && a->eqwtd[x] > a->riverbed[x])
          rcond = a->rivercond[x] * expf(a->pexp[x] * (a->wtd[x] - a->eqwtd[x]));
        else rcond = a->rivercond[x];
        qrf[x] = rcond * (a->wtd[x] - a->riverbed[x]) * deltat / a->area[x];
        qrf[x] = MAXF(qrf[x], 0.f);
      } else qrf[x] = 0.f;
    }
  for (int j = a->jts; j <= jtf; j++)                                               /* drv:1373-1461 */
    for (int i = a->its; i <= itf; i++) {
      size_t x = IX(i, j);
      int sl = a->isltyp[x];
      real bx = T->bb[sl - 1], smcmax = T->maxsmc[sl - 1];
      if (a->ivgtyp[x] == a->isurban) smcmax = 0.45f;
      real dwsat = T->satdw[sl - 1], dksat = T->satdk[sl - 1], psisat = -T->satpsi[sl - 1];
      if (bx > 0.0f && smcmax > 0.0f && -psisat > 0.0f) {
        real smceq[NOAHMP_NSOIL + 1];
        eqsmoisture(ns, zsoil, smcmax, dwsat, dksat, bx, smceq);
        for (int k = 1; k <= ns; k++) smoiseq[IX3(i, k, j)] = smceq[k];
        if (a->wtd[x] < zsoil[ns] - a->dzs[ns - 1]) {
          real expon = 2.f * bx + 3.f;
          real ddz = zsoil[ns] - a->wtd[x];
          real cc = psisat / ddz;
          real flux = (qlat[x] - qrf[x]) / deltat;
          real smc = 0.5f * smcmax;
          for (int iter = 1; iter <= 100; iter++) {
            real dd = (smc + smcmax) / (2.f * smcmax);
            real aa = -dksat * powf(dd, expon);
            real bbb = cc * (powf(smcmax / smc, bx) - 1.f) + 1.f;
            real func = aa * bbb - flux;
            real dfunc = -dksat * (expon / (2.f * smcmax)) * powf(dd, expon - 1.f) * bbb +
                         aa * cc * (-bx) * powf(smcmax, bx) * powf(smc, -bx - 1.f);
            real dx = func / dfunc;
            smc = smc - dx;
            if (fabsf(dx) < 1.E-6f) break;
          }
          a->smcwtd[x] = MAXF(smc, 1.E-4f);
        } else if (a->wtd[x] < zsoil[ns]) {
          real smceqdeep = smcmax * powf(psisat / (psisat - a->dzs[ns - 1]), 1.f / bx);
          smceqdeep = MAXF(smceqdeep, 1.E-4f);
          a->smcwtd[x] = smcmax * (a->wtd[x] - (zsoil[ns] - a->dzs[ns - 1])) + smceqdeep * (zsoil[ns] - a->wtd[x]);
        } else {
          a->smcwtd[x] = smcmax;
          for (int k = ns; k >= 2; k--) {
            if (a->wtd[x] >= zsoil[k - 1]) {
              real frliq = a->sh2oxy[IX3(i, k, j)] / a->smois[IX3(i, k, j)];
              a->smois[IX3(i, k, j)] = smcmax;
              a->sh2oxy[IX3(i, k, j)] = smcmax * frliq;
            } else {
              if (a->smois[IX3(i, k, j)] < smceq[k]) a->wtd[x] = zsoil[k];
              else a->wtd[x] = (a->smois[IX3(i, k, j)] * a->dzs[k - 1] - smceq[k] * zsoil[k - 1] + smcmax * zsoil[k]) /
                               (smcmax - smceq[k]);
              break;
            }
          }
        }
      } else {
        for (int k = 1; k <= ns; k++) smoiseq[IX3(i, k, j)] = smcmax;
        a->smcwtd[x] = smcmax;
        a->wtd[x] = 0.f;
      }
      a->deeprech[x] = 0.f; a->rech[x] = 0.f; a->qslat[x] = 0.f; a->qrfs[x] = 0.f; a->qsprings[x] = 0.f;
    }
  free(qlat); free(qrf); free(kcell); free(head); free(landmask);
  return 0;
}

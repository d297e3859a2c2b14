/* TEST INFRASTRUCTURE (oracle) -- CPU restatement of the cold start: NOAHMP_INIT (reference
 * phys/module_sf_noahmpdrv.F90:988-1134, "drv", restart=.false., table reading excluded) and SNOW_INIT
 * (drv:1182-1283).  Loop structure follows the reference (field by field over the tile).  Pinned bit-exact
 * against oracle/_ref (ref_noahmp_init) by tests/test_init.py.  Never linked into the product. */
#include <math.h>
#include <string.h>
#include "noahmp_oracle.h"
#include "nmp_internal.h"

const noahmp_tables* nmp_oracle_tables(void);

int nmp_oracle_init(const noahmp_step_args* a, int iswater, int fndsnowh, noahmp_status* st) {
  const noahmp_tables* T = nmp_oracle_tables();
  const real HLICE = 3.335E5f, GRAV_ = 9.81f, T0 = 273.15f;     /* drv:965-967 */
  const int ni = a->ime - a->ims + 1, ns = a->nsoil;
  const int itf = a->ite < a->ide - 1 ? a->ite : a->ide - 1;    /* drv:991-992 */
  const int jtf = a->jte < a->jde - 1 ? a->jte : a->jde - 1;
  (void)iswater;
  if (st) memset(st, 0, sizeof(*st));
  if (!T) return -1;
#define X2(i, j) ((size_t)((j) - a->jms) * ni + ((i) - a->ims))
#define X3(i, k, j, nk, k0) (((size_t)((j) - a->jms) * (nk) + ((k) - (k0))) * ni + ((i) - a->ims))
  if (!fndsnowh)                                                /* drv:997-1005 */
    for (int j = a->jts; j <= jtf; j++)
      for (int i = a->its; i <= itf; i++) a->snowh[X2(i, j)] = a->snow[X2(i, j)] * 0.005f;
  for (int j = a->jts; j <= jtf; j++)                           /* drv:1007-1020 */
    for (int i = a->its; i <= itf; i++)
      if (a->isltyp[X2(i, j)] < 1) {
        if (st) { st->code = NOAHMP_ERR_SOILTYP_RANGE; st->i = i; st->j = j; }
        return NOAHMP_ERR_SOILTYP_RANGE;
      }
  for (int j = a->jts; j <= jtf; j++)                           /* drv:1032-1069 */
    for (int i = a->its; i <= itf; i++) {
      size_t x = X2(i, j);
      if (a->ivgtyp[x] == a->isice && a->xice[x] <= 0.0f) {
        for (int k = 1; k <= ns; k++) {
          a->smois[X3(i, k, j, ns, 1)] = 1.0f;
          a->sh2o[X3(i, k, j, ns, 1)] = 0.0f;
          a->tslb[X3(i, k, j, ns, 1)] = MINF(a->tslb[X3(i, k, j, ns, 1)], 263.15f);
        }
        a->snow[x] = MAXF(a->snow[x], 10.0f);
        a->snowh[x] = a->snow[x] * 0.01f;
      } else {
        int sl = a->isltyp[x];
        real bx = T->bb[sl - 1], smcmax = T->maxsmc[sl - 1];
        for (int k = 1; k <= ns; k++)
          if (a->smois[X3(i, k, j, ns, 1)] > smcmax) a->smois[X3(i, k, j, ns, 1)] = smcmax;
        real psisat = T->satpsi[sl - 1];
        if (bx > 0.0f && smcmax > 0.0f && psisat > 0.0f) {
          for (int k = 1; k <= ns; k++) {
            real t = a->tslb[X3(i, k, j, ns, 1)];
            if (t < 273.149f) {
              real fk = powf((HLICE / (GRAV_ * (-psisat))) * ((t - T0) / t), -1.f / bx) * smcmax;
              fk = MAXF(fk, 0.02f);
              a->sh2o[X3(i, k, j, ns, 1)] = MINF(fk, a->smois[X3(i, k, j, ns, 1)]);
            } else a->sh2o[X3(i, k, j, ns, 1)] = a->smois[X3(i, k, j, ns, 1)];
          }
        } else {
          for (int k = 1; k <= ns; k++) a->sh2o[X3(i, k, j, ns, 1)] = a->smois[X3(i, k, j, ns, 1)];
        }
      }
    }
  for (int j = a->jts; j <= jtf; j++)                           /* drv:1073-1134 */
    for (int i = a->its; i <= itf; i++) {
      size_t x = X2(i, j);
      int warm = a->snow[x] > 0.0f && a->tsk[x] > 273.15f;
      a->tvxy[x] = a->tsk[x];   if (warm) a->tvxy[x] = 273.15f;
      a->tgxy[x] = a->tsk[x];   if (warm) a->tgxy[x] = 273.15f;
      a->canwat[x] = 0.0f;
      a->canliqxy[x] = a->canwat[x];
      a->canicexy[x] = 0.f;
      a->eahxy[x] = 2000.f;
      a->tahxy[x] = a->tsk[x];  if (warm) a->tahxy[x] = 273.15f;
      a->t2mvxy[x] = a->tsk[x]; if (warm) a->t2mvxy[x] = 273.15f;
      a->t2mbxy[x] = a->tsk[x]; if (warm) a->t2mbxy[x] = 273.15f;
      a->cmxy[x] = 0.0f; a->chxy[x] = 0.0f; a->fwetxy[x] = 0.0f; a->sneqvoxy[x] = 0.0f;
      a->alboldxy[x] = 0.65f; a->qsnowxy[x] = 0.0f; a->wslakexy[x] = 0.0f;
      if (a->iopt_run != 5) {
        a->waxy[x] = 4900.f;
        a->wtxy[x] = a->waxy[x];
        a->zwtxy[x] = (25.f + 2.0f) - a->waxy[x] / 1000 / 0.2f;
      } else {
        a->waxy[x] = 0.f;
        a->wtxy[x] = 0.f;
      }
      a->lfmassxy[x] = 50.f; a->stmassxy[x] = 50.0f; a->rtmassxy[x] = 500.0f; a->woodxy[x] = 500.0f;
      a->stblcpxy[x] = 1000.0f; a->fastcpxy[x] = 1000.0f; a->xsaixy[x] = 0.1f;
    }
  real zsoil[NOAHMP_NSOIL + 1];                                 /* drv:1139-1142, 1-based */
  zsoil[1] = -a->dzs[0];
  for (int k = 2; k <= ns; k++) zsoil[k] = zsoil[k - 1] - a->dzs[k - 1];
  /* SNOW_INIT drv:1182-1283 (NSNOW = 3) */
  real dzsno[3] = {0.f, 0.f, 0.f};                              /* DZSNO(-2:0); kept across columns like the reference's local */
  real dzsnso[3 + NOAHMP_NSOIL];                                /* DZSNSO(-2:NSOIL) */
#define DZ(iz) dzsno[(iz) + 2]
#define DS(iz) dzsnso[(iz) + 2]
  for (int j = a->jts; j <= jtf; j++)
    for (int i = a->its; i <= itf; i++) {
      size_t x = X2(i, j);
      real sd = a->snowh[x];
      int isn;
      if (sd < 0.025f) { isn = 0; DZ(-2) = 0.f; DZ(-1) = 0.f; DZ(0) = 0.f; }
      else if (sd >= 0.025f && sd <= 0.05f) { isn = -1; DZ(0) = sd; }
      else if (sd > 0.05f && sd <= 0.10f) { isn = -2; DZ(-1) = sd / 2.f; DZ(0) = sd / 2.f; }
      else if (sd > 0.10f && sd <= 0.25f) { isn = -2; DZ(-1) = 0.05f; DZ(0) = sd - DZ(-1); }
      else if (sd > 0.25f && sd <= 0.45f) { isn = -3; DZ(-2) = 0.05f; DZ(-1) = 0.5f * (sd - DZ(-2)); DZ(0) = 0.5f * (sd - DZ(-2)); }
      else if (sd > 0.45f) { isn = -3; DZ(-2) = 0.05f; DZ(-1) = 0.20f; DZ(0) = sd - DZ(-1) - DZ(-2); }
      else return -2;                                           /* drv:1245 wrf_error_fatal (NaN depth) */
      a->isnowxy[x] = isn;
      for (int iz = -2; iz <= 0; iz++) {
        a->tsnoxy[X3(i, iz, j, 3, -2)] = 0.f; a->snicexy[X3(i, iz, j, 3, -2)] = 0.f; a->snliqxy[X3(i, iz, j, 3, -2)] = 0.f;
      }
      for (int iz = isn + 1; iz <= 0; iz++) {
        a->tsnoxy[X3(i, iz, j, 3, -2)] = a->tgxy[x];
        a->snliqxy[X3(i, iz, j, 3, -2)] = 0.00f;
        a->snicexy[X3(i, iz, j, 3, -2)] = 1.00f * DZ(iz) * (a->snow[x] / sd);
      }
      for (int iz = isn + 1; iz <= 0; iz++) DS(iz) = -DZ(iz);
      DS(1) = zsoil[1];
      for (int iz = 2; iz <= ns; iz++) DS(iz) = zsoil[iz] - zsoil[iz - 1];
      a->zsnsoxy[X3(i, isn + 1, j, ns + 3, -2)] = DS(isn + 1);
      for (int iz = isn + 2; iz <= ns; iz++)
        a->zsnsoxy[X3(i, iz, j, ns + 3, -2)] = a->zsnsoxy[X3(i, iz - 1, j, ns + 3, -2)] + DS(iz);
    }
  if (st) st->n_land = (itf - a->its + 1) * (jtf - a->jts + 1);
  return 0;
}

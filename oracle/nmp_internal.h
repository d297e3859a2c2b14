/* TEST INFRASTRUCTURE (oracle): shared helpers of the C restatement. */
#ifndef NMP_INTERNAL_H
#define NMP_INTERNAL_H
#include "noahmp_oracle.h"

#define MINF(a, b) (((a) < (b)) ? (a) : (b))
#define MAXF(a, b) (((a) > (b)) ? (a) : (b))

/* x**n with INTEGER n: square-and-multiply, the order compiler-rt's __powisf2 (what flang
 * emits for REAL**INTEGER) uses, so T**4 == (T*T)*(T*T) bit-for-bit */
static inline real powi(real a, int b) {
  int recip = b < 0;
  real r = 1.f;
  for (;;) {
    if (b & 1) r *= a;
    b /= 2;
    if (b == 0) break;
    a *= a;
  }
  return recip ? 1.f / r : r;
}

/* intermediates that NOAHMP_SFLX hands from ATM/PHENOLOGY/ENERGY to WATER/ERROR (lsm:547-760) */
typedef struct {
  real thair, qair, eair, rhoair, qprecc, qprecl, solad[2], solai[2], swdown;
  real dzsnso[NL];
  real elai, esai, htop, igs, troot;
  real snicev[NL], snliqv[NL], epore[NL];
  real btrani[NL], btran, latheav, latheag, qmelt, fsrv, fsrg;
  int  imelt[NL], frozen_canopy, frozen_ground;
  real sice[NL];
} nmp_work;

typedef struct { real moz, fm, fh, fm2, fh2, fv; int mozsgn; } mo_state;
real nmp_tdc(real t);
void nmp_snow_age(real dt, real tg, real sneqvo, real sneqv, real* tauss, real* fage);
void nmp_sfcdif1(nmp_ctx* c, int iter, real sfctmp, real rhoair, real h, real qair, real zlvl, real zpd,
                 real z0m, real z0h, real ur, real mpe, mo_state* s, real* cm, real* ch, real* ch2);
void nmp_tsnosoi(const nmp_ctx* c, int isnow, real tbot, const real* zsnso, real ssoil, const real* df,
                 const real* hcpct, real zbot, real dt, real snowh, real* stc);
void nmp_combine(int glacier, int* isnow, real* sh2o, real* stc, real* snice, real* snliq, real* dzsnso,
                 real* sice, real* snowh, real* sneqv, real* ponding1, real* ponding2);
void nmp_divide(int nsnow, real dz2max, int* isnow, real* stc, real* snice, real* snliq, real* dzsnso);
void nmp_compact(real dt, const real* stc, const real* snice, const real* snliq, const int* imelt,
                 const real* ficeold, int isnow, real* dzsnso);
void nmp_snowh2o(const nmp_ctx* c, int glacier, real dt, real qsnfro, real qsnsub, real qrain, int* isnow,
                 real* dzsnso, real* snowh, real* sneqv, real* snice, real* snliq, real* sh2o, real* sice,
                 real* stc, real* qsnbot, real* ponding1, real* ponding2);
void nmp_esat(real t, real* esw, real* esi, real* desw, real* desi);
void nmp_rosr12(real* p, const real* a, const real* b, real* cc, const real* d, real* delta, int ntop,
                int nsoil);
void nmp_energy(nmp_ctx* c, nmp_column* s, nmp_work* w);
void nmp_water(nmp_ctx* c, nmp_column* s, nmp_work* w, real qvap, real qdew);
void nmp_carbon(nmp_ctx* c, nmp_column* s, nmp_work* w);

#endif

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

void nmp_esat(real t, real* esw, real* esi, real* desw, real* desi);
void nmp_rosr12(real* p, const real* a, const real* b, real* cc, const real* d, real* delta, int ntop,
                int nsoil);
void nmp_energy(nmp_ctx* c, nmp_column* s, nmp_work* w);
void nmp_water(nmp_ctx* c, nmp_column* s, nmp_work* w, real qvap, real qdew);
void nmp_carbon(nmp_ctx* c, nmp_column* s, nmp_work* w);

#endif

/* TEST INFRASTRUCTURE (oracle) -- the checker's EXP, pinned.
 *
 * The reference's EXP resolves to libm's expf (flang runtime -> glibc; SURVEY 8c "third-party arithmetic": glibc 2.35 in this
 * image, Ubuntu GLIBC 2.35-0ubuntu3.11).  glibc's x86-64 libm carries TWO builds of sysdeps/ieee754/flt-32/e_expf.c and picks
 * one per host at load time (ifunc, sysdeps/x86_64/fpu/multiarch/e_expf.c): `__expf_fma` (the same C source compiled with
 * -mfma -mavx2, so the compiler contracts a*b+c) on every CPU with FMA, `__expf_sse2` elsewhere.  An exhaustive comparison of
 * both over all 2^32 arguments (tools/expf_variants.c) shows: contraction inside the polynomial never changes a result; the one
 * contraction that does is  r = InvLn2N*xd - kd  ->  fma(InvLn2N, xd, -kd), and it changes exactly two arguments:
 *      x = 0x4202422f (32.5646324):  fma 0x56fc9f1c   sse2 0x56fc9f1b
 *      x = 0xc27c65d9 (-63.0994606): fma 0x11fa2993   sse2 0x11fa2992
 * So "the reference's bits" depend on the host at those two arguments.  The pin: this file restates e_expf.c as `__expf_fma`
 * evaluates it (the variant of the dev container's Xeon and of the MI355X boxes' EPYC 9575F, i.e. what the committed fixtures
 * were generated with and what the compiled reference returns on those hosts), with __builtin_fma so that the result does not
 * depend on the host the checker runs on.  The device code (noahmp_amd/csrc/nmp_libm.hpp::expf_) evaluates the same variant;
 * tests/test_libm.py compares both with the live libm over the argument space (0 mismatches on an FMA host, exactly the two
 * arguments above elsewhere) and checks the two discriminating arguments explicitly.
 *
 * Algorithm (glibc 2.35 e_expf.c:36-109, math_config.h EXP2F_TABLE_BITS = 5): exp(x) = 2^(k/32) * 2^(r/32) with
 * k = round(x * 32/ln2) by the shift trick, a 32-entry table of 2^(i/32) and a degree-3 polynomial, all in float64, one
 * rounding to float32 at the end; |x| beyond the overflow / underflow thresholds is resolved before (e_expf.c:52-66).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "nmp_pin_expf_tab.h"

static inline uint32_t pin_asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint64_t pin_asuint64(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double pin_asdouble(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

float nmp_pin_expf(float x) {
  const double xd = (double)x;
  const uint32_t abstop = (pin_asuint(x) >> 20) & 0x7ff;                         /* e_expf.c:48 top12(x) & 0x7ff */
  if (abstop >= 0x42b) {                                                         /* |x| >= 88 or x is nan (e_expf.c:50) */
    if (pin_asuint(x) == 0xff800000u) return 0.0f;                               /* exp(-inf) */
    if (abstop >= 0x7f8) return x + x;                                           /* nan, +inf */
    if (x > 0x1.62e42ep6f) return INFINITY;                                      /* x > log(0x1p128): overflow (e_expf.c:58) */
    if (x < -0x1.9fe368p6f) return 0.0f;                                         /* x < log(0x1p-150): underflow (e_expf.c:60) */
  }
  const double z = pin_invln2_scaled * xd;                                       /* e_expf.c:69 */
  double kd = z + pin_shift;                                                     /* e_expf.c:79-81 (no TOINT intrinsics on x86) */
  const uint64_t ki = pin_asuint64(kd);
  kd -= pin_shift;
  const double r = __builtin_fma(pin_invln2_scaled, xd, -kd);                    /* e_expf.c:83 as __expf_fma contracts it */
  uint64_t t = pin_exp2f_tab[ki % 32];                                           /* e_expf.c:86-88 */
  t += ki << (52 - 5);
  const double s = pin_asdouble(t);
  const double zz = pin_poly_scaled[0] * r + pin_poly_scaled[1];                 /* e_expf.c:89-93 (contraction here never shows) */
  const double r2 = r * r;
  double y = pin_poly_scaled[2] * r + 1.0;
  y = zz * r2 + y;
  y = y * s;
  return (float)y;
}

/* 1 if the host libm's expf is the pinned variant (decided by the two discriminating arguments), 0 if it is the other build */
int nmp_pin_expf_host_variant_is_pinned(void) {
  const uint32_t a[2] = {0x4202422fu, 0xc27c65d9u};
  for (int i = 0; i < 2; i++) {
    float x, y, p;
    memcpy(&x, &a[i], 4);
    y = expf(x);
    p = nmp_pin_expf(x);
    if (memcmp(&y, &p, 4)) return 0;
  }
  return 1;
}

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <pthread.h>
static const uint64_t T[32] = {
  0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
  0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
  0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
  0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
  0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
  0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
  0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
  0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
static const double SHIFT = 0x1.8p52, INV = 0x1.71547652b82fep+5;
static const double C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;
static inline uint64_t au(double d){uint64_t u;memcpy(&u,&d,8);return u;}
static inline double ad(uint64_t u){double d;memcpy(&d,&u,8);return d;}
// mode bit0: fma in polynomial; bit1: r = fma(INV, xd, -kd)
static inline float e(float x, int mode) {
  double xd = x, z = INV * xd, kd = z + SHIFT; uint64_t ki = au(kd); kd -= SHIFT;
  double r = (mode & 2) ? __builtin_fma(INV, xd, -kd) : z - kd;
  uint64_t t = T[ki % 32]; t += ki << 47; double s = ad(t);
  double zz, y, r2 = r * r;
  if (mode & 1) { zz = __builtin_fma(C0, r, C1); y = __builtin_fma(C2, r, 1.0); y = __builtin_fma(zz, r2, y); }
  else { zz = C0 * r + C1; y = C2 * r + 1.0; y = zz * r2 + y; }
  if (mode & 4) { /* y = z*r2 + y as fma, but y=C2*r+1 unfused */ }
  y = y * s;
  return (float)y;
}
typedef struct { uint32_t lo, hi; long bad[8]; uint32_t first[8][4]; } job;
static void* run(void* p) {
  job* j = p;
  for (uint64_t u = j->lo; u < j->hi; u++) {
    float x; uint32_t b = (uint32_t)u; memcpy(&x, &b, 4);
    if (!(x == x) || x > 88.7f || x < -103.9f) continue;
    float l = expf(x);
    for (int m = 0; m < 4; m++) {
      float v = e(x, m);
      if (memcmp(&v, &l, 4)) { if (j->bad[m] < 4) j->first[m][j->bad[m]] = b; j->bad[m]++; }
    }
    float v0 = e(x,0), v1 = e(x,1);
    if (memcmp(&v0,&v1,4)) { if (j->bad[4] < 4) j->first[4][j->bad[4]] = b; j->bad[4]++; }
  }
  return 0;
}
int main() {
  enum {N = 8}; pthread_t th[N]; static job jb[N];
  for (int i = 0; i < N; i++) { jb[i].lo = (uint32_t)((1ull << 32) / N * i); jb[i].hi = i == N-1 ? 0xFFFFFFFFu : (uint32_t)((1ull << 32) / N * (i + 1)); pthread_create(&th[i], 0, run, &jb[i]); }
  long tot[8] = {0};
  for (int i = 0; i < N; i++) { pthread_join(th[i], 0); for (int m = 0; m < 5; m++) { tot[m] += jb[i].bad[m]; for (int k = 0; k < jb[i].bad[m] && k < 4; k++) printf("mode %d mismatch at 0x%08x\n", m, jb[i].first[m][k]); } }
  for (int m = 0; m < 5; m++) printf("mode %d: %ld mismatches vs live libm (mode 4: nofma vs polyfma)\n", m, tot[m]);
  return 0;
}

// Host experiment (round 5; build: hipcc -O2 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Inoahmp_amd/csrc -Iinclude THIS -o powerr -lpthread):
// could X ** 0.5, X ** 0.25, X ** -0.25 (RAGRB's CWPC, SFCDIF1's / RAGRB's stability functions: four powf per canopy iteration) be evaluated by
// square roots and still return the reference libm's bits?  Over ALL positive normal float32 arguments: the relative error of the float64 value the
// reference powf forms before its final rounding, how often its result is not the correctly rounded power, and how many arguments have their
// true power within that error of a float32 rounding boundary (there a square-root evaluation cannot know which way the reference rounds).
// Result (profiles/r05_experiments.md section 6): error 2^-33.2; 0.064 % of results are not correctly rounded; 0.29 % of arguments are undecidable
// = 17 % of wavefronts per call (31 % for the two-argument call) would take the full powf anyway.  Estimated gain <= 1 % per exponent: not pursued.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <thread>
#include <vector>
#include "nmp_libm.hpp"
using namespace nmp::libm;
int main(int argc, char** argv) {
  const int nt = 8;
  const float ys[3] = {0.5f, 0.25f, -0.25f};
  for (int yi = 0; yi < 3; yi++) {
    const float y = ys[yi];
    std::vector<long double> maxrel(nt, 0);
    std::vector<long> ndiff(nt, 0), nuns30(nt, 0), nuns33(nt, 0), nuns36(nt, 0), cnt(nt, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([&, t]() {
      const uint64_t lo0 = 0x00800000ull, hi0 = 0x7f800000ull;
      const uint64_t lo = lo0 + (hi0 - lo0) * t / nt, hi = lo0 + (hi0 - lo0) * (t + 1) / nt;
      for (uint64_t u = lo; u < hi; u += 1) {
        const float x = asfloat((uint32_t)u);
        const double logx = powf_log2_inline((uint32_t)u);
        const double d = powf_exp2_inline((double)y * logx, 0);
        long double tr = sqrtl((long double)x);
        if (yi >= 1) tr = sqrtl(tr);
        if (yi == 2) tr = 1.0L / tr;
        const long double rel = fabsl(((long double)d - tr) / tr);
        if (rel > maxrel[t]) maxrel[t] = rel;
        const float g = (float)d, c = (float)tr;
        if (asuint(g) != asuint(c)) ndiff[t]++;
        const float up = nextafterf(c, INFINITY), dn = nextafterf(c, -INFINITY);
        const long double m1 = ((long double)c + up) / 2, m2 = ((long double)c + dn) / 2;
        const long double dist = fminl(fabsl(tr - m1), fabsl(tr - m2)) / tr;
        if (dist < 0x1p-30L) nuns30[t]++;
        if (dist < 0x1p-33L) nuns33[t]++;
        if (dist < 0x1p-36L) nuns36[t]++;
        cnt[t]++;
      }
    });
    for (auto& q : th) q.join();
    long double mr = 0; long nd = 0, n30 = 0, n33 = 0, n36 = 0, n = 0;
    for (int t = 0; t < nt; t++) { if (maxrel[t] > mr) mr = maxrel[t]; nd += ndiff[t]; n30 += nuns30[t]; n33 += nuns33[t]; n36 += nuns36[t]; n += cnt[t]; }
    printf("y=%g: n=%ld  max rel err of the double = 2^%.2Lf  results != correctly rounded: %ld (%.3e)  true value within 2^-30 / 2^-33 / 2^-36 of a midpoint: %.3e / %.3e / %.3e\n",
           y, n, log2l(mr), nd, (double)nd / n, (double)n30 / n, (double)n33 / n, (double)n36 / n);
  }
  return 0;
}

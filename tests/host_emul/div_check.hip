// TEST INFRASTRUCTURE -- checks the exact-division helpers of noahmp_amd/csrc/nmp_dev_common.hpp (div_rc / rc64) against
// IEEE float32 division.  Never shipped.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <thread>
#include <vector>
#include "nmp_dev_common.hpp"

using namespace nmp;

static inline uint32_t bits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float fromb(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }
static inline bool same(float a, float b) { return (a != a && b != b) || bits(a) == bits(b); }

// Host: all numerators x = start, start+stride, ... of the 2^32 bit patterns against one divisor c with r = 1/(double)c.
// out[0] = mismatches whose exact quotient is in the normal range or special (must be 0); out[1] = mismatches below the normal
// range (|x/c| < 2^-126: exact ties of the gradual-underflow grid, the documented exception); first_bad = a numerator of kind 0.
extern "C" void div_check_host(float c, uint32_t stride, int nthreads, long* out, uint32_t* first_bad) {
  std::vector<long> b0(nthreads, 0), b1(nthreads, 0);
  std::vector<uint32_t> fb(nthreads, 0);
  std::vector<std::thread> th;
  const double r = 1.0 / (double)c;
  for (int t = 0; t < nthreads; t++)
    th.emplace_back([&, t]() {
      const uint64_t lo = (uint64_t)t * (1ull << 32) / nthreads, hi = (uint64_t)(t + 1) * (1ull << 32) / nthreads;
      for (uint64_t u = lo + (stride - lo % stride) % stride; u < hi; u += stride) {
        const float x = fromb((uint32_t)u);
        volatile float q = x / c;
        const float p = div_rc(x, r);
        if (!same(p, q)) {
          if (fabs((double)x / (double)c) < 0x1p-126) b1[t]++;
          else { if (!b0[t]) fb[t] = (uint32_t)u; b0[t]++; }
        }
      }
    });
  for (auto& x : th) x.join();
  out[0] = out[1] = 0;
  for (int t = 0; t < nthreads; t++) { if (b0[t] && !out[0]) *first_bad = fb[t]; out[0] += b0[t]; out[1] += b1[t]; }
}

// GPU: rc64 over ALL 2^32 divisors -- max |y r - 1| (exact residual by fma) and the special values -- and div_rc(x, rc64(y)) against
// the device's own IEEE division for n pseudo-random (x, y) bit patterns plus every divisor paired with a few fixed numerators.
__global__ void rc64_kernel(double* maxerr, unsigned long long* bad) {
  double mx = 0.0;
  unsigned long long nb = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * blockDim.x) {
    const float y = __uint_as_float((unsigned)i);
    const double r = rc64(y);
    if (y != y) { if (r == r) nb++; }
    else if (y == 0.f) { if (!(isinf(r) && signbit(r) == signbit(y))) nb++; }
    else if (isinf(y)) { if (!(r == 0.0 && signbit(r) == signbit(y))) nb++; }
    else { const double e = fabs(__builtin_fma((double)y, r, -1.0)); if (e > mx) mx = e; }
    // three numerators per divisor through the whole helper
    const float xs[3] = {1.0f, 2.5104E06f, __uint_as_float((unsigned)(i * 2654435761u))};
    for (int k = 0; k < 3; k++) {
      const float q = xs[k] / y, p = div_rc(xs[k], r);
      const bool eq = (p != p && q != q) || __float_as_uint(p) == __float_as_uint(q);
      if (!eq && !(fabsf(q) < 0x1p-126f)) nb++;
    }
  }
  atomicMax((unsigned long long*)maxerr, (unsigned long long)__double_as_longlong(mx));
  if (nb) atomicAdd(bad, nb);
}

extern "C" long div_check_gpu(double* maxerr_out) {
  double* d_e = nullptr; unsigned long long* d_b = nullptr;
  if (hipMalloc(&d_e, 8) != hipSuccess || hipMalloc(&d_b, 8) != hipSuccess) return -1;
  if (hipMemset(d_e, 0, 8) != hipSuccess || hipMemset(d_b, 0, 8) != hipSuccess) return -1;
  hipLaunchKernelGGL(rc64_kernel, dim3(4096), dim3(256), 0, 0, d_e, d_b);
  unsigned long long b = 0;
  if (hipMemcpy(maxerr_out, d_e, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (hipMemcpy(&b, d_b, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  hipFree(d_e); hipFree(d_b);
  return (long)b;
}

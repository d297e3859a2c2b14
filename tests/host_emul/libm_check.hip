// TEST INFRASTRUCTURE -- compares noahmp_amd/csrc/nmp_libm.hpp (host compilation of the device source) with
// the live libm of this machine, bit for bit.  Never shipped.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <thread>
#include <vector>
#include "nmp_libm.hpp"

using namespace nmp::libm;

static inline bool same(float a, float b) {
  if (isnan(a) && isnan(b)) return true;
  return asuint(a) == asuint(b);
}

// the batched forms (logfN_<3>, expfN_<4>, powfN_<2>): the argument under test sits at position pos of the batch, the other
// positions hold arguments derived from its bits (specials included, since the walk covers every class of bit pattern)
__host__ __device__ static inline float batched_unary(int fn, uint32_t bits, int pos) {
  if (fn == 12) {
    float a[3], o[3];
    for (int q = 0; q < 3; q++) a[(pos + q) % 3] = asfloat(q == 0 ? bits : (q == 1 ? bits ^ 0x00400000u : bits * 2654435761u));
    logfN_<3>(a, o);
    return o[pos % 3];
  }
  float a[4], o[4];
  for (int q = 0; q < 4; q++) a[(pos + q) % 4] = asfloat(q == 0 ? bits : (q == 1 ? bits ^ 0x80000000u : (q == 2 ? bits ^ 0x00400000u : bits * 2654435761u)));
  expfN_<4>(a, o);
  return o[pos % 4];
}

// fn: 0 expf, 1 logf, 2 log10f, 3 atanf, 4 tanhf, 5 expm1f, 7 acosf, 8 tanf, 9 cosf, 10 sinf (|x| < 120), 11 atanf_ge1_ (SFCDIF1's form of atanf), 12 logfN_<3>, 13 expfN_<4> (batched forms; 14 = powfN_<2> on the GPU).  Walks bit patterns start, start+stride, ... over the whole 2^32 space.
extern "C" long libm_check_unary(int fn, uint32_t stride, int nthreads, uint32_t* first_bad) {
  std::vector<long> bad(nthreads, 0);
  std::vector<uint32_t> fb(nthreads, 0);
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++)
    th.emplace_back([&, t]() {
      const uint64_t lo = (uint64_t)t * (1ull << 32) / nthreads, hi = (uint64_t)(t + 1) * (1ull << 32) / nthreads;
      for (uint64_t u = lo + (stride - lo % stride) % stride; u < hi; u += stride) {
        const float x = asfloat((uint32_t)u);
        float a, b;
        switch (fn) {
          case 0: a = expf_(x); b = ::expf(x); break;
          case 1: a = logf_(x); b = ::logf(x); break;
          case 2: a = log10f_(x); b = ::log10f(x); break;
          case 3: a = atanf_(x); b = ::atanf(x); break;
          case 11: a = atanf_ge1_(x); b = ::atanf(x); break;
          case 12: a = batched_unary(12, (uint32_t)u, (int)(u / stride % 3)); b = ::logf(x); break;
          case 13: a = batched_unary(13, (uint32_t)u, (int)(u / stride % 4)); b = ::expf(x); break;
          case 4: a = tanhf_(x); b = ::tanhf(x); break;
          case 7: a = acosf_(x); b = ::acosf(x); break;
          case 8: a = tanf_(x); b = (fabsf(x) < 120.0f) ? ::tanf(x) : a; break;
          case 9: a = cosf_(x); b = (fabsf(x) < 120.0f) ? ::cosf(x) : a; break;
          case 10: a = sinf_(x); b = (fabsf(x) < 120.0f) ? ::sinf(x) : a; break;
          default: a = expm1f_(x); b = ::expm1f(x); break;
        }
        if (!same(a, b)) { if (!bad[t]) fb[t] = (uint32_t)u; bad[t]++; }
      }
    });
  for (auto& x : th) x.join();
  long n = 0;
  for (int t = 0; t < nthreads; t++) { if (bad[t] && !n) *first_bad = fb[t]; n += bad[t]; }
  return n;
}

static inline uint64_t splitmix(uint64_t& s) {
  uint64_t z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// mode 0: x, y = random bit patterns (all specials included); mode 1: x in (0, 2^20), y in (-32, 32)
// (the model's range: positive bases, moderate exponents); mode 2: specials cross product.
extern "C" long libm_check_pow(int mode, long n, int nthreads, uint32_t* bad_xy) {
  std::vector<long> bad(nthreads, 0);
  std::vector<uint32_t> bx(nthreads, 0), by(nthreads, 0);
  std::vector<std::thread> th;
  static const uint32_t sp[] = {0x00000000u, 0x80000000u, 0x3f800000u, 0xbf800000u, 0x7f800000u, 0xff800000u,
                                0x7fc00000u, 0x00000001u, 0x80000001u, 0x007fffffu, 0x00800000u, 0x7f7fffffu,
                                0xff7fffffu, 0x40000000u, 0xc0000000u, 0x40400000u, 0xc0400000u, 0x3f000000u,
                                0xbf000000u, 0x4b800000u, 0xcb800001u, 0x4b000001u, 0x3eaaaaabu, 0x42fc0000u,
                                0xc3160000u, 0x43000000u};
  const int nsp = sizeof(sp) / sizeof(sp[0]);
  for (int t = 0; t < nthreads; t++)
    th.emplace_back([&, t]() {
      uint64_t s = 0x1234567ull * (t + 1) + mode;
      // mode 3: the constant real exponents of the reference (X**2., **3., **4., **0.5, **0.25, **(-0.25), **1.7, **(2./3.))
      // over every n-th positive finite float
      static const float fixed_y[] = {2.0f, 3.0f, 4.0f, 0.5f, 0.25f, -0.25f, 1.7f, 2.f / 3.f};
      const long cnt = mode == 2 ? (long)nsp * nsp : mode == 3 ? (long)(0x7f800000u / (uint32_t)n) : n / nthreads;
      for (long c = 0; c < cnt; c++) {
        float x, y;
        if (mode == 2) {
          if (t) break;
          x = asfloat(sp[c / nsp]); y = asfloat(sp[c % nsp]);
        } else if (mode == 3) {
          if (c % nthreads != t) continue;
          x = asfloat((uint32_t)c * (uint32_t)n + 1u);
          for (int q = 0; q < 7; q++) {
            const float a = powf_(x, fixed_y[q]), b = ::powf(x, fixed_y[q]);
            if (!same(a, b)) { if (!bad[t]) { bx[t] = asuint(x); by[t] = asuint(fixed_y[q]); } bad[t]++; }
          }
          y = fixed_y[7];
        } else {
          const uint64_t r = splitmix(s);
          if (mode == 0) { x = asfloat((uint32_t)r); y = asfloat((uint32_t)(r >> 32)); }
          else {
            x = ldexpf((float)((uint32_t)r >> 8) * 0x1p-24f + 0x1p-25f, (int)((r >> 32) % 40) - 19);
            y = ((float)((uint32_t)(r >> 40)) * 0x1p-24f - 0.5f) * 64.f;
          }
        }
        const float a = powf_(x, y), b = ::powf(x, y);
        if (!same(a, b)) { if (!bad[t]) { bx[t] = asuint(x); by[t] = asuint(y); } bad[t]++; }
        // the batched forms on the same arguments: powfN_<2>, powf_pairN_<2> (shared log2 x), powf_constbaseN_<1> (log2 x given)
        const float xs[2] = {x, asfloat(asuint(x) ^ 0x00200000u)}, ys[2] = {y, -y};
        float o[2], o1[2], o2[2];
        powfN_<2>(xs, ys, o);
        powf_pairN_<2>(xs, y, ys[1], o1, o2);
        int nb = !same(o[0], b) + !same(o[1], ::powf(xs[1], ys[1])) + !same(o1[0], b) + !same(o1[1], ::powf(xs[1], y)) +
                 !same(o2[0], ::powf(x, ys[1])) + !same(o2[1], ::powf(xs[1], ys[1]));
        if (asuint(x) - 0x00800000u < 0x7f000000u) {
          const double l2[1] = {powf_log2_k(asuint(x))};
          float oc[1];
          powf_constbaseN_<1>(&x, l2, &y, oc);
          nb += !same(oc[0], b);
        }
        if (nb) { if (!bad[t]) { bx[t] = asuint(x); by[t] = asuint(y); } bad[t] += nb; }
      }
    });
  for (auto& x : th) x.join();
  long nb = 0;
  for (int t = 0; t < nthreads; t++) { if (bad[t] && !nb) { bad_xy[0] = bx[t]; bad_xy[1] = by[t]; } nb += bad[t]; }
  return nb;
}

// ---- the same routines evaluated ON THE GPU against this machine's libm (needs a device)
__global__ void libm_eval_kernel(int fn, uint32_t start, uint32_t stride, long n, const float* __restrict__ y,
                                 float* __restrict__ out) {
  libm_stage_tables();
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = asfloat(start + (uint32_t)i * stride);
  float r;
  switch (fn) {
    case 0: r = expf_(x); break;
    case 1: r = logf_(x); break;
    case 2: r = log10f_(x); break;
    case 3: r = atanf_(x); break;
    case 11: r = atanf_ge1_(x); break;
    case 12: r = batched_unary(12, start + (uint32_t)i * stride, (int)(i % 3)); break;
    case 13: r = batched_unary(13, start + (uint32_t)i * stride, (int)(i % 4)); break;
    case 14: {                                  // powfN_<2>: (x, y) at position i % 2, a derived pair at the other
      float xs[2], ys[2], o[2];
      const int p = (int)(i & 1);
      xs[p] = x; ys[p] = y[i]; xs[1 - p] = asfloat(asuint(x) ^ 0x00200000u); ys[1 - p] = -y[i];
      powfN_<2>(xs, ys, o);
      r = o[p];
      break;
    }
    case 4: r = tanhf_(x); break;
    case 5: r = expm1f_(x); break;
    case 7: r = acosf_(x); break;
    case 8: r = tanf_(x); break;
    case 9: r = cosf_(x); break;
    case 10: r = sinf_(x); break;
    default: r = powf_(x, y[i]); break;
  }
  out[i] = r;
}

// fn 0..5 unary as above, 6 = powf with y drawn from a splitmix stream.  Returns mismatches, -1 on HIP error.
extern "C" long libm_gpu_check(int fn, uint32_t start, uint32_t stride, long n, uint32_t* first_bad) {
  float *d_out = nullptr, *d_y = nullptr;
  std::vector<float> out(n), y;
  if (hipMalloc(&d_out, n * sizeof(float)) != hipSuccess) return -1;
  if (fn == 6 || fn == 14) {
    y.resize(n);
    uint64_t s = 99;
    for (long i = 0; i < n; i++) y[i] = ((float)((uint32_t)(splitmix(s) >> 40)) * 0x1p-24f - 0.5f) * 64.f;
    if (hipMalloc(&d_y, n * sizeof(float)) != hipSuccess) return -1;
    if (hipMemcpy(d_y, y.data(), n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return -1;
  }
  hipLaunchKernelGGL(libm_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, fn, start, stride, n, d_y, d_out);
  if (hipMemcpy(out.data(), d_out, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  hipFree(d_out);
  if (d_y) hipFree(d_y);
  long bad = 0;
  for (long i = 0; i < n; i++) {
    const float x = asfloat(start + (uint32_t)i * stride);
    float b;
    switch (fn) {
      case 0: b = ::expf(x); break;
      case 1: b = ::logf(x); break;
      case 2: b = ::log10f(x); break;
      case 3: b = ::atanf(x); break;
      case 11: b = ::atanf(x); break;
      case 12: b = ::logf(x); break;
      case 13: b = ::expf(x); break;
      case 4: b = ::tanhf(x); break;
      case 5: b = ::expm1f(x); break;
      case 7: b = ::acosf(x); break;
      case 8: b = (fabsf(x) < 120.0f) ? ::tanf(x) : out[i]; break;
      case 9: b = (fabsf(x) < 120.0f) ? ::cosf(x) : out[i]; break;
      case 10: b = (fabsf(x) < 120.0f) ? ::sinf(x) : out[i]; break;
      default: b = ::powf(x, y[i]); break;
    }
    if (!same(out[i], b)) { if (!bad) *first_bad = start + (uint32_t)i * stride; bad++; }
  }
  return bad;
}

// one argument, host compilation of the device header: the result's bit pattern (fn as in libm_check_unary)
extern "C" uint32_t libm_eval_unary(int fn, uint32_t xbits) {
  const float x = asfloat(xbits);
  float r = 0.f;
  switch (fn) {
    case 0: r = expf_(x); break;
    case 1: r = logf_(x); break;
    case 2: r = log10f_(x); break;
    case 3: r = atanf_(x); break;
    case 11: r = atanf_ge1_(x); break;
    case 12: r = batched_unary(12, xbits, 0); break;
    case 13: r = batched_unary(13, xbits, 0); break;
    case 4: r = tanhf_(x); break;
    case 5: r = expm1f_(x); break;
  }
  return asuint(r);
}

// one argument evaluated on the GPU: the result's bit pattern, -1 on a HIP error
extern "C" long libm_gpu_eval_unary(int fn, uint32_t xbits) {
  float* d_out = nullptr;
  float out = 0.f;
  if (hipMalloc(&d_out, sizeof(float)) != hipSuccess) return -1;
  hipLaunchKernelGGL(libm_eval_kernel, dim3(1), dim3(64), 0, 0, fn, xbits, 0u, 1L, (const float*)nullptr, d_out);
  if (hipMemcpy(&out, d_out, sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  hipFree(d_out);
  return (long)asuint(out);
}

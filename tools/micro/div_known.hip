// Micro-benchmark (round 3): what a division by a KNOWN divisor costs on MI355X when it is replaced by an exact sequence.
//  0 x / c                        the compiler's IEEE sequence (2 div_scale, rcp, 6 fma/mul, div_fmas, div_fixup)
//  1 (float)((double)x * rcd)     rcd = RN64(1/c): correctly rounded for EVERY x (error 2^-52 < 2^-49, the minimum distance of a
//                                 float32 quotient from a rounding boundary); cvt + mul_f64 + cvt
//  2 fma(x, ch, x*cl)             Brisebarre-Muller multiplication by the constant 1/c = ch + cl (needs a per-divisor proof)
//  3 q=x*rc; r=fma(-c,q,x); q=fma(r,rc,q)              Markstein, one correction
//  4 ... two corrections (5 operations)
//  5 v_cvt_f64_f32 alone   6 v_mul_f64 alone   7 v_cvt_f32_f64 alone
#include <hip/hip_runtime.h>
#include <stdio.h>
extern __shared__ float dyn[];
template <int OP>
__global__ void __launch_bounds__(64) k_op(float* out, int iters, float c, double rcd, float ch, float cl) {
  float x[8];
  double d[8];
  for (int i = 0; i < 8; i++) { x[i] = 1.0f + threadIdx.x + i; d[i] = x[i]; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (OP == 0) x[i] = x[i] / c;
        if (OP == 1) x[i] = (float)((double)x[i] * rcd);
        if (OP == 2) x[i] = __builtin_fmaf(x[i], ch, x[i] * cl);
        if (OP == 3) { float q = x[i] * ch; float r = __builtin_fmaf(-c, q, x[i]); x[i] = __builtin_fmaf(r, ch, q); }
        if (OP == 4) { float q = x[i] * ch; float r = __builtin_fmaf(-c, q, x[i]); q = __builtin_fmaf(r, ch, q);
                       r = __builtin_fmaf(-c, q, x[i]); x[i] = __builtin_fmaf(r, ch, q); }
        if (OP == 5) { d[i] = (double)x[i]; x[i] = __builtin_bit_cast(float, (int)__builtin_bit_cast(long long, d[i]) ^ __builtin_bit_cast(int, x[i])); }
        if (OP == 6) d[i] = d[i] * rcd;
        if (OP == 7) { x[i] = (float)d[i]; d[i] = __builtin_bit_cast(double, __builtin_bit_cast(long long, d[i]) ^ (long long)__builtin_bit_cast(int, x[i])); }
      }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += x[i] + (float)d[i];
  if (s == 12345.f) out[0] = s + dyn[0];
}
template <int OP>
double run(int waves, int iters, float* out) {
  const int lds = 160 * 1024 / (4 * waves) - 512;
  const int blocks = 256 * 4 * waves;
  const float c = 3600.0f;
  const double rcd = 1.0 / (double)c;
  const float ch = (float)rcd, cl = (float)(rcd - (double)ch);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(64), lds, 0, out, 10, c, rcd, ch, cl);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(64), lds, 0, out, iters, c, rcd, ch, cl);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 * 2.4e9 / ((double)iters * 64) / waves;      // SIMD cycles per operation (nominal clock)
}
int main() {
  float* out; (void)hipMalloc(&out, 4);
  const int iters = 5000;
  for (int w = 2; w <= 2; w++)
    printf("SIMD cycles per operation at %d waves/SIMD (nominal 2.4 GHz): x/c %.1f | f64 route %.1f | fma(x,ch,x*cl) %.1f | Markstein-3 %.1f | Markstein-5 %.1f | "
           "cvt_f64_f32(+xor) %.1f  mul_f64 %.1f  cvt_f32_f64(+xor) %.1f\n", w, run<0>(w, iters, out), run<1>(w, iters, out), run<2>(w, iters, out),
           run<3>(w, iters, out), run<4>(w, iters, out), run<5>(w, iters, out), run<6>(w, iters, out), run<7>(w, iters, out));
  return 0;
}

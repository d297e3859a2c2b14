// Micro-benchmark: issue cost of the instructions of the IEEE float32 division sequence on MI355X (2 waves per SIMD, independent streams).
#include <hip/hip_runtime.h>
#include <stdio.h>
extern __shared__ float dyn[];
template <int OP>
__global__ void __launch_bounds__(64) k_op(float* out, int iters, float b0) {
  float x[8], b[8];
  for (int i = 0; i < 8; i++) { x[i] = 1.0f + threadIdx.x + i; b[i] = b0 + 0.001f * i; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (OP == 0) x[i] = __builtin_fmaf(x[i], b[i], b[i]);
        if (OP == 1) x[i] = __builtin_amdgcn_rcpf(x[i]);
        if (OP == 2) { bool f; x[i] = __builtin_amdgcn_div_scalef(x[i], b[i], true, &f); }
        if (OP == 3) x[i] = __builtin_amdgcn_div_fmasf(x[i], b[i], b[i], (threadIdx.x & 1) != 0);
        if (OP == 4) x[i] = __builtin_amdgcn_div_fixupf(x[i], b[i], b[i]);
        if (OP == 5) x[i] = x[i] * b[i];
        if (OP == 6) x[i] = (x[i] < b[i]) ? x[i] + 1.0f : b[i];
      }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  if (s == 12345.f) out[0] = s + dyn[0];
}
template <int OP>
double run(int waves, int iters, float* out) {
  const int lds = 160 * 1024 / (4 * waves) - 512;
  const int blocks = 256 * 4 * waves;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(64), lds, 0, out, 10, 1.0001f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(64), lds, 0, out, iters, 1.0001f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 * 2.4e9 / ((double)iters * 64) / waves;      // SIMD cycles per instruction (nominal clock)
}
int main() {
  float* out; (void)hipMalloc(&out, 4);
  const int iters = 20000, w = 2;
  printf("SIMD cycles per wave64 instruction at %d waves/SIMD (nominal 2.4 GHz): v_fma_f32 %.2f  v_rcp_f32 %.2f  v_div_scale_f32 %.2f  v_div_fmas_f32(+vcc) %.2f  "
         "v_div_fixup_f32 %.2f  v_mul_f32 %.2f  cmp+cndmask+add %.2f\n", w, run<0>(w, iters, out), run<1>(w, iters, out), run<2>(w, iters, out),
         run<3>(w, iters, out), run<4>(w, iters, out), run<5>(w, iters, out), run<6>(w, iters, out));
  return 0;
}

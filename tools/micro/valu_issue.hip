// Micro-benchmark: VALU issue rate of one SIMD as a function of resident waves (MI355X).  Each wave runs a long loop of
// independent v_fma_f32 (8 accumulator chains); occupancy is set by the dynamic LDS request.  Prints wave-instructions per
// cycle per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/micro/valu_issue.hip -o valu_issue && ./valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
extern __shared__ float dyn[];
template <int DEP>
__global__ void __launch_bounds__(64) k_fma(float* out, int iters, float a, float b) {
  float x[8];
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) x[DEP ? 0 : i] = __builtin_fmaf(x[DEP ? 0 : i], a, b);
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  if (s == 12345.f) out[0] = s + dyn[0];
}
__global__ void __launch_bounds__(64) k_f64(float* out, int iters, double a, double b) {
  double x[8];
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = __builtin_fma(x[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  if (s == 12345.0) out[0] = (float)s + dyn[0];
}
__global__ void __launch_bounds__(64) k_rcp(float* out, int iters, float a) {
  float x[8];
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x + i + 1.5f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = __builtin_amdgcn_rcpf(x[i]);
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  if (s == 12345.f) out[0] = s + dyn[0];
}
template <class F>
double run(F launch, int waves_per_simd, int iters) {
  // one wave per block; LDS request limits blocks per CU: 160 KB / (4 SIMDs * waves)
  const int lds = 160 * 1024 / (4 * waves_per_simd) - 512;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 4 * waves_per_simd;      // exactly one resident set
  launch(blocks, lds, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch(blocks, lds, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  float* out; hipMalloc(&out, 4);
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("clock %d kHz\n", clk);
  const int iters = 20000;
  const double insts_per_wave = (double)iters * 16 * 8;
  for (int w : {1, 2, 3, 4, 8}) {
    double ms = run([&](int b, int lds, int it) { hipLaunchKernelGGL(k_fma<0>, dim3(b), dim3(64), lds, 0, out, it, 1.0001f, 0.5f); }, w, iters);
    double msd = run([&](int b, int lds, int it) { hipLaunchKernelGGL(k_fma<1>, dim3(b), dim3(64), lds, 0, out, it, 1.0001f, 0.5f); }, w, iters);
    double ms64 = run([&](int b, int lds, int it) { hipLaunchKernelGGL(k_f64, dim3(b), dim3(64), lds, 0, out, it, 1.0001, 0.5); }, w, iters);
    double msr = run([&](int b, int lds, int it) { hipLaunchKernelGGL(k_rcp, dim3(b), dim3(64), lds, 0, out, it, 1.0f); }, w, iters);
    const double cyc = (double)clk * 1e3;   // cycles per second at the reported clock
    printf("waves/SIMD %d: v_fma_f32 independent %.3f inst/cycle/SIMD (%.2f cycles/inst/wave) | dependent chain %.3f | v_fma_f64 %.3f | v_rcp_f32 %.3f\n", w,
           insts_per_wave * w / (ms * 1e-3 * cyc), (ms * 1e-3 * cyc) / insts_per_wave,
           insts_per_wave * w / (msd * 1e-3 * cyc), insts_per_wave * w / (ms64 * 1e-3 * cyc), insts_per_wave * w / (msr * 1e-3 * cyc));
  }
  return 0;
}

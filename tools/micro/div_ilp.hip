// Micro-benchmark: IEEE float32 division, dependent chain vs independent quotients, at 1/2/4 waves per SIMD (MI355X).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/div_ilp.hip -o div_ilp && ./div_ilp
#include <hip/hip_runtime.h>
#include <stdio.h>
extern __shared__ float dyn[];
template <int ILP>
__global__ void __launch_bounds__(64) k_div(float* out, int iters, float b0) {
  float x[8], b[8];
  for (int i = 0; i < 8; i++) { x[i] = 1.0f + threadIdx.x + i; b[i] = b0 + 0.001f * i; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (ILP == 1) {
#pragma unroll
        for (int i = 0; i < 8; i++) x[0] = x[0] / b[i];          // 8 divisions, each needs the previous quotient
      } else {
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = x[i] / b[i];          // 8 independent divisions
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  if (s == 12345.f) out[0] = s + dyn[0];
}
template <int ILP>
double run(int waves, int iters, float* out) {
  const int lds = 160 * 1024 / (4 * waves) - 512;
  const int blocks = 256 * 4 * waves;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_div<ILP>, dim3(blocks), dim3(64), lds, 0, out, 10, 1.0001f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k_div<ILP>, dim3(blocks), dim3(64), lds, 0, out, iters, 1.0001f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  float* out; (void)hipMalloc(&out, 4);
  const int iters = 20000;
  const double divs = (double)iters * 32;
  for (int w : {1, 2, 4}) {
    const double a = run<1>(w, iters, out), b = run<8>(w, iters, out);
    printf("waves/SIMD %d: dependent divisions %.1f cycles each per wave, independent %.1f cycles each per wave (2.4 GHz nominal)\n", w,
           a * 1e-3 * 2.4e9 / divs, b * 1e-3 * 2.4e9 / divs);
  }
  return 0;
}

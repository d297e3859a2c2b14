// Micro-benchmark: what one MI355X SIMD issues per cycle, as a function of resident waves -- the price list behind `roofline.valu`.
//
// Every kernel is a long loop of ONE instruction written as inline asm (so the compiler can neither pack two v_fma_f32 into a
// v_pk_fma_f32 nor fold anything nor reschedule): 8 independent accumulator chains, or ONE dependent chain.  Occupancy is set by the dynamic LDS
// request (one 64-lane wave per block, blocks per CU limited by LDS), the grid is exactly one resident set.  Two clocks: s_memtime
// inside the wave (shader cycles, MI355X_MICROARCH.md) and HIP events around the launch (x the reported clock rate).
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_issue.hip -o valu_issue && ./valu_issue
//   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only tools/micro/valu_issue.hip -o valu_issue.s      (the ISA of the loops)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
extern __shared__ float dyn[];

// One asm statement holds the whole unrolled body (UNROLL x CHAINS instructions): the compiler schedules nothing inside it and
// inserts no hazard s_nop between the lines (it does between separate asm statements, one per 8 instructions, which costs issue slots).
#define R16(X) X X X X X X X X X X X X X X X X
#define L3(T, N) T " %" #N ", %" #N ", %8, %9\n"
#define L2(T, N) T " %" #N ", %" #N ", %8\n"
#define L1(T, N) T " %" #N ", %" #N "\n"
#define LC(T, N) T " %" #N ", %" #N ", %8, vcc\n"
#define ALL8(L, T) L(T, 0) L(T, 1) L(T, 2) L(T, 3) L(T, 4) L(T, 5) L(T, 6) L(T, 7)
#define ONE8(L, T) L(T, 0) L(T, 0) L(T, 0) L(T, 0) L(T, 0) L(T, 0) L(T, 0) L(T, 0)
#define DEFOP(name, TYPE, text, L)                                                                                        \
  struct name { typedef TYPE type; static constexpr const char* label = text;                                              \
    template <int DEP> static __device__ __forceinline__ void body(TYPE* x, TYPE a, TYPE b) {                              \
      if (DEP) asm volatile(R16(ONE8(L, text)) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b) : "vcc"); \
      else asm volatile(R16(ALL8(L, text)) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b) : "vcc"); \
    } };
DEFOP(FmaF32, float, "v_fma_f32", L3)
DEFOP(MulF32, float, "v_mul_f32", L2)
DEFOP(AddF32, float, "v_add_f32", L2)
DEFOP(MaxF32, float, "v_max_f32", L2)
DEFOP(MovB32, float, "v_mov_b32", L1)
DEFOP(CndMask, float, "v_cndmask_b32", LC)       // the select of the Fortran MIN / MAX lowering
DEFOP(PkFmaF32, f2, "v_pk_fma_f32", L3)
DEFOP(PkMulF32, f2, "v_pk_mul_f32", L2)
DEFOP(PkAddF32, f2, "v_pk_add_f32", L2)
DEFOP(FmaF64, double, "v_fma_f64", L3)
DEFOP(MulF64, double, "v_mul_f64", L2)
DEFOP(AddF64, double, "v_add_f64", L2)
DEFOP(RcpF32, float, "v_rcp_f32", L1)
DEFOP(RcpF64, double, "v_rcp_f64", L1)
DEFOP(SqrtF32, float, "v_sqrt_f32", L1)

constexpr int UNROLL = 16, CHAINS = 8;

template <class OP, int DEP>
__global__ void __launch_bounds__(64) k_issue(float* out, int iters, float af, float bf, unsigned long long* cyc) {
  typedef typename OP::type T;
  T x[CHAINS];
  for (int i = 0; i < CHAINS; i++) x[i] = (T)(threadIdx.x + i + 1.5f);
  const T a = (T)af, b = (T)bf;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) OP::template body<DEP>(x, a, b);
  // the last results must have left the pipeline before the clock is read
  T s = x[0];
  for (int i = 1; i < CHAINS; i++) s += x[i];
  asm volatile("" ::"v"(s));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  float chk; memcpy(&chk, &s, 4);
  if (chk == 12345.f && iters < 0) out[0] = chk + dyn[0];
}

struct Res { double ipc_events, cyc_per_inst_wave; };

template <class OP, int DEP>
Res run(int w, int iters, float* out, unsigned long long* d_cyc, int clk_khz) {
  // one wave per block; the LDS request limits blocks per CU to 4 SIMDs x w waves
  const int lds = 160 * 1024 / (4 * w) - 512;
  const int blocks = 256 * 4 * w;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_issue<OP, DEP>), dim3(blocks), dim3(64), lds, 0, out, 10, 1.0001f, 0.5f, d_cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_issue<OP, DEP>), dim3(blocks), dim3(64), lds, 0, out, iters, 1.0001f, 0.5f, d_cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
  const double insts = (double)iters * UNROLL * CHAINS;          // per wave
  Res r;
  r.ipc_events = insts * w / (ms * 1e-3 * clk_khz * 1e3);       // wave-instructions per cycle per SIMD, wall clock x reported clock
  r.cyc_per_inst_wave = mean / insts;                           // s_memtime cycles one wave needs per instruction (w waves share the SIMD)
  hipEventDestroy(e0); hipEventDestroy(e1);
  return r;
}

template <class OP>
void row(float* out, unsigned long long* d_cyc, int clk) {
  const int iters = 4000;
  printf("%-38s", OP::label);
  for (int dep = 0; dep < 2; dep++) {
    printf(dep ? " | 1 dependent chain:" : " 8 independent chains:");
    for (int w : {1, 2, 4, 8}) {
      Res r = dep ? run<OP, 1>(w, iters, out, d_cyc, clk) : run<OP, 0>(w, iters, out, d_cyc, clk);
      // SIMD cycles per wave-instruction = cycles one wave sees per instruction / waves sharing the SIMD
      printf("  w%d %.2f (%.3f/cyc)", w, r.cyc_per_inst_wave / w, r.ipc_events);
    }
  }
  printf("\n");
}

int main() {
  float* out; hipMalloc(&out, 4);
  unsigned long long* d_cyc; hipMalloc(&d_cyc, 256 * 4 * 8 * 8);
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("%s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, clk);
  printf("per entry: SIMD cycles per wave64 instruction by s_memtime (= cycles a wave sees per instruction / resident waves per SIMD),\n"
         "in brackets wave-instructions per cycle per SIMD by HIP events x the reported clock; w = waves per SIMD\n");
  row<FmaF32>(out, d_cyc, clk);
  row<MulF32>(out, d_cyc, clk);
  row<AddF32>(out, d_cyc, clk);
  row<MaxF32>(out, d_cyc, clk);
  row<MovB32>(out, d_cyc, clk);
  row<CndMask>(out, d_cyc, clk);
  row<PkFmaF32>(out, d_cyc, clk);
  row<PkMulF32>(out, d_cyc, clk);
  row<PkAddF32>(out, d_cyc, clk);
  row<FmaF64>(out, d_cyc, clk);
  row<MulF64>(out, d_cyc, clk);
  row<AddF64>(out, d_cyc, clk);
  row<RcpF32>(out, d_cyc, clk);
  row<SqrtF32>(out, d_cyc, clk);
  row<RcpF64>(out, d_cyc, clk);
  return 0;
}

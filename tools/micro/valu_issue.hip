// Micro-benchmark: what one MI355X SIMD issues per cycle, as a function of resident waves -- the price list behind `roofline.valu`.
//
// Every kernel is a long loop of ONE instruction written as inline asm (so the compiler can neither pack two v_fma_f32 into a
// v_pk_fma_f32 nor fold anything nor reschedule): 8 independent accumulator chains, or ONE dependent chain.  Occupancy is set by the dynamic LDS
// request (blocks of four waves = one per SIMD, blocks per CU limited by LDS), the grid is exactly one resident set.  Two clocks: s_memtime
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
// line forms: %N = the chain's own register (N = 0..7), %8 / %9 = two loop-invariant VGPR operands, %10 = a 64-bit SGPR lane mask
#define L3(T, N) T " %" #N ", %" #N ", %8, %9\n"
#define L2(T, N) T " %" #N ", %" #N ", %8\n"
#define L1(T, N) T " %" #N ", %" #N "\n"
#define LC(T, N) T " %" #N ", %" #N ", %8, vcc\n"                                   /* select on VCC (never written in the loop) */
#define LCS(T, N) T " %" #N ", %" #N ", %8, %10\n"                                  /* select on an SGPR pair */
#define LCMP(T, N) T " vcc, %" #N ", %9\n"                                          /* compare -> VCC */
#define LCMPS(T, N) T " s[20:21], %" #N ", %9\n"                                    /* compare -> SGPR pair */
#define LCMPSEL(T, N) "v_cmp_gt_f32 vcc, %" #N ", %9\n" T " %" #N ", %" #N ", %8, vcc\n"     /* the Fortran MAX / MIN lowering: 2 instructions */
#define LFMAC(T, N) T " %" #N ", %8, %9\n"
#define LDSC(T, N) T " %" #N ", vcc, %" #N ", %8, %9\n"                             /* v_div_scale: writes VCC */
#define LRL(T, N) "v_readlane_b32 s20, %" #N ", 3\n"
#define LWL(T, N) "v_writelane_b32 %" #N ", s22, 3\n"
#define LNOP(T, N) "v_add_f32 %" #N ", %" #N ", %8\ns_nop 0\n"                       /* an s_nop between VALU instructions: 2 instructions */
#define LMIX(T, N) "v_mul_f32 %" #N ", %" #N ", %8\nv_mul_f64 v[40:41], v[40:41], v[42:43]\n"     /* float32 and float64 alternating: 2 */
#define LCVT(T, N) "v_cvt_f64_f32 v[40:41], %" #N "\nv_cvt_f32_f64 %" #N ", v[40:41]\n"           /* div_rc's conversions: 2, dependent */
#define LCVTI(T, N) "v_cvt_f64_f32 v[40:41], %" #N "\nv_cvt_f64_f32 v[42:43], %" #N "\n"          /* 2, independent */
#define ALL8(L, T) L(T, 0) L(T, 1) L(T, 2) L(T, 3) L(T, 4) L(T, 5) L(T, 6) L(T, 7)
#define ONE8(L, T) L(T, 0) L(T, 0) L(T, 0) L(T, 0) L(T, 0) L(T, 0) L(T, 0) L(T, 0)
#define NMP_IO : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b), "s"(m) \
               : "vcc", "s20", "s21", "s22", "v40", "v41", "v42", "v43"
#define DEFOPN(name, TYPE, labeltext, text, L, PERLINE)                                                                   \
  struct name { typedef TYPE type; static constexpr const char* label = labeltext; static constexpr int per = PERLINE;     \
    template <int DEP> static __device__ __forceinline__ void body(TYPE* x, TYPE a, TYPE b, unsigned long long m) {        \
      if (DEP) asm volatile(R16(ONE8(L, text)) NMP_IO);                                                                    \
      else asm volatile(R16(ALL8(L, text)) NMP_IO);                                                                        \
    } };
#define DEFOP(name, TYPE, text, L) DEFOPN(name, TYPE, text, text, L, 1)
DEFOP(FmaF32, float, "v_fma_f32", L3)
DEFOP(FmacF32, float, "v_fmac_f32", LFMAC)
DEFOP(MulF32, float, "v_mul_f32", L2)
DEFOP(AddF32, float, "v_add_f32", L2)
DEFOP(MaxF32, float, "v_max_f32", L2)
DEFOP(MinF32, float, "v_min_f32", L2)
DEFOP(MovB32, float, "v_mov_b32", L1)
DEFOP(AddU32, float, "v_add_u32", L2)
DEFOP(CndMask, float, "v_cndmask_b32", LC)
DEFOP(CndMaskS, float, "v_cndmask_b32_e64", LCS)
DEFOP(CmpVcc, float, "v_cmp_gt_f32", LCMP)
DEFOP(CmpS, float, "v_cmp_gt_f32_e64", LCMPS)
DEFOPN(CmpSel, float, "v_cmp_gt_f32 vcc + v_cndmask_b32 (pair)", "v_cndmask_b32", LCMPSEL, 2)
DEFOP(DivScale, float, "v_div_scale_f32", LDSC)
DEFOP(DivFmas, float, "v_div_fmas_f32", L3)
DEFOP(DivFixup, float, "v_div_fixup_f32", L3)
DEFOP(ReadLane, float, "v_readlane_b32", LRL)
DEFOP(WriteLane, float, "v_writelane_b32", LWL)
DEFOPN(AddNop, float, "v_add_f32 + s_nop 0 (pair)", "", LNOP, 2)
DEFOPN(Mix3264, float, "v_mul_f32 + v_mul_f64 (pair)", "", LMIX, 2)
DEFOPN(CvtRound, float, "v_cvt_f64_f32 + v_cvt_f32_f64 (dependent pair)", "", LCVT, 2)
DEFOPN(CvtInd, float, "v_cvt_f64_f32 x 2 (independent pair)", "", LCVTI, 2)
DEFOP(PkFmaF32, f2, "v_pk_fma_f32", L3)
DEFOP(PkMulF32, f2, "v_pk_mul_f32", L2)
DEFOP(PkAddF32, f2, "v_pk_add_f32", L2)
DEFOP(FmaF64, double, "v_fma_f64", L3)
DEFOP(MulF64, double, "v_mul_f64", L2)
DEFOP(AddF64, double, "v_add_f64", L2)
#define LLA(T, N) T " %" #N ", %" #N ", 2, %9\n"
DEFOP(LshlAddU64, double, "v_lshl_add_u64", LLA)
DEFOP(RcpF32, float, "v_rcp_f32", L1)
DEFOP(RcpF64, double, "v_rcp_f64", L1)
DEFOP(SqrtF32, float, "v_sqrt_f32", L1)

constexpr int UNROLL = 16, CHAINS = 8;

template <class OP, int DEP>
__global__ void __launch_bounds__(256) k_issue(float* out, int iters, float af, float bf, unsigned long long* cyc, unsigned long long mask) {
  typedef typename OP::type T;
  T x[CHAINS];
  for (int i = 0; i < CHAINS; i++) x[i] = (T)(threadIdx.x + i + 1.5f);
  const T a = (T)af, b = (T)bf;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) OP::template body<DEP>(x, a, b, mask);
  // the last results must have left the pipeline before the clock is read
  T s = x[0];
  for (int i = 1; i < CHAINS; i++) s += x[i];
  asm volatile("" ::"v"(s));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  float chk; memcpy(&chk, &s, 4);
  if (chk == 12345.f && iters < 0) out[0] = chk + dyn[0];
}

struct Res { double ipc_events, cyc_per_inst_wave; };

template <class OP, int DEP>
Res run(int w, int iters, float* out, unsigned long long* d_cyc, int clk_khz) {
  // one block = 4 waves = one wave per SIMD of a CU; the LDS request limits blocks per CU to w (160 KB per CU), so every SIMD holds
  // exactly w waves (blocks of one wave do not get there: at most 8 such blocks were resident per CU whatever the LDS request)
  const int lds = w == 1 ? 100 * 1024 : 160 * 1024 / w - 1024;
  const int blocks = 256 * w;
  if (hipFuncSetAttribute((const void*)k_issue<OP, DEP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) printf("[LDS attribute refused] ");
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_issue<OP, DEP>), dim3(blocks), dim3(256), lds, 0, out, 10, 1.0001f, 0.5f, d_cyc, 0x5555aaaa3333ccccULL);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_issue<OP, DEP>), dim3(blocks), dim3(256), lds, 0, out, iters, 1.0001f, 0.5f, d_cyc, 0x5555aaaa3333ccccULL);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) printf("[launch failed] ");
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), d_cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += (double)v; mean /= (blocks * 4);
  const double insts = (double)iters * UNROLL * CHAINS * OP::per;          // per wave
  Res r;
  r.ipc_events = insts * w / (ms * 1e-3 * clk_khz * 1e3);       // wave-instructions per cycle per SIMD, wall clock x reported clock
  r.cyc_per_inst_wave = mean / insts;                           // s_memtime cycles one wave needs per instruction (w waves share the SIMD)
  hipEventDestroy(e0); hipEventDestroy(e1);
  return r;
}

template <class OP>
void row(float* out, unsigned long long* d_cyc, int clk) {
  const int iters = 4000;
  printf("%-48s", OP::label);
  for (int dep = 0; dep < 2; dep++) {
    printf(dep ? " | 1 dependent chain:" : " 8 independent chains:");
    for (int w : {1, 2, 4, 8}) {
      Res r = dep ? run<OP, 1>(w, iters, out, d_cyc, clk) : run<OP, 0>(w, iters, out, d_cyc, clk);
      // SIMD cycles per wave-instruction = cycles one wave sees per instruction / waves sharing the SIMD
      printf("  w%d %.2f (%.3f/cyc)", w, r.cyc_per_inst_wave, r.ipc_events);
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
  printf("per entry: s_memtime cycles ONE WAVE needs per instruction while w waves share its SIMD; in brackets wave-instructions per cycle\n"
         "per SIMD from HIP events x the reported clock (a lower bound if the chip runs below that clock); w = waves requested per SIMD\n");
  row<FmaF32>(out, d_cyc, clk);
  row<FmacF32>(out, d_cyc, clk);
  row<MulF32>(out, d_cyc, clk);
  row<AddF32>(out, d_cyc, clk);
  row<MovB32>(out, d_cyc, clk);
  row<AddU32>(out, d_cyc, clk);
  row<MaxF32>(out, d_cyc, clk);
  row<MinF32>(out, d_cyc, clk);
  row<CmpVcc>(out, d_cyc, clk);
  row<CmpS>(out, d_cyc, clk);
  row<CndMask>(out, d_cyc, clk);
  row<CndMaskS>(out, d_cyc, clk);
  row<CmpSel>(out, d_cyc, clk);
  row<DivScale>(out, d_cyc, clk);
  row<DivFmas>(out, d_cyc, clk);
  row<DivFixup>(out, d_cyc, clk);
  row<ReadLane>(out, d_cyc, clk);
  row<WriteLane>(out, d_cyc, clk);
  row<AddNop>(out, d_cyc, clk);
  row<PkFmaF32>(out, d_cyc, clk);
  row<PkMulF32>(out, d_cyc, clk);
  row<PkAddF32>(out, d_cyc, clk);
  row<FmaF64>(out, d_cyc, clk);
  row<MulF64>(out, d_cyc, clk);
  row<AddF64>(out, d_cyc, clk);
  row<LshlAddU64>(out, d_cyc, clk);
  row<Mix3264>(out, d_cyc, clk);
  row<CvtRound>(out, d_cyc, clk);
  row<CvtInd>(out, d_cyc, clk);
  row<RcpF32>(out, d_cyc, clk);
  row<SqrtF32>(out, d_cyc, clk);
  row<RcpF64>(out, d_cyc, clk);
  return 0;
}

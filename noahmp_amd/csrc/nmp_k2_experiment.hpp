// Profiling build only (-DNMP_K2_EXPERIMENT, tools/experiments.sh k2): is there a third wave for the flux solvers?
//
// The land kernel needs ~250 VGPRs (two waves per SIMD).  VEGE_FLUX + BARE_FLUX + the flux blend (lsm:1700-1803) are about half of its
// time and read ~42 words per column.  This header (a) lets the land kernel dump exactly those words, per column, just before
// VEGE_FLUX (k2_dump, called from energy()), and (b) holds the flux solvers as a kernel of their own that reads the dump
// (k2_kernel<WAVES>), so that its register need and its duration at 1 / 2 / 3 / 4 waves per SIMD can be measured on the real
// inputs of a config-3 step.  Nothing here is part of the product library.
#ifndef NMP_K2_PART_A
#define NMP_K2_PART_A
// the words VEGE_FLUX / BARE_FLUX / the blend read that are not recomputed from these inside the kernel (float planes, SoA)
#define NMP_K2_FIELDS(X) \
  X(s.sfctmp) X(s.rhoair) X(s.qair) X(s.eair) X(s.canliq) X(s.canice) X(s.sav) X(s.sag) X(s.fwet) X(s.sfcprs) X(s.psfc) X(s.tv) \
  X(s.tg) X(s.tah) X(s.eah) X(s.ch) X(s.cm) X(s.lwdn) X(s.fveg) X(s.htop) X(s.igs) X(s.btran) X(s.snowh) \
  X(q.ur) X(q.vai) X(q.laisun) X(q.laisha) X(q.zlvl) X(q.zpd) X(q.z0m) X(q.z0mg) X(q.emv) X(q.emg) X(q.rsurf) X(q.rhsur) \
  X(q.parsun) X(q.parsha) X(q.df_top) X(q.dz_top) X(q.stc_top)
constexpr int NMP_K2_NIN = 40 + 2;      // + vegetation type, canopy flag
constexpr int NMP_K2_NOUT = 43;

#if defined(__HIP_DEVICE_COMPILE__) || !defined(__HIPCC_RTC__)
namespace nmp {
static __device__ float* g_k2_dump = nullptr;
static __device__ long g_k2_n = 0;
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
namespace nmp {
__device__ __forceinline__ void k2_dump(const Col& s, const VegIn& q, bool canopy) {
  float* o = g_k2_dump;
  const long n = g_k2_n, t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (!o || t >= n) return;
  int p = 0;
#define X(f) o[(size_t)(p++) * n + t] = f;
  NMP_K2_FIELDS(X)
#undef X
  o[(size_t)(p++) * n + t] = __int_as_float(s.vegtyp);
  o[(size_t)(p++) * n + t] = canopy ? 1.f : 0.f;
}
}  // namespace nmp
#endif
#endif  // NMP_K2_PART_A

#if defined(NMP_K2_KERNELS) && !defined(NMP_K2_PART_B)       // second inclusion: nmp_engine_fixed.inc, after nmp_kernel.hpp
#define NMP_K2_PART_B
namespace {
using namespace nmp;
struct K2Args { Ctx c; const float* in; float* out; long n; };

template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k2_kernel(const K2Args k) {
  libm::libm_stage_tables();
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= k.n) return;
  Col s = {};
  Parm P = {};
  VegIn q = {};
  int p = 0;
#define X(f) f = k.in[(size_t)(p++) * k.n + t];
  NMP_K2_FIELDS(X)
#undef X
  s.vegtyp = __float_as_int(k.in[(size_t)(p++) * k.n + t]);
  const bool canopy = k.in[(size_t)(p++) * k.n + t] != 0.f;
  s.thair = s.sfctmp; s.foln = 1.f; s.co2air = 395.e-06f * s.sfcprs; s.o2air = 0.209f * s.sfcprs;
  const int v = s.vegtyp - 1;
  P.czil = k.c.ts.czil; P.dleaf = k.c.T->dleaf[v]; q.cwp = k.c.T->cwpvt[v];
  if (s.tv > TFRZ) { s.latheav = HVAP; s.frozen_canopy = 0; } else { s.latheav = HSUB; s.frozen_canopy = 1; }
  q.gammav = div_rc(CPAIR * s.sfcprs, s.frozen_canopy ? NMP_RCC(0.622f * HSUB) : NMP_RCC(0.622f * HVAP));
  if (s.tg > TFRZ) { s.latheag = HVAP; s.frozen_ground = 0; } else { s.latheag = HSUB; s.frozen_ground = 1; }
  q.gammag = div_rc(CPAIR * s.sfcprs, s.frozen_ground ? NMP_RCC(0.622f * HSUB) : NMP_RCC(0.622f * HVAP));
  q.r_rhocp = rc64(s.rhoair * CPAIR); q.r_gammav = rc64(q.gammav); q.r_gammag = rc64(q.gammag);
  const float zpdg = s.snowh;
  float cmv = 0.f, cmb = 0.f, psnsun = 0.f, psnsha = 0.f;
  if (canopy) { s.tgv = s.tg; cmv = s.cm; s.chv = s.ch; }
  SimpleLoop runner;
  vege_flux(k.c, P, s, q, cmv, psnsun, psnsha, canopy, runner);
  s.tgb = s.tg; cmb = s.cm; s.chb = s.ch;
  bare_flux(k.c, P, s, q, zpdg, cmb);
  if (canopy) {                                       // lsm:1746-1766
    s.fira = s.fveg * s.irg + (1.0f - s.fveg) * s.irb + s.irc;
    s.fsh = s.fveg * s.shg + (1.0f - s.fveg) * s.shb + s.shc;
    s.fgev = s.fveg * s.evg + (1.0f - s.fveg) * s.evb;
    s.ssoil = s.fveg * s.ghv + (1.0f - s.fveg) * s.ghb;
    s.fcev = s.evc; s.fctr = s.tr;
    s.tg = s.fveg * s.tgv + (1.0f - s.fveg) * s.tgb;
    s.cm = s.fveg * cmv + (1.0f - s.fveg) * cmb;
    s.ch = s.fveg * s.chv + (1.0f - s.fveg) * s.chb;
  } else {
    s.fira = s.irb; s.fsh = s.shb; s.fgev = s.evb; s.ssoil = s.ghb; s.tg = s.tgb; s.fcev = 0.f; s.fctr = 0.f;
    s.cm = cmb; s.ch = s.chb; s.rssun = 0.0f; s.rssha = 0.0f; s.tgv = s.tgb; s.chv = s.chb;
  }
  float fire = s.lwdn + s.fira;
  if (fire <= 0.f) raise(s, NOAHMP_ERR_FIRE_NONPOSITIVE);
  s.emissi = s.fveg * (q.emg * (1 - q.emv) + q.emv + q.emv * (1 - q.emv) * (1 - q.emg)) + (1 - s.fveg) * q.emg;
  s.trad = pow_quarter((fire - (1 - s.emissi) * s.lwdn) / (s.emissi * SB));
  s.psn = psnsun * q.laisun + psnsha * q.laisha;
  p = 0;
#define O(f) k.out[(size_t)(p++) * k.n + t] = f;
  O(s.trad) O(s.fsh) O(s.ssoil) O(s.emissi) O(s.tg) O(s.eah) O(s.tah) O(s.cm) O(s.ch) O(s.t2mv) O(s.t2mb) O(s.q2v) O(s.q2b) O(s.fira)
  O(s.psn) O(s.rssun) O(s.rssha) O(s.tgv) O(s.tgb) O(s.chv) O(s.chb) O(s.irc) O(s.irg) O(s.shc) O(s.shg) O(s.evg) O(s.ghv) O(s.irb)
  O(s.shb) O(s.evb) O(s.ghb) O(s.tr) O(s.evc) O(s.chleaf) O(s.chuc) O(s.chv2) O(s.chb2) O(s.tv) O(s.fgev) O(s.fcev) O(s.fctr) O(s.qsfc)
  O(__int_as_float(s.err))
#undef O
}

static KArgs g_k2_last;                  // the kernel arguments of the last land-range launch (its Ctx)
static float* g_k2_in = nullptr; static float* g_k2_out = nullptr; static long g_k2_cols = 0;

template <int WAVES>
static float k2_time(const K2Args& a, size_t dyn_lds, int reps, hipStream_t s) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  if (dyn_lds > 48 * 1024) hipFuncSetAttribute((const void*)k2_kernel<WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds);
  const dim3 grid((unsigned)((a.n + 255) / 256)), block(256);
  hipLaunchKernelGGL(k2_kernel<WAVES>, grid, block, dyn_lds, s, a);            // warm-up
  hipEventRecord(e0, s);
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k2_kernel<WAVES>, grid, block, dyn_lds, s, a);
  hipEventRecord(e1, s);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return hipGetLastError() == hipSuccess ? ms / reps : -1.f;
}
}  // namespace

// what = 0: arm the dump (the next land-range launch writes its K2 inputs).  what = 1: disarm, time k2_kernel on the dump:
// out[0..] = ms of {2 waves; 3 waves; 4 waves; the 3-wave binary held at 2 waves/SIMD by an LDS request; the 2-wave binary held at 1;
// the 4-wave binary held at 3; the 4-wave binary held at 2}, then the number of columns.
extern "C" int noahmp_hip_debug_k2(int what, float* out, int nout) {
  using namespace nmp_host;
  if (what == 0) {
    const long n = g_k2_last.t_count;
    if (n <= 0) { g.last_error = "noahmp_hip_debug_k2: no land-range launch seen"; return -105; }
    if (g_k2_cols != n) {
      if (g_k2_in) hipFree(g_k2_in);
      if (g_k2_out) hipFree(g_k2_out);
      HIPCHK(hipMalloc(&g_k2_in, (size_t)NMP_K2_NIN * n * 4));
      HIPCHK(hipMalloc(&g_k2_out, (size_t)NMP_K2_NOUT * n * 4));
      g_k2_cols = n;
    }
    HIPCHK(hipMemset(g_k2_in, 0, (size_t)NMP_K2_NIN * n * 4));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(nmp::g_k2_n), &n, sizeof n));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(nmp::g_k2_dump), &g_k2_in, sizeof g_k2_in));
    return 0;
  }
  float* null = nullptr;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(nmp::g_k2_dump), &null, sizeof null));
  if (!g_k2_in) { g.last_error = "noahmp_hip_debug_k2: nothing dumped"; return -105; }
  K2Args a;
  a.c = g_k2_last.c; a.in = g_k2_in; a.out = g_k2_out; a.n = g_k2_cols;
  hipStream_t s = g.own_stream;
  const int reps = 5;
  const size_t two = 70 * 1024, one = 100 * 1024, three = 50 * 1024;     // LDS requests that leave room for 2 / 1 / 3 blocks of 256 per CU
  float r[8];
  r[0] = k2_time<2>(a, 0, reps, s);
  r[1] = k2_time<3>(a, 0, reps, s);
  r[2] = k2_time<4>(a, 0, reps, s);
  r[3] = k2_time<3>(a, two, reps, s);
  r[4] = k2_time<2>(a, one, reps, s);
  r[5] = k2_time<4>(a, three, reps, s);
  r[6] = k2_time<4>(a, two, reps, s);
  r[7] = (float)g_k2_cols;
  for (int i = 0; i < nout && i < 8; i++) out[i] = r[i];
  return 0;
}
#endif  // NMP_K2_KERNELS

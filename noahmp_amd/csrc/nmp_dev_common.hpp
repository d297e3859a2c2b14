// Noah-MP column engine for MI355X (gfx950) -- device-side common definitions.
//
// Execution model: one GPU thread advances one land column through one NOAHMP_SFLX step
// (reference phys/module_sf_noahmplsm.F90:518, "lsm").  Per-column scalars live in VGPRs;
// the 7-entry snow/soil layer arrays (index -2..4) live either in wavefront-private scratch
// or in LDS laid out [layer][thread] so that the runtime layer index (ISNOW-dependent) never
// causes a bank conflict: lane t always hits bank (t mod 32) whatever layer it addresses.
// All arithmetic is float32 in the reference's operation order; no fast-math, no contraction.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif
#include "noahmp_hip.h"

namespace nmp {

// __host__ too: tests/host_emul compiles the same source for the CPU to debug it without a GPU
#define NMP_DEV __host__ __device__ __forceinline__
#define L(i) ((i) + 2)   // layer -2..NSOIL -> slot 0..NL-1
#if defined(__HIP_DEVICE_COMPILE__)
#define NMP_ASSUME(c) __builtin_assume(c)
#else
#define NMP_ASSUME(c) ((void)0)
#endif
constexpr int NSOIL = NOAHMP_NSOIL;     // a build-time choice (include/noahmp_hip.h): 4 unless the library is built with -DNOAHMP_NSOIL=n
constexpr int NSNOW = NOAHMP_NSNOW;
constexpr int NL = NSNOW + NSOIL;
static_assert(NSNOW == 3 && NSOIL >= 2 && NSOIL <= 12, "layer slots: three snow layers, 2..12 soil layers");

// physical constants, lsm:12-28 and lsm:180-188
constexpr float GRAV = 9.80616f, SB = 5.67E-08f, VKC = 0.40f, TFRZ = 273.16f, HSUB = 2.8440E06f,
                HVAP = 2.5104E06f, HFUS = 0.3336E06f, CWAT = 4.188E06f, CICE = 2.094E06f,
                CPAIR = 1004.64f, TKWAT = 0.6f, TKICE = 2.2f, RAIR = 287.04f, RW = 461.269f,
                DENH2O = 1000.f, DENICE = 917.f;
constexpr float TIMEAN = 10.5f, FSATMX = 0.38f, M_MELT = 2.50f, Z0SNO = 0.002f, SSI = 0.03f,
                SWEMX = 1.00f;

// Layer array view.  STRIDE==1: plain per-thread array (registers / scratch).
// STRIDE==block size: LDS column-interleaved storage.
template <int STRIDE>
struct LArr {
  float* p;
  NMP_DEV float& operator[](int i) const { return p[i * STRIDE]; }
};

NMP_DEV float fmin2(float a, float b) { return (a < b) ? a : b; }   // Fortran MIN/MAX lowering
NMP_DEV float fmax2(float a, float b) { return (a > b) ? a : b; }

// REAL**INTEGER: square-and-multiply in the order flang/compiler-rt use (T**4 = (T*T)*(T*T)).
NMP_DEV float powi2(float a) { return a * a; }
NMP_DEV float powi3(float a) { return a * (a * a); }
NMP_DEV float powi4(float a) { float b = a * a; return b * b; }
NMP_DEV float powi5(float a) { float b = a * a; return a * (b * b); }

// Transcendentals.  NMP_EXACT_LIBM=1 (default): the reference libm's own algorithms (nmp_libm.hpp), so that
// EXP / LOG / ** / LOG10 / ATAN / TANH / TAN / ACOS / COS return the reference's bits on the GPU.  NMP_EXACT_LIBM=0: ocml's
// float32 routines (<= 1-2 ulp, a little faster, statistically equivalent results -- DESIGN.md section 5).
#ifndef NMP_EXACT_LIBM
#define NMP_EXACT_LIBM 1
#endif
}  // namespace nmp
#include "nmp_libm.hpp"
namespace nmp {
#if defined(NMP_LIBM_COUNT) && !defined(__HIP_DEVICE_COMPILE__)
// host-emulation instrumentation (tests/host_emul): dynamic call counts per routine
extern "C" long nmp_libm_calls[8];
#define NMP_CNT(i) (nmp_libm_calls[i]++)
#else
#define NMP_CNT(i) ((void)0)
#endif
// Fortran MAX / MIN as the reference's compilers lower them: a select on an ordered compare (x86 MAXSS / MINSS semantics), so a NaN
// in the SECOND operand propagates and a NaN in the first does not.  fmaxf / fminf (v_max_f32, IEEE maxNum) would swallow both; the
// difference only shows in columns that already carry NaNs (e.g. dynamic vegetation on a category without vegetation parameters),
// where the reference keeps running because its balance checks compare false.  Argument order = the reference's.
NMP_DEV float nmp_max(float a, float b) { return a > b ? a : b; }
NMP_DEV float nmp_min(float a, float b) { return a < b ? a : b; }

#if NMP_EXACT_LIBM
NMP_DEV float nmp_expf(float x) { NMP_CNT(0); return libm::expf_(x); }
NMP_DEV float nmp_expf_const(float x) { NMP_CNT(0); return libm::expf_k_(x); }     // compile-time argument: folds
NMP_DEV float nmp_logf(float x) { NMP_CNT(1); return libm::logf_(x); }
// N independent LOG / EXP evaluated together (same arithmetic; the table look-ups of the batch share one LDS latency)
template <int N> NMP_DEV void nmp_logfN(const float* x, float* out) { for (int n = 0; n < N; n++) NMP_CNT(1); libm::logfN_<N>(x, out); }
template <int N> NMP_DEV void nmp_expfN(const float* x, float* out) { for (int n = 0; n < N; n++) NMP_CNT(0); libm::expfN_<N>(x, out); }
template <int N> NMP_DEV void nmp_powfN(const float* x, const float* y, float* out) { for (int n = 0; n < N; n++) NMP_CNT(2); libm::powfN_<N>(x, y, out); }
// BASE ** y for compile-time positive bases (log2 BASE folds at compile time), and pairs x ** y1, x ** y2 with a shared log2 x
#define NMP_LOG2K(base) libm::powf_log2_k(libm::asuint(base))
template <int N> NMP_DEV void nmp_powf_constbaseN(const float* base, const double* log2base, const float* y, float* out) {
  for (int n = 0; n < N; n++) NMP_CNT(2);
  libm::powf_constbaseN_<N>(base, log2base, y, out);
}
template <int N> NMP_DEV void nmp_powf_pairN(const float* x, float y1, float y2, float* o1, float* o2) {
  for (int n = 0; n < 2 * N; n++) NMP_CNT(2);
  libm::powf_pairN_<N>(x, y1, y2, o1, o2);
}
NMP_DEV float nmp_powf(float x, float y) { NMP_CNT(2); return libm::powf_(x, y); }
NMP_DEV float nmp_log10f(float x) { NMP_CNT(3); return libm::log10f_(x); }
NMP_DEV float nmp_atanf(float x) { NMP_CNT(4); return libm::atanf_(x); }
NMP_DEV float nmp_atanf_ge1(float x) { NMP_CNT(4); return libm::atanf_ge1_(x); }     // x >= 1 (SFCDIF1)
NMP_DEV float nmp_tanhf(float x) { NMP_CNT(5); return libm::tanhf_(x); }
NMP_DEV float nmp_tanf(float x) { return libm::tanf_(x); }     // OPT_RAD=1 only (lsm:2531-2539)
NMP_DEV float nmp_acosf(float x) { return libm::acosf_(x); }
NMP_DEV float nmp_cosf(float x) { return libm::cosf_(x); }
// x**0.25, x**0.5, x**-0.25 with a literal exponent (SFCDIF1, RAGRB): the reference calls powf
NMP_DEV float pow_quarter(float x) { NMP_CNT(6); return libm::powf_(x, 0.25f); }
NMP_DEV void pow_quarter2(float x1, float x2, float& r1, float& r2) {
  NMP_CNT(6); NMP_CNT(6);
  const float x[2] = {x1, x2}, y[2] = {0.25f, 0.25f}; float o[2];
  libm::powfN_<2>(x, y, o);
  r1 = o[0]; r2 = o[1];
}
NMP_DEV float pow_half(float x) { NMP_CNT(6); return libm::powf_(x, 0.5f); }
// X**2. with a REAL exponent: the pinned reference build calls powf(X, 2.0), whose glibc result differs from the correctly
// rounded X*X by one ulp for 0.07 % of the arguments (lsm:1536, 2004, 5664, 6325; gla:490, 695).
NMP_DEV float pow_two(float x) { NMP_CNT(6); return libm::powf_(x, 2.0f); }
NMP_DEV float pow_neg_quarter(float x) { NMP_CNT(6); return libm::powf_(x, -0.25f); }
#else
NMP_DEV float nmp_expf(float x) { return expf(x); }
NMP_DEV float nmp_expf_const(float x) { return expf(x); }
NMP_DEV float nmp_logf(float x) { return logf(x); }
template <int N> NMP_DEV void nmp_logfN(const float* x, float* out) { for (int n = 0; n < N; n++) out[n] = logf(x[n]); }
template <int N> NMP_DEV void nmp_expfN(const float* x, float* out) { for (int n = 0; n < N; n++) out[n] = expf(x[n]); }
template <int N> NMP_DEV void nmp_powfN(const float* x, const float* y, float* out) { for (int n = 0; n < N; n++) out[n] = powf(x[n], y[n]); }
#define NMP_LOG2K(base) 0.0
template <int N> NMP_DEV void nmp_powf_constbaseN(const float* base, const double*, const float* y, float* out) { for (int n = 0; n < N; n++) out[n] = powf(base[n], y[n]); }
template <int N> NMP_DEV void nmp_powf_pairN(const float* x, float y1, float y2, float* o1, float* o2) { for (int n = 0; n < N; n++) { o1[n] = powf(x[n], y1); o2[n] = powf(x[n], y2); } }
NMP_DEV float nmp_powf(float x, float y) { return powf(x, y); }
NMP_DEV float nmp_log10f(float x) { return log10f(x); }
NMP_DEV float nmp_atanf(float x) { return atanf(x); }
NMP_DEV float nmp_atanf_ge1(float x) { return atanf(x); }
NMP_DEV float nmp_tanhf(float x) { return tanhf(x); }
NMP_DEV float nmp_tanf(float x) { return tanf(x); }
NMP_DEV float nmp_acosf(float x) { return acosf(x); }
NMP_DEV float nmp_cosf(float x) { return cosf(x); }
// IEEE sqrt chains (each step correctly rounded, total <= 0.75 ulp, ~10 VALU ops) instead of ocml powf
NMP_DEV float pow_quarter(float x) { return sqrtf(sqrtf(x)); }
NMP_DEV void pow_quarter2(float x1, float x2, float& r1, float& r2) { r1 = sqrtf(sqrtf(x1)); r2 = sqrtf(sqrtf(x2)); }
NMP_DEV float pow_half(float x) { return sqrtf(x); }
NMP_DEV float pow_two(float x) { return x * x; }
NMP_DEV float pow_neg_quarter(float x) { return 1.0f / sqrtf(sqrtf(x)); }
#endif

// Phase timers (profiling build only, -DNMP_PHASE_TIMERS): the first active lane of every wave adds the shader-clock ticks since
// its previous NMP_TIC to a per-phase counter (spread over 256 slots).  
// With 2 waves interleaved per SIMD a tick interval contains the other wave's issue slots too, so the numbers
// are shares of the kernel's time, not instruction counts.
#if defined(NMP_PHASE_TIMERS)
constexpr int NMP_NPHASE = 32;
static __device__ unsigned long long g_nmp_prof[NMP_NPHASE * 256];
#endif
#if defined(NMP_PHASE_TIMERS) && defined(__HIP_DEVICE_COMPILE__)
static __shared__ long long s_nmp_last[8];
#define NMP_TIC(ph) do { if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) { const long long now_ = clock64(); \
    const int w_ = threadIdx.x >> 6; \
    atomicAdd(&g_nmp_prof[(ph) * 256 + (blockIdx.x & 255)], (unsigned long long)(now_ - s_nmp_last[w_])); \
    s_nmp_last[w_] = now_; } } while (0)
#define NMP_TIC0() do { if ((threadIdx.x & 63) == 0) s_nmp_last[threadIdx.x >> 6] = clock64(); } while (0)
#else
#define NMP_TIC(ph) ((void)0)
#define NMP_TIC0() ((void)0)
#endif

#ifdef NMP_FIXED_DVEG          // option-specialised translation unit (nmp_engine_fixed.inc, or compiled at run time by
                               // noahmp_jit.hip): the options are compile-time constants -- every other alternative's code folds
                               // away.  Unset ones default to the reference's namelist (run/namelist.hrldas).
#ifndef NMP_FIXED_RUN
#define NMP_FIXED_RUN 1
#endif
#ifndef NMP_FIXED_CRS
#define NMP_FIXED_CRS 1
#define NMP_FIXED_BTR 1
#define NMP_FIXED_SFC 1
#define NMP_FIXED_FRZ 1
#define NMP_FIXED_INF 1
#define NMP_FIXED_RAD 3
#define NMP_FIXED_ALB 2
#define NMP_FIXED_SNF 1
#define NMP_FIXED_TBOT 2
#define NMP_FIXED_STC 1
#endif
struct Opt {
  static constexpr int dveg = NMP_FIXED_DVEG, crs = NMP_FIXED_CRS, btr = NMP_FIXED_BTR, run = NMP_FIXED_RUN, sfc = NMP_FIXED_SFC,
                       frz = NMP_FIXED_FRZ, inf = NMP_FIXED_INF, rad = NMP_FIXED_RAD, alb = NMP_FIXED_ALB, snf = NMP_FIXED_SNF,
                       tbot = NMP_FIXED_TBOT, stc = NMP_FIXED_STC;
};
#else
struct Opt {   // the 12 option integers, uniform over the grid (drv:15-17)
  int dveg, crs, btr, run, sfc, frz, inf, rad, alb, snf, tbot, stc;
};
#endif

// Per-type constants that follow from the parameter tables alone: what every column of a soil type (x urban override, lsm:9294-9300)
// or vegetation type would evaluate again and again -- THKS**(1-SMCMAX) (three powf, lsm:2076-2094), THKDRY, the dry-layer part of
// RSURF (a powf, lsm:1655), KDT, FRZX, the leaf-orientation integrals of TWOSTREAM (a logf and two divisions, lsm:2891-2897).
// (The float64 reciprocals of SMCMAX, SMCREF-SMCWLT, PSISAT are formed from registers by rc64 instead.)  Filled once per noahmp_hip_set_tables on the HOST by
// derive_tables() (nmp_dev_sflx.hpp) with the very functions and float32 operations the device code used to run per column
// (nmp_libm is host + device and held to the reference libm bit for bit on both).
constexpr int NSLT = 30, NVEGT = 27;
struct Derived {
  float thks_pow[2][NSLT], thkdry[2][NSLT], d_rsurf[2][NSLT], frzx[2][NSLT];      // [urban override][soil type - 1]
  float kdt[NSLT];
  float chil[NVEGT], phi1[NVEGT], phi2[NVEGT], avmu[NVEGT];                       // [vegetation type - 1]
};

// the device image of the tables: the ABI struct, then what derive_tables() made of it
struct TablesDev { noahmp_tables t; Derived d; };
NMP_DEV const Derived* derived_of(const noahmp_tables* T) { return &reinterpret_cast<const TablesDev*>(T)->d; }

struct Parm {  // REDPRM output (lsm:9282-9335): per-column, in registers instead of module globals
  int nroot;
  int st, u;     // soil type - 1 as REDPRM validated it, urban override 0 / 1: the column's row of Derived
  float rgl, rsmin, hs, rsmax, topt;
  float bexp, smcmax, smcref, psisat, dksat, dwsat, smcwlt, quartz;
  float slope, csoil, zbot, czil, kdt, frzx;
  // per-type constants of `Derived`, fetched in REDPRM's own batch of table gathers (a gather that is waited on later, on its own,
  // costs a ~1500-cycle round trip at two waves per SIMD; the float64 reciprocals are cheaper to form from registers, rc64: 37 cycles)
  float thks_pow, thkdry, d_rsurf, chil, phi1, phi2, avmu;
  // rows of the vegetation tables that later phases read right at their start, fetched in the same batch: PHENOLOGY's monthly
  // LAI / SAI at the column's two months and HVT / HVB / TMIN, ENERGY's Z0MVT / CWPVT, VEGE_FLUX's DLEAF
  float lai1, lai2, sai1, sai2, hvt, hvb, tmin, z0mvt, cwpvt, dleaf;
  float ph_wt1;                     // PHENOLOGY's interpolation weight WT1 (lsm:1065)
  float ch2op;                      // CANWATER's CH2OP(VEGTYP), fetched with the water-phase rows (redprm_water)
};

// ---- Exact division through a float64 reciprocal (round 3) --------------------------------------------------------------
// The reference divides in float32 (IEEE, round to nearest even); the compiler's sequence for that costs ~50 SIMD cycles
// (2 v_div_scale, v_rcp, 6 FMA/MUL, v_div_fmas, v_div_fixup: tools/micro/div_known.hip).  For a divisor y whose float64
// reciprocal r = (1/y)(1 + e), |e| <= 2^-50, is at hand,
//        RN32(x / y) == (float)((double)x * r)            for EVERY float32 x                              (12.5 cycles)
// because (i) the product carries a relative error below 2^-50 + 2^-53 < 2^-49, and (ii) a float32 quotient X/Y (24-bit
// integers X, Y) that is not itself representable lies at least 1/(Y 2^24) > 2^-49 |X/Y| away from every float32 rounding
// boundary M/2^24 (M odd): |X 2^24 - M Y| is a non-zero integer.  So the float64 value and the exact quotient round to the same
// float32; zeros keep their sign, infinities and NaNs propagate as in the division, overflow rounds to infinity in the final
// conversion, and gradual underflow is rounded once (the only inputs that can differ are exact ties BELOW the normal range,
// |x| < |y| 2^-126).  tests/test_libm.py sweeps all 2^32 numerators for every constant divisor used below, and all 2^32
// divisors for rc64().
// Where r comes from: compile-time constants (NMP_RCC: folded by the compiler), scalars uniform over the grid (Urc, filled by
// the host: DT, the soil-layer geometry), per-column parameters (Parm: once per column-step), and run-time divisors that serve
// three or more divisions (rc64: v_rcp_f64 + one cubic Newton step, 2^-53; ~37 cycles).
NMP_DEV float div_rc(float x, double r) { return (float)((double)x * r); }
#define NMP_RCC(c) (1.0 / (double)(c))
NMP_DEV double rc64(float y) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double yd = (double)y;
  const double r0 = __builtin_amdgcn_rcp(yd);            // 2^-24.4 (measured over all float32 y, tools/micro/rc64.hip)
  const double e = __builtin_fma(-yd, r0, 1.0);
  const double t = __builtin_fma(e, e, e);               // r0 (1 + e + e^2): error e^3 = 2^-73, then one rounding
  return __builtin_amdgcn_div_fixup(__builtin_fma(r0, t, r0), yd, 1.0);     // y = 0, Inf, NaN as 1/y
#else
  return 1.0 / (double)y;
#endif
}

// float64 reciprocals of the divisors that are uniform over the grid (host: ctx_fill_uniform)
struct Urc {
  double dt;             // 1 / DT
  double dt_hfus;        // 1 / (DT*HFUS), the float32 product the reference divides by (lsm:6836, 6843)
  double dz[NL];         // 1 / DZ(k), k = 1..4: DZ(1) = -ZSOIL(1), DZ(k) = ZSOIL(k-1) - ZSOIL(k)     (slots L(1)..L(4))
  double dz2[NL];        // 1 / (ZSOIL(k-1) - ZSOIL(k+1)), k = 1..3, ZSOIL(0) = 0 (SRT's TEMP1, lsm:8150-8163)
  double dzmm[NL];       // 1 / (DZ(k)*1000.)
  double zs[NL];         // 1 / (-ZSOIL(k)): the root-zone depth ZROOT of a column whose roots reach layer k (lsm:1619, 8897)
  double one_m_ea4;      // 1 / (1 - EXP(-4.)) (SOILWATER's FCR, lsm:7770)
};

// The scalars of the parameter tables (category indices and counts, the GENPARM values): uniform over the grid, so they travel as
// kernel arguments (scalar registers).  Read through the table pointer they were VECTOR loads -- the compiler cannot prove that no
// store of the kernel clobbers them -- and PHENOLOGY's `VEGTYP == ISWATER .OR. VEGTYP == ISBARREN .OR. ...` (lsm:1087) was a chain of
// dependent memory round trips; REDPRM's range check of the soil / vegetation type (lsm:9266-9279) delayed its whole batch of gathers.
struct TabScalars {
  int iswater, isbarren, issnow, eblforest, lucats, slcats;
  float csoil, zbot, czil, topt, rsmax, slope0;      // CSOIL_DATA, ZBOT_DATA, CZIL_DATA, TOPT_DATA, RSMAX_DATA, SLOPE_DATA(1) (SLOPETYP = 1, drv:525)
  // NOAHMP_RAD_PARAMETERS (lsm:409-447) as the land path reads them: soil colour class ISC = 4 and surface type IST = 1 are fixed by the
  // driver (drv:526-527), so ALBSAT / ALBDRY(ISC, band) and EG(IST) are grid-uniform too
  float albsat4[2], albdry4[2], omegas[2], betads, betais, eg1;
};
constexpr int NMP_TS_INTS = 6, NMP_TS_FLOATS = 15;    // members of TabScalars as the type-free LaunchDesc carries them (nmp_engine_host.hpp)
NMP_DEV TabScalars tab_scalars(const noahmp_tables& t) {
  TabScalars r;
  r.iswater = t.iswater; r.isbarren = t.isbarren; r.issnow = t.issnow; r.eblforest = t.eblforest; r.lucats = t.lucats; r.slcats = t.slcats;
  r.csoil = t.csoil_data; r.zbot = t.zbot_data; r.czil = t.czil_data; r.topt = t.topt_data; r.rsmax = t.rsmax_data; r.slope0 = t.slope_data[0];
  for (int ib = 0; ib < 2; ib++) { r.albsat4[ib] = t.albsat[ib][3]; r.albdry4[ib] = t.albdry[ib][3]; r.omegas[ib] = t.omegas[ib]; }
  r.betads = t.betads; r.betais = t.betais; r.eg1 = t.eg[0];
  return r;
}
// TabScalars <-> the two plain arrays of LaunchDesc / Engine (host)
NMP_DEV void tab_scalars_pack(const TabScalars& r, int* i, float* f) {
  i[0] = r.iswater; i[1] = r.isbarren; i[2] = r.issnow; i[3] = r.eblforest; i[4] = r.lucats; i[5] = r.slcats;
  f[0] = r.csoil; f[1] = r.zbot; f[2] = r.czil; f[3] = r.topt; f[4] = r.rsmax; f[5] = r.slope0;
  f[6] = r.albsat4[0]; f[7] = r.albsat4[1]; f[8] = r.albdry4[0]; f[9] = r.albdry4[1]; f[10] = r.omegas[0]; f[11] = r.omegas[1];
  f[12] = r.betads; f[13] = r.betais; f[14] = r.eg1;
}
NMP_DEV TabScalars tab_scalars_unpack(const int* i, const float* f) {
  TabScalars r;
  r.iswater = i[0]; r.isbarren = i[1]; r.issnow = i[2]; r.eblforest = i[3]; r.lucats = i[4]; r.slcats = i[5];
  r.csoil = f[0]; r.zbot = f[1]; r.czil = f[2]; r.topt = f[3]; r.rsmax = f[4]; r.slope0 = f[5];
  r.albsat4[0] = f[6]; r.albsat4[1] = f[7]; r.albdry4[0] = f[8]; r.albdry4[1] = f[9]; r.omegas[0] = f[10]; r.omegas[1] = f[11];
  r.betads = f[12]; r.betais = f[13]; r.eg1 = f[14];
  return r;
}

// launch-uniform context (kernel argument, lands in SGPRs)
struct Ctx {
  const noahmp_tables* __restrict__ T;
  const Derived* __restrict__ D;
  Opt O;
  float dt;
  float zsoil[NL];   // zsoil[L(1..4)], drv:392-395
  int isurban;
  TabScalars ts;     // host: tab_scalars(the tables passed to noahmp_hip_set_tables)
  Urc u;
  // optional cost record of this launch (libraries built with -DNMP_COST_RECORD, set_option "record_cost"; NULL = off): two bytes per column, [2 t] = iterations of VEGE_FLUX's
  // canopy loop (lsm:3234; 0 without a canopy), [2 t + 1] = STOMATA bisection steps of both leaves (lsm:5413) -- the two trip counts
  // that differ from column to column.  Read back by the column sort (NOAHMP_SORT_COST): a wavefront runs as long as its slowest lane.
  unsigned char* cost;
};
NMP_DEV void record_cost(const Ctx& c, int which, int n) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(NMP_COST_RECORD)       // experiment builds only: the stores cost the land kernel 1.2 % (r05)
  if (c.cost) c.cost[2 * (size_t)(blockIdx.x * blockDim.x + threadIdx.x) + which] = (unsigned char)n;
#else
  (void)c; (void)which; (void)n;
#endif
}

// Element k (0..NSOIL: a root depth) of a grid-uniform array that lives in the kernel arguments, for a k that differs per lane: a chain of
// selects on scalar operands.  Indexed directly, such an array is read by a vector load from the kernel-argument segment -- a dependent
// memory round trip (~500 cycles exposed) for a word the scalar unit already holds.
// uniform_value(x): x itself for a value that is uniform over the wavefront, as a VALUE the optimiser cannot trace back to its address
// (v_readfirstlane): keeps select(load a, load b) from becoming load(select(&a, &b))
#if defined(__HIP_DEVICE_COMPILE__)
NMP_DEV float uniform_value(float x) { int i; memcpy(&i, &x, 4); i = __builtin_amdgcn_readfirstlane(i); memcpy(&x, &i, 4); return x; }
NMP_DEV double uniform_value(double x) {
  int i[2]; memcpy(i, &x, 8);
  i[0] = __builtin_amdgcn_readfirstlane(i[0]); i[1] = __builtin_amdgcn_readfirstlane(i[1]);
  memcpy(&x, i, 8); return x;
}
#else
NMP_DEV float uniform_value(float x) { return x; }
NMP_DEV double uniform_value(double x) { return x; }
#endif
template <class T> NMP_DEV T pick_layer(const T* a, int k) {
  // (no local array of the values: the float64 instantiation's one stayed in scratch memory -- three 16-byte stores and loads per lane)
  T r = uniform_value(a[L(0)]);
#pragma unroll
  for (int j = 1; j < NSOIL; j++) { const T vj = uniform_value(a[L(j)]); r = (k == j) ? vj : r; }
  const T vn = uniform_value(a[L(NSOIL)]);
  r = (k >= NSOIL) ? vn : r;
  return r;
}

// a[L(k)] of a per-column soil array, 1 <= k <= NSOIL, as nested selects (k == 1) ? a1 : (k == 2) ? a2 : ... : aN -- no run-time indexing: the
// arrays are registers.  (Written as a loop that starts from aN and overrides it, the compiler folded the selects of loads into one load at a
// selected address: three arrays went to scratch memory, 48 bytes of stores per column -- found in the write counters, round 5.)
template <int J, class A> NMP_DEV float pick_soil_from(const A& a, int k) {
  if constexpr (J >= NSOIL) return a[L(NSOIL)];
  else return (k == J) ? a[L(J)] : pick_soil_from<J + 1>(a, k);
}
template <class A> NMP_DEV float pick_soil(const A& a, int k) { return pick_soil_from<1>(a, k); }
// a[L(k + 1)] for 1 <= k <= NSOIL - 2, the same way: (k == 1) ? a2 : (k == 2) ? a3 : ... : a(NSOIL-1)
template <int J, class A> NMP_DEV float pick_soil_below_from(const A& a, int k) {
  if constexpr (J >= NSOIL - 2) return a[L(NSOIL - 1)];
  else return (k == J) ? a[L(J + 1)] : pick_soil_below_from<J + 1>(a, k);
}
template <class A> NMP_DEV float pick_soil_below(const A& a, int k) { return pick_soil_below_from<1>(a, k); }
// ANY(X(1:4) ...) of PHASECHANGE_GLACIER (gla:1804, 1829, 1854, 1883): the reference tests layers 1..4 whatever NSOIL is
template <class A, class P> NMP_DEV bool any_of_layers_1_to_4(const A& a, P pred) {
  bool r = false;
#pragma unroll
  for (int j = 1; j <= (NSOIL < 4 ? NSOIL : 4); j++) r = r || pred(a[L(j)]);
  return r;
}

// soil-layer thickness as SOILWATER sees it (DZSNSO(1..4) after SNOWWATER rebuilt the layer geometry, lsm:6978-6994): uniform
NMP_DEV float dz_soil(const Ctx& c, int k) { return k == 1 ? -c.zsoil[L(1)] : (c.zsoil[L(k - 1)] - c.zsoil[L(k)]); }

// host: everything of Ctx that follows from dt and zsoil
NMP_DEV void ctx_fill_uniform(Ctx& c) {
  c.u.dt = 1.0 / (double)c.dt;
  c.u.dt_hfus = 1.0 / (double)(c.dt * HFUS);
  for (int k = 1; k <= NSOIL; k++) {
    const float dz = dz_soil(c, k);
    c.u.dz[L(k)] = 1.0 / (double)dz;
    c.u.dzmm[L(k)] = 1.0 / (double)(dz * 1000.f);
    c.u.zs[L(k)] = 1.0 / (double)(-c.zsoil[L(k)]);
    if (k < NSOIL) c.u.dz2[L(k)] = 1.0 / (double)(k == 1 ? -c.zsoil[L(2)] : (c.zsoil[L(k - 1)] - c.zsoil[L(k + 1)]));
  }
  c.u.one_m_ea4 = 1.0 / (double)(1.0f - nmp_expf(-4.0f));
}

// per-column scalar state of one NOAHMP_SFLX call (lsm:518-543); layer arrays are separate
struct Col {
  // in
  float lat, julian, cosz, shdfac, shdmax, sfctmp, sfcprs, psfc, uu, vv, q2, soldn, lwdn, prcp,
        tbot, co2air, o2air, foln, zlvl;
  int yearlen, ice, ist, isc, vegtyp;
  // inout
  float albold, sneqvo, tah, eah, fwet, canliq, canice, tv, tg, qsfc, qsnow;
  int isnow;
  float snowh, sneqv, zwt, wa, wt, wslake, lfmass, rtmass, stmass, wood, stblcp, fastcp, lai, sai,
        cm, ch, tauss, smcwtd, deeprech, rech;
  // the six accumulators the final scatter adds to (drv:731-732, 751-752, 833-834): their old values travel with the WATER phase's
  // inputs (gather_water_state), so that the scatter is stores only -- read there, each one was a dependent memory round trip behind
  // `s_waitcnt vmcnt(0)` at the very end of the wave (the compiler cannot move a load above earlier stores to other arrays)
  float acc_sfcrunoff, acc_udrunoff, acc_acsnow, acc_acsnom, acc_rech, acc_deeprech;
  // out
  float fsa, fsr, fira, fsh, ssoil, fcev, fgev, fctr, ecan, etran, edir, trad, tgb, tgv, t2mv, t2mb,
        q2v, q2b, runsrf, runsub, apar, psn, sav, sag, fsno, nee, gpp, npp, fveg, albedo, qsnbot,
        ponding, ponding1, ponding2, rssun, rssha, bgap, wgap, chv, chb, emissi, shg, shc, shb, evg,
        evb, ghv, ghb, irg, irc, irb, tr, evc, chleaf, chuc, chv2, chb2, fpice;
  // SFLX-internal hand-offs ENERGY -> WATER/ERROR (lsm:547-760)
  float thair, qair, eair, rhoair, qprecc, qprecl, solad0, solad1, solai0, solai1, swdown;
  float elai, esai, htop, igs, btran, latheav, latheag, qmelt, fsrv, fsrg;
  int frozen_canopy, frozen_ground;
  int err;
};

template <class A>
struct Lay {        // the layer arrays of a column
  A stc, zsnso, dzsnso;              // -2..4
  A smc, sh2o, sice, smceq, btrani;  // 1..4 (stored in slots L(1)..L(4))
  A snice, snliq, ficeold;           // -2..0
  A imelt;                           // -2..4, phase-change flag stored as float 0/1/2
};

NMP_DEV void raise(Col& s, int code) { if (!s.err) s.err = code; }

}  // namespace nmp

/* TEST INFRASTRUCTURE (oracle) -- CPU restatement of the reference's column physics.
 *
 * Plain C99, float32, scalar, one column at a time, same operation order as the
 * reference Fortran (phys/module_sf_noahmplsm.F90 = "lsm", phys/module_sf_noahmpdrv.F90 = "drv").
 * Every function cites the reference lines it follows.  This code is NEVER linked into
 * the product (noahmp_amd/csrc); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.
 *
 * Parity pin: checked against the reference itself (oracle/_ref, compiled from
 * /root/reference) by tests/test_oracle_vs_ref.py and against the committed fixtures in
 * tests/golden/ that the same reference build produced.
 */
#ifndef NOAHMP_ORACLE_H
#define NOAHMP_ORACLE_H
#include "noahmp_hip.h"
#include <math.h>
/* EXP of the restatement = the pinned build of glibc's expf (nmp_pin_expf.c: the host libm picks one of two builds that differ
 * at two arguments; the checker must not depend on the host it runs on) */
float nmp_pin_expf(float x);
int nmp_pin_expf_host_variant_is_pinned(void);
#ifndef NMP_PIN_EXPF_IMPL
#define expf nmp_pin_expf
#endif

typedef float real;

/* layer index -2..NSOIL -> C index 0..6 (snow -2..0, soil 1..4) */
#define L(i) ((i) + 2)
#define NL 7

/* physical constants, lsm:12-28 */
#define GRAV   9.80616f
#define SB     5.67E-08f
#define VKC    0.40f
#define TFRZ   273.16f
#define HSUB   2.8440E06f
#define HVAP   2.5104E06f
#define HFUS   0.3336E06f
#define CWAT   4.188E06f
#define CICE   2.094E06f
#define CPAIR  1004.64f
#define TKWAT  0.6f
#define TKICE  2.2f
#define TKAIR  0.023f
#define RAIR   287.04f
#define RW     461.269f
#define DENH2O 1000.f
#define DENICE 917.f
/* lsm:180-188 */
#define TIMEAN 10.5f
#define FSATMX 0.38f
#define M_MELT 2.50f
#define Z0SNO  0.002f
#define SSI    0.03f
#define SWEMX  1.00f

typedef struct {
  int dveg, opt_crs, opt_btr, opt_run, opt_sfc, opt_frz, opt_inf, opt_rad, opt_alb, opt_snf,
      opt_tbot, opt_stc;                         /* lsm:112-177 */
} nmp_opt;

typedef struct {                                 /* per-column parameters, REDPRM lsm:9282-9335 */
  int  nroot;
  real rgl, rsmin, hs, rsmax, topt;
  real bexp, smcdry, f1, smcmax, smcref, psisat, dksat, dwsat, smcwlt, quartz;
  real slope, csoil, zbot, czil, kdt, frzx;
} nmp_parm;

typedef struct {
  const noahmp_tables* T;
  nmp_opt  O;
  nmp_parm P;
  real dt;
  int  nsoil, nsnow;
  int  vegtyp;      /* 1-based category */
  int  isurban;
  int  err;         /* first NOAHMP_ERR_* raised in this column */
  real zsoil[NL];   /* zsoil[L(1..nsoil)], drv:392-395 */
} nmp_ctx;

/* all in/out scalars of one NOAHMP_SFLX call (lsm:518-543) */
typedef struct {
  /* in */
  real lat, julian, cosz, dx, dz8w, shdfac, shdmax, sfctmp, sfcprs, psfc, uu, vv, q2, soldn, lwdn,
       prcp, tbot, co2air, o2air, foln, zlvl;
  int  yearlen, ice, ist, isc;
  real smceq[NL], ficeold[NL];
  /* inout */
  real albold, sneqvo, stc[NL], sh2o[NL], smc[NL], tah, eah, fwet, canliq, canice, tv, tg, qsfc,
       qsnow;
  int  isnow;
  real zsnso[NL], snowh, sneqv, snice[NL], snliq[NL], zwt, wa, wt, wslake, lfmass, rtmass, stmass,
       wood, stblcp, fastcp, lai, sai, cm, ch, tauss, smcwtd, deeprech, rech;
  /* out */
  real fsa, fsr, fira, fsh, ssoil, fcev, fgev, fctr, ecan, etran, edir, trad, tgb, tgv, t2mv, t2mb,
       q2v, q2b, runsrf, runsub, apar, psn, sav, sag, fsno, nee, gpp, npp, fveg, albedo, qsnbot,
       ponding, ponding1, ponding2, rssun, rssha, bgap, wgap, chv, chb, emissi, shg, shc, shb, evg,
       evb, ghv, ghb, irg, irc, irb, tr, evc, chleaf, chuc, chv2, chb2, fpice;
} nmp_column;

void nmp_redprm(nmp_ctx* c, int vegtyp, int soiltyp, int slopetyp);
void nmp_sflx(nmp_ctx* c, nmp_column* s);

/* same ABI as the HIP engine / the reference harness (host pointers only) */
int nmp_oracle_set_tables(const noahmp_tables* t);
int nmp_oracle_step(const noahmp_step_args* a, noahmp_status* st);
int nmp_oracle_init(const noahmp_step_args* a, int iswater, int fndsnowh, noahmp_status* st);   /* drv:847-1283 */
int nmp_oracle_forcing_prep(const noahmp_step_args* a, const float* lon2d, const float* rain_rate, int iday, int ihour,
                            int iminute, int isecond, float zlvl, int scale_vegfra, float* julian_out);   /* hdrv:336-354, 813-863 */
int nmp_oracle_wtable_mmf(const noahmp_wtable_args* a, noahmp_status* st);
int nmp_oracle_groundwater_init(const noahmp_wtable_args* a, int iswater, noahmp_status* st);   /* drv:1286-1522 */   /* gw:14-198 */

#endif

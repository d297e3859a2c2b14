// Noah-MP column engine for MI355X -- one column-step: the ILOOP body of noahmplsm (drv:424-837).
// Shared by the GPU kernel (noahmp_engine.hip) and by the host-compiled emulation used only by
// tests/ (tests/host_emul) to debug the device source without a GPU.
#pragma once
#ifndef __HIPCC_RTC__
#include <string.h>
#endif
#include "nmp_dev_sflx.hpp"
#include "nmp_dev_glacier.hpp"

namespace nmp {

struct KArgs {
  noahmp_step_args a;     // array members hold DEVICE pointers here
  Ctx c;
  int ni, nka;            // memory extents: ni = ime-ims+1, nka = kme-kms+1
  int nti, ntj;           // tile extents
  int k1;                 // 0-based slot of atmospheric level 1 inside (kms:kme)
  int kp_lo, kp_hi;       // slots of P8W3D(kts), P8W3D(kts+1)
  int yearlen;
  unsigned long long* err;   // min over columns of ((linear index + 1) << 8 | code)
  int* counts;               // [0]=land [1]=glacier [2]=skipped
  unsigned long long err_base;   // step ordinal << 40 for asynchronous stepping (0 otherwise)
  long t_offset;                 // tile index of this launch's first column when a tile is advanced in row chunks
  long t_first, t_count;         // class-range launches (sorted layout): this launch covers tile indices [t_first, t_first + t_count)
  long r_land, r_ice, r_skip;    // noahmp_ranges_kernel: columns of the land, land-ice and skipped range (in this order in the arrays)
};

constexpr int LAY_SLOTS = 4 * NL + 5 * NSOIL + 3 * NSNOW;   // stc,zsnso,dzsnso,imelt | smc,sh2o,sice,smceq,btrani | snice,snliq,ficeold

template <int STRIDE>
NMP_DEV Lay<LArr<STRIDE>> make_lay(float* base) {
  Lay<LArr<STRIDE>> y;
  int o = 0;
  y.stc.p = base + o * STRIDE; o += NL;
  y.zsnso.p = base + o * STRIDE; o += NL;
  y.dzsnso.p = base + o * STRIDE; o += NL;
  y.imelt.p = base + o * STRIDE; o += NL;
  // soil-only arrays keep slots L(1)..L(NSOIL): bias the base pointer by -L(1) slots
  y.smc.p = base + (o - L(1)) * STRIDE; o += NSOIL;
  y.sh2o.p = base + (o - L(1)) * STRIDE; o += NSOIL;
  y.sice.p = base + (o - L(1)) * STRIDE; o += NSOIL;
  y.smceq.p = base + (o - L(1)) * STRIDE; o += NSOIL;
  y.btrani.p = base + (o - L(1)) * STRIDE; o += NSOIL;
  y.snice.p = base + o * STRIDE; o += NSNOW;
  y.snliq.p = base + o * STRIDE; o += NSNOW;
  y.ficeold.p = base + o * STRIDE; o += NSNOW;
  return y;
}

// Memory position of a column's word in one of the caller's arrays.  NMP_WIDE_INDEX (the generic translation unit): 64-bit element
// indices, any array size.  Otherwise (option-specialised units): 32-bit BYTE offsets from the array's base, which is a kernel argument in
// scalar registers -- an access can be `global_load_dword v, v_offset, s[base:base+1]` with no vector address arithmetic, and the offsets
// that stay live are single registers instead of pairs (land kernel: 13 spilled registers -> 0, -1.4 %).  The explicit raw-buffer form
// (`buffer_load_dword ... offen` for every access) measured no better.  The host sends a call whose largest array reaches 4 GiB to the
// generic unit (noahmp_hip_index_width, noahmp_engine.hip: fixed_level).
#ifdef NMP_WIDE_INDEX
typedef size_t nmp_ij_t;
#define G2(f) k.a.f[ij]
#define G3(f, lev, nk) k.a.f[((size_t)jj * (nk) + (lev)) * k.ni + ii]
#else
typedef uint32_t nmp_ij_t;
template <class T> NMP_DEV T& at32(T* base, uint32_t idx) {
  const uint32_t o = idx * (uint32_t)sizeof(T);
  return *(T*)((char*)base + (size_t)o);
}
#define G2(f) nmp::at32(k.a.f, ij)
#define G3(f, lev, nk) nmp::at32(k.a.f, ((uint32_t)jj * (uint32_t)(nk) + (uint32_t)(lev)) * (uint32_t)k.ni + (uint32_t)ii)
#endif


// memory position of tile column t (no memory access); false: outside the tile
NMP_DEV bool column_index(const KArgs& k, long t, int& ii, int& jj, nmp_ij_t& ij) {
  if (t >= (long)k.nti * k.ntj) return false;
  const int tj = (int)(t / k.nti), ti = (int)(t - (long)tj * k.nti);
  ii = k.a.its - k.a.ims + ti;
  jj = k.a.jts - k.a.jms + tj;
  ij = (nmp_ij_t)jj * (nmp_ij_t)k.ni + (nmp_ij_t)ii;
  return true;
}

// Classify a column from its XLAND, XICE, IVGTYP and apply the water / sea-ice shortcuts (drv:399-441).
// returns 0 land, 1 glacier, 2 skipped
NMP_DEV int column_classify_values(const KArgs& k, float xland, float xice, int ivg, int ii, int jj, nmp_ij_t ij) {
  int ice = (xice >= k.a.xice_thres) ? 1 : ((ivg == k.a.isice) ? -1 : 0);      // drv:426-432
  const bool water = (xland - 1.5f) >= 0.f;
  if (k.a.itimestep == 1) {                                                      // drv:399-419
    if (water) {
      G2(smstav) = 1.0f; G2(smstot) = 1.0f;
      for (int l = 0; l < NSOIL; l++) { G3(smois, l, NSOIL) = 1.0f; G3(tslb, l, NSOIL) = 273.16f; }
    } else if (xice == 1.f) {
      G2(smstav) = 1.0f; G2(smstot) = 1.0f;
      for (int l = 0; l < NSOIL; l++) G3(smois, l, NSOIL) = 1.0f;
    }
  }
  if (water) return 2;
  if (ice == 1) {                                                                // drv:436-441
    for (int l = 0; l < NSOIL; l++) G3(sh2o, l, NSOIL) = 1.0f;
    G2(xlaixy) = 0.01f;
    return 2;
  }
  return (ice == -1) ? 1 : 0;
}

// Classify column t (loads its XLAND, XICE, IVGTYP).  returns 0 land, 1 glacier, 2 skipped, 3 outside the tile
NMP_DEV int column_classify(const KArgs& k, long t, int& ii, int& jj, nmp_ij_t& ij) {
  if (!column_index(k, t, ii, jj, ij)) return 3;
  const float xland = G2(xland), xice = G2(xice);
  const int ivg = G2(ivgtyp);
  return column_classify_values(k, xland, xice, ivg, ii, jj, ij);
}

// Outputs that are final once the ENERGY phase is done (nothing in WATER / CARBON / the SFLX tail touches
// them).  Land columns store them right after ENERGY so that their ~48 registers are free during the
// water phase (fewer spills at 2 waves/SIMD); glacier columns store them with everything else.
NMP_DEV void scatter_energy_outputs(const KArgs& k, const Col& s, nmp_ij_t ij) {
  G2(tsk) = s.trad; G2(hfx) = s.fsh; G2(grdflx) = s.ssoil;                             // drv:728-730
  if (s.albedo > -999) G2(albedo) = s.albedo;                                          // drv:741
  G2(snowc) = s.fsno; G2(emiss) = s.emissi;
  G2(tgxy) = s.tg; G2(eahxy) = s.eah; G2(tahxy) = s.tah; G2(cmxy) = s.cm; G2(chxy) = s.ch;
  G2(alboldxy) = s.albold; G2(taussxy) = s.tauss; G2(sneqvoxy) = s.sneqvo;
  G2(t2mvxy) = s.t2mv; G2(t2mbxy) = s.t2mb; G2(q2mvxy) = s.q2v / (1.0f - s.q2v);       // drv:789-791
  G2(tradxy) = s.trad; G2(fvegxy) = s.fveg; G2(fsaxy) = s.fsa; G2(firaxy) = s.fira;
  G2(aparxy) = s.apar; G2(psnxy) = s.psn; G2(savxy) = s.sav; G2(sagxy) = s.sag;
  G2(rssunxy) = s.rssun; G2(rsshaxy) = s.rssha; G2(bgapxy) = s.bgap; G2(wgapxy) = s.wgap;
  G2(tgvxy) = s.tgv; G2(tgbxy) = s.tgb; G2(chvxy) = s.chv; G2(chbxy) = s.chb;
  G2(ircxy) = s.irc; G2(irgxy) = s.irg; G2(shcxy) = s.shc; G2(shgxy) = s.shg; G2(evgxy) = s.evg;
  G2(ghvxy) = s.ghv; G2(irbxy) = s.irb; G2(shbxy) = s.shb; G2(evbxy) = s.evb; G2(ghbxy) = s.ghb;
  G2(trxy) = s.tr; G2(evcxy) = s.evc; G2(chleafxy) = s.chleaf; G2(chucxy) = s.chuc;
  G2(chv2xy) = s.chv2; G2(chb2xy) = s.chb2;
}

// The part of the gather (drv:449-545) that only WATER, CARBON and the final pass-through scatter read.
NMP_DEV void gather_water_state(const KArgs& k, Col& s, nmp_ij_t ij) {
  s.wslake = G2(wslakexy); s.zwt = G2(zwtxy); s.wt = G2(wtxy);
  s.lfmass = G2(lfmassxy); s.rtmass = G2(rtmassxy); s.stmass = G2(stmassxy); s.wood = G2(woodxy);
  s.stblcp = G2(stblcpxy); s.fastcp = G2(fastcpxy);
  s.smcwtd = G2(smcwtdxy);
  s.acc_sfcrunoff = G2(sfcrunoff); s.acc_udrunoff = G2(udrunoff); s.acc_acsnow = G2(acsnow); s.acc_acsnom = G2(acsnom);
  s.acc_rech = G2(rechxy); s.acc_deeprech = G2(deeprechxy);
}

// Gather -> REDPRM -> NOAHMP_SFLX | NOAHMP_GLACIER -> scatter for one land / glacier column.
// Returns the column's status word (0 = ok).  A column that fails before or inside the ENERGY phase is
// left untouched; one that fails the closing water-balance check has already stored its energy-phase
// outputs (the reference STOPs at that point, so nothing downstream can observe the difference).
//
// `runner` executes iterations 2..20 of the canopy loop (SimpleLoop: each lane its own column).  The interface
// lets a runner synchronise the workgroup, which is why threads without a column (cls == 2) may call this
// function too and every phase is guarded by `live`; see DESIGN.md section 6 for the lane-compaction runner that
// was measured and dropped.
// MODE 0: any mix of land and glacier columns; 1: land columns only (no glacier code in the kernel); 2: glacier columns only.
// EARLY (class-range kernels, MODE 1 / 2): the caller has NOT classified the column; its XLAND / XICE / IVGTYP travel with the gather,
// one memory round trip instead of two in a row at the start of every wave, and the class comes back through *cls_out -- a column
// of another class than the range was declared to hold is left untouched (the caller raises NOAHMP_ERR_CLASS_RANGE).
template <int STRIDE, int MODE = 0, bool EARLY = false, class Runner>
NMP_DEV int column_step(const KArgs& k, int cls, int ii, int jj, nmp_ij_t ij, float* base, Runner& runner, int* cls_out = nullptr) {
  Lay<LArr<STRIDE>> y = make_lay<STRIDE>(base);
  // value-initialise (NOT memset(): HIP's device memset is a byte loop through a pointer PHI, which
  // pins the whole struct in scratch and defeats scalar replacement -- 556 B/lane of scratch traffic)
  Col s = {};
  Parm P = {};
  bool live = (cls == 0);          // advances through NOAHMP_SFLX
  int failed = 0;
  int soiltyp_w = 1;               // the validated soil type, kept for redprm_water
  NMP_TIC0();
  float c_xland = 0.f, c_xice = 0.f;
  int c_ivg = 0;
  if (EARLY) { c_xland = G2(xland); c_xice = G2(xice); c_ivg = G2(ivgtyp); }
  if (EARLY || cls <= 1) {
  // ---- gather, drv:449-545.  Three parts, in this order, so that the wave pays ONE round trip to HBM and REDPRM's table gathers travel
  // under it: (A) every load of the column's 87 words, nothing else (a branch or a store to LDS between two loads splits the batch: the
  // loads behind it are not issued before the values in front of it have arrived); (B) classification, the type remaps and REDPRM, which
  // need only the first words (vegetation / soil type, XLAND, XICE, latitude); (C) conversions and the layer arrays' way into LDS.
  // (A)
  int vegtyp = EARLY ? c_ivg : G2(ivgtyp), soiltyp = G2(isltyp);
  const float xice_in = EARLY ? c_xice : G2(xice);
  s.lat = G2(xlatin);
  s.cosz = G2(coszin);
  const float dz8w_in = G3(dz8w, k.k1, k.nka), vegfra_in = G2(vegfra), vegmax_in = G2(vegmax);
  s.tbot = G2(tmn);
  s.sfctmp = G3(t3d, k.k1, k.nka);
  const float qv_in = G3(qv3d, k.k1, k.nka);
  s.uu = G3(u_phy, k.k1, k.nka); s.vv = G3(v_phy, k.k1, k.nka);
  s.soldn = G2(swdown); s.lwdn = G2(glw);
  const float p8w_hi = G3(p8w3d, k.kp_hi, k.nka), p8w_lo = G3(p8w3d, k.kp_lo, k.nka);
  s.psfc = G3(p8w3d, k.k1, k.nka);
  const float rainbl_in = G2(rainbl);
  int isnow_in = G2(isnowxy);
  float smc_in[NSOIL], sh2o_in[NSOIL], tslb_in[NSOIL], smceq_in[NSOIL], tsno_in[3], snice_in[3], snliq_in[3], zsnso_in[NL];
#pragma unroll
  for (int l = 1; l <= NSOIL; l++) {
    smc_in[l - 1] = G3(smois, l - 1, NSOIL); sh2o_in[l - 1] = G3(sh2o, l - 1, NSOIL);
    tslb_in[l - 1] = G3(tslb, l - 1, NSOIL); smceq_in[l - 1] = G3(smoiseq, l - 1, NSOIL);
  }
#pragma unroll
  for (int l = -2; l <= 0; l++) {
    snice_in[l + 2] = G3(snicexy, l + 2, 3); snliq_in[l + 2] = G3(snliqxy, l + 2, 3); tsno_in[l + 2] = G3(tsnoxy, l + 2, 3);
  }
#pragma unroll
  for (int l = -2; l <= NSOIL; l++) zsnso_in[L(l)] = G3(zsnsoxy, l + 2, NSOIL + 3);
  s.sneqv = G2(snow); s.snowh = G2(snowh); s.qsfc = G2(qsfc);
  s.tv = G2(tvxy); s.tg = G2(tgxy); s.canliq = G2(canliqxy); s.canice = G2(canicexy);
  s.eah = G2(eahxy); s.tah = G2(tahxy); s.cm = G2(cmxy); s.ch = G2(chxy); s.fwet = G2(fwetxy);
  s.sneqvo = G2(sneqvoxy); s.albold = G2(alboldxy); s.qsnow = G2(qsnowxy);
  s.lai = G2(xlaixy); s.sai = G2(xsaixy);
  s.tauss = G2(taussxy); s.wa = G2(waxy);                  // WA enters the water balance taken at the start (lsm:703)
  // WSLAKE, ZWT, WT, SMCWTD, the carbon pools and the accumulators are first read by the WATER / CARBON phase: they are gathered there
  // (gather_water_state), not here, so that they do not occupy registers through the ENERGY phase
  // (B)
  if (EARLY) {                                   // everything above was loads: now the class
    cls = column_classify_values(k, c_xland, c_xice, c_ivg, ii, jj, ij);
    if (cls_out) *cls_out = cls;
    if (cls != MODE - 1) return 0;
    live = (cls == 0);
  }
  s.ist = 1; s.isc = 4; s.ice = (cls == 1) ? -1 : 0;
  s.yearlen = k.yearlen; s.julian = k.a.julian;
  // ISNOWXY is -NSNOW .. 0 by construction of the model (lsm:7044-7343 keep it there); the reference indexes its (-2:4) layer arrays
  // with it unchecked, here another value is a reported error like REDPRM's type checks (the column is left untouched).  With the range
  // stated, the compiler resolves every `layer > ISNOW` of a SOIL layer at compile time -- no branch, and the layer loops' LDS reads of
  // the four soil layers become one batch -- and keeps the branches of the three snow layers only.
  if (isnow_in < -NSNOW || isnow_in > 0) raise(s, NOAHMP_ERR_ISNOW_RANGE);
  s.isnow = isnow_in < -NSNOW ? -NSNOW : (isnow_in > 0 ? 0 : isnow_in);
  NMP_ASSUME(s.isnow >= -NSNOW && s.isnow <= 0);
  if (soiltyp == 14 && xice_in == 0.f) soiltyp = 7;                                 // drv:530-534
  if (vegtyp == k.a.isurban || vegtyp == 31 || vegtyp == 32 || vegtyp == 33) vegtyp = k.a.isurban;
  NMP_TIC(0);    // gather
  redprm(k.c, s, P, vegtyp, soiltyp);
  soiltyp_w = soiltyp;
  NMP_TIC(1);    // redprm
  // (C)
  s.zlvl = 0.5f * dz8w_in;
  s.shdfac = div_rc(vegfra_in, NMP_RCC(100.f));
  s.shdmax = div_rc(vegmax_in, NMP_RCC(100.f));
  s.q2 = qv_in / (1.0f + qv_in);
  s.sfcprs = (p8w_hi + p8w_lo) * 0.5f;
  s.prcp = div_rc(rainbl_in, k.c.u.dt);
#pragma unroll
  for (int l = 1; l <= NSOIL; l++) {
    y.smc[L(l)] = smc_in[l - 1]; y.sh2o[L(l)] = sh2o_in[l - 1];
    y.stc[L(l)] = tslb_in[l - 1]; y.smceq[L(l)] = smceq_in[l - 1];
    y.sice[L(l)] = 0.f; y.btrani[L(l)] = 0.f;
  }
#pragma unroll
  for (int l = -2; l <= 0; l++) {
    y.stc[L(l)] = tsno_in[l + 2]; y.snice[L(l)] = snice_in[l + 2]; y.snliq[L(l)] = snliq_in[l + 2];
    y.ficeold[L(l)] = (l > s.isnow) ? snice_in[l + 2] / (snice_in[l + 2] + snliq_in[l + 2]) : 0.f;     // drv:516-518
  }
#pragma unroll
  for (int l = -2; l <= NSOIL; l++) { y.zsnso[L(l)] = zsnso_in[L(l)]; y.dzsnso[L(l)] = 0.f; y.imelt[L(l)] = 0.f; }
  s.rech = 0.f; s.deeprech = 0.f;
  s.co2air = 395.e-06f * s.sfcprs;
  s.o2air = 0.209f * s.sfcprs;
  s.foln = 1.0f;
  if (vegtyp == 25 || vegtyp == 26 || vegtyp == 27) { s.shdfac = 0.0f; s.lai = 0.0f; }      // drv:540-545
  s.vegtyp = (vegtyp >= 1 && vegtyp <= k.c.ts.lucats) ? vegtyp : 1;
  if (s.err) { failed = s.err; live = false; }                                     // REDPRM fatals, lsm:9266-9344
  }  // cls <= 1

  float qfx_out = 0.f, lh_out = 0.f;
  if constexpr (MODE != 1)
  if (cls == 1 && !failed) {
    s.tbot = nmp_min(s.tbot, 263.15f);                                               // drv:555
    glacier(k.c, s, y);
    if (s.err) failed = s.err;
    else {
      // NOAHMP_GLACIER reads none of the WATER-phase words: SMCWTD and the accumulators are passed through, the rest is overwritten by
      // glacier_fill_undefined (drv:571-625).  Gathered here, as one batch, they are not live across the glacier physics (the land-ice
      // kernel needs all 256 registers; held from the start they were spilled one by one, each behind its own wait)
      gather_water_state(k, s, ij);
      glacier_fill_undefined(s);                                                   // drv:571-625
      qfx_out = s.edir; lh_out = s.fgev;                                           // drv:627-628
      scatter_energy_outputs(k, s, ij);
    }
  }
  if constexpr (MODE != 2) {
    float beg_wb = 0.f;
    // the WATER phase's own inputs (state words only it reads, its rows of the soil tables) are requested before TSNOSOI and
    // consumed after PHASECHANGE: their round trip is hidden (they used to be gathered, and waited for, at the start of WATER)
    bool water_inputs = false;
    auto prefetch_water = [&]() {
      gather_water_state(k, s, ij);
      redprm_water(k.c, P, soiltyp_w, s.vegtyp);
      water_inputs = true;
    };
    sflx_energy(k.c, P, s, y, beg_wb, live, runner, prefetch_water);               // all threads (see above)
    if (live) {
      if (s.err) failed = s.err;
      else {
        lh_out = s.fcev + s.fgev + s.fctr;                                         // drv:714
        scatter_energy_outputs(k, s, ij);
        if (!water_inputs) prefetch_water();          // (an ENERGY that returned before TSNOSOI)
        NMP_TIC(11);   // energy tail + early scatter
        sflx_water(k.c, P, s, y, beg_wb);
        if (s.err) failed = s.err;
        qfx_out = s.ecan + s.edir + s.etran;                                       // drv:713
      }
    }
  }
  if (cls > 1 || failed) return failed;
  // ---- scatter of everything the water phase (or the glacier tail) produced, drv:728-835
  G2(qfx) = qfx_out; G2(lh) = lh_out;
  G2(smstav) = 0.0f; G2(smstot) = 0.0f;
  G2(sfcrunoff) = s.acc_sfcrunoff + s.runsrf * k.a.dt;
  G2(udrunoff) = s.acc_udrunoff + s.runsub * k.a.dt;
#pragma unroll
  for (int l = 1; l <= NSOIL; l++) {
    G3(smois, l - 1, NSOIL) = y.smc[L(l)]; G3(sh2o, l - 1, NSOIL) = y.sh2o[L(l)];
    G3(tslb, l - 1, NSOIL) = y.stc[L(l)];
  }
  G2(snow) = s.sneqv; G2(snowh) = s.snowh;
  G2(canwat) = s.canliq + s.canice;
  G2(acsnow) = s.acc_acsnow + s.prcp * s.fpice;                                    // no *DT (drv:751)
  G2(acsnom) = s.acc_acsnom + s.qsnbot * k.a.dt + s.ponding + s.ponding1 + s.ponding2;
  G2(qsfc) = s.qsfc;
  G2(isnowxy) = s.isnow; G2(tvxy) = s.tv; G2(canliqxy) = s.canliq; G2(canicexy) = s.canice;
  G2(fwetxy) = s.fwet; G2(qsnowxy) = s.qsnow;
  G2(wslakexy) = s.wslake; G2(zwtxy) = s.zwt; G2(waxy) = s.wa; G2(wtxy) = s.wt;
#pragma unroll
  for (int l = -2; l <= 0; l++) {
    G3(tsnoxy, l + 2, 3) = y.stc[L(l)]; G3(snicexy, l + 2, 3) = y.snice[L(l)];
    G3(snliqxy, l + 2, 3) = y.snliq[L(l)];
  }
#pragma unroll
  for (int l = -2; l <= NSOIL; l++) G3(zsnsoxy, l + 2, NSOIL + 3) = y.zsnso[L(l)];
  G2(lfmassxy) = s.lfmass; G2(rtmassxy) = s.rtmass; G2(stmassxy) = s.stmass; G2(woodxy) = s.wood;
  G2(stblcpxy) = s.stblcp; G2(fastcpxy) = s.fastcp; G2(xlaixy) = s.lai; G2(xsaixy) = s.sai;
  G2(q2mbxy) = s.q2b / (1.0f - s.q2b);                                             // drv:792 (urban fix is late)
  G2(neexy) = s.nee; G2(gppxy) = s.gpp; G2(nppxy) = s.npp;
  G2(runsfxy) = s.runsrf; G2(runsbxy) = s.runsub; G2(ecanxy) = s.ecan;
  G2(edirxy) = s.edir; G2(etranxy) = s.etran;
  G2(rechxy) = s.acc_rech + s.rech * 1.E3f;
  G2(deeprechxy) = s.acc_deeprech + s.deeprech;
  G2(smcwtdxy) = s.smcwtd;
  NMP_TIC(15);   // water tail + final scatter
  return 0;
}

}  // namespace nmp

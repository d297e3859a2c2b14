"""Randomised single-step parity: broad random states and forcing (every USGS category incl. urban / glacier / water, every soil
type, snow from none to deep, 230-320 K, calm to storm, night to zenith, dry to saturated, drizzle to downpour), cold start, then
a few steps, every output compared bit for bit after every step.

  fuzz_parity.py ref  [nseeds ncol] [opt=val ...]   C restatement vs the COMPILED REFERENCE (dev container; one subprocess per
                                                     seed because the reference STOPs the process on a fatal column)
  fuzz_parity.py emul [nseeds ncol] [opt=val ...]   device source compiled for the host vs the restatement (no GPU needed)
  fuzz_parity.py gpu  [nseeds ncol] [opt=val ...]   HIP engine vs the restatement (GPU box)
  further keys: scalars=1 (DT / DZS / YR / JULIAN / ZLVL drawn per seed), modis=1 (MODIS tables and categories), steps=N (default 3),
  nan=1 (NaN / Inf / huge / denormal forcing words on 3 % of the columns; nan=2: state words too), soil=1 (soil classes 13, 15-19 on land as well)

Columns on which the restatement reports a fatal code (energy / water balance stops of the reference) are replaced by a benign
column before the comparison and counted.
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from noahmp_amd import synth  # noqa: E402
from noahmp_amd.state import ModelConfig  # noqa: E402
from noahmp_amd.tables import load_tables  # noqa: E402
from tools.compare import exact_check  # noqa: E402

F = np.float32
NSTEPS = 3
WIDE_SOIL = 0
YR, JUL = 2000, 180.0
SFC2_UNDEF = ("t2mvxy", "t2mbxy", "q2mvxy", "q2mbxy", "chv2xy", "chb2xy")


def random_tile(tb, ncol, seed, cfg):
    r = np.random.Generator(np.random.Philox(seed))
    ni, nj = ncol, 1
    s = synth._base_store(ni, nj, cfg)
    a = s.a
    shp = (nj, ni)
    a["ivgtyp"][...] = r.integers(1, int(tb.get("lucats", 27)) + 1, size=shp)
    a["isltyp"][...] = r.integers(1, 13, size=shp)
    if WIDE_SOIL:                                 # soil=1: every class of the table except water (14: FRZX of soil 14 is the documented difference)
        a["isltyp"][...] = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 16, 17, 18, 19])[r.integers(0, 18, size=shp)]
    ice = a["ivgtyp"] == cfg.isice
    water = a["ivgtyp"] == cfg.iswater
    a["isltyp"][ice] = 16
    a["isltyp"][water] = 14
    a["xland"][water] = 2.0
    a["xlatin"][...] = r.uniform(-80, 80, size=shp).astype(F)
    a["vegfra"][...] = np.where(r.random(shp) < 0.1, r.choice([0.0, 1.0, 100.0], size=shp), r.uniform(1, 99, size=shp)).astype(F)
    a["vegmax"][...] = np.maximum(a["vegfra"], r.uniform(50, 100, size=shp).astype(F))
    tair = r.uniform(233.0, 318.0, size=shp).astype(F)
    tair[ice] = np.minimum(tair[ice], F(270.0))
    a["tmn"][...] = (tair + r.uniform(-8, 8, size=shp)).astype(F)
    tsk = (tair + r.uniform(-6, 6, size=shp)).astype(F)
    a["tsk"][...] = tsk
    snowy = (r.random(shp) < 0.5) & (tsk < 274.0) | ice
    depth = np.where(snowy, np.exp(r.uniform(np.log(0.003), np.log(2.5), size=shp)), 0.0).astype(F)
    a["snowh"][...] = depth
    a["snow"][...] = (depth * r.uniform(60, 400, size=shp)).astype(F)
    for k in range(4):
        a["tslb"][:, k, :] = (tsk * F(0.5) + a["tmn"] * F(0.5) + r.uniform(-3, 3, size=shp)).astype(F)
        a["smois"][:, k, :] = r.uniform(0.03, 0.46, size=shp).astype(F)
    # forcing
    cosz = np.where(r.random(shp) < 0.4, 0.0, r.uniform(0.0, 1.0, size=shp)).astype(F)
    a["coszin"][...] = cosz
    a["swdown"][...] = (cosz * r.uniform(200, 1250, size=shp)).astype(F)
    a["glw"][...] = r.uniform(120, 480, size=shp).astype(F)
    psfc = r.uniform(5.2e4, 1.04e5, size=shp).astype(F)
    es = 611.2 * np.exp(17.67 * (tair - 273.15) / (tair - 29.65))
    q = (r.uniform(0.05, 1.0, size=shp) * 0.622 * es / (psfc - es)).astype(F)
    wind = np.where(r.random(shp) < 0.05, 0.0, np.exp(r.uniform(np.log(0.05), np.log(35.0), size=shp))).astype(F)
    ang = r.uniform(0, 2 * np.pi, size=shp)
    for lev in range(2):
        a["t3d"][:, lev, :] = tair
        a["qv3d"][:, lev, :] = q
        a["u_phy"][:, lev, :] = (wind * np.cos(ang)).astype(F)
        a["v_phy"][:, lev, :] = (wind * np.sin(ang)).astype(F)
        a["p8w3d"][:, lev, :] = psfc
    a["dz8w"][...] = F(2.0 * cfg.zlvl)
    a["rainbl"][...] = np.where(r.random(shp) < 0.35, np.exp(r.uniform(np.log(0.01), np.log(25.0), size=shp)), 0.0).astype(F)
    return s


def clean(port, s, it0):
    """Replace columns that raise a fatal code in the restatement within NSTEPS steps by column 0; return how many."""
    nrep = 0
    codes = {}
    for _ in range(max(64, s.ni // 8)):
        t = s.copy()
        bad = None
        for it in range(it0, it0 + NSTEPS):
            st = port.noahmplsm(t, it, YR, JUL)
            if st.code:
                bad = st.i - 1
                codes[st.code] = codes.get(st.code, 0) + 1
                if os.environ.get("NMP_FUZZ_VERBOSE") and nrep < 8:
                    c = bad
                    print("   fatal %d at col %d step %d: veg %d soil %d isnow %d snowh %.4f swe %.2f tsk %.2f t %.2f wind %.2f cosz %.3f sw %.1f rain %.3f"
                          % (st.code, c, it, s["ivgtyp"][0, c], s["isltyp"][0, c], s["isnowxy"][0, c], s["snowh"][0, c], s["snow"][0, c],
                             s["tsk"][0, c], s.a["t3d"][0, 0, c], float(np.hypot(s.a["u_phy"][0, 0, c], s.a["v_phy"][0, 0, c])),
                             s["coszin"][0, c], s["swdown"][0, c], s["rainbl"][0, c]))
                break
        if bad is None:
            if codes:
                print("   fatal codes replaced:", codes)
            return nrep
        assert bad != 0, "the benign column itself is fatal"
        for k, v in s.a.items():
            if k != "dzs":
                v[..., bad] = v[..., 0]
        nrep += 1
    raise RuntimeError("too many fatal columns: %s" % codes)


def draw_scalars(seed):
    """scalars=1: the uniform scalars of the boundary (drv:51-83) drawn per seed -- DT, DZS, YR, JULIAN, DZ8W (= 2 ZLVL)."""
    r = np.random.Generator(np.random.Philox(seed + 7919))
    dt = float(r.choice([600.0, 900.0, 1800.0, 3600.0]))
    dzs = [(0.1, 0.3, 0.6, 1.0), (0.05, 0.25, 0.7, 1.5), (0.07, 0.21, 0.72, 1.0)][int(r.integers(0, 3))]
    yr = int(r.choice([2000, 2001, 2004, 2100]))
    jul = float(np.float32(r.uniform(1.0, 366.0 if yr in (2000, 2004) else 365.0)))
    zlvl = float(r.choice([2.0, 10.0, 30.0]))
    return dict(dt=dt, dzs=dzs, zlvl=zlvl), yr, jul


def one_seed(mode, seed, ncol, kw):
    global YR, JUL, NSTEPS, WIDE_SOIL
    kw = dict(kw)
    YR, JUL = 2000, 180.0
    NSTEPS = kw.pop("steps", 3)
    poison = kw.pop("nan", 0)
    WIDE_SOIL = kw.pop("soil", 0)                   # steps=N: a longer free run under the same (constant) forcing
    if kw.pop("scalars", 0):
        sc, YR, JUL = draw_scalars(seed)
        kw.update(sc)
        print("seed %d scalars: dt %g dzs %s zlvl %g yr %d julian %.3f" % (seed, sc["dt"], sc["dzs"], sc["zlvl"], YR, JUL))
    modis = kw.pop("modis", 0)                    # modis=1: the MODIFIED_IGBP_MODIS_NOAH tables and category indices (hdrv:130-143)
    T, tb = load_tables("modis" if modis else "usgs")
    if modis:
        kw.update(isurban=tb["isurban"], isice=tb["issnow"], iswater=tb["iswater"])
    from oracle.portlib import PortLib
    port = PortLib(autobuild=False)
    port.set_tables(T)
    cfg = ModelConfig(**kw)
    s = random_tile(tb, ncol, seed, cfg)
    s["ivgtyp"][0, 0], s["isltyp"][0, 0], s["xland"][0, 0] = 7, 6, 1.0          # column 0: benign grass / loam
    for k in ("tsk", "tmn"):
        s[k][0, 0] = 285.0
    s.a["tslb"][0, :, 0] = 285.0
    s.a["smois"][0, :, 0] = 0.28
    s["snow"][0, 0] = s["snowh"][0, 0] = 0.0
    s.a["t3d"][0, :, 0] = 286.0
    s.a["qv3d"][0, :, 0] = 0.006
    s.a["u_phy"][0, :, 0], s.a["v_phy"][0, :, 0] = 3.0, 1.0
    s.a["p8w3d"][0, :, 0] = 95000.0
    s["swdown"][0, 0], s["coszin"][0, 0], s["glw"][0, 0], s["rainbl"][0, 0] = 400.0, 0.5, 330.0, 0.0
    if kw.get("iopt_run") == 5:
        init_cfg = ModelConfig(**dict(kw, iopt_run=1))
    else:
        init_cfg = cfg
    s.cfg = init_cfg
    rc, _ = port.noahmp_init(s, fndsnowh=True)
    assert rc == 0
    s.cfg = cfg
    if kw.get("iopt_run") == 5:
        s.a["smoiseq"][...] = s.a["smois"]
        s["smcwtdxy"] = s.a["smois"][:, -1, :]
        s["zwtxy"] = -6.0
        for k in ("waxy", "wtxy", "deeprechxy", "rechxy"):
            s[k] = 0.0
    synth.first_step_fixups(s)
    if poison:                                    # nan=1: 3 % of the columns get one forcing word no model run should see
        r = np.random.Generator(np.random.Philox(seed + 104729))
        cols = np.flatnonzero(r.random(ncol) < 0.03)
        cols = cols[cols != 0]
        keys = ("t3d", "qv3d", "u_phy", "v_phy", "p8w3d", "swdown", "glw", "rainbl", "coszin")
        if poison >= 2:                           # nan=2: any float INOUT word of the state as well
            from noahmp_amd.abi import FIELD_INFO
            keys = keys + tuple(k for k in s.a if k in FIELD_INFO and FIELD_INFO[k][2] == "inout" and s.a[k].dtype == np.float32)
        vals = (np.nan, np.inf, -np.inf, 0.0, -1.0e30, 1.0e30, 1.0e-42, -0.0)
        for c in cols:
            k = keys[r.integers(len(keys))]
            if s.a[k].ndim == 3 and poison >= 2:
                s.a[k][0, r.integers(s.a[k].shape[1]), c] = F(vals[r.integers(len(vals))])      # one level
            else:
                s.a[k][0, ..., c] = F(vals[r.integers(len(vals))])
    nrep = clean(port, s, 1)
    skip = SFC2_UNDEF if kw.get("iopt_sfc") == 2 else ()
    if mode == "ref":
        from oracle.reflib import RefLib
        other = RefLib("O0")
        other.set_tables(T)
        step = lambda x, it: other.noahmplsm(x, it, YR, JUL)
    elif mode == "emul":
        from host_emul.emullib import EmulLib
        other = EmulLib()
        other.set_tables(T)
        step = lambda x, it: other.noahmplsm(x, it, YR, JUL)
    else:
        import torch  # noqa: F401
        from noahmp_amd.driver import Engine
        other = Engine(T, device=0)
        if os.environ.get("NMP_NO_JIT"):                    # debugging: option sets outside the ahead-of-time kernels take the generic kernel
            other.lib.noahmp_hip_set_option(b"jit_option_kernels", 0)
        step = lambda x, it: other.noahmplsm(x, it, YR, JUL, check=False)
    nbad = 0
    a = s.copy()
    for it in range(1, NSTEPS + 1):
        b = a.copy()
        port.noahmplsm(a, it, YR, JUL)
        step(b, it)
        ok, lines = exact_check(a, b, skip=skip)
        if not ok:
            nbad += int(lines[0].split()[0])
            print("seed %d step %d: %s" % (seed, it, "\n   ".join(lines[:6])))
            j = 0
            ne = None
            for k in a.a:
                x, y = a.a[k], b.a[k]
                if x.dtype.kind == "f" and k not in skip and k != "dzs":
                    m = ~((x == y) | ((x != x) & (y != y)))
                    m = m.any(axis=1) if m.ndim == 3 else m
                    ne = m if ne is None else ne | m
            cols = np.argwhere(ne[0])[:4, 0]
            for c in cols:
                print("   col %d: veg %d soil %d isnow %d snowh %.4f tsk %.2f t %.2f wind %.2f cosz %.3f rain %.3f"
                      % (c, s["ivgtyp"][0, c], s["isltyp"][0, c], s["isnowxy"][0, c], s["snowh"][0, c], s["tsk"][0, c],
                         s.a["t3d"][0, 0, c], float(np.hypot(s.a["u_phy"][0, 0, c], s.a["v_phy"][0, 0, c])), s["coszin"][0, c],
                         s["rainbl"][0, c]))
    isn = np.bincount(-s["isnowxy"].ravel(), minlength=4).tolist()
    print("seed %d: %d columns x %d steps, %d fatal columns replaced, ISNOW 0..-3 counts %s: %s"
          % (seed, ncol, NSTEPS, nrep, isn, "bit-identical" if nbad == 0 else "%d column-steps DIFFER" % nbad))
    return nbad


def main():
    mode = sys.argv[1]
    pos = [x for x in sys.argv[2:] if "=" not in x and not x.startswith("--")]
    kw = {x.split("=")[0]: int(x.split("=")[1]) for x in sys.argv[2:] if "=" in x and not x.startswith("--")}
    if "--seed" in sys.argv[-1]:
        seed = int(sys.argv[-1].split(":")[1])
        sys.exit(1 if one_seed(mode, seed, int(pos[1]), kw) else 0)
    nseeds, ncol = (int(pos[0]), int(pos[1])) if len(pos) >= 2 else (4, 4096)
    bad = 0
    for seed in range(1, nseeds + 1):
        if mode == "ref":          # the reference STOPs the process on a fatal column: isolate every seed
            rc = subprocess.call([sys.executable, os.path.abspath(__file__), mode, str(nseeds), str(ncol)] +
                                 ["%s=%d" % kv for kv in kw.items()] + ["--seed:%d" % seed])
            if rc not in (0, 1):
                print("seed %d: the reference process ended with code %d" % (seed, rc))
            bad += rc != 0
        else:
            bad += one_seed(mode, seed, ncol, kw) != 0
    print("%s, options %s: %d of %d seeds with differences" % (mode, kw, bad, nseeds))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

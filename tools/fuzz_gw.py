"""Dev-container check: WTABLE_mmf_noahmp on broad random planes (every soil class, FDEPTH 3..1600 m incl. 0, PEXP 0.5..5, water table
1 cm..80 m, AREA 1e3..1e8, pending recharge up to 5 m), three successive calls: compiled reference vs C restatement vs the device source
compiled for the host, bit for bit."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from test_groundwater import gw_store, GW_OUT
from noahmp_amd.tables import load_tables
from oracle.portlib import PortLib
from oracle.reflib import RefLib
from host_emul.emullib import EmulLib
tables = load_tables("usgs")
T = tables[0]
port = PortLib(); port.set_tables(T)
ref = RefLib("O0"); ref.set_tables(T)
em = EmulLib(); em.set_tables(T)
F = np.float32
nbad = 0
for seed in range(1, 25):
    r = np.random.default_rng(seed)
    s0 = gw_store(tables, ni=96, nj=64, seed=seed, stress=float(r.choice([0.0, 0.01, 0.1, 1.0, 5.0])), area=float(10 ** r.uniform(3, 8)))
    a0 = s0.a
    shp = a0["zwtxy"].shape
    a0["isltyp"][...] = r.integers(1, 20, size=shp)
    a0["fdepth"][...] = np.where(r.random(shp) < 0.05, 0.0, 10 ** r.uniform(0.5, 3.2, size=shp)).astype(F)
    a0["pexp"][...] = r.uniform(0.5, 5.0, size=shp).astype(F)
    a0["rivercond"][...] = (10 ** r.uniform(-5, 0.5, size=shp)).astype(F)
    a0["zwtxy"][...] = -(10 ** r.uniform(-2, 1.9, size=shp)).astype(F)
    a0["eqwtd"][...] = -(10 ** r.uniform(-1, 1.7, size=shp)).astype(F)
    a0["riverbed"][...] = a0["eqwtd"] - r.uniform(0, 3, size=shp).astype(F)
    a0["xice"][...] = np.where(r.random(shp) < 0.02, 1.0, 0.0).astype(F)
    if os.environ.get("NMP_FUZZ_POISON"):         # NaN / Inf / huge / denormal in one word of 2 % of the cells
        vals = (np.nan, np.inf, -np.inf, 0.0, -1.0e30, 1.0e30, 1.0e-42, -0.0)
        keys = ("zwtxy", "eqwtd", "riverbed", "fdepth", "pexp", "rivercond", "smois", "sh2o", "smoiseq", "deeprechxy", "rechxy", "smcwtdxy")
        keys = [k for k in keys if k in a0]
        for cidx in np.argwhere(r.random(shp) < 0.02):
            k = keys[r.integers(len(keys))]
            v = F(vals[r.integers(len(vals))])
            if a0[k].ndim == 3:
                a0[k][cidx[0], r.integers(a0[k].shape[1]), cidx[1]] = v
            else:
                a0[k][cidx[0], cidx[1]] = v
    a, b, c = s0.copy(), s0.copy(), s0.copy()
    for it in range(3):
        ref.wtable_mmf(a); port.wtable_mmf(b); em.wtable_mmf(c)
        for n in GW_OUT:
            for nm, o in (("port", b), ("emul", c)):
                if not np.array_equal(a.a[n], o.a[n], equal_nan=True):
                    nbad += 1
                    bad = np.argwhere(~((a.a[n] == o.a[n]) | (np.isnan(a.a[n]) & np.isnan(o.a[n]))))
                    print("seed", seed, "call", it, nm, n, len(bad), bad[0], a.a[n][tuple(bad[0])], o.a[n][tuple(bad[0])])
        for o in (a, b, c): o.a["deeprechxy"][...] = s0.a["deeprechxy"]
print("groundwater fuzz 24 seeds x 6144 cells x 3 calls:", "bit-identical" if nbad == 0 else "%d mismatches" % nbad)

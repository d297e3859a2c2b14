import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import load_store, GOLDEN
from noahmp_amd.driver import Engine
from noahmp_amd.tables import load_tables
from noahmp_amd.state import ModelConfig
T, tb = load_tables("usgs")
eng = Engine(T, device=0)
g = np.load(os.path.join(GOLDEN, "golden_opts.npz"))
sweep = [eval(x) for x in g["sweep"]]
base = load_store(g, "init", 32, 4)
for n, kw in enumerate(sweep):
    s = base.copy(); s.cfg = ModelConfig(**kw)
    if kw.get("iopt_run") == 5: s["waxy"] = 0.0; s["wtxy"] = 0.0
    st = eng.noahmplsm(s, 1, 2000, 180.0)
    ref = g["opt%02d/wgapxy" % n]
    hn, gn = np.isnan(s['wgapxy']), np.isnan(ref)
    if hn.any() or gn.any() or kw.get("iopt_rad") == 1:
        print(kw, "hip nan", int(hn.sum()), "gold nan", int(gn.sum()), "hip inf", int(np.isinf(s['wgapxy']).sum()), "gold inf", int(np.isinf(ref).sum()))
        idx = np.argwhere(hn != gn)
        for j, i in idx[:3]:
            print("   ", j, i, "veg", s['ivgtyp'][j, i], "fveg", s['fvegxy'][j, i], "hip b/w", s['bgapxy'][j, i], s['wgapxy'][j, i], "gold", g["opt%02d/bgapxy" % n][j, i], ref[j, i])

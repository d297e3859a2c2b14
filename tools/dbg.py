import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import load_store, GOLDEN
from noahmp_amd.driver import Engine
from noahmp_amd.tables import load_tables
from noahmp_amd.state import ModelConfig
from oracle.portlib import PortLib
T, tb = load_tables("usgs")
eng = Engine(T, device=0)
port = PortLib(autobuild=False); port.set_tables(T)
g = np.load(os.path.join(GOLDEN, "golden_opts.npz"))
base = load_store(g, "init", 32, 4)
s = base.copy(); s.cfg = ModelConfig(iopt_rad=1)
so = s.copy()
st = eng.noahmplsm(s, 1, 2000, 180.0)
port.noahmplsm(so, 1, 2000, 180.0)
n = [i for i, x in enumerate(g["sweep"]) if "iopt_rad': 1" in x][0]
ref = g["opt%02d/bgapxy" % n]
print("nan counts hip/port/golden:", np.isnan(s['bgapxy']).sum(), np.isnan(so['bgapxy']).sum(), np.isnan(ref).sum())
idx = np.argwhere(np.isnan(s['bgapxy']) != np.isnan(ref))
for j, i in idx[:5]:
    print(j, i, "veg", s['ivgtyp'][j, i], "fveg", s['fvegxy'][j,i], "hip", s['bgapxy'][j, i], s['wgapxy'][j,i], "port", so['bgapxy'][j, i], so['wgapxy'][j,i], "gold", ref[j, i])

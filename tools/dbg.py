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
s = base.copy(); s.cfg = ModelConfig(idveg=2)
so = s.copy()
st = eng.noahmplsm(s, 1, 2000, 180.0)
port.noahmplsm(so, 1, 2000, 180.0)
d = np.abs(s['fastcpxy'] - so['fastcpxy'])
idx = np.argwhere(d > 1)
print(st.code, len(idx))
for j, i in idx[:6]:
    print(j, i, "veg", s['ivgtyp'][j, i], "hip", s['fastcpxy'][j, i], "port", so['fastcpxy'][j, i], "lai", s['xlaixy'][j,i], so['xlaixy'][j,i], "lfmass", s['lfmassxy'][j,i], so['lfmassxy'][j,i])
from noahmp_amd.abi import FIELD_INFO
j, i = idx[0]
for n in s.a:
    if FIELD_INFO[n][2] != "in" and s.a[n].ndim == 2:
        a, b = s.a[n][j, i], so.a[n][j, i]
        if a != b: print(n, a, b)
for lds in (0, 1):
    eng.set_option("lds", lds)
    s2 = base.copy(); s2.cfg = ModelConfig(idveg=2)
    eng.noahmplsm(s2, 1, 2000, 180.0)
    print("lds", lds, s2['fastcpxy'][j, i], s2['stblcpxy'][j, i])
print("pre  :", [(n, float(base.a[n][j, i])) for n in ("lfmassxy","rtmassxy","stmassxy","woodxy","stblcpxy","fastcpxy","xlaixy","xsaixy")])
print("hip  :", [(n, float(s.a[n][j, i])) for n in ("lfmassxy","rtmassxy","stmassxy","woodxy","stblcpxy","fastcpxy","xlaixy","xsaixy")])
print("port :", [(n, float(so.a[n][j, i])) for n in ("lfmassxy","rtmassxy","stmassxy","woodxy","stblcpxy","fastcpxy","xlaixy","xsaixy")])
d = s.copy(); d.a['fastcpxy'][...] = 777.0
dd = base.copy(); dd.cfg = ModelConfig(idveg=2); dd.a['fastcpxy'][...] = 777.0
ddv = dd.to_device("cuda:0")
eng.noahmplsm(ddv, 1, 2000, 180.0)
print("device-mode fastcp/stblcp at urban:", float(ddv.a['fastcpxy'][j, i]), float(ddv.a['stblcpxy'][j, i]))

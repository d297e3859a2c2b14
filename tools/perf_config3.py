"""Kernel time on BASELINE config 3 (snow / urban / glacier mix) at a given size.  usage: perf_config3.py [ni nj]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from noahmp_amd import synth  # noqa: E402
from noahmp_amd.driver import Engine  # noqa: E402
from noahmp_amd.tables import load_tables  # noqa: E402

ni, nj = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2048, 512)
T, tb = load_tables("usgs")
eng = Engine(T, device=0, lib_path=os.environ.get("NMP_LIB"))
for gf in (0.01, 0.0):
    s = synth.config3(tb, ni=ni, nj=nj, glacier_frac=gf)
    synth.first_step_fixups(s)
    synth.diurnal_forcing(s, 12, t_offset=s.t_offset)
    d = s.to_device("cuda:0")
    ms = []
    for it in range(1, 7):
        st = eng.noahmplsm(d, it, 2000, 180.0, check=False)
        ms.append(st.kernel_ms)
    best = min(ms[1:])
    n = st.n_land + st.n_glacier
    print("config3 %dx%d glacier_frac %.2f: kernel %.3f ms -> %.3e col-steps/s (land %d glacier %d code %d)"
          % (ni, nj, gf, best, n / best * 1e3, st.n_land, st.n_glacier, st.code))

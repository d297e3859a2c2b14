"""Run N device-resident config-2 steps at noon (profiling target).  usage: steps_run.py [nsteps] [ni nj]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from noahmp_amd import synth  # noqa: E402
from noahmp_amd.driver import Engine  # noqa: E402
from noahmp_amd.state import ModelConfig  # noqa: E402
from noahmp_amd.tables import load_tables  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ni, nj = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1024, 1024)
T, tb = load_tables("usgs")
eng = Engine(T, device=0, lib_path=os.environ.get("NMP_LIB"))
s = synth.config2(tb, ni=ni, nj=nj, cfg=ModelConfig(idveg=1))
synth.first_step_fixups(s)
synth.diurnal_forcing(s, 12, t_offset=s.t_offset)
d = s.to_device("cuda:0")
ms = []
for it in range(1, n + 1):
    ms.append(eng.noahmplsm(d, it, 2000, 180.0).kernel_ms)
print("kernel ms:", " ".join("%.3f" % m for m in ms))

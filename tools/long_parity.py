"""Long free-running bit-identity check on the GPU: a mixed tile (snow, urban, glacier, all ISNOW states) advanced for N
hourly steps by the HIP engine and by the oracle from the same start, compared bit for bit every step.
usage: long_parity.py [nsteps] [ni nj] [key=value ...]   (ModelConfig options)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
from noahmp_amd import synth  # noqa: E402
from noahmp_amd.driver import Engine  # noqa: E402
from noahmp_amd.state import ModelConfig  # noqa: E402
from noahmp_amd.tables import load_tables  # noqa: E402
from oracle.portlib import PortLib  # noqa: E402
from tools.compare import exact_check  # noqa: E402

pos = [a for a in sys.argv[1:] if "=" not in a]
kw = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
nsteps = int(pos[0]) if pos else 72
ni, nj = (int(pos[1]), int(pos[2])) if len(pos) > 2 else (256, 64)
T, tb = load_tables("usgs")
port = PortLib(autobuild=not os.path.exists(os.path.join(ROOT, "oracle", "_build", "libnoahmp_oracle.so")))
port.set_tables(T)
eng = Engine(T, device=0)
s = synth.mixed_small(tb, ni=ni, nj=nj, seed=31, cfg=ModelConfig(**kw))
synth.first_step_fixups(s)
o = s.copy()
d = s.to_device("cuda:0")
first_bad = None
for it in range(1, nsteps + 1):
    synth.diurnal_forcing(o, (it - 1) % 24, t_offset=s.t_offset)
    for k in ("coszin", "swdown", "glw", "t3d", "rainbl"):
        d.a[k].copy_(torch.from_numpy(o.a[k]))
    so = port.noahmplsm(o, it, 2000, 180.0 + it / 24.0)
    sd = eng.noahmplsm(d, it, 2000, 180.0 + it / 24.0, check=False)
    ok, lines = exact_check(o, d.to_host())
    if (not ok or so.code != sd.code) and first_bad is None:
        first_bad = it
        print("step", it, "codes", so.code, sd.code, "\n".join(lines[:6]))
print("config %s, %d columns x %d steps: %s; ISNOW states seen %s" % (
    kw, ni * nj, nsteps, "BIT-IDENTICAL" if first_bad is None else "first difference at step %d" % first_bad,
    sorted(set(np.unique(o.a["isnowxy"]).tolist()))))

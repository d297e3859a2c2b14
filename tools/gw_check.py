"""Ad-hoc GPU check of the MMF groundwater kernels against the oracle (C restatement).

  python tools/gw_check.py parity        deviation statistics on the test cases
  python tools/gw_check.py perf [ni nj]  kernel timing, device-resident
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from noahmp_amd import synth  # noqa: E402
from noahmp_amd.driver import Engine  # noqa: E402
from noahmp_amd.state import ModelConfig  # noqa: E402
from noahmp_amd.tables import load_tables  # noqa: E402
from oracle.portlib import PortLib  # noqa: E402
from test_groundwater import gw_store, GW_OUT, CASES  # noqa: E402


def parity():
    tabs = load_tables("usgs")
    port = PortLib(autobuild=not os.path.exists(os.path.join(ROOT, "oracle", "_build", "libnoahmp_oracle.so")))
    port.set_tables(tabs[0])
    eng = Engine(tabs[0], device=0)
    for case in CASES:
        s0 = gw_store(tabs, ni=256, nj=128, **case)
        a, b = s0.copy(), s0.copy()
        for call in range(3):
            b = a.copy()                       # restart the GPU from the oracle's state each call
            port.wtable_mmf(a)
            st = eng.wtable_mmf(b)
            line = []
            for n in GW_OUT:
                x, y = a.a[n].astype(np.float64), b.a[n].astype(np.float64)
                d = np.abs(x - y)
                rel = d / np.maximum(np.abs(x), 1e-6)
                line.append("%s: ne=%d maxabs=%.2e maxrel=%.2e" % (n, int((x != y).sum()), d.max(), rel.max()))
            print(case, "call", call, "n_land", st.n_land, "kernel_ms %.3f" % st.kernel_ms)
            print("   " + "\n   ".join(line))
            a.a["deeprechxy"][...] = s0.a["deeprechxy"]


def perf(ni=4608, nj=1536):
    tabs = load_tables("usgs")
    eng = Engine(tabs[0], device=0)
    cfg = ModelConfig(iopt_run=5)
    s = synth.config2(tabs[1], ni=ni, nj=nj, cfg=cfg)
    synth.groundwater_fields(s, tabs[1], stress=0.02)
    d = s.to_device()
    ts = []
    for it in range(12):
        st = eng.wtable_mmf(d)
        ts.append(st.kernel_ms)
    ms = float(np.median(ts[2:]))
    n = ni * nj
    print("gw %dx%d: kernels %.3f ms  -> %.3e cells/s, %.1f GB/s at 216 B/cell (n_land %d)"
          % (ni, nj, ms, n / ms * 1e3, n * 216 / ms / 1e6, st.n_land))


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "parity"
    if mode == "parity":
        parity()
    else:
        perf(*[int(v) for v in sys.argv[2:4]])

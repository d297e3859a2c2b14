"""Write tests/golden/golden_gw.npz: one WTABLE_mmf_noahmp call of the COMPILED REFERENCE (oracle/_ref,
built from /root/reference by oracle/Makefile) on a seeded 40x32 tile.  Dev container only.

    python tests/golden/make_golden_gw.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from noahmp_amd.tables import load_tables  # noqa: E402
from oracle.reflib import RefLib  # noqa: E402
from test_groundwater import gw_store, GW_OUT  # noqa: E402


def main():
    tables = load_tables("usgs")
    ref = RefLib("O0")
    ref.set_tables(tables[0])
    s = gw_store(tables, ni=40, nj=32, seed=11, stress=0.05, area=1.0e6)
    out = {"ni": 40, "nj": 32}
    from noahmp_amd.state import GW_ALIAS, GW_EXTRA
    for k in sorted(set(GW_ALIAS.values()) | set(GW_EXTRA)):
        out["in/" + k] = np.array(s.a[k], copy=True)
    ref.wtable_mmf(s)
    for k in GW_OUT:
        out["out/" + k] = s.a[k]
    p = os.path.join(ROOT, "tests", "golden", "golden_gw.npz")
    np.savez_compressed(p, **out)
    print("wrote", p, os.path.getsize(p), "bytes")


if __name__ == "__main__":
    main()

"""Write tests/golden/golden_init.npz: NOAHMP_INIT of the COMPILED REFERENCE (oracle/_ref) on a seeded 48x8 tile.
Dev container only.      python tests/golden/make_golden_init.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from noahmp_amd.tables import load_tables  # noqa: E402
from oracle.reflib import RefLib  # noqa: E402
from test_init import raw_store, INIT_FIELDS  # noqa: E402


def main():
    tables = load_tables("usgs")
    ref = RefLib("O0")
    ref.set_tables(tables[0])
    s = raw_store(tables, ni=48, nj=8, seed=12)
    out = {"ni": 48, "nj": 8}
    for k in INIT_FIELDS + ["isltyp", "ivgtyp", "xice", "tsk", "tmn"]:
        out["in/" + k] = np.array(s.a[k], copy=True)
    ref.noahmp_init(s)
    for k in INIT_FIELDS:
        out["out/" + k] = s.a[k]
    p = os.path.join(ROOT, "tests", "golden", "golden_init.npz")
    np.savez_compressed(p, **out)
    print("wrote", p, os.path.getsize(p), "bytes")


if __name__ == "__main__":
    main()

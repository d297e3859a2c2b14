"""Generate the committed parameter-table fixtures from the REFERENCE's own table readers.

Runs only in the dev container (needs /root/reference and oracle/_ref).  The reference's
read_mp_veg_parameters (lsm:274) and SOIL_VEG_GEN_PARM (drv:1528) parse run/*.TBL; the resulting
module arrays are dumped through ref_get_tables and stored as JSON (data, not source):
    noahmp_amd/data/tables_usgs.json , noahmp_amd/data/tables_modis.json
Usage:  python tests/golden/make_tables.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.reflib import RefLib  # noqa: E402


def dump(modis, isurban, out):
    ref = RefLib("O0")
    ref.read_tables(modis=modis)
    d = ref.get_tables_dict(isurban=isurban)
    js = {}
    for k, v in d.items():
        if isinstance(v, np.ndarray):
            # float32 -> shortest repr that round-trips through float32
            js[k] = v.tolist() if v.dtype.kind == "i" else [
                float(np.format_float_scientific(x, unique=True)) for x in v.ravel()]
            js[k] = {"shape": list(v.shape), "data": js[k]}
        else:
            js[k] = float(np.format_float_scientific(np.float32(v), unique=True)) if isinstance(v, float) else int(v)
    with open(out, "w") as f:
        json.dump(js, f, indent=0, separators=(",", ":"))
    print("wrote", out)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "usgs"
    # one dataset per process: the reference keeps the tables in module globals
    if which == "usgs":
        dump(False, 1, os.path.join(ROOT, "noahmp_amd", "data", "tables_usgs.json"))
        os.system("%s %s modis" % (sys.executable, os.path.abspath(__file__)))
    else:
        dump(True, 13, os.path.join(ROOT, "noahmp_amd", "data", "tables_modis.json"))

"""Parameter tables for the engine.

In the Fortran deployment the shim uploads the reference's own module arrays (filled by
NOAHMP_INIT -> read_mp_veg_parameters / SOIL_VEG_GEN_PARM).  For the Python host side the
same arrays are kept as JSON images under ``noahmp_amd/data`` (produced from the reference's
table readers by tests/golden/make_tables.py).
"""
import json
import os

import numpy as np

from .abi import tables_from_dict
from .abi_spec import TABLE_FIELDS

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def load_tables_dict(dataset="usgs"):
    with open(os.path.join(_DATA, "tables_%s.json" % dataset)) as f:
        js = json.load(f)
    out = {}
    for n, k, s, src in TABLE_FIELDS:
        v = js[n]
        if isinstance(v, dict):
            dt = np.int32 if k == "i" else np.float32
            out[n] = np.asarray(v["data"], dtype=dt).reshape(v["shape"])
        else:
            out[n] = v
    return out


def load_tables(dataset="usgs"):
    """-> (ctypes noahmp_tables, dict of numpy arrays)"""
    d = load_tables_dict(dataset)
    return tables_from_dict(d), d

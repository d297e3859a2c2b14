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


def bare_categories(tables_dict):
    """Vegetation categories that never carry a canopy: LAI + SAI zero in every month, the urban and the barren category (lsm:776-777:
    FVEG = 0 there).  Measured in round 6 (tools/veg_cost.py, profiles/r06_experiments.md): the land kernel's time per column is bimodal
    in the category -- 0.58-0.60 ns per column-step for every category with a canopy, 0.33-0.34 for exactly these (USGS: 1, 19, 23,
    25, 26, 27) -- so "the expensive categories first" (Engine.set_veg_order) means "these last"."""
    laim, saim = np.asarray(tables_dict["laim"]), np.asarray(tables_dict["saim"])
    n = int(tables_dict.get("lucats", laim.shape[0]))
    bare = {v + 1 for v in range(min(n, laim.shape[0])) if not (laim[v] + saim[v]).any()}
    bare |= {int(tables_dict["isurban"]), int(tables_dict["isbarren"])}
    return sorted(bare)


def canopy_first_order(tables_dict):
    """Order of the vegetation categories for the sort key (noahmp_hip_sort_set_veg_order): categories with a canopy in numeric order,
    then the bare ones -- workgroups start in key order, so the cheap waves form the tail of the launch (longest-first list scheduling)."""
    n = int(tables_dict.get("lucats", 27))
    bare = set(bare_categories(tables_dict))
    return [v for v in range(1, n + 1) if v not in bare] + [v for v in range(1, n + 1) if v in bare]

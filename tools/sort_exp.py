"""Experiment: what would sorting the columns by (class, vegetation type, snow-layer count) buy?  The synthetic tile is
physically re-ordered so that equal keys are adjacent (an upper bound: no gather cost), and the kernel is timed on both.
usage: sort_exp.py [config2|config3]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from noahmp_amd import synth  # noqa: E402
from noahmp_amd.driver import Engine  # noqa: E402
from noahmp_amd.state import ModelConfig  # noqa: E402
from noahmp_amd.tables import load_tables  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "config2"
T, tb = load_tables("usgs")
eng = Engine(T, device=0)
if which == "config2":
    s = synth.config2(tb, cfg=ModelConfig(idveg=1))
else:
    s = synth.config3(tb, ni=2048, nj=512)
synth.first_step_fixups(s)
synth.diurnal_forcing(s, 12, t_offset=s.t_offset)


def run(store, label):
    d = store.to_device("cuda:0")
    ms = [eng.noahmplsm(d, it, 2000, 180.0, check=False).kernel_ms for it in range(1, 7)]
    print("%-44s kernel %.3f ms" % (label, min(ms[1:])))


def permuted(store, key):
    p = np.argsort(key.ravel(), kind="stable")
    o = store.copy()
    for k, v in store.a.items():
        if k == "dzs":
            continue
        if v.ndim == 2:
            o.a[k][...] = v.ravel()[p].reshape(v.shape)
        else:
            nj, nk, ni = v.shape
            o.a[k][...] = v.transpose(1, 0, 2).reshape(nk, -1)[:, p].reshape(nk, nj, ni).transpose(1, 0, 2)
    return o


a = s.a
glacier = (a["ivgtyp"] == s.cfg.isice).astype(np.int64)
veg = a["ivgtyp"].astype(np.int64)
isn = (-a["isnowxy"]).astype(np.int64)
bare = ((a["ivgtyp"] == 19) | (a["ivgtyp"] == s.cfg.isurban)).astype(np.int64)
run(s, which + " as generated (random order)")
run(permuted(s, glacier), "sorted by class (land | glacier)")
run(permuted(s, glacier * 10 + bare), "sorted by class, vegetated | bare")
run(permuted(s, glacier * 1000 + veg), "sorted by class, vegetation type")
run(permuted(s, glacier * 1000 + isn * 100 + bare), "sorted by class, snow layers, veg | bare")
run(permuted(s, glacier * 10000 + veg * 10 + isn), "sorted by class, vegetation type, snow layers")
soil = a["isltyp"].astype(np.int64)
vfb = np.clip((a["vegfra"] / 10.0).astype(np.int64), 0, 9)
tsb = np.clip(((a["tsk"] - 270.0) / 3.0).astype(np.int64), 0, 15)
run(permuted(s, glacier * 100000 + veg * 64 + soil), "sorted by class, vegetation type, soil type")
run(permuted(s, glacier * 100000 + veg * 16 + vfb), "sorted by class, vegetation type, VEGFRA decile")
run(permuted(s, glacier * 100000 + veg * 16 + tsb), "sorted by class, vegetation type, TSK 3-K bin")
run(permuted(s, glacier * 100000 + tsb * 64 + veg), "sorted by class, TSK 3-K bin, vegetation type")
run(permuted(s, glacier * 1000000 + veg * 256 + tsb * 16 + vfb), "sorted by class, veg type, TSK bin, VEGFRA decile")
tsb1 = np.clip(((a["tsk"] - 270.0) / 1.0).astype(np.int64), 0, 47)
run(permuted(s, glacier * 1000000 + veg * 64 + tsb1), "sorted by class, vegetation type, TSK 1-K bin")
run(permuted(s, glacier * 1000000 + veg * 100000 + (a["tsk"] * 100).astype(np.int64) % 100000), "sorted by class, vegetation type, TSK")
tp = os.path.join(ROOT, "trips_tmp.npy")
if which == "config2" and os.path.exists(tp):      # oracle experiment: canopy-loop trip counts of THIS step (host emulation)
    trips = np.load(tp).reshape(veg.shape).astype(np.int64)
    run(permuted(s, trips), "sorted by this step's canopy trip count")
    run(permuted(s, veg * 32 + trips), "sorted by vegetation type, then trip count")
    run(permuted(s, trips * 32 + veg), "sorted by trip count, then vegetation type")

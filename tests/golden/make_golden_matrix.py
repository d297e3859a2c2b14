"""Write tests/golden/golden_vegsnow.npz and golden_snowedge.npz by RUNNING THE COMPILED REFERENCE (oracle/_ref, -O0 float32):
SURVEY 8c fixtures (2) every USGS category x {0..3 snow layers} and (4) snow depths one ulp around every layering threshold,
24-hour free runs with the config-1 forcing (rain / snowfall at hours 10-12), snapshots after steps 1, 6, 12, 18, 24.
The cold start itself is the reference's NOAHMP_INIT (snow layering by SNOW_INIT).  Dev container only.
    python tests/golden/make_golden_matrix.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from noahmp_amd import synth  # noqa: E402
from noahmp_amd.tables import load_tables  # noqa: E402
from oracle.reflib import RefLib  # noqa: E402

SNAP = (1, 6, 12, 18, 24)


def run(ref, maker, tb, path):
    captured = []
    orig = synth.noahmp_init

    def spy(store, tables, fndsnowh=True):          # cold start by the reference itself, not by the synthetic helper
        captured.append(store.copy())
        ref.noahmp_init(store, fndsnowh=fndsnowh)
    synth.noahmp_init = spy
    try:
        s = maker(tb)
    finally:
        synth.noahmp_init = orig
    synth.first_step_fixups(s)
    out = {"t_offset": s.t_offset, "ni": s.ni, "nj": s.nj}
    for k, v in captured[0].a.items():
        out["raw/" + k] = v.copy()
    for k, v in s.a.items():
        out["init/" + k] = v.copy()
    seen = set()
    for it in range(1, 25):
        synth.diurnal_forcing(s, (it - 1) % 24, t_offset=s.t_offset)
        ref.noahmplsm(s, it, 2000, 180.0)
        seen.update(np.unique(s["isnowxy"]).tolist())
        if it in SNAP:
            for k, v in s.a.items():
                out["step%02d/%s" % (it, k)] = v.copy()
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; ISNOW states", sorted(seen))


def main():
    T, tb = load_tables("usgs")
    ref = RefLib("O0")
    ref.set_tables(T)
    here = os.path.dirname(os.path.abspath(__file__))
    run(ref, synth.veg_snow_matrix, tb, os.path.join(here, "golden_vegsnow.npz"))
    run(ref, synth.snow_edges, tb, os.path.join(here, "golden_snowedge.npz"))


if __name__ == "__main__":
    main()

"""Generate golden input/output vectors by RUNNING THE REFERENCE (oracle/_ref, -O0 float32 build).

Dev-container only (needs /root/reference for the .TBL files and oracle/_ref/libnoahmp_ref.so).
Writes small compressed fixtures (data only) next to this script:
  golden_config1.npz    BASELINE config 1: single column, 24 hourly steps, every output each step
  golden_mixed.npz      64x4 mixed tile (snow 0..3 layers, urban, barren, 9 veg types): initial
                        state + full state after steps 1, 12, 24 of a free run
  golden_opts.npz       same tile, one step at hour 12, for each entry of the OPT_* sweep
Usage:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from noahmp_amd import synth  # noqa: E402
from noahmp_amd.state import ModelConfig  # noqa: E402
from noahmp_amd.tables import load_tables  # noqa: E402
from oracle.reflib import RefLib  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SWEEP = [dict(), dict(idveg=1), dict(idveg=2), dict(idveg=4), dict(idveg=5), dict(iopt_crs=2),
         dict(iopt_btr=2), dict(iopt_btr=3), dict(iopt_run=2), dict(iopt_run=3), dict(iopt_run=4),
         dict(iopt_run=5), dict(iopt_sfc=2), dict(iopt_frz=2), dict(iopt_inf=2), dict(iopt_rad=1),
         dict(iopt_rad=2), dict(iopt_alb=1), dict(iopt_snf=2), dict(iopt_snf=3), dict(iopt_tbot=1),
         dict(iopt_stc=2)]


def pack(prefix, store, out):
    for k, v in store.a.items():
        out["%s/%s" % (prefix, k)] = v.copy()


def main():
    ref = RefLib("O0")
    ref.read_tables()
    T, tb = load_tables("usgs")

    # ---- config 1
    s = synth.config1(tb)
    synth.first_step_fixups(s)
    out = {}
    pack("init", s, out)
    traj = {k: [] for k in s.a}
    for it in range(1, 25):
        synth.diurnal_forcing(s, (it - 1) % 24)
        ref.noahmplsm(s, it, 2000, 180.0)
        for k, v in s.a.items():
            traj[k].append(v.copy())
    for k, v in traj.items():
        out["traj/%s" % k] = np.stack(v)          # leading axis = step 1..24
    np.savez_compressed(os.path.join(HERE, "golden_config1.npz"), **out)

    # ---- mixed tile, free run
    s = synth.mixed_small(tb, ni=64, nj=4)
    synth.first_step_fixups(s)
    out = {"t_offset": s.t_offset}
    pack("init", s, out)
    for it in range(1, 25):
        synth.diurnal_forcing(s, (it - 1) % 24, t_offset=s.t_offset)
        ref.noahmplsm(s, it, 2000, 180.0)
        if it in (1, 12, 24):
            pack("step%02d" % it, s, out)
    np.savez_compressed(os.path.join(HERE, "golden_mixed.npz"), **out)

    # ---- option sweep, single step at noon from the same initial state
    out = {}
    base = synth.mixed_small(tb, ni=32, nj=4)
    synth.first_step_fixups(base)
    synth.diurnal_forcing(base, 12, t_offset=base.t_offset)
    pack("init", base, out)                       # one shared pre-step state for every option set
    from noahmp_amd.abi import FIELD_INFO
    for n, kw in enumerate(SWEEP):
        s = base.copy()
        s.cfg = ModelConfig(**kw)
        if kw.get("iopt_run") == 5:               # MMF keeps no aquifer store (drv:1107-1108)
            s["waxy"] = 0.0
            s["wtxy"] = 0.0
        ref.noahmplsm(s, 1, 2000, 180.0)
        for k, v in s.a.items():
            if FIELD_INFO[k][2] != "in":
                out["opt%02d/%s" % (n, k)] = v.copy()
    out["sweep"] = np.array([repr(k) for k in SWEEP])
    np.savez_compressed(os.path.join(HERE, "golden_opts.npz"), **out)
    for f in ("golden_config1.npz", "golden_mixed.npz", "golden_opts.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()

"""Write tests/golden/golden_scalars.npz by RUNNING THE COMPILED REFERENCE (oracle/_ref, -O0 float32): the sweep over the scalars
that are uniform over the grid at the noahmplsm boundary (drv:51-83) -- DT (NITER doubling lsm:7850-7857, FACT, the accumulators,
COMPACT, PHASECHANGE), DZS (ZSOIL drv:392-395: root fractions, the tridiagonal coefficients, ZWTEQ), YR (YEARLEN drv:381-390),
JULIAN on both hemispheres (lsm:1054-1071), DZ8W (the forcing height).  One cold start by the reference's NOAHMP_INIT per case,
then a free run with the config-1 forcing sampled at the case's time step; snapshots at the listed steps.  Dev container only.
    python tests/golden/make_golden_scalars.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from noahmp_amd import synth  # noqa: E402
from noahmp_amd.state import ModelConfig  # noqa: E402
from noahmp_amd.tables import load_tables  # noqa: E402
from oracle.reflib import RefLib  # noqa: E402

DZS2 = (0.05, 0.25, 0.7, 1.5)
# name -> cfg overrides, YR, JULIAN at the first step, steps, local hour of the first step, rain [mm/h] in hours 10-12, snapshots
CASES = {
    "dt600":        dict(cfg=dict(dt=600.0), yr=2000, jul=180.0, nsteps=30, h0=8.0, rain=6.0, snap=(1, 15, 30)),
    "dt900":        dict(cfg=dict(dt=900.0), yr=2000, jul=180.0, nsteps=24, h0=8.0, rain=6.0, snap=(1, 12, 24)),
    "dt1800_pour":  dict(cfg=dict(dt=1800.0), yr=2000, jul=180.0, nsteps=16, h0=7.0, rain=130.0, snap=(1, 8, 16)),
    "dt3600_dzs2":  dict(cfg=dict(dzs=DZS2), yr=2000, jul=180.0, nsteps=24, h0=0.0, rain=2.0, snap=(1, 12, 24)),
    "dzs2_pour":    dict(cfg=dict(dzs=DZS2), yr=2000, jul=180.0, nsteps=8, h0=8.0, rain=45.0, snap=(3, 8)),
    "dt900_dzs2":   dict(cfg=dict(dt=900.0, dzs=DZS2), yr=2001, jul=300.0, nsteps=12, h0=9.0, rain=30.0, snap=(6, 12)),
    "y2001_j015":   dict(cfg=dict(), yr=2001, jul=15.0, nsteps=6, h0=9.0, rain=2.0, snap=(1, 6)),
    "y2001_j100":   dict(cfg=dict(), yr=2001, jul=100.25, nsteps=6, h0=9.0, rain=2.0, snap=(1, 6)),
    "y2001_j260":   dict(cfg=dict(), yr=2001, jul=260.5, nsteps=6, h0=9.0, rain=2.0, snap=(1, 6)),
    "y2001_j350":   dict(cfg=dict(), yr=2001, jul=350.0, nsteps=6, h0=9.0, rain=2.0, snap=(1, 6)),
    "y2100_j060":   dict(cfg=dict(), yr=2100, jul=60.0, nsteps=6, h0=9.0, rain=2.0, snap=(1, 6)),
    "y2000_j001":   dict(cfg=dict(), yr=2000, jul=1.0, nsteps=4, h0=10.0, rain=2.0, snap=(1, 4)),
    "y2000_j366":   dict(cfg=dict(), yr=2000, jul=366.0, nsteps=4, h0=10.0, rain=2.0, snap=(1, 4)),
    "y2001_j365":   dict(cfg=dict(idveg=4), yr=2001, jul=365.0, nsteps=4, h0=10.0, rain=2.0, snap=(1, 4)),
    "zlvl10_dt1800": dict(cfg=dict(dt=1800.0, zlvl=10.0), yr=2001, jul=200.0, nsteps=12, h0=8.0, rain=4.0, snap=(1, 12)),
    "zlvl2_dveg1":  dict(cfg=dict(zlvl=4.0, idveg=1), yr=2001, jul=150.0, nsteps=12, h0=4.0, rain=4.0, snap=(1, 12)),
}


def case_config(name):
    return ModelConfig(**CASES[name]["cfg"])


def case_forcing(store, name, it):
    """Forcing of step `it` (1-based) of case `name`: the config-1 diurnal cycle at the case's local time; rain as a rate."""
    c = CASES[name]
    dt = store.cfg.dt
    hour = c["h0"] + (it - 1) * dt / 3600.0
    synth.diurnal_forcing(store, hour, t_offset=store.t_offset, rain_mm=c["rain"] * dt / 3600.0)
    return c["yr"], np.float32(c["jul"] + (it - 1) * dt / 86400.0)


def main():
    T, tb = load_tables("usgs")
    ref = RefLib("O0")
    ref.set_tables(T)
    out = {}
    for name, c in CASES.items():
        cfg = case_config(name)
        captured = []
        orig = synth.noahmp_init

        def spy(store, tables, fndsnowh=True):          # cold start by the reference itself
            captured.append(store.copy())
            ref.noahmp_init(store, fndsnowh=fndsnowh)
        synth.noahmp_init = spy
        try:
            s = synth.scalar_tile(tb, cfg)
        finally:
            synth.noahmp_init = orig
        case_forcing(s, name, 1)
        synth.first_step_fixups(s)
        out[name + "/t_offset"] = s.t_offset
        for k, v in captured[0].a.items():
            out["%s/raw/%s" % (name, k)] = v.copy()
        for k, v in s.a.items():
            out["%s/init/%s" % (name, k)] = v.copy()
        seen = set()
        for it in range(1, c["nsteps"] + 1):
            yr, jul = case_forcing(s, name, it)
            ref.noahmplsm(s, it, yr, jul)
            seen.update(np.unique(s["isnowxy"]).tolist())
            if it in c["snap"]:
                for k, v in s.a.items():
                    out["%s/step%02d/%s" % (name, it, k)] = v.copy()
        print("%-14s %2d steps, ISNOW states %s, SMOIS(1) %.3f..%.3f" % (name, c["nsteps"], sorted(seen),
              s.a["smois"][:, 0, :].min(), s.a["smois"][:, 0, :].max()))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_scalars.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

"""Ad-hoc GPU check: HIP engine vs the oracle (C restatement; optionally the compiled reference).

  python tools/gpu_check.py parity [nsteps]      per-step-restart + free-run parity on a mixed tile
  python tools/gpu_check.py sweep                single-step parity over the OPT_* sweep
  python tools/gpu_check.py perf [ni nj]         kernel timing for block/LDS variants
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from noahmp_amd import synth  # noqa: E402
from noahmp_amd.driver import Engine  # noqa: E402
from noahmp_amd.state import ModelConfig  # noqa: E402
from noahmp_amd.tables import load_tables  # noqa: E402
from oracle.portlib import PortLib  # noqa: E402
from tools.compare import compare, report  # noqa: E402

SWEEP = [dict(), dict(idveg=1), dict(idveg=2), dict(idveg=4), dict(idveg=5), dict(iopt_crs=2),
         dict(iopt_btr=2), dict(iopt_btr=3), dict(iopt_run=2), dict(iopt_run=3), dict(iopt_run=4),
         dict(iopt_run=5), dict(iopt_sfc=2), dict(iopt_frz=2), dict(iopt_inf=2), dict(iopt_rad=1),
         dict(iopt_rad=2), dict(iopt_alb=1), dict(iopt_snf=2), dict(iopt_snf=3), dict(iopt_tbot=1),
         dict(iopt_stc=2)]


def setup():
    T, tb = load_tables("usgs")
    port = PortLib(autobuild=not os.path.exists(os.path.join(ROOT, "oracle", "_build", "libnoahmp_oracle.so")))
    port.set_tables(T)
    eng = Engine(T, device=0, lib_path=os.environ.get("NMP_LIB"))
    return T, tb, port, eng


def parity(nsteps=24, cfgkw=None, glacier_frac=0.0, quiet=False):
    T, tb, port, eng = setup()
    cfg = ModelConfig(**(cfgkw or {}))
    s = synth.mixed_small(tb, ni=64, nj=8, cfg=cfg, glacier_frac=glacier_frac)
    synth.first_step_fixups(s)
    so = s.copy()
    free = s.copy()
    nbad_restart = 0
    for it in range(1, nsteps + 1):
        for st_ in (so, free):
            synth.diurnal_forcing(st_, (it - 1) % 24, t_offset=s.t_offset)
        sd = so.copy()                       # HIP restarts from the oracle's state each step
        port.noahmplsm(so, it, 2000, 180.0)
        st = eng.noahmplsm(sd, it, 2000, 180.0, check=False)
        if st.code:
            print("step", it, "HIP fatal", st.code, st.i, st.j)
        bad = compare(so, sd, steps=1)
        if bad:
            nbad_restart += 1
            if not quiet or nbad_restart < 3:
                print("step %d (restart) violations:\n%s" % (it, report(bad)))
        eng.noahmplsm(free, it, 2000, 180.0, check=False)
    badf = compare(so, free, steps=nsteps)
    print("cfg %s: restart-violating steps %d/%d; free-run violations after %d steps: %d"
          % (cfgkw, nbad_restart, nsteps, nsteps, len(badf)))
    if badf:
        print(report(badf))
    tot = ex = 0
    for n in so.a:
        if so.a[n].dtype.kind == "f":
            tot += so.a[n].size
            ex += int((so.a[n] == sd.a[n]).sum())
    print("  bit-exact fraction (last restart step): %.4f" % (ex / tot))
    return nbad_restart, len(badf)


def perf(ni=1024, nj=1024):
    T, tb, port, eng = setup()
    s = synth.config2(tb, ni=ni, nj=nj)
    synth.first_step_fixups(s)
    synth.diurnal_forcing(s, 12, t_offset=s.t_offset)
    hour = int(os.environ.get("NMP_HOUR", "12"))
    synth.diurnal_forcing(s, hour, t_offset=s.t_offset)
    ref = None
    for block, lds in ((64, 1), (64, 0), (128, 1), (256, 1), (256, 0)):
        eng.set_option("block", block)
        eng.set_option("lds", lds)
        d = s.to_device("cuda:0")
        ms = []
        for it in range(1, 6):
            st = eng.noahmplsm(d, it, 2000, 180.0, check=False)
            ms.append(st.kernel_ms)
        best = min(ms[1:])
        chk = float(d.a["tslb"].double().sum().item()) + float(d.a["hfx"].double().sum().item())
        ref = chk if ref is None else ref
        print("block %3d lds %d: kernel %.3f ms  -> %.3e col-steps/s (n_land %d, code %d)%s"
              % (block, lds, best, st.n_land / best * 1e3, st.n_land, st.code,
                 "" if chk == ref else "  CHECKSUM DIFFERS"))


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "parity"
    if mode == "parity":
        parity(int(sys.argv[2]) if len(sys.argv) > 2 else 24)
    elif mode == "sweep":
        for kw in SWEEP:
            parity(6, kw, quiet=True)
    elif mode == "perf":
        perf(*(int(x) for x in sys.argv[2:4]))

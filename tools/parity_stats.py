"""Deviation statistics (max / 99.9th percentile / fraction beyond TIGHT) per field for
  gpu  : HIP engine vs the oracle              (run on the GPU box)
  ref  : reference -O2 vs reference -O0        (run in the dev container; the reference's own floor)
one step from an identical state, repeated over 24 hourly steps on a 128x64 mixed tile."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from noahmp_amd import synth  # noqa: E402
from noahmp_amd.abi import FIELD_INFO  # noqa: E402
from noahmp_amd.tables import load_tables  # noqa: E402
from tools.compare import tolerance  # noqa: E402


def main(mode):
    T, tb = load_tables("usgs")
    if mode == "gpu":
        from noahmp_amd.driver import Engine
        from oracle.portlib import PortLib
        a = PortLib(autobuild=False); a.set_tables(T)
        b = Engine(T, device=0, lib_path=os.environ.get("NMP_LIB"))
        stepa = lambda s, it: a.noahmplsm(s, it, 2000, 180.0)            # noqa: E731
        stepb = lambda s, it: b.noahmplsm(s, it, 2000, 180.0, check=False)   # noqa: E731
    else:
        from oracle.reflib import RefLib
        a = RefLib("O0"); a.read_tables()
        b = RefLib("O2"); b.read_tables()
        stepa = lambda s, it: a.noahmplsm(s, it, 2000, 180.0)            # noqa: E731
        stepb = lambda s, it: b.noahmplsm(s, it, 2000, 180.0)            # noqa: E731
    s = synth.mixed_small(tb, ni=128, nj=64, seed=21, glacier_frac=0.0)
    synth.first_step_fixups(s)
    so = s.copy()
    names = [n for n in so.a if FIELD_INFO[n][2] != "in" and so.a[n].dtype.kind == "f"]
    acc = {n: [] for n in names}
    for it in range(1, 25):
        synth.diurnal_forcing(so, (it - 1) % 24, t_offset=s.t_offset)
        sd = so.copy()
        stepa(so, it)
        stepb(sd, it)
        for n in names:
            acc[n].append(np.abs(so.a[n].astype(np.float64) - sd.a[n]).ravel())
            acc[n + "/mag"] = acc.get(n + "/mag", []) + [np.maximum(np.abs(so.a[n]), np.abs(sd.a[n])).ravel().astype(np.float64)]
    out = {}
    for n in names:
        d = np.concatenate(acc[n]); m = np.concatenate(acc[n + "/mag"])
        rt, at = tolerance(n, 1)
        tight = float((d > at + rt * m).mean())
        out[n] = dict(max=float(d.max()), p999=float(np.quantile(d, 0.999)), frac_tight=tight, frac_ne=float((d > 0).mean()),
                      maxrel=float((d / np.maximum(m, 1e-30))[d > at].max()) if (d > at).any() else 0.0)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "parity_stats_%s%s.json" % (mode, os.environ.get("NMP_TAG", ""))), "w"), indent=1)
    for n in names:
        o = out[n]
        if o["max"] > 0:
            print("%-12s max %.3e  p99.9 %.3e  maxrel %.2e  frac>tight %.4f  frac!= %.5f"
                  % (n, o["max"], o["p999"], o["maxrel"], o["frac_tight"], o["frac_ne"]))
    print("fields with any differing entry: %d of %d; overall differing fraction %.6f"
          % (sum(1 for n in names if out[n]["max"] > 0), len(names), float(np.mean([out[n]["frac_ne"] for n in names]))))


if __name__ == "__main__":
    main(sys.argv[1])

"""Dev-container check: NOAHMP_INIT / SNOW_INIT on broad random inputs (every soil class, sea ice, glacier, FNDSNOWH both ways):
compiled reference vs C restatement vs the device source compiled for the host, bit for bit."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tools')]
from noahmp_amd import synth
from noahmp_amd.state import ModelConfig
from noahmp_amd.tables import load_tables
from oracle.portlib import PortLib
from oracle.reflib import RefLib
from host_emul.emullib import EmulLib
import fuzz_parity as fz
T, tb = load_tables("usgs")
port = PortLib(); port.set_tables(T)
ref = RefLib("O0"); ref.set_tables(T)
em = EmulLib(); em.set_tables(T)
F = np.float32
nbad = 0
for seed in range(1, 13):
    cfg = ModelConfig()
    s = fz.random_tile(tb, 8192, seed, cfg)
    r = np.random.default_rng(seed)
    s["isltyp"] = r.integers(1, 20, size=s["isltyp"].shape).astype(np.int32)      # every soil class incl. 14 water, 15 bedrock, 16 ice
    s["isltyp"][s["ivgtyp"] == cfg.isice] = 16
    s["xice"] = np.where(r.random(s["xice"].shape) < 0.03, 1.0, 0.0).astype(F)
    if os.environ.get("NMP_FUZZ_POISON"):         # NaN / Inf / huge / denormal in one input word of 3 % of the columns
        vals = (np.nan, np.inf, -np.inf, 0.0, -1.0e30, 1.0e30, 1.0e-42, -0.0)
        keys = ("tsk", "tmn", "canwat", "tslb", "smois")          # (a NaN snow depth is a fatal of the reference: SNOW_INIT stops the process)
        for c in np.flatnonzero(r.random(8192) < 0.03):
            k = keys[r.integers(len(keys))]
            if s.a[k].ndim == 3:
                s.a[k][0, r.integers(s.a[k].shape[1]), c] = F(vals[r.integers(len(vals))])
            else:
                s.a[k][0, c] = F(vals[r.integers(len(vals))])
    for fnd in (True, False):
        a, b, c = s.copy(), s.copy(), s.copy()
        ref.noahmp_init(a, fndsnowh=fnd); ref.set_tables(T)
        rc, _ = port.noahmp_init(b, fndsnowh=fnd)
        rc2, _ = em.noahmp_init(c, fndsnowh=fnd)
        for k in a.a:
            if k == "dzs": continue
            for nm, o in (("port", b), ("emul", c)):
                x, y = a.a[k], o.a[k]
                if not (np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y)):
                    nbad += 1; print("seed", seed, fnd, nm, k, np.argwhere(x != y)[:3].tolist())
    print("seed", seed, "done, isnow", np.bincount(-a["isnowxy"].ravel(), minlength=4).tolist())
print("init fuzz:", "bit-identical" if nbad == 0 else "%d field mismatches" % nbad)

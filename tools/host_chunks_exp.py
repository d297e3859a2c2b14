"""The unedited drop-in at config-3 size (noahmp_hip_step(NOAHMP_MEM_HOST), the Fortran shim's defaults: pin_host_arrays + trust_out_mirror) by the
number of row chunks of the upload | kernel | download pipeline (set_option host_chunks).  usage: host_chunks_exp.py [chunks ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")
import torch  # noqa: E402,F401
from noahmp_amd import synth  # noqa: E402
from noahmp_amd.driver import Engine  # noqa: E402
from noahmp_amd.state import ModelConfig  # noqa: E402
from noahmp_amd.tables import load_tables  # noqa: E402

T, tb = load_tables("usgs")
eng = Engine(T, device=0)
s = synth.config3_tile(tb, 4608, 1536, cfg=ModelConfig(idveg=3))
synth.first_step_fixups(s)
eng.set_option("pin_host_arrays", 1)
eng.set_option("trust_out_mirror", 1)
it = 1
for _ in range(3):
    eng.noahmplsm(s, it, 2000, 180.0); it += 1
for nc in [int(a) for a in sys.argv[1:]] or [3, 6, 12, 24]:
    eng.set_option("host_chunks", nc)
    eng.noahmplsm(s, it, 2000, 180.0); it += 1
    t0 = time.perf_counter()
    km = 0.0
    for _ in range(3):
        km += eng.noahmplsm(s, it, 2000, 180.0).kernel_ms; it += 1
    dt = (time.perf_counter() - t0) / 3
    print("host_chunks %2d: %.1f ms per call = %.3g column-steps/s, kernels %.2f ms" % (nc, dt * 1e3, s.ncol / dt, km / 3), flush=True)

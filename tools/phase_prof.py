"""Phase shares of the column kernel from the -DNMP_PHASE_TIMERS build (noahmp_amd/csrc/variants/lib_prof.so)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from noahmp_amd import synth  # noqa: E402
from noahmp_amd.driver import Engine  # noqa: E402
from noahmp_amd.state import ModelConfig  # noqa: E402
from noahmp_amd.tables import load_tables  # noqa: E402

NAMES = {0: "gather", 1: "redprm", 2: "atm+phenology+energy preamble", 3: "thermoprop", 4: "radiation",
         5: "btran/rsurf", 6: "vege_flux (rest: setup, loop2, t2m)", 7: "bare_flux", 8: "flux merge", 9: "tsnosoi",
         10: "phasechange", 11: "energy tail + early scatter", 12: "water preamble", 13: "water: groundwater + rest", 14: "carbon",
         22: "water: canwater", 23: "water: snowwater", 24: "water: soilwater",
         15: "final scatter", 16: "vege loop1: sfcdif", 17: "vege loop1: ragrb", 18: "vege loop1: esat",
         19: "vege loop1: stomata", 20: "vege loop1: flux solve", 21: "vege loop1: exit"}
if __name__ != "__main__":
    pass
else:
    hour = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    T, tb = load_tables("usgs")
    eng = Engine(T, device=0, lib_path=os.path.join(ROOT, "noahmp_amd", "csrc", "variants", "lib_prof.so"))
    s = synth.config2(tb, cfg=ModelConfig(idveg=1))
    synth.first_step_fixups(s)
    synth.diurnal_forcing(s, hour, t_offset=s.t_offset)
    d = s.to_device("cuda:0")
    out = (C.c_ulonglong * 32)()
    for it in range(1, 4):
        eng.noahmplsm(d, it, 2000, 180.0)
    eng.lib.noahmp_hip_debug_phase_ticks(out, 32)          # clear
    ms = 0.0
    for it in range(4, 10):
        ms += eng.noahmplsm(d, it, 2000, 180.0).kernel_ms
    eng.lib.noahmp_hip_debug_phase_ticks(out, 32)
    tot = float(sum(out))
    print("hour %d: kernel %.3f ms/step (profiling build)" % (hour, ms / 6))
    for p in sorted(NAMES, key=lambda p: -out[p]):
        print("  %-42s %5.1f %%" % (NAMES[p], 100.0 * out[p] / tot))


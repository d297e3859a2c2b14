"""Run-time compiled vs generic kernel on config 2 with an option set that has no ahead-of-time kernel.  usage: jit_speed.py [opt=val ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noahmp_amd import synth
from noahmp_amd.driver import Engine
from noahmp_amd.state import ModelConfig
from noahmp_amd.tables import load_tables
kw = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[1:]} or dict(idveg=1, iopt_btr=2)
T, tb = load_tables("usgs")
eng = Engine(T, device=0)
s = synth.config2(tb, cfg=ModelConfig(**kw))
synth.first_step_fixups(s)
synth.diurnal_forcing(s, 12, t_offset=s.t_offset)
for jit in (0, 1):
    eng.set_option("jit_option_kernels", jit)
    d = s.to_device("cuda:0")
    eng.sort_store(d)
    t0 = time.perf_counter()
    first = eng.noahmplsm(d, 1, 2000, 180.0).kernel_ms
    t_first = time.perf_counter() - t0
    ms = [eng.noahmplsm(d, it, 2000, 180.0).kernel_ms for it in range(2, 8)]
    print("options %s, jit %d: first call %.1f s, kernel %.3f ms (sorted layout, noon)  %s"
          % (kw, jit, t_first, min(ms), eng.lib.noahmp_hip_last_error().decode()[:80]))
eng.set_option("jit_option_kernels", 0)

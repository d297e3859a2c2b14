"""Experiment: a step as two launches (row blocks) on two streams, so that the tail of one block overlaps the body of
the other.  usage: overlap_exp.py [frac_of_rows_in_first_block]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from noahmp_amd import synth
from noahmp_amd.driver import Engine
from noahmp_amd.state import ModelConfig
from noahmp_amd.tables import load_tables

T, tb = load_tables("usgs")
eng = Engine(T, device=0)
s = synth.config2(tb, cfg=ModelConfig(idveg=1))
synth.first_step_fixups(s)
fkeys = ("coszin", "swdown", "glw", "t3d", "rainbl")
forcing = []
for h in range(24):
    synth.diurnal_forcing(s, h, t_offset=s.t_offset)
    forcing.append({k: torch.from_numpy(s.a[k].copy()).cuda() for k in fkeys})
fptr = [{k: f[k].data_ptr() for k in fkeys} for f in forcing]
for frac in [float(x) for x in sys.argv[1:]] or [0.0, 0.5, 0.33, 0.25]:
    d = s.to_device("cuda:0")
    nj = s.nj
    cut = int(nj * frac)
    blocks = [(1, nj)] if cut == 0 else [(1, cut), (cut + 1, nj)]
    streams = [torch.cuda.Stream() for _ in blocks]
    args = []
    for (j0, j1) in blocks:
        a = d.step_args(1, 2000, 180.0)
        a.jts, a.jte = j0, j1
        args.append(a)

    def step(it):
        for a, st in zip(args, streams):
            for k_, p_ in fptr[(it + 5) % 24].items():
                setattr(a, k_, p_)
            a.itimestep = it
            eng.noahmplsm_async(a, stream=st.cuda_stream)
    for it in range(1, 7):
        step(it)
    eng.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 96
    for it in range(7, 7 + n):
        step(it)
    st, _ = eng.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    chk = float(d.a["tslb"].double().sum().item())
    print("first block %.2f of rows: %.4f ms/step -> %.3e col-steps/s  (checksum %.6f)" % (frac, dt * 1e3, s.ncol / dt, chk))

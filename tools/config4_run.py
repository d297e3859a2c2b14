"""SURVEY 8d config 4 on one GPU: the config-3 grid (4608 x 1536 = 7.08 M cells, 30 % snow, urban, glacier) with OPT_RUN = 5 --
every step the column kernel, every STEPWTD steps WTABLE_mmf_noahmp (LATERALFLOW stencil + UPDATEWTD), device-resident, tile
order (the stencil needs the (i,j) neighbourhood, so this configuration is not column-sorted).  With WORLD_SIZE > 1 (torchrun) the grid
is split by the mpp rule and ZWTXY's ring is exchanged before every groundwater call (parallel.exchange_halo); that branch has
not been run this round (one-GPU boxes only; the exchange itself is covered by the gloo tests).
usage: config4_run.py [ni nj [nsteps]]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from noahmp_amd import synth  # noqa: E402
from noahmp_amd.driver import Engine  # noqa: E402
from noahmp_amd.parallel import Comm  # noqa: E402
from noahmp_amd.state import ModelConfig  # noqa: E402
from noahmp_amd.tables import load_tables  # noqa: E402

a = [int(x) for x in sys.argv[1:]]
gni, gnj = (a[0], a[1]) if len(a) >= 2 else (4608, 1536)
nsteps = a[2] if len(a) > 2 else 24
comm = Comm()
T, tb = load_tables("usgs")
eng = Engine(T, device=comm.local_rank)
cfg = ModelConfig(iopt_run=5)
stepwtd = max(int(cfg.wtddt * 60.0 / cfg.dt + 0.5), 1)               # hdrv:1227 NINT (round half away from zero)
if comm.world == 1:
    s = synth.config3(tb, ni=gni, nj=gnj, cfg=cfg)
    geom = None
else:                                                                 # every rank builds its own tile + ring (weak-scaling style)
    geom = comm.my_geometry(gni * 1, gnj * 1)
    s = synth.config3(tb, ni=geom["ime"] - geom["ims"] + 1, nj=geom["jme"] - geom["jms"] + 1, cfg=cfg, seed=3 + comm.rank)
    s.set_index(**{k: geom[k] for k in ("ids", "ide", "jds", "jde", "ims", "ime", "jms", "jme", "its", "ite", "jts", "jte")})
synth.groundwater_fields(s, tb)
synth.first_step_fixups(s)
d = s.to_device("cuda:%d" % comm.local_rank)
forc = {}
for h in range(24):
    synth.diurnal_forcing(s, h, t_offset=s.t_offset)
    forc[h] = {k: torch.from_numpy(s.a[k].copy()).to(d.device) for k in ("coszin", "swdown", "glw", "t3d", "rainbl")}
if geom is not None:
    comm.exchange_halo([d.a["fdepth"], d.a["topo"]], geom)                            # static planes once
    comm.exchange_halo([d.a["isltyp"]], geom)


def step(it):
    eng.stream_sync()                                      # the previous step's kernel may still read the planes overwritten below
    for k, v in forc[(it - 1) % 24].items():
        d.a[k].copy_(v)
    torch.cuda.current_stream().synchronize()
    eng.noahmplsm_async(d.step_args(it, 2000, 180.0))
    if it % stepwtd == 0:
        st, _ = eng.sync()
        if geom is not None:
            comm.exchange_halo([d.a["zwtxy"]], geom)
        g = eng.wtable_mmf(d)
        return st.kernel_ms, g.kernel_ms, st.n_land + st.n_glacier
    return 0.0, 0.0, 0


for it in range(1, 4):
    step(it)
eng.sync()
comm.barrier()
torch.cuda.synchronize()
t0 = time.perf_counter()
kc = kg = 0.0
ncol = 0
for it in range(4, 4 + nsteps):
    c, g, n = step(it)
    kc += c
    kg += g
    ncol = n or ncol
st, _ = eng.sync()
kc += st.kernel_ms
torch.cuda.synchronize()
comm.barrier()
wall = comm.reduce_max(time.perf_counter() - t0)
tot = comm.reduce_sum(ncol)
if comm.rank == 0:
    print(json.dumps(dict(config="4 (config-3 grid, OPT_RUN=5)", grid=[gni, gnj], n_gpus=comm.world, steps=nsteps, stepwtd=stepwtd,
                          columns=int(tot), ms_per_step=round(wall / nsteps * 1e3, 3),
                          column_kernel_ms=round(kc / nsteps, 3), wtable_ms_per_call=round(kg / max(nsteps // stepwtd, 1), 3),
                          column_steps_per_s=tot * nsteps / wall)))
comm.close()

"""Land-kernel cost per vegetation type (round 6: input of the longest-first column order, noahmp_hip_sort_set_veg_order).

For every vegetation category of the table set (USGS 1..27 / MODIS 1..20, without water and land ice) a config-3-style tile in which
EVERY land column has that category (same snow mix, soil types, temperatures as the bench workload), sorted like the bench sorts
(snow-layer count, TSK bin), advanced through one diurnal cycle on the device; the land kernel's own time per step by forcing hour.

usage: veg_cost.py [usgs|modis] [ni nj] [dveg]      ->  one JSON line (per type: 24-h mean, day mean, night mean in ns per column-step)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")
import bench  # noqa: E402
from noahmp_amd import synth  # noqa: E402
from noahmp_amd.driver import Engine  # noqa: E402
from noahmp_amd.state import ModelConfig  # noqa: E402
from noahmp_amd.tables import load_tables  # noqa: E402


def main():
    dataset = sys.argv[1] if len(sys.argv) > 1 else "usgs"
    ni, nj = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1024, 512)
    dveg = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    import torch
    T, tb = load_tables(dataset)
    eng = Engine(T, device=0, lib_path=os.environ.get("NMP_LIB"))
    cfg = ModelConfig(idveg=dveg, isurban=tb["isurban"], isice=tb["issnow"], iswater=tb["iswater"])
    ntypes = 27 if dataset == "usgs" else 20
    out = {}
    stream = torch.cuda.Stream()
    for v in range(1, ntypes + 1):
        if v in (cfg.iswater, cfg.isice):
            continue
        saved = synth.CONUS_VEG
        synth.CONUS_VEG = np.array([v], dtype=np.int32)
        try:
            s = synth.config3(tb, ni=ni, nj=nj, seed=11, cfg=cfg, urban_frac=0.0, glacier_frac=0.0)
        finally:
            synth.CONUS_VEG = saved
        synth.first_step_fixups(s)
        forcing = []
        for h in range(24):
            synth.diurnal_forcing(s, h, t_offset=s.t_offset)
            forcing.append({k: torch.from_numpy(s.a[k].copy()).cuda() for k in bench.FKEYS})
        d = s.to_device("cuda:0")
        perm = eng.sort_store(d, tsk_bin=1.0)
        work = {k: d.a[k] for k in bench.FKEYS}
        lvl1 = tuple(i for i, k in enumerate(bench.FKEYS) if work[k].dim() == 3)
        scat = eng.scatter([work[k] for k in bench.FKEYS], [forcing[0][k] for k in bench.FKEYS], perm, ni, nj, first_level_only=lvl1)
        sarg = d.step_args(1, 2000, 180.0)
        hours = {}
        for it in range(1, 31):                                 # 6 warm-up steps (06:00 ..), then 24
            h = bench.forcing_hour(it)
            scat.set_sources([forcing[h][k] for k in bench.FKEYS])
            scat(stream.cuda_stream)
            sarg.itimestep = it
            eng.noahmplsm_async(sarg, stream.cuda_stream)
            if it == 6 or it == 30:
                st, _ = eng.sync(check=False)
                if st.code:
                    print("type %d: code %d" % (v, st.code), file=sys.stderr)
                if it == 30:
                    for k, ms in enumerate(eng.sync_step_timing()):
                        hours[bench.forcing_hour(7 + k)] = ms
        ncol = d.class_ranges[0]
        per = {h: ms * 1e6 / ncol for h, ms in hours.items()}  # ns per column-step
        day = [per[h] for h in per if 6 < h < 18]
        night = [per[h] for h in per if not 6 < h < 18]
        out[v] = {"mean": sum(per.values()) / len(per), "day": sum(day) / len(day), "night": sum(night) / len(night), "columns": ncol}
        print("type %2d  mean %.3f  day %.3f  night %.3f ns/column-step  (%d land columns)" % (v, out[v]["mean"], out[v]["day"], out[v]["night"], ncol),
              file=sys.stderr)
        del d, forcing, scat, work
        torch.cuda.empty_cache()
    order = sorted(out, key=lambda v: -out[v]["mean"])
    print(json.dumps({"dataset": dataset, "grid": [ni, nj], "dveg": dveg, "ns_per_column_step": out, "longest_first": order}))


if __name__ == "__main__":
    main()

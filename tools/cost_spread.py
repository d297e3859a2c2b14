"""Within-wavefront spread of the two data-dependent trip counts of the land kernel (set_option record_cost: iterations of VEGE_FLUX's
canopy loop, STOMATA bisection steps) under the CURRENT column order of a bench run, and under the order a sort by (current key, cost
bucket) would give -- the measurement behind bench.py --cost-key (profiles/r05_experiments.md).  Called by bench.py when
NMP_COST_SPREAD=1 (after the timed region, on the box); prints one JSON line per sampled step to stderr."""
import json
import sys

import numpy as np


def wave_stats(iters, bis, tag):
    n = iters.size // 64 * 64
    it, bs = iters[:n].reshape(-1, 64).astype(np.float64), bis[:n].reshape(-1, 64).astype(np.float64)
    cost = 8.0 * it + bs
    out = {"order": tag, "waves": int(it.shape[0]),
           "canopy_iterations_mean": float(it.mean()), "canopy_iterations_wave_max_mean": float(it.max(axis=1).mean()),
           "bisections_mean": float(bs.mean()), "bisections_wave_max_mean": float(bs.max(axis=1).mean()),
           # a wave runs each loop as long as its slowest lane: useful lane-iterations / issued lane-iterations
           "canopy_loop_lane_use": float(it.sum() / max(it.max(axis=1).sum() * 64, 1)),
           "bisection_lane_use": float(bs.sum() / max(bs.max(axis=1).sum() * 64, 1)),
           "cost_lane_use": float(cost.sum() / max(cost.max(axis=1).sum() * 64, 1)),
           "waves_without_canopy": float((it.max(axis=1) == 0).mean())}
    return out


_earlier = []          # (hour, order by that hour's cost) of the previous samples of this run: does a record predict later steps?


def report(eng, run, hour, nland):
    n = run.d.ncol
    cost = np.zeros(2 * n, dtype=np.uint8)
    got = eng.lib.noahmp_hip_fetch_cost(cost.ctypes.data, n, None)
    if got != n:
        print("COSTSPREAD nothing recorded", file=sys.stderr)
        return None
    iters, bis = cost[0::2][:nland].astype(np.int64), cost[1::2][:nland].astype(np.int64)
    keys = run.d.sort_keys.cpu().numpy().view(np.uint32).astype(np.int64)[:nland]
    cur = wave_stats(iters, bis, "current")
    c = 8 * iters + bis
    bucket = np.where(c == 0, 0, np.minimum(1 + c // 14, 15))
    group = keys >> 12                                     # class | vegetation type, snow layers | band
    order = np.lexsort((keys & 0xFF, bucket, group))       # what NOAHMP_SORT_COST would do with THIS step's record
    srt = wave_stats(iters[order], bis[order], "group, cost bucket, tsk bin")
    order2 = np.lexsort((c, group))
    full = wave_stats(iters[order2], bis[order2], "group, exact cost")
    pred = [wave_stats(iters[o], bis[o], "group, cost bucket of hour %d, tsk bin" % h) for h, o in _earlier]
    _earlier.append((hour, order))
    res = {"hour": hour, "land_columns": int(nland), "groups": int(np.unique(group).size), "orders": [cur, srt, full] + pred}
    print("COSTSPREAD " + json.dumps(res), file=sys.stderr)
    return order

"""SURVEY 8d config 5 on the GPU, end to end on device-resident arrays: cold start (noahmp_hip_init), then per hourly
step forcing interpolation between 3-hourly records (noahmp_hip_forcing_interpolate), forcing preparation with the
solar zenith angle (noahmp_hip_forcing_prep) and the column step (noahmp_hip_step_async) -- nothing returns to the host
between the cold start and the end of the run.

Parity at full size: columns are independent, so a random SAMPLE of columns is advanced by the oracle (C restatement)
through the same chain from the same raw state and the same forcing records, and compared bit for bit with the same
columns of the full-size device run at a few checkpoints and at the end.

usage: config5_run.py [ni nj [nsteps [nsample]]] [option=value ...]     default 3600 1800 720 4096, ModelConfig options
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from noahmp_amd import synth5  # noqa: E402
from noahmp_amd.driver import Engine, NoahMPFatal  # noqa: E402
from noahmp_amd.state import ColumnStore  # noqa: E402
from noahmp_amd.tables import load_tables  # noqa: E402
from oracle.portlib import PortLib  # noqa: E402
from tools.compare import exact_check  # noqa: E402


def extract(store, flat, n=None):
    """Columns `flat` (linear j*ni+i indices) of a host or device store as an n x 1 host ColumnStore."""
    flat = np.asarray(flat, dtype=np.int64)
    out = ColumnStore(len(flat), 1, store.cfg)
    for k, v in store.a.items():
        if k == "dzs" or k not in out.a:
            continue
        if not isinstance(v, np.ndarray):
            idx = torch.from_numpy(flat).to(v.device)
            if v.dim() == 3:
                col = v.permute(0, 2, 1).reshape(-1, v.shape[1])[idx].cpu().numpy()
            else:
                col = v.reshape(-1)[idx].cpu().numpy()
        elif v.ndim == 3:
            col = v.transpose(0, 2, 1).reshape(-1, v.shape[1])[flat]
        else:
            col = v.reshape(-1)[flat]
        out.a[k][...] = col.T[None] if col.ndim == 2 else col[None]
    return out


def fatal_column_vs_oracle(port, raw, lon, recs_tile, tcol, fail_step, e, cfgkw, ni, nj, verbose):
    """The oracle on the one tile column the device run stopped at, from the same raw state through the same chain up to the failing step."""
    osamp = extract(raw, [tcol])
    lon_s = lon.reshape(-1)[[tcol]][None].copy()
    nrec = (fail_step + synth5.RECORD_HOURS - 1) // synth5.RECORD_HOURS + 1
    rec_s = []
    for i in range(nrec):
        rec_s.append({k: (v.reshape(-1)[tcol:tcol + 1].cpu().numpy()[None].copy() if v is not None else None) for k, v in recs_tile.at(i).items()})
    port.noahmp_init(osamp, fndsnowh=True)
    rain_s = np.zeros((1, 1), np.float32)
    codes = []
    for n in range(fail_step):
        ri, k = divmod(n, synth5.RECORD_HOURS)
        port.forcing_interpolate(osamp, rec_s[ri], rec_s[ri + 1] if k else None, 3600 * k, 3600 * synth5.RECORD_HOURS, rain_s)
        iday, ihour = synth5.step_time(n)
        jul = port.forcing_prep(osamp, lon_s, rain_s, iday, ihour, first_step=(n == 0))
        codes.append(int(port.noahmplsm(osamp, n + 1, 2000, jul).code))
        if codes[-1]:
            break
    res = dict(options=cfgkw or {}, grid=[ni, nj], device_fatal=dict(code=int(e.code), step=fail_step, sorted_i=int(e.i), sorted_j=int(e.j), tile_column=tcol,
                                                                  vegtyp=int(raw.a["ivgtyp"].reshape(-1)[tcol]), soiltyp=int(raw.a["isltyp"].reshape(-1)[tcol])),
               oracle_codes_of_that_column=[c for c in codes if c] or [0], oracle_stopped_at_step=(len(codes) if codes and codes[-1] else None),
               oracle_same_fatal=bool(codes and codes[-1] == int(e.code) and len(codes) == fail_step),
               sample_bit_identical=None, checkpoints=[], device_status_max=int(e.code))
    if verbose:
        print(json.dumps(res))
    return res


def run(ni=3600, nj=1800, nsteps=720, nsample=4096, seed=5, verbose=True, checkpoints=(1, 24, 240), restart_path=None, cfgkw=None,
        resort_every=24, resort_frac=0.10, lon_band=15.0):
    T, tb = load_tables("usgs")
    port = PortLib(autobuild=not os.path.exists(os.path.join(ROOT, "oracle", "_build", "libnoahmp_oracle.so")))
    port.set_tables(T)
    eng = Engine(T, device=0)
    from noahmp_amd.state import ModelConfig
    raw, lon, static = synth5.config5_raw(ni, nj, seed=seed, cfg=ModelConfig(**cfgkw) if cfgkw else None)
    dev = torch.device("cuda", 0)
    d = raw.to_device("cuda:0")
    eng.noahmp_init(d, fndsnowh=True)                                  # cold start on the device
    skw = {}
    if lon_band:                                                       # longitude band (hours of local solar time) as a sub-key of the order
        d.a["lonband"] = ((torch.from_numpy(lon).to(dev) + 180.0) / float(lon_band)).floor().clamp_(0, 31).to(torch.int32).contiguous()
        skw["band"] = "lonband"
    perm = eng.sort_store(d, **skw)                                    # (class, vegetation type, snow layers, [band,] TSK bin) order
    lon_t = torch.from_numpy(lon).to(dev).reshape(-1)
    static_t = {k: torch.from_numpy(v).to(dev).reshape(-1) for k, v in static.items()}

    def sorted_side(perm):
        """what lives outside the store but in its column order: longitude and the static fields of the forcing records"""
        pl = perm.long()
        lon_d = lon_t[pl].reshape(nj, ni).contiguous()
        return lon_d, synth5.Records(d.a["xlatin"], lon_d, {k: v[pl].reshape(nj, ni).contiguous() for k, v in static_t.items()})

    lon_d, recs = sorted_side(perm)
    rain_d = torch.zeros((nj, ni), dtype=torch.float32, device=dev)
    n_land = int(((d.a["xland"] < 1.5) & (d.a["xice"] < raw.cfg.xice_thres)).sum().item())        # soil + glacier columns

    # ---- the sample: sorted positions, their tile columns, their raw state and forcing records
    r = np.random.Generator(np.random.Philox(seed + 100))
    nsample = min(nsample, ni * nj)
    pos = np.sort(r.choice(ni * nj, size=nsample, replace=False))
    tile_cols = perm.cpu().numpy().astype(np.int64)[pos]
    tile_cols_t = torch.from_numpy(tile_cols).to(dev)
    osamp = extract(raw, tile_cols)
    lon_s = lon.reshape(-1)[tile_cols][None].copy()
    pos_t = torch.from_numpy(pos).to(dev)
    nrec = (nsteps + synth5.RECORD_HOURS - 1) // synth5.RECORD_HOURS + 1
    rec_s = []
    for i in range(nrec):
        rec_s.append({k: (v.reshape(-1)[pos_t].cpu().numpy()[None].copy() if v is not None else None)
                      for k, v in recs.at(i).items()})
    checkpoints = sorted(set([c for c in checkpoints if c < nsteps] + [nsteps]))

    # ---- device run: record evaluation (torch) and the engine's kernels share one stream, so they are ordered
    snaps = {}
    resorts, stale_seen = 0, []
    dev_codes, n_done = [], 0
    ts = torch.cuda.Stream(device=dev)
    sp = ts.cuda_stream
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    rec_a = rec_b = None
    last_sync = 0
    recs_tile = synth5.Records(torch.from_numpy(np.ascontiguousarray(raw.a["xlatin"])).to(dev).reshape(-1), lon_t, static_t)   # tile order (fatal_column_vs_oracle)
    with torch.cuda.stream(ts):
        for n in range(nsteps):
            ri, k = divmod(n, synth5.RECORD_HOURS)
            if k == 0:
                rec_a = rec_b if rec_b is not None else recs.at(ri)
                rec_b = recs.at(ri + 1)
            eng.forcing_interpolate(d, rec_a, rec_b if k else None, 3600 * k, 3600 * synth5.RECORD_HOURS, rain_d, stream=sp, wait=False)
            iday, ihour = synth5.step_time(n)
            jul = eng.forcing_prep(d, lon_d, rain_d, iday, ihour, first_step=(n == 0), stream=sp, wait=False)
            eng.noahmplsm_async(d.step_args(n + 1, 2000, jul), stream=sp)
            resort_due = bool(resort_every) and (n + 1) % resort_every == 0 and n + 1 < nsteps
            if n + 1 in checkpoints or (n + 1) % 24 == 0 or resort_due:
                try:
                    st, _ = eng.sync()
                except NoahMPFatal as e:
                    # A column stopped the model (the reference would STOP in wrf_error_fatal): is it the model or the engine?  The oracle takes
                    # the very column through the same chain; the same code at the same step means the synthetic state ran into the model's own check.
                    ts.synchronize()
                    fail_step = last_sync + int(str(e).split("step +")[1].split(":")[0]) + 1
                    tcol = int(perm[(e.j - 1) * ni + (e.i - 1)].item())
                    return fatal_column_vs_oracle(port, raw, lon, recs_tile, tcol, fail_step, e, cfgkw, ni, nj, verbose)
                last_sync = n + 1
                kernel_ms += st.kernel_ms
                dev_codes.append(int(st.code))                           # 0 = every column passed the model's own balance checks
                n_done += sum(eng.sync_counts())
                if n + 1 in checkpoints:
                    t_hold = time.perf_counter()
                    inv = torch.empty(ni * nj, dtype=torch.int64, device=dev)
                    inv[perm.long()] = torch.arange(ni * nj, device=dev)
                    snaps[n + 1] = extract(d, inv[tile_cols_t].cpu().numpy())         # the sample's current sorted positions
                    ts.synchronize()
                    t0 += time.perf_counter() - t_hold                   # snapshots are not part of the run
                if resort_due:                                         # inside the timed run: staleness, re-sort
                    ts.synchronize()
                    stale = eng.sort_staleness(d)
                    stale_seen.append(stale)
                    if stale > resort_frac * ni * nj:
                        perm = eng.sort_store(d, **skw)
                        lon_d, recs = sorted_side(perm)
                        ri2, k2 = divmod(n + 1, synth5.RECORD_HOURS)       # records are in the store's column order: evaluate them again
                        rec_a, rec_b = (recs.at(ri2), recs.at(ri2 + 1)) if k2 else (None, None)
                        resorts += 1
        ts.synchronize()
    wall = time.perf_counter() - t0
    restart_s = None
    if restart_path:                                                   # checkpoint of the sorted device state (8f-4)
        from noahmp_amd import restart
        t2 = time.perf_counter()
        names = [f for _, f, _ in restart.RESTART_VARS]
        vals = restart.fetch(eng, d, names, perm=perm, mask=[f for _, f, lay in restart.RESTART_VARS if lay])
        fetch_s = time.perf_counter() - t2
        restart.write_restart(restart_path, d, "2000-07-20_00:00:00", "2000-06-20_00:00:00", engine=eng, perm=perm)
        restart_s = dict(fetch_s=round(fetch_s, 3), fetch_and_write_s=round(time.perf_counter() - t2 - fetch_s, 3),
                         bytes=os.path.getsize(restart_path), host_bytes=sum(v.nbytes for v in vals.values()))

    # ---- oracle on the sample
    t1 = time.perf_counter()
    port.noahmp_init(osamp, fndsnowh=True)
    rain_s = np.zeros((1, nsample), np.float32)
    report, ok_all = [], True
    for n in range(nsteps):
        ri, k = divmod(n, synth5.RECORD_HOURS)
        port.forcing_interpolate(osamp, rec_s[ri], rec_s[ri + 1] if k else None, 3600 * k, 3600 * synth5.RECORD_HOURS, rain_s)
        iday, ihour = synth5.step_time(n)
        jul = port.forcing_prep(osamp, lon_s, rain_s, iday, ihour, first_step=(n == 0))
        so = port.noahmplsm(osamp, n + 1, 2000, jul)
        assert so.code == 0, "oracle: fatal code %d" % so.code
        if n + 1 in snaps:
            skip = ("t2mvxy", "t2mbxy", "q2mvxy", "q2mbxy", "chv2xy", "chb2xy") if (cfgkw or {}).get("iopt_sfc") == 2 else ()
            ok, lines = exact_check(osamp, snaps[n + 1], skip=skip)      # OPT_SFC=2: FH2 undefined in the reference
            ok_all &= ok
            report.append((n + 1, ok, lines[:4]))
    t_or = time.perf_counter() - t1
    isn = sorted(set(np.unique(osamp.a["isnowxy"]).tolist()))
    res = dict(options=cfgkw or {}, grid=[ni, nj], steps=nsteps, land_columns=n_land, wall_s=round(wall, 3), ms_per_step=round(wall / nsteps * 1e3, 3),
               column_steps_per_s=n_land * nsteps / wall, column_kernel_ms_per_step=round(kernel_ms / nsteps, 3),
               resorts=resorts, stale_columns_seen=stale_seen[-3:], device_status_max=max(dev_codes) if dev_codes else None,
               cells_stepped=int(n_done),
               sample=nsample, sample_bit_identical=bool(ok_all), checkpoints=[c for c, _, _ in report],
               isnow_states_in_sample=isn, oracle_sample_s=round(t_or, 1),
               glacier_in_sample=int((osamp.a["ivgtyp"] == raw.cfg.isice).sum()),
               water_in_sample=int((osamp.a["xland"] > 1.5).sum()), restart=restart_s)
    if verbose:
        for c, ok, lines in report:
            print("checkpoint step %d: %s" % (c, "BIT-IDENTICAL" if ok else "DIFFERS\n  " + "\n  ".join(lines)))
        print(json.dumps(res))
    return res


if __name__ == "__main__":
    kw = {x.split("=")[0]: int(x.split("=")[1]) for x in sys.argv[1:] if "=" in x}
    a = [int(x) for x in sys.argv[1:] if "=" not in x]
    res = run(*(a[:2] if len(a) >= 2 else (3600, 1800)), nsteps=a[2] if len(a) > 2 else 720,
              nsample=a[3] if len(a) > 3 else 4096, restart_path=os.environ.get("NMP_RESTART_PATH"),
              cfgkw=kw or None, resort_frac=float(os.environ.get("NMP_RESORT_FRAC", "0.10")),
              resort_every=int(os.environ.get("NMP_RESORT_EVERY", "24")))
    sys.exit(0 if res["sample_bit_identical"] else 1)

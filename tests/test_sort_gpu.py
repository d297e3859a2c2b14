"""The device-side column order (noahmp_hip_sort_columns / _sort_staleness / _scatter_plan / _permute_step_arrays) and the
BASELINE configs[2] workload at its full size.  north_star: columns sorted by (vegetation type, snow-layer count) before launch;
the snow-layer count changes at lsm:7044 (COMBINE), 7110 (DIVIDE), 7177 (COMBO), 7294-7343 (SNOWH2O), so a run re-sorts."""
import ctypes as C

import numpy as np
import pytest

from noahmp_amd import synth, abi
from noahmp_amd.abi import FIELD_INFO
from noahmp_amd.state import ModelConfig

pytestmark = pytest.mark.gpu
FKEYS = ("coszin", "swdown", "glw", "t3d", "rainbl")


def _outs(store):
    return [k for k in store.a if k in FIELD_INFO and FIELD_INFO[k][2] != "in"]


def _cols(v, idx=None):
    """(nj, nk, ni) or (nj, ni) -> per-column rows, optionally gathered."""
    v = v.transpose(0, 2, 1).reshape(-1, v.shape[1]) if v.ndim == 3 else v.reshape(-1)
    return v if idx is None else v[idx]


def _up(x):
    """numpy -> device.  (Round 4 staged these multi-megabyte copies through page-locked tensors because every few full `-m gpu` runs one
    of torch's pageable copies died with "Memory access fault by GPU ... Write access to a read-only page".  Round 5 found the cause -- the HIP
    runtime's in-place page-locking of pageable buffers, tests/conftest.py -- and the staging is gone.)"""
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _down(t):
    return t.cpu().numpy()


def _host_key(s, tsk_bin=1.0, veg=True, snow=True, snow_first=False, band=None):
    a = s.a
    ivg = a["ivgtyp"].ravel().astype(np.int64)
    cls = np.where((a["xland"].ravel() - np.float32(1.5) >= 0) | (a["xice"].ravel() >= np.float32(s.cfg.xice_thres)), 2,
                   np.where(ivg == s.cfg.isice, 1, 0))
    vk = np.where((cls == 0) & veg, np.clip(ivg, 0, 63), 0)
    sk = np.where(snow, np.clip(-a["isnowxy"].ravel().astype(np.int64), 0, 3), 0)
    tb = np.zeros_like(ivg)
    if tsk_bin:
        t = a["tsk"].ravel().astype(np.float32)
        t = np.where(np.isnan(t), np.float32(250.0), t)
        tb = np.clip(((t - np.float32(230.0)) * np.float32(1000.0 / int(round(tsk_bin * 1000)))).astype(np.int64), 0, 255)
    hi = (sk << 6 | vk) if snow_first else (vk << 2 | sk)
    bd = np.clip(band.ravel().astype(np.int64), 0, 31) if band is not None else np.zeros_like(ivg)
    return np.where(cls == 2, 2 << 25, cls << 25 | hi << 17 | bd << 12 | tb), cls     # class(2) | veg, snow(8) | band(5) | cost(4) = 0 | tsk bin(8)


@pytest.mark.parametrize("kw", [dict(), dict(tsk_bin=0), dict(snow_first=True, tsk_bin=0.5), dict(veg=False), dict(snow=False), dict(band="lonband")],
                         ids=["default", "no_tsk", "snow_first", "no_veg", "no_snow", "band"])
def test_device_sort_is_the_stable_sort_of_the_key(engine, tables, kw):
    import torch
    s = synth.mixed_small(tables[1], ni=160, nj=48, glacier_frac=0.06, seed=71)
    s["xland"][3, :17] = 2.0
    s["xice"][5, 40:47] = 1.0
    s["tsk"][7, 3] = np.nan
    d = s.to_device("cuda:0")
    band = None
    if kw.get("band"):          # a caller-defined static sub-key plane (noahmp_hip_sort_set_band): here 24 "longitude bands" + out-of-range values
        band = (np.arange(160)[None, :] * 24 // 160 + np.zeros((48, 1), np.int64)).astype(np.int32)
        band[11, 5], band[12, 6] = -3, 77
        d.a["lonband"] = torch.from_numpy(band).cuda()
    perm = engine.sort_store(d, **kw).cpu().numpy().astype(np.int64)
    key, cls = _host_key(s, **{"tsk_bin": 1.0, **kw, "band": band})
    want = np.argsort(key, kind="stable")
    np.testing.assert_array_equal(perm, want)
    np.testing.assert_array_equal(d.sort_keys.cpu().numpy().view(np.uint32).astype(np.int64), key[want])
    assert d.class_ranges == (int((cls == 0).sum()), int((cls == 1).sum()))
    h = d.to_host()
    for k, v in s.a.items():
        if k != "dzs":
            assert np.array_equal(_cols(v, perm), _cols(h.a[k]), equal_nan=True), k
    if band is not None:        # the band plane travelled with the state
        assert np.array_equal(band.reshape(-1)[perm], d.a["lonband"].cpu().numpy().reshape(-1))
    assert engine.sort_staleness(d) == 0


def test_vegetation_order_of_the_key(engine, tables):
    """noahmp_hip_sort_set_veg_order: the categories named first run first inside the land range (longest-first list scheduling puts the
    cheap categories at the tail of the launch); a pure relabelling of the vegetation field of the key -- staleness 0 with the same order,
    identity again after a reset."""
    s = synth.mixed_small(tables[1], ni=160, nj=48, glacier_frac=0.06, seed=72)
    first = [14, 15, 11, 2]                                       # forests and crops first; everything else follows in numeric order
    seq = first + [v for v in range(64) if v not in first]
    rank = np.empty(64, np.int64)
    rank[seq] = np.arange(64)
    try:
        engine.set_veg_order(first)
        d = s.to_device("cuda:0")
        perm = engine.sort_store(d).cpu().numpy().astype(np.int64)
        key, cls = _host_key(s)
        vk = (key >> 19) & 63
        key2 = np.where(cls == 0, (key & ~(63 << 19)) | (rank[vk] << 19), key)
        np.testing.assert_array_equal(perm, np.argsort(key2, kind="stable"))
        land = s.a["ivgtyp"].ravel()[perm][:d.class_ranges[0]]
        seen = [v for i, v in enumerate(land) if i == 0 or land[i - 1] != v]
        assert seen[:4] == [v for v in first if (s.a["ivgtyp"] == v).any()][:4] and len(seen) == len(set(seen))
        assert engine.sort_staleness(d) == 0
    finally:
        engine.set_veg_order(None)
    d2 = s.to_device("cuda:0")
    np.testing.assert_array_equal(engine.sort_store(d2).cpu().numpy().astype(np.int64), np.argsort(_host_key(s)[0], kind="stable"))
    bad = (C.c_int32 * 4)(0, 1, 64, 3)
    assert engine.lib.noahmp_hip_sort_set_veg_order(bad, 4) == -105


@pytest.mark.parametrize("ni,nj", [(333, 37), (1111, 517), (1153, 769), (2051, 1601), (3100, 2049)])     # chunks of 1024 / 2048 / 4096 / 16384 / 16384 columns
def test_scatter_plan_on_device_equals_host_plan(engine, tables, ni, nj):
    import torch
    r = np.random.Generator(np.random.Philox(5))
    n = ni * nj                                               # not a multiple of the chunk
    p = r.permutation(n).astype(np.int32)
    perm = _up(p)
    src = [_up(r.normal(size=(nj, ni)).astype(np.float32)), _up(r.normal(size=(nj, 2, ni)).astype(np.float32))]
    dst = [torch.zeros_like(t) for t in src]
    sc = engine.scatter(dst, src, perm, ni, nj)
    chunk = engine.lib.noahmp_hip_scatter_chunk_of(ni, nj)
    assert chunk == {333: 1024, 1111: 2048, 1153: 4096, 2051: 16384, 3100: 16384}[ni]
    inv = np.empty(n, dtype=np.int64)
    inv[p] = np.arange(n)
    npad = (n + chunk - 1) // chunk * chunk
    invp = np.full(npad, np.iinfo(np.int64).max, dtype=np.int64)
    invp[:n] = inv
    invp = invp.reshape(-1, chunk)
    order = np.argsort(invp, axis=1, kind="stable")
    dpos = np.take_along_axis(invp, order, axis=1)
    np.testing.assert_array_equal(_down(sc.order).view(np.uint16), order.astype(np.uint16).ravel()[:n])
    np.testing.assert_array_equal(_down(sc.dpos), dpos.ravel()[:n].astype(np.int32))
    sc()
    engine.stream_sync()
    for t, u in zip(src, dst):
        np.testing.assert_array_equal(_cols(_down(t), p), _cols(_down(u)))
    # level arrays of which only the first level travels (nlev < 0 at the C-ABI): level 1 moved, level 2 of the destination untouched
    dst2 = [torch.full_like(t, -7.0) for t in src]
    sc2 = engine.scatter(dst2, src, perm, ni, nj, first_level_only=(1,))
    sc2()
    sc2.exchange([dst[1]], [src[1]], False, stream=None, first_level_only=(0,))     # the sorted_exchange entry, level 1 over the full result: no change
    engine.stream_sync()
    np.testing.assert_array_equal(_cols(_down(src[0]), p), _cols(_down(dst2[0])))
    got = _down(dst2[1])
    np.testing.assert_array_equal(_cols(_down(src[1])[:, :1, :], p), _cols(got[:, :1, :]))
    assert (got[:, 1, :] == -7.0).all()
    np.testing.assert_array_equal(_cols(_down(src[1]), p), _cols(_down(dst[1])))


def test_sort_refuses_a_tile_with_a_halo(engine, tables):
    s = synth.mixed_small(tables[1], ni=32, nj=8)
    s.set_index(its=2, ite=31, jts=2, jte=7)
    d = s.to_device("cuda:0")
    with pytest.raises(RuntimeError, match="memory block must be the tile"):
        engine.sort_store(d)


def test_resorted_run_over_snow_accumulation_equals_tile_order(engine, tables):
    """A day of snowfall on thin snow packs and melt on others: layers appear, divide, combine and vanish.  The sorted run checks its
    staleness every 4 steps and re-sorts; un-permuted with the composed permutation it equals the tile-order run bit for bit."""
    import torch
    cfg = ModelConfig()
    s = synth.mixed_small(tables[1], ni=128, nj=40, glacier_frac=0.05, snow_frac=0.6, seed=83, cfg=cfg)
    synth.first_step_fixups(s)
    r = np.random.Generator(np.random.Philox(84))
    warm = r.random(size=s.t_offset.shape) < 0.5
    toff = np.where(warm, s.t_offset + np.float32(6.0), s.t_offset - np.float32(4.0)).astype(np.float32)   # melt here, accumulate there
    nsteps = 24
    forc = []
    for it in range(1, nsteps + 1):
        synth.diurnal_forcing(s, (it + 5) % 24, t_offset=toff, rain_hours=tuple(range(24)), rain_mm=3.0)
        forc.append({k: torch.from_numpy(s.a[k].copy()).cuda() for k in FKEYS})
    plain, srt = s.to_device("cuda:0"), s.to_device("cuda:0")
    isn0 = s.a["isnowxy"].copy()
    perm = engine.sort_store(srt)
    work = {k: srt.a[k] for k in FKEYS}
    sc = engine.scatter([work[k] for k in FKEYS], [forc[0][k] for k in FKEYS], perm, s.ni, s.nj)
    args = srt.step_args(1, 2000, 180.0)
    stale, resorts = [], 0
    for it in range(1, nsteps + 1):
        plain.a.update(forc[it - 1])
        assert engine.noahmplsm(plain, it, 2000, 180.0).code == 0
    for it in range(1, nsteps + 1):
        sc.set_sources([forc[it - 1][k] for k in FKEYS])
        sc()
        args.itimestep = it
        engine.noahmplsm_async(args)
        if it % 4 == 0:
            st, _ = engine.sync()
            assert st.code == 0
            engine.sort_staleness_async(srt)                       # the count without a wait: enqueued, read later
            stale.append(engine.sort_staleness(srt))
            assert engine.sort_staleness_result(wait=True) == stale[-1]
            if stale[-1] > 0:
                perm = engine.sort_store(srt)
                assert engine.sort_staleness(srt) == 0
                work = {k: srt.a[k] for k in FKEYS}
                sc = engine.scatter([work[k] for k in FKEYS], [forc[0][k] for k in FKEYS], perm, s.ni, s.nj)
                args = srt.step_args(it, 2000, 180.0)
                resorts += 1
    st, _ = engine.sync()
    assert st.code == 0
    p = perm.cpu().numpy().astype(np.int64)
    assert sorted(p.tolist()) == list(range(s.ncol))
    hp, hs = plain.to_host(), srt.to_host()
    assert (hp.a["isnowxy"] != isn0).sum() > 50 and resorts >= 2, (stale, resorts)        # the layering really changed
    for k in _outs(hp):
        if k not in FKEYS:
            assert np.array_equal(_cols(hp.a[k], p), _cols(hs.a[k]), equal_nan=True), k


def test_config3_full_size_sorted_sample_bit_identical(engine, port, tables):
    """BASELINE configs[2] at its full size (4608 x 1536 = 7 077 888 columns; the grid bench.py times): 24 hourly steps on the
    device-sorted layout with staleness checks and re-sorts, every step status 0 (the model's own SW / energy / water balance checks
    for every column, lsm:1185-1221, gla:2939-2968); a fixed random sample of 4096 tile columns is advanced by the oracle
    from the same state with the same forcing and compared bit for bit at steps 1, 12 and 24."""
    import torch
    from tools.config5_run import extract
    gx, gy = 4608, 1536
    cfg = ModelConfig()
    s = synth.config3_tile(tables[1], gx, gy, cfg=cfg)
    synth.first_step_fixups(s)
    r = np.random.Generator(np.random.Philox(31))
    cols = np.sort(r.choice(gx * gy, size=4096, replace=False))
    osamp = extract(s, cols)
    toff_s = s.t_offset.reshape(-1)[cols][None].copy()
    d = s.to_device("cuda:0")
    n_cells = s.ncol
    forc = {}
    for h in range(24):
        synth.diurnal_forcing(s, h, t_offset=s.t_offset)
        forc[h] = {k: torch.from_numpy(s.a[k].copy()).cuda() for k in FKEYS}
    del s
    perm = engine.sort_store(d)
    work = {k: d.a[k] for k in FKEYS}
    sc = engine.scatter([work[k] for k in FKEYS], [forc[0][k] for k in FKEYS], perm, gx, gy)
    args = d.step_args(1, 2000, 180.0)
    snaps, resorts = {}, 0
    cols_t = torch.from_numpy(cols).cuda()
    for it in range(1, 25):
        sc.set_sources([forc[(it + 5) % 24][k] for k in FKEYS])
        sc()
        args.itimestep = it
        engine.noahmplsm_async(args)
        if it in (1, 12, 24):
            st, _ = engine.sync()
            assert st.code == 0 and st.n_land + st.n_glacier + st.n_skipped == n_cells * (1 if it == 1 else (11 if it == 12 else 12))
            inv = torch.empty_like(perm, dtype=torch.int64)
            inv[perm.long()] = torch.arange(perm.numel(), device=perm.device)
            snaps[it] = extract(d, inv[cols_t].cpu().numpy())
            if it == 12 and engine.sort_staleness(d) > 0.01 * n_cells:
                perm = engine.sort_store(d)
                work = {k: d.a[k] for k in FKEYS}
                sc = engine.scatter([work[k] for k in FKEYS], [forc[0][k] for k in FKEYS], perm, gx, gy)
                args = d.step_args(it, 2000, 180.0)
                resorts += 1
    for k in ("tsk", "hfx", "lh", "tslb", "smois", "snow", "snowh"):
        assert bool(torch.isfinite(d.a[k]).all()), k
    assert int(d.a["isnowxy"].min()) >= -3 and int(d.a["isnowxy"].max()) <= 0
    # the oracle on the sample, same chain
    for it in range(1, 25):
        synth.diurnal_forcing(osamp, (it + 5) % 24, t_offset=toff_s)
        so = port.noahmplsm(osamp, it, 2000, 180.0)
        assert so.code == 0
        if it in snaps:
            from tools.compare import exact_check
            ok, lines = exact_check(osamp, snaps[it])
            assert ok, "step %d:\n%s" % (it, "\n".join(lines[:6]))
    isn = set(np.unique(osamp.a["isnowxy"]).tolist())
    assert {0, -3} <= isn and (osamp.a["ivgtyp"] == cfg.isice).sum() > 10 and (osamp.a["ivgtyp"] == cfg.isurban).sum() > 20


def test_cost_key_sorts_by_recorded_trip_counts(engine, tables):
    """(Experiment builds only, -DNMP_COST_RECORD: the default library has no cost record -- its stores cost the land kernel 1.2 % and the
    key does not pay, profiles/r05_experiments.md section 2 -- and this test skips.)  set_option record_cost: a device-resident step leaves every land column's canopy-loop iterations and STOMATA bisection steps in an
    engine-owned plane (tile's current order); noahmp_hip_sort_columns(NOAHMP_SORT_COST) puts a bucket of them into the key, between the
    band and the temperature bin.  The order is the stable sort of that key, staleness ignores the bucket, a permutation invalidates the
    record, and the sorted store advances to the same bits as the tile-order one."""
    import torch
    s = synth.mixed_small(tables[1], ni=192, nj=40, glacier_frac=0.05, seed=73)
    synth.first_step_fixups(s)
    synth.diurnal_forcing(s, 13, t_offset=s.t_offset)
    d, ref = s.to_device("cuda:0"), s.to_device("cuda:0")
    prev = engine.set_option("record_cost", 1)
    if prev < 0:
        pytest.skip("library built without -DNMP_COST_RECORD")
    try:
        engine.noahmplsm(d, 1, 2000, 180.0)
        engine.noahmplsm(ref, 1, 2000, 180.0)
        n = s.ni * s.nj
        cost = np.zeros(2 * n, dtype=np.uint8)
        assert engine.lib.noahmp_hip_fetch_cost(cost.ctypes.data, n, None) == n
        iters, bis = cost[0::2].astype(np.int64), cost[1::2].astype(np.int64)
        h = d.to_host()
        veg_cols = (h.a["fvegxy"].ravel() > 0) & (h.a["ivgtyp"].ravel() != s.cfg.isice)
        assert iters.max() <= 20 and (iters[veg_cols] >= 6).all() and bis.max() <= 40 and (bis > 0).any()      # loop 1 runs >= 6 iterations (lsm:3453)
        key, cls = _host_key(h, tsk_bin=1.0)
        assert (iters[cls != 0] == 0).all()
        c = 8 * iters + bis
        bucket = np.where(c == 0, 0, np.minimum(1 + c // 14, 15))
        key = np.where(cls == 0, key | bucket << 8, key)
        perm = engine.sort_store(d, cost=True).cpu().numpy().astype(np.int64)
        np.testing.assert_array_equal(perm, np.argsort(key, kind="stable"))
        assert len(np.unique(bucket[cls == 0])) >= 3
        assert engine.sort_staleness(d) == 0                       # the bucket is not part of what "stale" means
        assert engine.lib.noahmp_hip_fetch_cost(cost.ctypes.data, n, None) == 0     # the record died with the permutation
        synth.diurnal_forcing(s, 14, t_offset=s.t_offset)
        for k in ("coszin", "swdown", "glw", "t3d", "rainbl"):
            ref.a[k] = torch.from_numpy(s.a[k].copy()).cuda()
            v = s.a[k]
            vs = _cols(v, perm)
            d.a[k] = torch.from_numpy(np.ascontiguousarray((vs.reshape(v.shape[0], v.shape[2], v.shape[1]).transpose(0, 2, 1) if v.ndim == 3
                                                            else vs.reshape(v.shape)))).cuda()
        engine.noahmplsm(d, 2, 2000, 180.0)
        engine.noahmplsm(ref, 2, 2000, 180.0)
        hd, hr = d.to_host(), ref.to_host()
        for k in _outs(hr):
            assert np.array_equal(_cols(hr.a[k], perm), _cols(hd.a[k]), equal_nan=True), k
    finally:
        engine.set_option("record_cost", prev)
        engine._apply_ranges(None)

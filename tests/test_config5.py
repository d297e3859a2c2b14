"""global lat/lon grid, cold start -> [forcing interpolation -> forcing preparation -> column step] x n,
entirely on device-resident arrays (noahmp_amd/synth5.py generates the workload, tools/config5_run.py drives it).

CPU: the generator and the oracle chain on a coarse grid.  GPU: the chain through the C-ABI at 1 degree, a random sample of
columns bit-identical to the oracle at every checkpoint (columns are independent, so a sample of a run is a run of the sample;
tools/config5_run.py does the same at the full 3600 x 1800 size, profiles/r01_parity_stats.md)."""
import numpy as np
import pytest

from noahmp_amd import synth5


def test_generator_classes_and_oracle_chain(tables, port):
    import torch
    ni, nj = 72, 36
    s, lon, static = synth5.config5_raw(ni, nj)
    cfg = s.cfg
    veg = s["ivgtyp"]
    assert (veg == cfg.isice).any() and (veg == cfg.isurban).any() and (veg == cfg.iswater).any()
    assert (np.abs(s["xlatin"][veg == cfg.isice]) > 66.0).all()
    assert (s["snow"] > 0).any() and (s["snow"] == 0).any()
    recs = synth5.Records(torch.from_numpy(s["xlatin"]), torch.from_numpy(lon), {k: torch.from_numpy(v) for k, v in static.items()})
    rc, _ = port.noahmp_init(s, fndsnowh=True)
    assert rc == 0
    rain = np.zeros((nj, ni), np.float32)
    host = lambda r: {k: (v.numpy() if v is not None else None) for k, v in r.items()}
    for n in range(6):
        ri, k = divmod(n, synth5.RECORD_HOURS)
        port.forcing_interpolate(s, host(recs.at(ri)), host(recs.at(ri + 1)) if k else None, 3600 * k, 10800, rain)
        jul = port.forcing_prep(s, lon, rain, *synth5.step_time(n), first_step=(n == 0))
        st = port.noahmplsm(s, n + 1, 2000, jul)
        assert st.code == 0
    assert st.n_land + st.n_glacier == int((veg != cfg.iswater).sum()) and st.n_skipped == int((veg == cfg.iswater).sum())
    assert set(np.unique(s["isnowxy"])) >= {0, -3}
    assert np.isfinite(s["hfx"]).all() and 200.0 < s["tsk"].min() and s["tsk"].max() < 340.0


def test_extract_columns():
    from tools.config5_run import extract
    s, _, _ = synth5.config5_raw(24, 12)
    flat = np.array([0, 5, 24 * 3 + 7, 24 * 12 - 1])
    e = extract(s, flat)
    assert e.ni == 4 and e.nj == 1
    np.testing.assert_array_equal(e["tsk"][0], s["tsk"].reshape(-1)[flat])
    np.testing.assert_array_equal(e.a["tslb"][0, :, 2], s.a["tslb"][3, :, 7])
    np.testing.assert_array_equal(e["ivgtyp"][0], s["ivgtyp"].reshape(-1)[flat])


@pytest.mark.gpu
def test_gpu_config5_chain_sample_bit_identical():
    from tools.config5_run import run
    res = run(360, 180, nsteps=30, nsample=2048, verbose=False, checkpoints=(1, 7, 24))
    assert res["sample_bit_identical"], res
    assert res["checkpoints"] == [1, 7, 24, 30]
    assert res["glacier_in_sample"] > 0 and res["water_in_sample"] > 0
    assert set(res["isnow_states_in_sample"]) >= {0, -3}


@pytest.mark.gpu
def test_gpu_config5_chain_with_resorts_bit_identical():
    """The same chain with the column order re-established every 6 steps whatever the staleness (re-sort of the state, of the
    longitude / static record fields that live in the store's column order, re-evaluation of the cached forcing records): the
    sample, followed through every permutation, stays bit-identical to the oracle."""
    from tools.config5_run import run
    res = run(360, 180, nsteps=30, nsample=2048, verbose=False, checkpoints=(1, 7, 24), resort_every=6, resort_frac=-1.0)
    assert res["sample_bit_identical"], res
    assert res["resorts"] == 4 and res["checkpoints"] == [1, 7, 24, 30]


@pytest.mark.gpu
def test_gpu_config5_full_size_sample_bit_identical():
    """BASELINE configs[4] at its full size (3600 x 1800 = 6 480 000 cells; the grid `bench.py --workload config5` and the default
    line's `config5_reference` leg time): cold start on the device, 24 hourly steps of interpolate -> prepare -> step on the sorted
    layout, every sync status 0 (the model's own SW / energy / water balance checks for every column, lsm:1185-1221,
    gla:2939-2968) and every cell visited every step; a fixed random sample of 4096 columns is advanced by the oracle through the
    same chain from the same raw state and compared bit for bit at steps 1, 12 and 24 -- the twin of
    test_config3_full_size_sorted_sample_bit_identical."""
    from tools.config5_run import run
    res = run(3600, 1800, nsteps=24, nsample=4096, verbose=False, checkpoints=(1, 12))
    assert res["device_status_max"] == 0, res
    assert res["cells_stepped"] == 3600 * 1800 * 24, res
    assert res["sample_bit_identical"], res
    assert res["checkpoints"] == [1, 12, 24]
    assert res["glacier_in_sample"] > 10 and res["water_in_sample"] > 10
    assert set(res["isnow_states_in_sample"]) >= {0, -3}
    assert res["land_columns"] > 6_000_000


@pytest.mark.parametrize("level", [1, 2])
def test_smooth_generators_keep_marginals_cells_and_decomposition(level):
    """Round 6's second / third config-5 generator (synth5.smooth_uniform: forcing factors -- and with level 2 the state's per-cell noise --
    as spatially smooth fields): a tile is the same cells whatever the decomposition, the marginal ranges are the default generator's,
    neighbouring cells are close (the default's are independent), and what is documented as unchanged is unchanged."""
    gx, gy = 360, 180
    raw0, lon0, st0 = synth5.config5_tile(gx, gy)
    raw, lon, st = synth5.config5_tile(gx, gy, smooth=level)
    part = synth5.config5_tile(gx, gy, 100, 40, 90, 50, smooth=level)
    for k in st:
        assert np.array_equal(st[k][40:90, 100:190], part[2][k]), k
    for k in ("isltyp", "tsk", "snow", "smois", "ivgtyp", "xland"):
        assert np.array_equal(raw.a[k][40:90, ..., 100:190], part[0].a[k]), k
    for k, (lo, hi) in dict(cloud=(0.4, 0.9), rh=(0.4, 0.9), uwind=(1.0, 8.0), vwind=(-3.0, 3.0), phase=(0, 15)).items():
        assert lo <= st[k].min() and st[k].max() <= hi and st[k].std() > 0.1 * (hi - lo), k
        rough = lambda f: np.abs(np.diff(f, axis=1)).mean()
        assert rough(st[k]) < 0.6 * rough(st0[k]), k                       # smooth along a row where the default is white noise
    assert np.array_equal(raw.a["ivgtyp"], raw0.a["ivgtyp"]) and np.array_equal(raw.a["xland"], raw0.a["xland"])
    if level == 1:
        for k in raw.a:
            if k != "dzs":
                assert np.array_equal(raw.a[k], raw0.a[k]), k                # the state is the default's
        assert np.array_equal(st["tbase"], st0["tbase"])
    else:
        land = (raw.a["xland"] < 1.5) & (raw.a["ivgtyp"] != 24)
        assert set(np.unique(raw.a["isltyp"][land])) <= set(range(1, 13)) and len(np.unique(raw.a["isltyp"][land])) == 12
        assert abs(float((raw.a["snow"] > 0).mean()) - float((raw0.a["snow"] > 0).mean())) < 0.02
        assert np.abs(np.diff(raw.a["tsk"], axis=1)).mean() < 0.5 * np.abs(np.diff(raw0.a["tsk"], axis=1)).mean()

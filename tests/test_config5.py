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

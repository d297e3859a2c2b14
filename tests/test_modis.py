"""The other land-use data set the reference supports: MODIFIED_IGBP_MODIS_NOAH (20 categories, urban 13, snow/ice 15, barren 16,
water 17; hdrv:130-143, MPTABLE.TBL / VEGPARM.TBL second blocks).  The engine takes the category indices as arguments and the
tables as an image, so nothing in the kernels is USGS-specific: cold start + 24-hour free run on a tile holding every MODIS category
with 0..3 snow layers, the C restatement against the compiled reference (live) and the GPU against the restatement, bit for bit."""
import numpy as np
import pytest

from noahmp_amd import synth
from noahmp_amd.abi import FIELD_INFO
from noahmp_amd.state import ModelConfig
from noahmp_amd.tables import load_tables


def modis_raw(modis):
    """veg_snow_matrix before its cold start, categories folded into MODIS 1..20 -> (raw store, t_offset)."""
    tb = modis[1]
    cfg = ModelConfig(isurban=tb["isurban"], isice=tb["issnow"], iswater=tb["iswater"])
    captured = []
    orig = synth.noahmp_init
    synth.noahmp_init = lambda store, tables, fndsnowh=True: captured.append((store.copy(), store.t_offset.copy()))
    try:
        synth.veg_snow_matrix(tb, cfg=cfg)
    finally:
        synth.noahmp_init = orig
    raw, toff = captured[0]
    raw.cfg = cfg
    raw["ivgtyp"] = ((raw["ivgtyp"] - 1) % 20) + 1
    water, ice = raw["ivgtyp"] == cfg.iswater, raw["ivgtyp"] == cfg.isice
    raw["xland"] = np.where(water, 2.0, 1.0).astype(np.float32)
    raw["isltyp"] = np.where(water, 14, np.where(ice, 16, np.where(raw["isltyp"] > 12, 6, raw["isltyp"]))).astype(np.int32)
    return raw, toff, cfg, water


def _outs(s):
    return [k for k in s.a if FIELD_INFO[k][2] != "in"]


@pytest.fixture(scope="module")
def modis():
    return load_tables("modis")


def test_port_matches_reference_with_modis_tables(reflib, modis):
    from oracle.portlib import PortLib
    port = PortLib(autobuild=True)
    port.set_tables(modis[0])
    reflib.set_tables(modis[0])
    try:
        raw, toff, cfg, water = modis_raw(modis)
        a, b = raw.copy(), raw.copy()
        reflib.noahmp_init(a, fndsnowh=True)       # the harness passes MMINLU = 'USGS' to NOAHMP_INIT, which re-reads the tables:
        reflib.set_tables(modis[0])                # put the MODIS image back (the per-column part uses only the soil tables)
        rc, _ = port.noahmp_init(b, fndsnowh=True)
        assert rc == 0
        for s in (a, b):
            synth.first_step_fixups(s)
        glac = 0
        for it in range(1, 25):
            for s in (a, b):
                synth.diurnal_forcing(s, (it - 1) % 24, t_offset=toff)
            reflib.noahmplsm(a, it, 2000, 180.0)
            st = port.noahmplsm(b, it, 2000, 180.0)
            assert st.code == 0
            glac = st.n_glacier
            for k in _outs(a):
                x, y = a.a[k], b.a[k]
                assert np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y), (k, it)
        assert glac == int((raw["ivgtyp"] == cfg.isice).sum()) > 0 and st.n_skipped == int(water.sum()) > 0
        assert sorted(np.unique(raw["ivgtyp"]).tolist()) == list(range(1, 21))
    finally:                                     # both libraries keep their tables in process-wide module state
        reflib.set_tables(load_tables("usgs")[0])
        port.set_tables(load_tables("usgs")[0])


@pytest.mark.gpu
def test_gpu_matches_oracle_with_modis_tables(modis):
    import ctypes as C
    import torch
    from noahmp_amd.driver import Engine
    from oracle.portlib import PortLib
    port = PortLib(autobuild=True)
    port.set_tables(modis[0])
    eng = Engine(modis[0], device=0)
    try:
        raw, toff, cfg, water = modis_raw(modis)
        o = raw.copy()
        port.noahmp_init(o, fndsnowh=True)
        d = raw.to_device("cuda:0")
        eng.noahmp_init(d, fndsnowh=True)
        synth.first_step_fixups(o)
        for k in ("eahxy", "tahxy", "chxy", "cmxy"):
            d.a[k].copy_(torch.from_numpy(o.a[k]))
        for it in range(1, 25):
            synth.diurnal_forcing(o, (it - 1) % 24, t_offset=toff)
            for k in ("coszin", "swdown", "glw", "t3d", "rainbl", "qv3d", "u_phy", "v_phy", "p8w3d", "dz8w"):
                d.a[k].copy_(torch.from_numpy(o.a[k]))
            port.noahmplsm(o, it, 2000, 180.0)
            st = eng.noahmplsm(d, it, 2000, 180.0)
            assert st.code == 0
        h = d.to_host()
        for k in _outs(o):
            x, y = o.a[k], h.a[k]
            assert np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y), k
    finally:
        eng.lib.noahmp_hip_set_tables(C.byref(load_tables("usgs")[0]))     # the engine is a process-wide singleton
        port.set_tables(load_tables("usgs")[0])

"""Cold start: NOAHMP_INIT's per-column part + SNOW_INIT, reference phys/module_sf_noahmpdrv.F90:988-1283.

CPU: the C restatement against the compiled reference (bit-exact), the device source compiled for the host against the
restatement (bit-exact), the committed fixture.  GPU: noahmp_hip_init through the C-ABI against the restatement (bit-exact),
host-memory vs device-resident path."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from noahmp_amd import synth
from noahmp_amd.abi import FIELD_INFO
from noahmp_amd.state import ColumnStore, ModelConfig

INIT_FIELDS = ["snow", "snowh", "canwat", "tslb", "smois", "sh2o", "isnowxy", "tvxy", "tgxy", "canicexy", "canliqxy",
               "eahxy", "tahxy", "cmxy", "chxy", "fwetxy", "sneqvoxy", "alboldxy", "qsnowxy", "wslakexy", "zwtxy", "waxy",
               "wtxy", "tsnoxy", "zsnsoxy", "snicexy", "snliqxy", "lfmassxy", "rtmassxy", "stmassxy", "woodxy", "stblcpxy",
               "fastcpxy", "xsaixy", "t2mvxy", "t2mbxy"]


def raw_store(tables, ni=64, nj=12, seed=5, cfg=None, marker=-777.0):
    """The state a caller hands to NOAHMP_INIT: forcing-independent inputs set, everything else a marker value."""
    import noahmp_amd.init as ini
    captured = []
    orig = synth.noahmp_init

    def spy(store, tb, fndsnowh=True):
        captured.append(store.copy())
        return orig(store, tb, fndsnowh)
    synth.noahmp_init = spy
    try:
        synth.mixed_small(tables[1], ni=ni, nj=nj, seed=seed, cfg=cfg)
    finally:
        synth.noahmp_init = orig
    s = captured[0]
    for k in INIT_FIELDS:
        if k not in ("snow", "snowh", "tslb", "smois", "sh2o"):
            s.a[k][...] = marker if s.a[k].dtype.kind == "f" else 7
    return s


def same(a, b, what):
    for k in a.a:
        if k == "dzs":
            continue
        x, y = a.a[k], b.a[k]
        ok = np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y)
        assert ok, "%s: %s differs at %s" % (what, k, np.argwhere(x != y)[:3].tolist())


@pytest.mark.parametrize("fnd", [True, False])
@pytest.mark.parametrize("run", [1, 5])
def test_port_matches_reference_bitexact(tables, port, reflib, fnd, run):
    reflib.set_tables(tables[0])
    s = raw_store(tables, cfg=ModelConfig(iopt_run=run if run != 5 else 1))
    s.cfg = ModelConfig(iopt_run=run)
    if run == 5:
        pytest.skip("reference NOAHMP_INIT needs the optional MMF arguments under OPT_RUN=5 (drv:1154-1176)")
    a, b = s.copy(), s.copy()
    reflib.noahmp_init(a, fndsnowh=fnd)
    rc, st = port.noahmp_init(b, fndsnowh=fnd)
    assert rc == 0
    same(a, b, "fndsnowh=%s" % fnd)
    assert set(np.unique(b["isnowxy"])) == {0, -1, -2, -3}
    # the marker survives exactly where the reference writes nothing (inactive ZSNSOXY entries, drv:1275)
    assert (b["zsnsoxy"][:, :3, :] == -777.0).any()


@pytest.mark.parametrize("fnd", [True, False])
@pytest.mark.parametrize("run", [1, 5])
def test_device_source_on_host_matches_port(tables, port, fnd, run):
    from host_emul.emullib import EmulLib
    em = EmulLib()
    em.set_tables(tables[0])
    s = raw_store(tables)
    s.cfg = ModelConfig(iopt_run=run)
    a, b = s.copy(), s.copy()
    rc, _ = port.noahmp_init(a, fndsnowh=fnd)
    rc2, _ = em.noahmp_init(b, fndsnowh=fnd)
    assert rc == 0 and rc2 == 0
    same(a, b, "emul")


def test_numpy_mirror_agrees(tables, port):
    """noahmp_amd/init.py (what synth uses to build test states) against the restatement."""
    import noahmp_amd.init as ini
    s = raw_store(tables)
    a, b = s.copy(), s.copy()
    port.noahmp_init(a)
    ini.noahmp_init(b, tables[1])
    for k in INIT_FIELDS:
        if k == "sh2o":       # numpy's powf is not glibc's
            np.testing.assert_allclose(a.a[k], b.a[k], rtol=3e-7, atol=0)
        elif k == "zsnsoxy":
            m = b.a[k] != -777.0
            np.testing.assert_array_equal(a.a[k][m], b.a[k][m])
        else:
            np.testing.assert_array_equal(a.a[k], b.a[k], err_msg=k)


def test_bad_soil_type_is_fatal(tables, port):
    s = raw_store(tables, ni=16, nj=4)
    s["isltyp"][2, 5] = 0
    rc, st = port.noahmp_init(s)
    assert rc == 1 and (st.i, st.j) == (6, 3)


def test_golden_fixture(tables, port):
    """tests/golden/make_golden_init.py wrote this from the compiled reference."""
    z = np.load(os.path.join(GOLDEN, "golden_init.npz"))
    ni, nj = int(z["ni"]), int(z["nj"])
    s = ColumnStore(ni, nj, ModelConfig())
    for k in s.a:
        if "in/" + k in z:
            s.a[k][...] = z["in/" + k]
    port.noahmp_init(s)
    for k in INIT_FIELDS:
        np.testing.assert_array_equal(s.a[k], z["out/" + k], err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("fnd", [True, False])
@pytest.mark.parametrize("run", [1, 5])
def test_gpu_init_bit_identical_to_oracle(engine, port, tables, fnd, run):
    s = raw_store(tables, ni=256, nj=32, seed=8)
    s.cfg = ModelConfig(iopt_run=run)
    a, h = s.copy(), s.copy()
    d = s.to_device("cuda:0")
    port.noahmp_init(a, fndsnowh=fnd)
    st = engine.noahmp_init(h, fndsnowh=fnd)               # host-memory path of the C-ABI
    engine.noahmp_init(d, fndsnowh=fnd)                    # device-resident path
    assert st.code == 0 and st.n_land == 256 * 32
    same(a, h, "host path")
    same(a, d.to_host(), "device path")


@pytest.mark.gpu
def test_gpu_init_golden_and_fatal(engine, tables):
    from noahmp_amd.driver import NoahMPFatal
    z = np.load(os.path.join(GOLDEN, "golden_init.npz"))
    ni, nj = int(z["ni"]), int(z["nj"])
    s = ColumnStore(ni, nj, ModelConfig())
    for k in s.a:
        if "in/" + k in z:
            s.a[k][...] = z["in/" + k]
    bad = s.copy()
    engine.noahmp_init(s)
    for k in INIT_FIELDS:
        np.testing.assert_array_equal(s.a[k], z["out/" + k], err_msg=k)
    bad["isltyp"][1, 3] = 0
    with pytest.raises(NoahMPFatal) as e:
        engine.noahmp_init(bad)
    assert (e.value.code, e.value.i, e.value.j) == (1, 4, 2)


@pytest.mark.gpu
def test_gpu_cold_start_then_steps_match_oracle(engine, port, tables):
    """Device-resident from the first moment: init on the GPU, then 6 steps, against the oracle doing the same."""
    from tools.compare import exact_check
    s = raw_store(tables, ni=128, nj=16, seed=9)
    synth.diurnal_forcing(s, 9, t_offset=getattr(s, "t_offset", None))
    o = s.copy()
    d = s.to_device("cuda:0")
    port.noahmp_init(o)
    engine.noahmp_init(d)
    h = d.to_host()
    for st_ in (o, h):
        synth.first_step_fixups(st_)
    d = h.to_device("cuda:0")
    import torch
    for it in range(1, 7):
        synth.diurnal_forcing(o, 8 + it, t_offset=getattr(s, "t_offset", None))
        for k in ("coszin", "swdown", "glw", "t3d", "rainbl"):
            d.a[k].copy_(torch.from_numpy(o.a[k]))
        port.noahmplsm(o, it, 2000, 180.0)
        engine.noahmplsm(d, it, 2000, 180.0)
    ok, lines = exact_check(o, d.to_host())
    assert ok or not engine.exact_libm, "\n".join(lines)


# ------------------------------------------------------------------------------------------------
# GROUNDWATER_INIT + EQSMOISTURE (drv:1286-1522), the OPT_RUN=5 half of the cold start
GWI_OUT = ["smois", "sh2o", "smoiseq", "smcwtdxy", "zwtxy", "deeprechxy", "rechxy", "qslat", "qrfs", "qsprings"]


def gwi_store(tables, ni=48, nj=24, seed=21):
    """Raw cold-start inputs + MMF planes; AREA = DX*DY as NOAHMP_INIT sets it (drv:1117)."""
    s = raw_store(tables, ni=ni, nj=nj, seed=seed)
    s.cfg = ModelConfig(iopt_run=5)
    synth.groundwater_fields(s, tables[1], seed=seed + 1, area=s.cfg.dx * s.cfg.dx)
    r = np.random.default_rng(seed)
    s["ivgtyp"][r.random(size=(nj, ni)) < 0.03] = s.cfg.iswater          # the init's land mask uses ISWATER
    s["smoiseq"][...] = -777.0
    for k in ("smcwtdxy", "deeprechxy", "rechxy", "qslat", "qrfs", "qsprings"):
        s.a[k][...] = -777.0
    return s


def test_groundwater_init_port_matches_reference(tables, port, reflib):
    reflib.set_tables(tables[0])
    s = gwi_store(tables)
    a, b = s.copy(), s.copy()
    reflib.noahmp_init_mmf(a)                       # NOAHMP_INIT(OPT_RUN=5, MMF arguments present)
    rc, _ = port.noahmp_init(b)
    assert rc == 0
    port.groundwater_init(b)
    same(a, b, "mmf init")
    w0, w1 = s["zwtxy"], b["zwtxy"]
    deep = w0 < -3.0
    assert deep.any() and ((w0 >= -2.0) & (w1 != w0)).any()       # Newton branch and in-column adjustment both hit
    assert (b["smoiseq"] != -777.0).all()


def test_groundwater_init_device_source_on_host(tables, port):
    from host_emul.emullib import EmulLib
    em = EmulLib()
    em.set_tables(tables[0])
    s = gwi_store(tables, seed=23)
    port.noahmp_init(s)
    a, b = s.copy(), s.copy()
    port.groundwater_init(a)
    em.groundwater_init(b)
    same(a, b, "emul gw init")


@pytest.mark.gpu
def test_gpu_groundwater_init_bit_identical(engine, port, tables):
    s = gwi_store(tables, ni=192, nj=64, seed=25)
    port.noahmp_init(s)
    a, h = s.copy(), s.copy()
    d = s.to_device("cuda:0")
    port.groundwater_init(a)
    engine.groundwater_init(h)
    engine.groundwater_init(d)
    if engine.exact_libm:
        same(a, h, "host path")
        same(a, d.to_host(), "device path")
    else:
        for k in GWI_OUT:
            np.testing.assert_allclose(h.a[k], a.a[k], rtol=2e-5, atol=1e-6, err_msg=k)


@pytest.mark.gpu
def test_gpu_mmf_cold_start_to_first_groundwater_call(engine, port, tables):
    """OPT_RUN=5 entirely on the device: init, groundwater init, one step, one WTABLE call; vs the oracle."""
    from tools.compare import exact_check
    if not engine.exact_libm:
        pytest.skip("ocml build")
    s = gwi_store(tables, ni=96, nj=32, seed=27)
    synth.diurnal_forcing(s, 10, t_offset=getattr(s, "t_offset", None))
    o, d = s.copy(), s.to_device("cuda:0")
    port.noahmp_init(o); port.groundwater_init(o)
    engine.noahmp_init(d); engine.groundwater_init(d)
    h = d.to_host()
    for st_ in (o, h):
        synth.first_step_fixups(st_)
    d = h.to_device("cuda:0")
    so = port.noahmplsm(o, 1, 2000, 180.0)
    sd = engine.noahmplsm(d, 1, 2000, 180.0, check=False)
    assert so.code == sd.code
    port.wtable_mmf(o)
    engine.wtable_mmf(d)
    ok, lines = exact_check(o, d.to_host())
    assert ok, "\n".join(lines)
    hh = d.to_host()
    for k in ("qrf", "qspring", "qslat", "qrfs", "qsprings", "smoiseq"):
        np.testing.assert_array_equal(o.a[k], hh.a[k], err_msg=k)

"""Forcing preparation on the device: driver/module_hrldas_noahmp_driver.F90:336-354 + CALC_DECLIN (hdrv:813-863).

COSZEN / JULIAN are PINNED: CALC_DECLIN is an external subroutine that needs only util/module_date_utilities.F, so `make -C oracle
declin` compiles it unmodified (cut out of the driver file at build time, nothing committed) and the oracle restatement
(oracle/nmp_forcing.c) is held to it bit for bit -- live where oracle/_ref exists, and through tests/golden/golden_declin.npz
(produced by tests/golden/make_golden_declin.py from that build) everywhere.  The copies / unit scalings of hdrv:336-354 are
checked against their numpy float32 statement; the temporal interpolation (netcdf_io:1369-1403, module-internal, needs NetCDF)
stays a restatement."""
import ctypes as C

import numpy as np
import pytest

from noahmp_amd import abi
from noahmp_amd.state import ColumnStore, ModelConfig

OUT = ["t3d", "qv3d", "u_phy", "v_phy", "p8w3d", "rainbl", "vegfra", "dz8w", "coszin", "eahxy", "tahxy", "chxy", "cmxy"]


def case(ni=96, nj=40, seed=3):
    r = np.random.default_rng(seed)
    s = ColumnStore(ni, nj, ModelConfig())
    s["xlatin"] = r.uniform(-89.0, 89.0, size=(nj, ni)).astype(np.float32)
    lon = r.uniform(-180.0, 180.0, size=(nj, ni)).astype(np.float32)
    rain = r.uniform(0.0, 2e-3, size=(nj, ni)).astype(np.float32)
    for k, lo, hi in (("t3d", 250, 310), ("qv3d", 1e-4, 2e-2), ("u_phy", -10, 10), ("v_phy", -10, 10), ("p8w3d", 6e4, 1.02e5)):
        s.a[k][:, 0, :] = r.uniform(lo, hi, size=(nj, ni)).astype(np.float32)
        s.a[k][:, 1, :] = -777.0
    s["vegfra"] = r.uniform(0.0, 1.0, size=(nj, ni)).astype(np.float32)
    for k in ("coszin", "dz8w", "rainbl", "eahxy", "tahxy", "chxy", "cmxy"):
        s[k] = -777.0
    return s, lon, rain


def _day_of_year(y, mo, d):
    import datetime
    return (datetime.date(y, mo, d) - datetime.date(y, 1, 1)).days            # GETH_IDTS(date, YYYY-01-01): days since 1 January


def _oracle_cosz(port, lat, lon, when):
    y, mo, d, h, mi, sec = [int(x) for x in when]
    s = ColumnStore(lat.size, 1, ModelConfig())
    s["xlatin"] = lat[None, :]
    jul = port.forcing_prep(s, lon[None, :].copy(), np.zeros((1, lat.size), np.float32), _day_of_year(y, mo, d), h, mi, sec)
    return s["coszin"][0].copy(), np.float32(jul)


def test_oracle_cosz_equals_reference_calc_declin_fixture(port):
    """oracle/nmp_forcing.c == the reference's CALC_DECLIN (golden_declin.npz: 11 dates x 4096 points), COSZ and JULIAN bit for bit."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "golden_declin.npz"))
    for i, when in enumerate(g["when"]):
        cosz, jul = _oracle_cosz(port, g["lat"], g["lon"], when)
        assert jul == g["julian%02d" % i], when
        np.testing.assert_array_equal(cosz, g["cosz%02d" % i], err_msg=str(when))


def test_oracle_cosz_equals_compiled_calc_declin_live(port):
    """The same against the compiled reference subroutine itself, at random dates and points (dev container only)."""
    import os
    from golden import make_golden_declin as m
    if not os.path.exists(m.LIB):
        pytest.skip("oracle/_ref/libnoahmp_declin_ref.so not built (needs /root/reference)")
    lib = m.ref_lib()
    r = np.random.Generator(np.random.Philox(23))
    for _ in range(40):
        y = int(r.choice([1999, 2000, 2001, 2004, 2100]))
        mo, d = int(r.integers(1, 13)), int(r.integers(1, 29))
        when = (y, mo, d, int(r.integers(0, 24)), int(r.integers(0, 60)), int(r.integers(0, 60)))
        lat, lon = m.grid(seed=int(r.integers(1, 1 << 30)), n=512)
        want, wjul = m.run_reference(lib, when, lat, lon)
        cosz, jul = _oracle_cosz(port, lat, lon, when)
        assert jul == wjul, when
        np.testing.assert_array_equal(cosz, want, err_msg=str(when))


TIMES = [(0, 0, 0, 0), (45, 13, 30, 0), (79, 23, 59, 59), (80, 0, 0, 1), (171, 12, 0, 0), (200, 6, 15, 30), (364, 18, 45, 12)]


@pytest.mark.parametrize("when", TIMES)
def test_oracle_against_float64_formula(port, when):
    iday, h, m, sec = when
    s, lon, rain = case()
    jul = port.forcing_prep(s, lon, rain, iday, h, m, sec, scale_vegfra=True)
    assert jul == np.float32(iday) + np.float32(h) / np.float32(24.0)
    d2r = np.float64(np.float32(3.14159265) / np.float32(180.0))
    sx = (360.0 / 365.0) * ((jul - 80.0) if jul >= 80.0 else (jul + 285.0)) * d2r
    decl = np.arcsin(np.sin(23.5 * d2r) * np.sin(sx))
    tloc = np.mod(h + m / 60.0 + sec / 3600.0 + lon.astype(np.float64) / 15.0 + 24.0, 24.0)
    lat = s["xlatin"].astype(np.float64) * d2r
    want = np.sin(lat) * np.sin(decl) + np.cos(lat) * np.cos(decl) * np.cos(15.0 * (tloc - 12.0) * d2r)
    np.testing.assert_allclose(s["coszin"], want, atol=3e-6, rtol=0)
    for k in ("t3d", "qv3d", "u_phy", "v_phy", "p8w3d"):
        np.testing.assert_array_equal(s.a[k][:, 1, :], s.a[k][:, 0, :])
    np.testing.assert_array_equal(s["rainbl"], rain * np.float32(s.cfg.dt))
    assert (s["dz8w"] == np.float32(2.0 * s.cfg.zlvl)).all() and s["vegfra"].max() > 1.0
    assert (s["eahxy"] == -777.0).all() and (s["chxy"] == -777.0).all()          # only with first_step


def test_first_step_guesses(port):
    from noahmp_amd import synth
    s, lon, rain = case()
    want = s.copy()
    port.forcing_prep(s, lon, rain, 171, 0, first_step=True)
    synth.first_step_fixups(want)                                                # numpy float32 statement of hdrv:374-384
    for k in ("eahxy", "tahxy", "chxy", "cmxy"):
        np.testing.assert_array_equal(s[k], want[k], err_msg=k)


@pytest.mark.parametrize("when", TIMES)
def test_device_source_on_host_matches_oracle(port, when):
    from host_emul.emullib import EmulLib
    em = EmulLib()
    lib = abi.load_library()                    # host-only entry of the product library (no GPU call)
    iday, h, m, sec = when
    s, lon, rain = case(seed=5)
    a, b = s.copy(), s.copy()
    first = iday % 2 == 1
    port.forcing_prep(a, lon, rain, iday, h, m, sec, scale_vegfra=True, first_step=first)
    sd, cd = C.c_float(0), C.c_float(0)
    jul = lib.noahmp_hip_declination(iday, h, C.byref(sd), C.byref(cd))
    hour = np.float32(np.float32(np.float32(h) + np.float32(m) / np.float32(60.0)) + np.float32(sec) / np.float32(3600.0))
    em.forcing_prep(b, lon, rain, float(hour), sd.value, cd.value, scale_vegfra=True, first_step=first)
    assert jul == np.float32(iday) + np.float32(h) / np.float32(24.0)
    for k in OUT:
        np.testing.assert_array_equal(a.a[k], b.a[k], err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("when", TIMES[1::2])
def test_gpu_forcing_prep_bit_identical(engine, port, when):
    import torch
    iday, h, m, sec = when
    s, lon, rain = case(ni=512, nj=64, seed=7)
    a = s.copy()
    first = iday % 2 == 1
    ja = port.forcing_prep(a, lon, rain, iday, h, m, sec, scale_vegfra=True, first_step=first)
    d = s.to_device("cuda:0")
    jd = engine.forcing_prep(d, torch.from_numpy(lon).cuda(), torch.from_numpy(rain).cuda(), iday, h, m, sec, scale_vegfra=True,
                             first_step=first)
    assert ja == jd
    hst = d.to_host()
    for k in OUT:
        np.testing.assert_array_equal(a.a[k], hst.a[k], err_msg=k)
    with pytest.raises(RuntimeError):
        engine.lib.noahmp_hip_forcing_prep.restype = C.c_int
        rc = engine.lib.noahmp_hip_forcing_prep(C.byref(s.step_args(1, 2000, 1.0)), None, None, 0, 0, 0, 0, 30.0, 0, None,
                                                abi.MEM_HOST, None, None)
        assert rc == -104
        raise RuntimeError("host arrays are refused")


# ---- temporal interpolation between two forcing records (hrldas_input_interpolate, netcdf_io:1369-1403)
IOUT = ["t3d", "qv3d", "u_phy", "v_phy", "p8w3d", "glw", "swdown", "vegfra", "xlaixy"]
BRACKETS = [(0, 10800), (1800, 10800), (3600, 10800), (7200, 10800), (10800, 10800), (1, 3), (3599, 3600)]


def records(ni, nj, seed, with_veg=True):
    r = np.random.default_rng(seed)
    rng = dict(t=(250, 310), q=(1e-4, 2e-2), u=(-10, 10), v=(-10, 10), p=(6e4, 1.02e5), lw=(150, 450), sw=(0, 1000),
               pcp=(0, 2e-3), fpar=(0, 1), lai=(0, 6))
    out = []
    for _ in range(2):
        d = {k: r.uniform(lo, hi, size=(nj, ni)).astype(np.float32) for k, (lo, hi) in rng.items()}
        if not with_veg:
            d["fpar"] = d["lai"] = None
        out.append(d)
    return out


def icase(ni, nj, seed):
    s, _, _ = case(ni, nj, seed)
    for k in ("glw", "swdown", "xlaixy"):
        s[k] = -777.0
    return s


@pytest.mark.parametrize("idts,idts2", BRACKETS)
def test_interpolate_oracle_properties(port, idts, idts2):
    ni, nj = 96, 40
    ra, rb = records(ni, nj, 11)
    s = icase(ni, nj, 3)
    rain = np.full((nj, ni), -1.0, np.float32)
    port.forcing_interpolate(s, ra, rb, idts, idts2, rain)
    f = np.float32(idts2 - idts) / np.float32(idts2)
    g = np.float32(1.0) - f
    for k3, k in (("t3d", "t"), ("qv3d", "q"), ("u_phy", "u"), ("v_phy", "v"), ("p8w3d", "p")):
        np.testing.assert_array_equal(s.a[k3][:, 0, :], ra[k] * f + rb[k] * g, err_msg=k)     # numpy float32, no FMA
        assert (s.a[k3][:, 1, :] == -777.0).all()                                             # level 2 is forcing_prep's
    np.testing.assert_array_equal(s["glw"], ra["lw"] * f + rb["lw"] * g)
    np.testing.assert_array_equal(s["swdown"], ra["sw"] * f + rb["sw"] * g)
    np.testing.assert_array_equal(rain, ra["pcp"])                                            # not interpolated
    np.testing.assert_array_equal(s["vegfra"], ra["fpar"])
    np.testing.assert_array_equal(s["xlaixy"], ra["lai"])
    if idts == 0:
        np.testing.assert_array_equal(s.a["t3d"][:, 0, :], ra["t"])
    if idts == idts2:
        np.testing.assert_array_equal(s.a["t3d"][:, 0, :], rb["t"])


def test_interpolate_copy_and_missing_vegetation(port):
    ni, nj = 64, 16
    ra, _ = records(ni, nj, 12, with_veg=False)
    s = icase(ni, nj, 4)
    veg0, lai0 = s["vegfra"].copy(), s["xlaixy"].copy()
    rain = np.zeros((nj, ni), np.float32)
    port.forcing_interpolate(s, ra, None, 0, 0, rain)                  # hrldas_input_copy
    np.testing.assert_array_equal(s.a["qv3d"][:, 0, :], ra["q"])
    np.testing.assert_array_equal(s["swdown"], ra["sw"])
    np.testing.assert_array_equal(s["vegfra"], veg0)                   # variable absent: carried over
    np.testing.assert_array_equal(s["xlaixy"], lai0)


@pytest.mark.parametrize("idts,idts2", BRACKETS[1::2])
def test_interpolate_device_source_on_host_matches_oracle(port, idts, idts2):
    from host_emul.emullib import EmulLib
    em = EmulLib()
    ni, nj = 96, 40
    ra, rb = records(ni, nj, 13)
    a, b = icase(ni, nj, 5), icase(ni, nj, 5)
    raina, rainb = np.zeros((nj, ni), np.float32), np.zeros((nj, ni), np.float32)
    port.forcing_interpolate(a, ra, rb, idts, idts2, raina)
    em.forcing_interpolate(b, ra, rb, idts, idts2, rainb)
    for k in IOUT:
        np.testing.assert_array_equal(a.a[k], b.a[k], err_msg=k)
    np.testing.assert_array_equal(raina, rainb)


@pytest.mark.gpu
@pytest.mark.parametrize("idts,idts2", [(1800, 10800), (3599, 3600), (None, None)])
def test_gpu_forcing_interpolate_bit_identical(engine, port, idts, idts2):
    import torch
    ni, nj = 512, 64
    ra, rb = records(ni, nj, 14)
    if idts is None:
        rb, idts, idts2 = None, 0, 0
    a, s = icase(ni, nj, 6), icase(ni, nj, 6)
    rain = np.zeros((nj, ni), np.float32)
    port.forcing_interpolate(a, ra, rb, idts, idts2, rain)
    d = s.to_device("cuda:0")
    dev = lambda r: None if r is None else {k: torch.from_numpy(v).cuda() for k, v in r.items()}
    raind = torch.zeros((nj, ni), dtype=torch.float32, device="cuda:0")
    engine.forcing_interpolate(d, dev(ra), dev(rb), idts, idts2, raind)
    hst = d.to_host()
    for k in IOUT:
        np.testing.assert_array_equal(a.a[k], hst.a[k], err_msg=k)
    np.testing.assert_array_equal(rain, raind.cpu().numpy())
    with pytest.raises(RuntimeError):                                   # target outside the bracket is refused
        engine.forcing_interpolate(d, dev(ra), dev(ra), 7200, 3600, raind)


@pytest.mark.gpu
@pytest.mark.parametrize("idts,idts2,first", [(1800, 10800, False), (3599, 3600, True), (None, None, False)])
def test_gpu_forcing_interpolate_prep_is_the_two_calls(engine, port, idts, idts2, first):
    """noahmp_hip_forcing_interpolate_prep (one launch) = the oracle's hrldas_input_interpolate followed by its forcing preparation, bit for
    bit: every array either call writes, the rain-rate scratch plane and JULIAN."""
    import torch
    ni, nj = 512, 64
    ra, rb = records(ni, nj, 15)
    if idts is None:
        rb, idts, idts2 = None, 0, 0
    _, lon, _ = case(ni=ni, nj=nj, seed=9)
    a, s = icase(ni, nj, 8), icase(ni, nj, 8)
    rain = np.zeros((nj, ni), np.float32)
    iday, h, m, sec = 171, 14, 30, 0
    port.forcing_interpolate(a, ra, rb, idts, idts2, rain)
    ja = port.forcing_prep(a, lon, rain, iday, h, m, sec, scale_vegfra=True, first_step=first)
    d = s.to_device("cuda:0")
    dev = lambda r: None if r is None else {k: torch.from_numpy(v).cuda() for k, v in r.items()}
    raind = torch.zeros((nj, ni), dtype=torch.float32, device="cuda:0")
    jd = engine.forcing_interpolate_prep(d, dev(ra), dev(rb), idts, idts2, raind, torch.from_numpy(lon).cuda(), iday, h, m, sec,
                                         scale_vegfra=True, first_step=first)
    assert ja == jd
    hst = d.to_host()
    for k in sorted(set(IOUT) | set(OUT)):
        np.testing.assert_array_equal(a.a[k], hst.a[k], err_msg=k)
    np.testing.assert_array_equal(rain, raind.cpu().numpy())

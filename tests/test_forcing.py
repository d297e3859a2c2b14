"""Forcing preparation on the device: driver/module_hrldas_noahmp_driver.F90:336-354 + CALC_DECLIN (hdrv:813-863).

The oracle (oracle/nmp_forcing.c) is a restatement only: the driver file cannot be compiled here (NetCDF), so parity with a
reference build is UNPINNED for this entry; the restatement is cross-checked against an independent float64 evaluation."""
import ctypes as C

import numpy as np
import pytest

from noahmp_amd import abi
from noahmp_amd.state import ColumnStore, ModelConfig

OUT = ["t3d", "qv3d", "u_phy", "v_phy", "p8w3d", "rainbl", "vegfra", "dz8w", "coszin"]


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
    s["coszin"] = -777.0
    s["dz8w"] = -777.0
    s["rainbl"] = -777.0
    return s, lon, rain


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


@pytest.mark.parametrize("when", TIMES)
def test_device_source_on_host_matches_oracle(port, when):
    from host_emul.emullib import EmulLib
    em = EmulLib()
    lib = abi.load_library()                    # host-only entry of the product library (no GPU call)
    iday, h, m, sec = when
    s, lon, rain = case(seed=5)
    a, b = s.copy(), s.copy()
    port.forcing_prep(a, lon, rain, iday, h, m, sec, scale_vegfra=True)
    sd, cd = C.c_float(0), C.c_float(0)
    jul = lib.noahmp_hip_declination(iday, h, C.byref(sd), C.byref(cd))
    hour = np.float32(np.float32(np.float32(h) + np.float32(m) / np.float32(60.0)) + np.float32(sec) / np.float32(3600.0))
    em.forcing_prep(b, lon, rain, float(hour), sd.value, cd.value, scale_vegfra=True)
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
    ja = port.forcing_prep(a, lon, rain, iday, h, m, sec, scale_vegfra=True)
    d = s.to_device("cuda:0")
    jd = engine.forcing_prep(d, torch.from_numpy(lon).cuda(), torch.from_numpy(rain).cuda(), iday, h, m, sec, scale_vegfra=True)
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

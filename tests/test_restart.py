"""Restart files: the reference's variable list, dimensions and masking (hdrv:596-672, netcdf_io:1950-2523)
written in classic NetCDF, and checkpoint -> resume continuity of a run."""
import numpy as np
import pytest

from noahmp_amd import restart, synth
from noahmp_amd.state import ColumnStore, ModelConfig

STATIC = ["ivgtyp", "isltyp", "tmn", "xland", "xice", "xlatin", "dzs"]


def small(tables, nj=6, ni=40):
    s = synth.mixed_small(tables[1], ni=ni, nj=nj, seed=11)
    s["ivgtyp"][0, :5] = s.cfg.iswater
    s["xland"][0, :5] = 2.0
    synth.first_step_fixups(s)
    return s


def advance(port, s, first, last):
    for it in range(first, last + 1):
        synth.diurnal_forcing(s, (it - 1) % 24, t_offset=s.t_offset)
        st = port.noahmplsm(s, it, 2000, 180.0 + it / 24.0)
        assert st.code == 0


def test_file_layout_and_roundtrip(tables, port, tmp_path):
    from scipy.io import netcdf_file
    s = small(tables)
    advance(port, s, 1, 3)
    p = restart.write_restart(str(tmp_path / "restart.2000010103"), s, "2000-01-01_03:00:00", "2000-01-01_00:00:00")
    f = netcdf_file(p, "r", mmap=False)
    assert f.dimensions["Time"] is None and f.dimensions["DateStrLen"] == 19
    assert f.dimensions["west_east"] == s.ni and f.dimensions["south_north"] == s.nj
    assert f.dimensions["west_east_stag"] == s.ni + 1 and f.dimensions["sosn_layers"] == 3 + s.cfg.nsoil
    assert f.TITLE.startswith(b"RESTART FILE FROM HRLDAS") and f.START_DATE == b"2000-01-01_00:00:00"
    assert f.variables["Times"][0].tobytes() == b"2000-01-01_03:00:00"
    assert [n for n, _, _ in restart.RESTART_VARS] == [n for n in f.variables if n != "Times"]       # hdrv:610-671 order
    assert f.variables["SOIL_T"].dimensions == ("Time", "south_north", "soil_layers_stag", "west_east")
    assert f.variables["ZSNSO"].dimensions == ("Time", "south_north", "sosn_layers", "west_east")
    assert f.variables["SNICE"].dimensions == ("Time", "south_north", "snow_layers", "west_east")
    assert f.variables["TG"].dimensions == ("Time", "south_north", "west_east")
    assert f.variables["SOIL_T"].MemoryOrder == b"XZY" and f.variables["TG"].MemoryOrder == b"XY "
    assert f.variables["ISNOW"].data.dtype.kind == "i"
    water = s["ivgtyp"] == s.cfg.iswater
    soil_t, tg = np.array(f.variables["SOIL_T"][0]), np.array(f.variables["TG"][0])
    assert (soil_t[:, 0, :][water] == restart.MISSING).all()                 # put_var_3d masks (netcdf_io:2039-2042)
    np.testing.assert_array_equal(soil_t[:, 0, :][~water], s.a["tslb"][:, 0, :][~water])
    np.testing.assert_array_equal(tg, s["tgxy"])                            # put_var_2d with restart_flag does not
    assert (np.array(f.variables["GVFMIN"][0]) == restart.UNDEFINED).all()  # driver-only array, not in the step block
    f.close()
    b = ColumnStore(s.ni, s.nj, s.cfg)
    assert restart.read_restart(p, b) == "2000-01-01_03:00:00"
    for name, field, lay in restart.RESTART_VARS:
        if field in s.a and not lay:
            np.testing.assert_array_equal(b.a[field], s.a[field], err_msg=name)


def test_resume_continues_bit_identically(tables, port, tmp_path):
    """6 steps straight == 3 steps, restart file, fresh arrays + static inputs + file, 3 more steps (land columns)."""
    a = small(tables)
    advance(port, a, 1, 6)
    b = small(tables)
    advance(port, b, 1, 3)
    p = restart.write_restart(str(tmp_path / "restart.nc"), b, "2000-01-01_03:00:00")
    c = ColumnStore(b.ni, b.nj, b.cfg)
    for k, v in c.a.items():                                                  # hdrv:262-: everything else undefined
        if v.dtype.kind == "f":
            v[...] = restart.UNDEFINED
    for k in STATIC:
        c.a[k][...] = b.a[k]
    c.t_offset = b.t_offset
    restart.read_restart(p, c)
    advance(port, c, 4, 6)
    land = a["ivgtyp"] != a.cfg.iswater
    for name, field, lay in restart.RESTART_VARS:
        if field not in a.a:
            continue
        x, y = a.a[field], c.a[field]
        m = np.broadcast_to(land[:, None, :] if x.ndim == 3 else land, x.shape)
        assert np.array_equal(x[m], y[m], equal_nan=True), name
    for k in ("hfx", "lh", "grdflx", "tsk", "t2mvxy", "emiss"):
        assert np.array_equal(a.a[k][land], c.a[k][land], equal_nan=True), k
    # ALBEDO is the one output that is not a function of the restart list: at night noahmplsm leaves the previous value
    # (`IF ( SALB > -999 )`, phys/module_sf_noahmpdrv.F90:741) and ALBEDO is not in the reference's restart file, so a resumed run shows
    # undefined_real there until the first daylight step -- in the reference as here.
    assert (c.a["albedo"][land] == restart.UNDEFINED).any()


@pytest.mark.gpu
def test_gpu_fetch_from_sorted_layout_and_restart(tables, engine, port, tmp_path):
    s = small(tables, nj=16, ni=128)
    advance(port, s, 1, 2)
    d = s.to_device("cuda:0")
    perm = engine.sort_store(d)
    names = [f for _, f, _ in restart.RESTART_VARS if f in s.a]
    layered = [f for _, f, lay in restart.RESTART_VARS if lay]
    got = restart.fetch(engine, d, names, perm=perm, mask=layered)
    want = restart.fetch(None, s, names, mask=layered)
    for n in names:
        np.testing.assert_array_equal(got[n], want[n], err_msg=n)
    pd = restart.write_restart(str(tmp_path / "dev.nc"), d, "2000-01-01_02:00:00", engine=engine, perm=perm)
    ph = restart.write_restart(str(tmp_path / "host.nc"), s, "2000-01-01_02:00:00")
    assert open(pd, "rb").read() == open(ph, "rb").read()                    # same bytes from either residence
    rate = np.full((s.nj, s.ni), 5e-4, np.float32)
    od = restart.write_output(str(tmp_path / "odev.nc"), d, "2000-01-01_02:00:00", engine=engine, perm=perm, extra=dict(rainrate=rate))
    oh = restart.write_output(str(tmp_path / "ohost.nc"), s, "2000-01-01_02:00:00", extra=dict(rainrate=rate))
    assert open(od, "rb").read() == open(oh, "rb").read()


def test_output_file(tables, port, tmp_path):
    from scipy.io import netcdf_file
    s = small(tables)
    advance(port, s, 1, 2)
    rate = np.full((s.nj, s.ni), 5e-4, np.float32)
    p = restart.write_output(str(tmp_path / "output.nc"), s, "2000-01-01_02:00:00", extra=dict(rainrate=rate))
    f = netcdf_file(p, "r", mmap=False)
    assert f.TITLE.startswith(b"OUTPUT FROM HRLDAS") and "sosn_layers" not in f.dimensions
    assert [n for n, _, _, _ in restart.OUTPUT_VARS] == [n for n in f.variables if n != "Times"]
    water = s["ivgtyp"] == s.cfg.iswater
    for name, field in (("HFX", "hfx"), ("TG", "tgxy"), ("RAINRATE", None)):
        a = np.array(f.variables[name][0])
        assert (a[water] == restart.MISSING).all() and (a[~water] != restart.MISSING).all(), name     # netcdf_io:1971
        if field:
            np.testing.assert_array_equal(a[~water], s[field][~water])
    np.testing.assert_array_equal(np.array(f.variables["IVGTYP"][0]), s["ivgtyp"])                    # integers are not masked
    z = np.array(f.variables["ZSNSO_SN"][0])
    assert z.shape == (s.nj, 3, s.ni)
    np.testing.assert_array_equal(z[:, :, ~water[0]][0], s.a["zsnsoxy"][0, :3, :][:, ~water[0]])
    assert f.variables["HFX"].units == b"W m{-2}" and f.variables["SOIL_M"].dimensions[2] == "soil_layers_stag"
    f.close()

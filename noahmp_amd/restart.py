"""Restart / output files for a run that lives on the device.

What the reference writes: `hrldas_noahmp_vars_write_restart` (driver/module_hrldas_noahmp_driver.F90:596-672, "hdrv") and the
matching read list (hdrv:185-245), through driver/module_hrldas_netcdf_io.F90 ("netcdf_io") `hrldas_restart_prepare`
(2106-2230: dimensions, global attributes, `Times`), `hrldas_restart_add` (2286-2523: one variable, dimensions
(west_east, [layers,] south_north, Time) in Fortran order, attributes MemoryOrder / description / units / stagger) and
`put_var_2d` / `put_var_3d` (1950-2051: water points of LAYERED restart variables and of every real output variable
become -1.E33).

The reference creates its files as NetCDF-4 (HDF5).  Neither libnetcdf / libhdf5 nor h5py exist in this image, so this
module writes the same data model in the classic 64-bit-offset format (CDF-2) with scipy.io.netcdf_file: `nf90_open`
reads classic and NetCDF-4 files alike, so `hrldas_restart_read` / `hrldas_restart_get` accept these files unchanged.
Reading a restart file the FORTRAN driver wrote (HDF5) is not possible here; a Fortran caller reads it with its own
library and hands the arrays over as it always does.

Device side: the variables are brought to the host by `noahmp_hip_output_fields` (one gather per <= 32 fields: sorted ->
tile order and the water masking fused), so only the restart / output list crosses PCIe, not the whole state.
"""
import ctypes as C

import numpy as np

from . import abi
from .abi_spec import NSNOW

MISSING = np.float32(-1.0e33)            # netcdf_io:2189 global attribute and the mask value
UNDEFINED = np.float32(-1.0e20)          # undefined_real, driver/module_hrldas_noahmp_vars.F90:6

# (NetCDF name, step-block / groundwater field, layers) in the order of hdrv:610-671.  Fields the step block does not
# carry (the driver's GVFMIN / GVFMAX) are looked up in `extra` and written as undefined_real when absent.
RESTART_VARS = [
    ("SOIL_T", "tslb", "SOIL"), ("SNOW_T", "tsnoxy", "SNOW"), ("SMC", "smois", "SOIL"), ("SH2O", "sh2o", "SOIL"),
    ("ZSNSO", "zsnsoxy", "SOSN"), ("SNICE", "snicexy", "SNOW"), ("SNLIQ", "snliqxy", "SNOW"), ("QSNOW", "qsnowxy", None),
    ("FWET", "fwetxy", None), ("SNEQVO", "sneqvoxy", None), ("EAH", "eahxy", None), ("TAH", "tahxy", None),
    ("ALBOLD", "alboldxy", None), ("CM", "cmxy", None), ("CH", "chxy", None), ("ISNOW", "isnowxy", None),
    ("CANLIQ", "canliqxy", None), ("CANICE", "canicexy", None), ("SNEQV", "snow", None), ("SNOWH", "snowh", None),
    ("TV", "tvxy", None), ("TG", "tgxy", None), ("ZWT", "zwtxy", None), ("WA", "waxy", None), ("WT", "wtxy", None),
    ("WSLAKE", "wslakexy", None), ("LFMASS", "lfmassxy", None), ("RTMASS", "rtmassxy", None), ("STMASS", "stmassxy", None),
    ("WOOD", "woodxy", None), ("STBLCP", "stblcpxy", None), ("FASTCP", "fastcpxy", None), ("LAI", "xlaixy", None),
    ("SAI", "xsaixy", None), ("FPAR", "vegfra", None), ("GVFMIN", "gvfmin", None), ("GVFMAX", "gvfmax", None),
    ("SHDMAX", "vegmax", None), ("ACMELT", "acsnom", None), ("ACSNOW", "acsnow", None), ("TAUSS", "taussxy", None),
    ("QSFC", "qsfc", None), ("SFCRUNOFF", "sfcrunoff", None), ("UDRUNOFF", "udrunoff", None),
    # "below for opt_run = 5" (hdrv:654-670)
    ("SMOISEQ", "smoiseq", "SOIL"), ("AREAXY", "area", None), ("SMCWTDXY", "smcwtdxy", None),
    ("DEEPRECHXY", "deeprechxy", None), ("QSLATXY", "qslat", None), ("QRFSXY", "qrfs", None), ("QSPRINGSXY", "qsprings", None),
    ("RECHXY", "rechxy", None), ("QRFXY", "qrf", None), ("QSPRINGXY", "qspring", None), ("FDEPTHXY", "fdepth", None),
    ("RIVERCONDXY", "rivercond", None), ("RIVERBEDXY", "riverbed", None), ("EQZWT", "eqwtd", None), ("PEXPXY", "pexp", None),
]
LAYER_DIM = {"SOIL": "soil_layers_stag", "SNOW": "snow_layers", "SOSN": "sosn_layers"}

# Output (LDASOUT) list: (NetCDF name, field, layers, units) in the order of hrldas_noahmp_vars_write_output (hdrv:690-812).
# "rainrate" = RAINBL_tmp, the driver's precipitation rate (the rain_rate plane of noahmp_hip_forcing_prep), passed in
# `extra`; ZSNSO_SN is the snow part of ZSNSOXY (hdrv: ZSNSOXY(:,-nsnow+1:0,:)).  The free-text descriptions of the
# reference are not reproduced: `description` carries "-" like the reference's own restart variables.
OUTPUT_VARS = [
    ("IVGTYP", "ivgtyp", None, "category"), ("ISLTYP", "isltyp", None, "category"), ("FVEG", "fvegxy", None, "-"),
    ("LAI", "xlaixy", None, "-"), ("SAI", "xsaixy", None, "-"), ("SWFORC", "swdown", None, "W m{-2}"),
    ("COSZ", "coszin", None, "W m{-2}"), ("LWFORC", "glw", None, "W m{-2}"), ("RAINRATE", "rainrate", None, "kg m{-2} s{-1}"),
    ("EMISS", "emiss", None, ""), ("FSA", "fsaxy", None, "W m{-2}"), ("FIRA", "firaxy", None, "W m{-2}"),
    ("GRDFLX", "grdflx", None, "W m{-2}"), ("HFX", "hfx", None, "W m{-2}"), ("LH", "lh", None, "W m{-2}"),
    ("ECAN", "ecanxy", None, "kg m{-2} s{-1}"), ("ETRAN", "etranxy", None, "kg m{-2} s{-1}"),
    ("EDIR", "edirxy", None, "kg m{-2} s{-1}"), ("ALBEDO", "albedo", None, "-"), ("UGDRNOFF", "udrunoff", None, "mm"),
    ("SFCRNOFF", "sfcrunoff", None, "mm"), ("CANLIQ", "canliqxy", None, "mm"), ("CANICE", "canicexy", None, "mm"),
    ("ZWT", "zwtxy", None, "m"), ("WA", "waxy", None, "kg m{-2}"), ("WT", "wtxy", None, "kg m{-2}"),
    ("SAV", "savxy", None, "W m{-2}"), ("TR", "trxy", None, "W m{-2}"), ("EVC", "evcxy", None, "W m{-2}"),
    ("IRC", "ircxy", None, "W m{-2}"), ("SHC", "shcxy", None, "W m{-2}"), ("IRG", "irgxy", None, "W m{-2}"),
    ("SHG", "shgxy", None, "W m{-2}"), ("EVG", "evgxy", None, "W m{-2}"), ("GHV", "ghvxy", None, "W m{-2}"),
    ("SAG", "sagxy", None, "W m{-2}"), ("IRB", "irbxy", None, "W m{-2}"), ("SHB", "shbxy", None, "W m{-2}"),
    ("EVB", "evbxy", None, "W m{-2}"), ("GHB", "ghbxy", None, "W m{-2}"), ("TRAD", "tradxy", None, "K"),
    ("TG", "tgxy", None, "K"), ("TV", "tvxy", None, "K"), ("TAH", "tahxy", None, "K"), ("TGV", "tgvxy", None, "K"),
    ("TGB", "tgbxy", None, "K"), ("T2MV", "t2mvxy", None, "K"), ("T2MB", "t2mbxy", None, "K"),
    ("Q2MV", "q2mvxy", None, "kg/kg"), ("Q2MB", "q2mbxy", None, "kg/kg"), ("EAH", "eahxy", None, "Pa"),
    ("FWET", "fwetxy", None, "fraction"), ("ZSNSO_SN", "zsnsoxy", "SNOW", "m"), ("SNICE", "snicexy", "SNOW", "mm"),
    ("SNLIQ", "snliqxy", "SNOW", "mm"), ("SOIL_T", "tslb", "SOIL", "K"), ("SOIL_M", "smois", "SOIL", "m{3} m{-3}"),
    ("SOIL_W", "sh2o", "SOIL", "m3 m-3"), ("SNOW_T", "tsnoxy", "SNOW", "K"), ("SNOWH", "snowh", None, "m"),
    ("SNEQV", "snow", None, "kg m{-2}"), ("QSNOW", "qsnowxy", None, "mm s{-1}"), ("ISNOW", "isnowxy", None, "count"),
    ("FSNO", "snowc", None, ""), ("ACSNOW", "acsnow", None, "mm"), ("ACSNOM", "acsnom", None, "mm"), ("CM", "cmxy", None, ""),
    ("CH", "chxy", None, ""), ("CHV", "chvxy", None, "m s{-1}"), ("CHB", "chbxy", None, "m s{-1}"),
    ("CHLEAF", "chleafxy", None, "m s{-1}"), ("CHUC", "chucxy", None, "m s{-1}"), ("CHV2", "chv2xy", None, "m s{-1}"),
    ("CHB2", "chb2xy", None, "m s{-1}"), ("LFMASS", "lfmassxy", None, "g m{-2}"), ("RTMASS", "rtmassxy", None, "g m{-2}"),
    ("STMASS", "stmassxy", None, "g m{-2}"), ("WOOD", "woodxy", None, "g m{-2}"), ("STBLCP", "stblcpxy", None, "g m{-2}"),
    ("FASTCP", "fastcpxy", None, "g m{-2}"), ("NEE", "neexy", None, "g m{-2} s{-1} CO2"), ("GPP", "gppxy", None, "g m{-2} s{-1} C"),
    ("NPP", "nppxy", None, "g m{-2} s{-1} C"), ("PSN", "psnxy", None, "umol CO@ m{-2} s{-1}"), ("APAR", "aparxy", None, "W m{-2}"),
    ("SMCWTD", "smcwtdxy", None, "g m{-2}"), ("RECH", "rechxy", None, "g m{-2}"), ("QRFS", "qrfs", None, "g m{-2}"),
    ("QSPRINGS", "qsprings", None, "g m{-2}"), ("QSLAT", "qslat", None, "g m{-2}"),
]


def fetch(engine, store, names, perm=None, mask=()):
    """{name: host array in TILE order} for fields of a DeviceColumnStore (or a host store: plain copies + numpy mask).

    `perm` = the permutation sort_store returned (sorted position p holds tile column perm[p]) or None; `mask` = names that
    get -1.E33 on water points.  Device stores go through noahmp_hip_output_fields, 32 fields per launch."""
    names = [n for n in names if n in store.a]
    if getattr(store, "device", None) is None:
        assert perm is None
        out = {n: store.a[n].copy() for n in names}
        water = store.a["ivgtyp"] == store.cfg.iswater
        for n in mask:
            if n in out and out[n].dtype.kind == "f":
                out[n][np.broadcast_to(water[:, None, :] if out[n].ndim == 3 else water, out[n].shape)] = MISSING
        return out
    import torch
    out = {}
    inv = None
    if perm is not None:
        inv = torch.empty_like(perm)
        inv[perm.long()] = torch.arange(perm.numel(), dtype=perm.dtype, device=perm.device)
        torch.cuda.current_stream().synchronize()      # `inv` is written on torch's stream and read on the engine's own stream
    veg = store.a["ivgtyp"]
    for i in range(0, len(names), 32):
        chunk = names[i:i + 32]
        src = [store.a[n] for n in chunk]
        dst = [torch.empty_like(t) for t in src]
        n = len(chunk)
        bits = 0
        for f, nm in enumerate(chunk):
            if nm in mask and src[f].dtype == torch.float32:
                bits |= 1 << f
        rc = engine.lib.noahmp_hip_output_fields(
            n, (C.c_void_p * n)(*[t.data_ptr() for t in dst]), (C.c_void_p * n)(*[t.data_ptr() for t in src]),
            (C.c_int * n)(*[(t.shape[1] if t.dim() == 3 else 1) for t in src]), inv.data_ptr() if inv is not None else None,
            veg.data_ptr(), store.cfg.iswater, bits, store.ni, store.nj, None)
        if rc:
            raise RuntimeError("noahmp_hip_output_fields: rc=%d %s" % (rc, engine.lib.noahmp_hip_last_error().decode()))
        torch.cuda.synchronize()
        for nm, t in zip(chunk, dst):
            out[nm] = t.cpu().numpy()
    return out


def _date19(s):
    d = bytearray(b"0000-00-00_00:00:00")               # netcdf_io:2191-2192
    b = s.encode()[:19]
    d[:len(b)] = b
    return bytes(d)


def write_restart(path, store, olddate, startdate=None, engine=None, perm=None, extra=None, **attrs):
    """restart.<date> with the reference's dimensions, attributes and variable list; returns the path."""
    layered = [f for _, f, lay in RESTART_VARS if lay]
    vals = fetch(engine, store, [f for _, f, _ in RESTART_VARS], perm=perm, mask=layered)      # put_var_3d masks, put_var_2d not
    return _write(path, store, [(n, f, lay, "-") for n, f, lay in RESTART_VARS], vals, extra, olddate, startdate or olddate,
                  "RESTART FILE FROM HRLDAS ", restart=True, **attrs)


def write_output(path, store, date, startdate=None, engine=None, perm=None, extra=None, **attrs):
    """output.<date> / *.LDASOUT_DOMAIN* record (hrldas_output_prepare + the hrldas_output_add list, hdrv:690-812):
    every real variable masked to -1.E33 on water points (put_var_2d with restart_flag false, put_var_3d)."""
    names = [f for _, f, _, _ in OUTPUT_VARS]
    vals = fetch(engine, store, names, perm=perm, mask=names)
    if "zsnsoxy" in vals:
        vals = dict(vals, zsnsoxy=vals["zsnsoxy"][:, :NSNOW, :])
    return _write(path, store, OUTPUT_VARS, vals, extra, date, startdate or date, "OUTPUT FROM HRLDAS ", restart=False, **attrs)


def _write(path, store, varlist, vals, extra, date, startdate, title, restart, version="v20150506", llanduse="USGS",
           dx=1000.0, dy=1000.0, mapproj=0, truelat1=0.0, truelat2=0.0, cen_lon=0.0):
    from scipy.io import netcdf_file
    extra = extra or {}
    nsoil = store.cfg.nsoil
    olddate = date
    f = netcdf_file(path, "w", version=2)
    try:
        f.createDimension("Time", None)
        f.createDimension("DateStrLen", 19)
        f.createDimension("west_east", store.ni)
        f.createDimension("south_north", store.nj)
        f.createDimension("west_east_stag", store.ni + 1)
        f.createDimension("south_north_stag", store.nj + 1)
        f.createDimension("soil_layers_stag", nsoil)
        f.createDimension("snow_layers", NSNOW)
        if restart:
            f.createDimension("sosn_layers", NSNOW + nsoil)                  # netcdf_io:2186; the output file has none
        f.TITLE = title.encode() + version.encode()
        f.missing_value = MISSING
        f.START_DATE = _date19(startdate)
        f.MAP_PROJ = np.int32(mapproj)
        f.DX, f.DY = np.float32(dx), np.float32(dy)
        f.TRUELAT1, f.TRUELAT2, f.STAND_LON = np.float32(truelat1), np.float32(truelat2), np.float32(cen_lon)
        f.MMINLU = llanduse.encode()
        tv = f.createVariable("Times", "c", ("Time", "DateStrLen"))
        tv[0] = np.frombuffer(_date19(olddate), dtype="S1")
        for name, field, lay, units in varlist:
            a = vals.get(field, extra.get(field))
            if a is None:
                a = np.full((store.nj, store.ni), UNDEFINED, np.float32)
            a = np.asarray(a)
            if field not in vals and not restart and a.dtype.kind == "f" and "ivgtyp" in vals:   # an `extra` plane (tile order)
                a = np.where(vals["ivgtyp"] == store.cfg.iswater, MISSING, a).astype(np.float32)
            dims = ("Time", "south_north", LAYER_DIM[lay], "west_east") if lay else ("Time", "south_north", "west_east")
            v = f.createVariable(name, "i" if a.dtype.kind == "i" else "f", dims)
            v.MemoryOrder = b"XZY" if lay else b"XY "
            v.description = b"-"
            v.units = units.encode()
            v.stagger = b"Z" if lay else b"-"
            v[0] = a.astype(np.int32 if a.dtype.kind == "i" else np.float32)
    finally:
        f.close()
    return path


def read_restart(path, store, extra=None):
    """hrldas_restart_read + the hrldas_restart_get list (hdrv:185-245) into a HOST store; returns OLDDATE.

    Every variable of the list must be present (the reference stops otherwise, netcdf_io:2695) unless the store has no
    plane for it (groundwater planes of a run without OPT_RUN = 5, the driver-only GVFMIN / GVFMAX -> `extra`)."""
    from scipy.io import netcdf_file
    f = netcdf_file(path, "r", mmap=False)
    try:
        olddate = bytes(f.variables["Times"][0].tobytes()).decode()
        for name, field, lay in RESTART_VARS:
            if field not in store.a and extra is None:
                continue
            if name not in f.variables:
                raise KeyError("ERROR[HRLDAS_RESTART_GET]: Problems finding %s in file %s" % (name, path))
            a = np.array(f.variables[name][0])
            if field in store.a:
                store.a[field][...] = a
            elif extra is not None:
                extra[field] = a
    finally:
        f.close()
    return olddate

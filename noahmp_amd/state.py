"""Column store: every array of the noahmplsm interface, in the caller's Fortran layout.

A field declared ``(ims:ime, k0:k1, jms:jme)`` in the reference (drv:51-205) is held here as a
C-ordered array of shape ``(nj, nk, ni)`` -- byte-for-byte the same memory image, i fastest.
That layout is already structure-of-arrays with a coalescable i-run per (k, j), so the HIP
kernels read it directly; no transposition happens at the boundary.

``ColumnStore`` keeps numpy arrays (host);  ``to_device`` returns a twin backed by torch
tensors in HBM whose ``data_ptr()`` are handed to the C-ABI with ``NOAHMP_MEM_DEVICE``.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from .abi import StepArgs, WtableArgs, FIELD_INFO, ARRAY_FIELDS, nlev
from .abi_spec import STEP_FIELDS, WTABLE_FIELDS


@dataclass
class ModelConfig:
    """Scalars that are uniform over the grid (drv:51-83). Defaults = run/noahmp.namelist:20-31."""
    dt: float = 3600.0
    dx: float = 1000.0
    dzs: tuple = (0.1, 0.3, 0.6, 1.0)
    nsoil: int = 4
    xice_thres: float = 0.5
    isice: int = 24          # USGS (hdrv:140-143)
    isurban: int = 1
    iswater: int = 16
    idveg: int = 3
    iopt_crs: int = 1
    iopt_btr: int = 1
    iopt_run: int = 1
    iopt_sfc: int = 1
    iopt_frz: int = 1
    iopt_inf: int = 1
    iopt_rad: int = 3
    iopt_alb: int = 2
    iopt_snf: int = 1
    iopt_tbot: int = 2
    iopt_stc: int = 1
    iz0tlnd: int = 0
    zlvl: float = 30.0       # namelist.F90:52 (always 30: SURVEY section 5 config bug)
    wtddt: float = 30.0      # minutes between WTABLE_mmf_noahmp calls (hdrv:1227 STEPWTD = nint(WTDDT*60/DTBL))

    def options(self):
        return dict(idveg=self.idveg, iopt_crs=self.iopt_crs, iopt_btr=self.iopt_btr,
                    iopt_run=self.iopt_run, iopt_sfc=self.iopt_sfc, iopt_frz=self.iopt_frz,
                    iopt_inf=self.iopt_inf, iopt_rad=self.iopt_rad, iopt_alb=self.iopt_alb,
                    iopt_snf=self.iopt_snf, iopt_tbot=self.iopt_tbot, iopt_stc=self.iopt_stc)


def field_shape(name, ni, nj, nsoil):
    kind, lev, io = FIELD_INFO[name]
    if lev == "vec":
        return (nsoil,)
    if lev is None:
        return (nj, ni)
    return (nj, nlev(lev, nsoil), ni)


# WTABLE_mmf_noahmp dummy -> store array, as the reference driver wires them (hdrv:420-436).
GW_ALIAS = dict(smois="smois", sh2oxy="sh2o", smcwtd="smcwtdxy", wtd="zwtxy", deeprech="deeprechxy",
                rech="rechxy", smoiseq="smoiseq", isltyp="isltyp", ivgtyp="ivgtyp", xland="xland",
                xice="xice", dzs="dzs")
# MMF-only planes (hdrv:240-252 FDEPTHXY, AREAXY, TERRAIN, RIVERCONDXY, RIVERBEDXY, EQZWT, PEXPXY, Q*XY)
GW_EXTRA = ("fdepth", "area", "topo", "rivercond", "riverbed", "eqwtd", "pexp", "qrf", "qspring", "qslat",
            "qrfs", "qsprings")


def field_dtype(name):
    return np.int32 if FIELD_INFO[name][0] == "pi" else np.float32


class ColumnStore:
    """Host-side owner of all noahmplsm arrays for an ni x nj tile (memory == tile, hdrv:112-129)."""

    def __init__(self, ni, nj, cfg=None, fill=0.0):
        self.ni, self.nj = int(ni), int(nj)
        self.cfg = cfg or ModelConfig()
        ns = self.cfg.nsoil
        self.a = {}
        for n in ARRAY_FIELDS:
            self.a[n] = np.full(field_shape(n, ni, nj, ns), fill, dtype=field_dtype(n))
        self.a["dzs"][:] = np.asarray(self.cfg.dzs, dtype=np.float32)
        self.device = None
        self.idx = None

    # ------------------------------------------------------------------ index block
    def set_index(self, **kw):
        """Override the WRF index block (ids..kte).  Default: domain == memory == tile (hdrv:112-129).
        A rank of a decomposed domain keeps a halo: ims = its-1 etc., ids..ide = the global domain."""
        d = self.index()
        d.update(kw)
        assert d["ime"] - d["ims"] + 1 == self.ni and d["jme"] - d["jms"] + 1 == self.nj, "memory dims are fixed"
        self.idx = d
        return self

    def index(self):
        if getattr(self, "idx", None):
            return dict(self.idx)
        return dict(ids=1, ide=self.ni, jds=1, jde=self.nj, kds=1, kde=2,
                    ims=1, ime=self.ni, jms=1, jme=self.nj, kms=1, kme=2,
                    its=1, ite=self.ni, jts=1, jte=self.nj, kts=1, kte=1)

    def add_groundwater(self):
        """Allocate the planes only WTABLE_mmf_noahmp uses (OPT_RUN = 5)."""
        for n in GW_EXTRA:
            if n not in self.a:
                self.a[n] = np.zeros((self.nj, self.ni), dtype=np.float32)
        return self

    def wtable_args(self):
        """Pack a noahmp_wtable_args block (field order = WTABLE_mmf_noahmp dummy order, gw:14-22)."""
        cfg = self.cfg
        w = WtableArgs()
        scal = dict(nsoil=cfg.nsoil, xice_threshold=cfg.xice_thres, isice=cfg.isice, wtddt=cfg.wtddt,
                    isurban=cfg.isurban)
        scal.update(self.index())
        for n, k, lev, io, ln in WTABLE_FIELDS:
            if k in ("pf", "pi"):
                setattr(w, n, self.ptr(GW_ALIAS.get(n, n)))
            else:
                setattr(w, n, scal[n])
        return w

    def __getitem__(self, k):
        return self.a[k]

    def __setitem__(self, k, v):
        self.a[k][...] = v

    @property
    def ncol(self):
        return self.ni * self.nj

    def copy(self):
        o = ColumnStore.__new__(ColumnStore)
        o.ni, o.nj, o.cfg, o.device = self.ni, self.nj, self.cfg, None
        o.idx = dict(self.idx) if self.idx else None
        o.a = {k: np.array(v, copy=True) for k, v in self.a.items()}
        return o

    def ptr(self, name):
        return self.a[name].ctypes.data

    def step_args(self, itimestep, yr, julian):
        """Pack a noahmp_step_args block (field order = noahmplsm dummy order, drv:11-44)."""
        cfg = self.cfg
        s = StepArgs()
        scal = dict(itimestep=itimestep, yr=yr, julian=julian, dt=cfg.dt, nsoil=cfg.nsoil, dx=cfg.dx,
                    xice_thres=cfg.xice_thres, isice=cfg.isice, isurban=cfg.isurban,
                    iz0tlnd=cfg.iz0tlnd, **cfg.options())
        scal.update(self.index())
        for n, k, lev, io, ln in STEP_FIELDS:
            if k in ("pf", "pi"):
                setattr(s, n, self.ptr(n))
            else:
                setattr(s, n, scal[n])
        s._ranges = getattr(self, "class_ranges", None)       # (n_land, n_glacier) of a class-sorted device store, or None
        return s

    # ------------------------------------------------------------------ device twin
    def to_device(self, device="cuda:0"):
        import torch
        d = DeviceColumnStore.__new__(DeviceColumnStore)
        d.ni, d.nj, d.cfg, d.device = self.ni, self.nj, self.cfg, torch.device(device)
        d.idx = dict(self.idx) if self.idx else None
        d.a = {k: torch.from_numpy(np.ascontiguousarray(v)).to(d.device) for k, v in self.a.items()}
        d.a["dzs"] = np.array(self.a["dzs"], copy=True)   # config vector: always host memory at the ABI
        return d


class DeviceColumnStore(ColumnStore):
    """Same fields, resident in HBM as torch tensors (torch is only the allocator here)."""

    def ptr(self, name):
        v = self.a[name]
        return v.ctypes.data if isinstance(v, np.ndarray) else v.data_ptr()

    def to_host(self):
        h = ColumnStore.__new__(ColumnStore)
        h.ni, h.nj, h.cfg, h.device = self.ni, self.nj, self.cfg, None
        h.idx = dict(self.idx) if self.idx else None
        h.a = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v.detach().cpu().numpy())
               for k, v in self.a.items()}
        return h

    def copy(self):
        o = DeviceColumnStore.__new__(DeviceColumnStore)
        o.ni, o.nj, o.cfg, o.device = self.ni, self.nj, self.cfg, self.device
        o.idx = dict(self.idx) if self.idx else None
        o.a = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v.clone())
               for k, v in self.a.items()}
        if hasattr(self, "class_ranges"):
            o.class_ranges = self.class_ranges
        return o

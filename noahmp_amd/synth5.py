"""a global lat/lon grid (0.1 degree = 3600 x 1800 at full size) with a land mask, glaciers in the
polar caps, 1 % urban, latitude-dependent climate, run from a cold start for 30 days x 24 steps with forcing records
every 3 hours that are interpolated to the model step and a solar zenith angle from CALC_DECLIN.

This is a WORKLOAD GENERATOR (seeded, synthetic): static fields come from numpy (Philox), the 3-hourly forcing records
are evaluated on the device with float32 torch elementwise ops.  The product path it drives is
noahmp_hip_init -> [noahmp_hip_forcing_interpolate -> noahmp_hip_forcing_prep -> noahmp_hip_step_async] x nsteps.
"""
import math

import numpy as np

from .state import ColumnStore, ModelConfig
from .synth import CONUS_VEG, _base_store, _rng

F = np.float32
TROPIC_VEG = np.array([2, 7, 10, 13], dtype=np.int32)          # cropland, grassland, savanna, evergreen broadleaf
BOREAL_VEG = np.array([7, 14, 15, 21], dtype=np.int32)         # grassland, evergreen needleleaf, mixed forest, wooded tundra
POLAR_VEG = np.array([20, 22, 23], dtype=np.int32)             # herbaceous / mixed / bare-ground tundra
DAY0 = 171                                                     # 20 June: days since 1 January (GETH_IDTS convention)
RECORD_HOURS = 3


def config5_raw(ni=3600, nj=1800, seed=5, cfg=None, water_frac=0.03, urban_frac=0.01, polar_glacier=0.30):
    """The state handed to NOAHMP_INIT (nothing the cold start computes is set) + lon + the static forcing factors."""
    return _config5_rows(ni, nj, 0, nj, seed, cfg, water_frac, urban_frac, polar_glacier)


ROW_BLOCK = 60


def config5_tile(gx, gy, x0=0, y0=0, nx=None, ny=None, seed=5, cfg=None, **kw):
    """Cells [x0, x0+nx) x [y0, y0+ny) (0-based) of ONE global gx x gy config-5 grid: the grid is generated in blocks of ROW_BLOCK
    full rows, block b from the Philox key (seed, b), so that a tile is the same cells whatever the decomposition
    (mpp_land_partition_calc, mpp:227-288) -- what N ranks cut is what one rank holds.  -> (raw ColumnStore, lon, static)."""
    nx = gx - x0 if nx is None else nx
    ny = gy - y0 if ny is None else ny
    assert 0 <= x0 and x0 + nx <= gx and 0 <= y0 and y0 + ny <= gy
    cfg = cfg or ModelConfig(idveg=1)
    out = _base_store(nx, ny, cfg)
    lon = np.zeros((ny, nx), dtype=F)
    static = None
    for b in range(y0 // ROW_BLOCK, (y0 + ny - 1) // ROW_BLOCK + 1):
        r0 = b * ROW_BLOCK
        rows = min(ROW_BLOCK, gy - r0)
        blk, blon, bst = _config5_rows(gx, gy, r0, rows, [seed, b], cfg, **kw)
        lo, hi = max(y0, r0), min(y0 + ny, r0 + rows)
        for k, v in blk.a.items():
            if k != "dzs":
                out.a[k][lo - y0:hi - y0] = v[lo - r0:hi - r0, ..., x0:x0 + nx]
        lon[lo - y0:hi - y0] = blon[lo - r0:hi - r0, x0:x0 + nx]
        if static is None:
            static = {k: np.zeros((ny, nx), dtype=F) for k in bst}
        for k, v in bst.items():
            static[k][lo - y0:hi - y0] = v[lo - r0:hi - r0, x0:x0 + nx]
    return out, lon, static


def smooth_uniform(ni, gy, r0, nj, seed, nmodes=24):
    """A spatially smooth random field with U(0, 1) marginals on rows r0 .. r0+nj-1 of the global ni x gy lat/lon grid, as a closed form
    of the GLOBAL cell indices (so a tile is the same cells whatever the decomposition): a sum of `nmodes` plane waves with random
    directions, phases and wavelengths of 1 000 .. 10 000 km at the equator (zonal wavenumbers 4 .. 40: periodic in longitude;
    correlation length a few hundred km -- synoptic scale, what reanalysis forcing interpolated to 0.1 degree looks like,
    hdrv:332-366, netcdf_io:1369-1404), mapped through the normal CDF."""
    r = _rng(seed)
    m = r.integers(4, 41, size=nmodes).astype(np.float64) * r.choice([-1.0, 1.0], size=nmodes)      # zonal wavenumber (cycles per 360 degrees)
    n = r.uniform(-20.0, 20.0, size=nmodes)                                                         # meridional cycles per 180 degrees
    ph = r.uniform(0.0, 2.0 * math.pi, size=nmodes)
    x = (np.arange(ni, dtype=np.float64) + 0.5) * (2.0 * math.pi / ni)
    y = (np.arange(r0, r0 + nj, dtype=np.float64) + 0.5) * (math.pi / gy)
    f = np.zeros((nj, ni))
    for k in range(nmodes):
        f += np.cos(m[k] * x[None, :] + 2.0 * n[k] * y[:, None] + ph[k])
    f *= math.sqrt(2.0 / nmodes)                                                                    # unit variance
    from scipy.special import ndtr
    return ndtr(f)


def smooth_normal(ni, gy, r0, nj, seed, nmodes=24):
    """The same field before the CDF: N(0, 1) marginals."""
    from scipy.special import ndtri
    return ndtri(np.clip(smooth_uniform(ni, gy, r0, nj, seed, nmodes), 1e-9, 1.0 - 1e-9))


def _config5_rows(ni, gy, r0, nj, seed, cfg, water_frac=0.03, urban_frac=0.01, polar_glacier=0.30, smooth=False):
    """Rows r0 .. r0+nj-1 (all ni columns) of the global ni x gy grid.  smooth: the forcing factors (sky transmissivity, relative humidity,
    surface pressure / elevation, wind, rain timing) are spatially smooth fields (smooth_uniform) with the marginal distributions of the
    default's i.i.d. draws -- round 6's second generator, to tell what of config 5's lane utilisation is the model and what is white noise
    in the forcing; everything else (vegetation, soil, snow, temperatures) is the same cells.  smooth = 2: the STATE's per-cell noise is
    smooth too -- soil category, the 3-K temperature scatter, vegetation fraction, soil moisture, snow water equivalent and density come from
    smooth fields with the default's marginals (vegetation category, water / urban / land-ice masks stay as they are: the column order
    groups by them anyway) -- the upper bound of what spatial coherence of the inputs can buy."""
    cfg = cfg or ModelConfig(idveg=1)
    r = _rng(seed)
    s = _base_store(ni, nj, cfg)
    a = s.a
    shp = (nj, ni)
    lat1 = (np.arange(r0, r0 + nj, dtype=np.float64) + 0.5) * (180.0 / gy) - 90.0
    lon1 = (np.arange(ni, dtype=np.float64) + 0.5) * (360.0 / ni) - 180.0
    lat = np.broadcast_to(lat1[:, None], shp).astype(F)
    lon = np.broadcast_to(lon1[None, :], shp).astype(F).copy()
    a["xlatin"][...] = lat
    alat = np.abs(lat)
    u = r.random(size=shp)
    pick = lambda pool: pool[r.integers(0, len(pool), size=shp)]
    veg = np.where(alat < 23.0, pick(TROPIC_VEG), np.where(alat < 50.0, pick(CONUS_VEG),
                                                            np.where(alat < 66.0, pick(BOREAL_VEG), pick(POLAR_VEG))))
    a["ivgtyp"][...] = veg.astype(np.int32)
    a["isltyp"][...] = r.integers(1, 13, size=shp).astype(np.int32)
    a["ivgtyp"][u < urban_frac] = cfg.isurban
    glacier = (alat > 66.0) & (r.random(size=shp) < polar_glacier)
    a["ivgtyp"][glacier] = cfg.isice
    a["isltyp"][glacier] = 16
    water = (u >= urban_frac) & (u < urban_frac + water_frac)
    a["ivgtyp"][water] = cfg.iswater
    a["isltyp"][water] = 14
    a["xland"][water] = 2.0
    a["vegfra"][...] = r.uniform(20.0, 90.0, size=shp).astype(F)
    a["vegmax"][...] = np.maximum(a["vegfra"], F(90.0))
    sin2 = np.sin(np.deg2rad(lat.astype(np.float64))) ** 2
    tbase = (300.0 - 50.0 * sin2 + np.clip(r.normal(0.0, 3.0, size=shp), -8.0, 8.0)).astype(F)
    tbase[glacier] = np.minimum(tbase[glacier], F(262.0))
    a["tmn"][...] = (tbase - F(1.0)).astype(F)
    a["tsk"][...] = tbase
    cold = (tbase < F(272.0)) | glacier
    swe = r.uniform(5.0, 300.0, size=shp).astype(F)
    rho = r.uniform(100.0, 350.0, size=shp).astype(F)
    a["snow"][...] = np.where(cold, swe, F(0.0))
    a["snowh"][...] = np.where(cold, swe / rho, F(0.0))
    for k, (dt_, sm) in enumerate(zip((0.0, 0.3, 0.6, 0.9), (0.25, 0.27, 0.30, 0.31))):
        a["tslb"][:, k, :] = a["tsk"] * F(0.5) + a["tmn"] * F(0.5) + F(dt_)
        a["smois"][:, k, :] = F(sm) + r.uniform(-0.05, 0.05, size=shp).astype(F)
    static = dict(
        tbase=tbase,
        cloud=r.uniform(0.4, 0.9, size=shp).astype(F),             # transmissivity of the column's sky
        rh=r.uniform(0.4, 0.9, size=shp).astype(F),
        psfc=(101325.0 * np.exp(-r.uniform(0.0, 2500.0, size=shp) / 8000.0)).astype(F),
        uwind=r.uniform(1.0, 8.0, size=shp).astype(F),
        vwind=r.uniform(-3.0, 3.0, size=shp).astype(F),
        phase=r.integers(0, 16, size=shp).astype(F),               # when this column's rain events come
    )
    if smooth and int(smooth) >= 2:
        base = int(seed[0]) if isinstance(seed, (list, tuple)) else int(seed)
        su = lambda i: smooth_uniform(ni, gy, r0, nj, [base, 950 + i])
        land = ~(water | glacier)
        a["isltyp"][land] = np.minimum(np.floor(12.0 * su(0)) + 1, 12).astype(np.int32)[land]
        a["vegfra"][...] = (20.0 + 70.0 * su(1)).astype(F)
        a["vegmax"][...] = np.maximum(a["vegfra"], F(90.0))
        tb2 = (300.0 - 50.0 * sin2 + np.clip(3.0 * smooth_normal(ni, gy, r0, nj, [base, 952]), -8.0, 8.0)).astype(F)
        tb2[glacier] = np.minimum(tb2[glacier], F(262.0))
        tbase = tb2
        a["tmn"][...] = (tbase - F(1.0)).astype(F)
        a["tsk"][...] = tbase
        cold = (tbase < F(272.0)) | glacier
        swe = (5.0 + 295.0 * su(3)).astype(F)
        rho = (100.0 + 250.0 * su(4)).astype(F)
        a["snow"][...] = np.where(cold, swe, F(0.0))
        a["snowh"][...] = np.where(cold, swe / rho, F(0.0))
        dsm = (-0.05 + 0.10 * su(5)).astype(F)
        for k, (dt_, sm) in enumerate(zip((0.0, 0.3, 0.6, 0.9), (0.25, 0.27, 0.30, 0.31))):
            a["tslb"][:, k, :] = a["tsk"] * F(0.5) + a["tmn"] * F(0.5) + F(dt_)
            a["smois"][:, k, :] = F(sm) + dsm
        static["tbase"] = tbase
    if smooth:
        base = int(seed[0]) if isinstance(seed, (list, tuple)) else int(seed)
        u = {k: smooth_uniform(ni, gy, r0, nj, [base, 900 + i]) for i, k in enumerate(("cloud", "rh", "psfc", "uwind", "vwind", "phase"))}
        static.update(
            cloud=(0.4 + 0.5 * u["cloud"]).astype(F), rh=(0.4 + 0.5 * u["rh"]).astype(F),
            psfc=(101325.0 * np.exp(-(2500.0 * u["psfc"]) / 8000.0)).astype(F),
            uwind=(1.0 + 7.0 * u["uwind"]).astype(F), vwind=(-3.0 + 6.0 * u["vwind"]).astype(F),
            phase=np.minimum(np.floor(16.0 * u["phase"]), 15.0).astype(F))                        # rain comes in fronts, not cell by cell
    return s, lon, static


def declination(iday, ihour):
    """Float64 restatement of the uniform part of CALC_DECLIN (hdrv:826-854), used only to shape the synthetic records."""
    julian = iday + ihour / 24.0
    d2r = math.pi / 180.0
    sx = (360.0 / 365.0) * ((julian - 80.0) if julian >= 80.0 else (julian + 285.0)) * d2r
    return math.asin(math.sin(23.5 * d2r) * math.sin(sx))


class Records:
    """3-hourly forcing records on the device (torch float32), functions of position and record time only."""

    def __init__(self, lat, lon, static):
        import torch
        self.t = torch
        self.lat, self.lon, self.s = lat, lon, static
        d2r = math.pi / 180.0
        self.sinlat, self.coslat = torch.sin(lat * d2r), torch.cos(lat * d2r)

    def at(self, rec_index):
        """Record number `rec_index` (valid at hour RECORD_HOURS * rec_index since the start of the run)."""
        t = self.t
        hours = RECORD_HOURS * rec_index
        iday, ihour = DAY0 + hours // 24, hours % 24
        decl = declination(iday, ihour)
        loc = t.remainder(ihour + self.lon / 15.0 + 24.0, 24.0)
        cosz = t.clamp(self.sinlat * math.sin(decl) + self.coslat * math.cos(decl)
                       * t.cos((loc - 12.0) * (15.0 * math.pi / 180.0)), min=0.0)
        s = self.s
        tair = s["tbase"] + 5.0 * t.cos((loc - 15.0) * (2.0 * math.pi / 24.0)) * s["cloud"]
        es = 611.2 * t.exp(17.67 * (tair - 273.15) / (tair - 29.65))
        q = s["rh"] * 0.622 * es / (s["psfc"] - es)
        lw = (0.65 + 0.3 * (1.0 - s["cloud"])) * 5.67e-8 * tair ** 4
        raining = t.remainder(s["phase"] + float(rec_index), 16.0) < 1.0
        return dict(t=tair, q=q, u=s["uwind"], v=s["vwind"], p=s["psfc"], lw=lw, sw=1000.0 * cosz * s["cloud"],
                    pcp=t.where(raining, t.full_like(tair, 5.0e-4), t.zeros_like(tair)), fpar=None, lai=None)


def step_time(n):
    """(iday, ihour) of 0-based model step n (one-hour steps from DAY0 00 UTC)."""
    return DAY0 + n // 24, n % 24

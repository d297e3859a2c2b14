"""Seeded synthetic workloads: the BASELINE.json configs as SURVEY.md section 8(d) specifies them.

RNG: ``numpy.random.Generator(Philox(seed))`` (counter-based, reproducible across platforms).
All fields are float32 / int32 in the column store's Fortran layout.
"""
import math

import numpy as np

from .state import ColumnStore, ModelConfig
from .init import noahmp_init

F = np.float32
CONUS_VEG = np.array([2, 5, 7, 8, 10, 11, 14, 15, 19], dtype=np.int32)


def _rng(seed):
    return np.random.Generator(np.random.Philox(seed))


def diurnal_forcing(store, hour, t_offset=None, t_base=283.0, rain_hours=(10, 11, 12), rain_mm=2.0):
    """SURVEY 8d config-1 forcing at local hour `hour` (0..23), broadcast over the tile."""
    a = store.a
    cfg = store.cfg
    cosz = max(0.0, math.sin(math.pi * (hour - 6.0) / 12.0))
    a["coszin"][...] = F(cosz)
    a["swdown"][...] = F(800.0 * cosz)
    a["glw"][...] = F(320.0 + 20.0 * cosz)
    t = F(t_base + 8.0 * cosz)
    if t_offset is None:
        a["t3d"][...] = t
    else:
        a["t3d"][...] = (t + t_offset)[:, None, :].astype(F)
    a["qv3d"][...] = F(0.006)
    a["u_phy"][...] = F(3.0)
    a["v_phy"][...] = F(1.0)
    a["p8w3d"][...] = F(95000.0)
    a["dz8w"][...] = F(2.0 * cfg.zlvl)                       # hdrv:344
    a["rainbl"][...] = F(rain_mm if (int(math.floor(hour)) % 24) in rain_hours else 0.0)   # mm per step (hdrv:343)
    return cosz


def _base_store(ni, nj, cfg):
    s = ColumnStore(ni, nj, cfg)
    a = s.a
    a["xland"][...] = 1.0
    a["xice"][...] = 0.0
    a["xlatin"][...] = 30.0
    a["vegmax"][...] = 90.0
    # what the HRLDAS cold start leaves at -1e20 and the harness must zero
    for nm in ("sfcrunoff", "udrunoff", "acsnom", "acsnow", "rechxy", "deeprechxy", "qsfc", "albedo",
               "smoiseq", "smcwtdxy", "xlaixy", "taussxy"):
        a[nm][...] = 0.0
    a["zsnsoxy"][...] = 0.0
    return s


def first_step_fixups(store):
    """hdrv:374-384: on the first step HRLDAS overwrites EAH/TAH/CH/CM."""
    a = store.a
    a["eahxy"][...] = (a["p8w3d"][:, 0, :] * a["qv3d"][:, 0, :]) / (F(0.622) + a["qv3d"][:, 0, :])
    a["tahxy"][...] = a["t3d"][:, 0, :]
    a["chxy"][...] = 0.1
    a["cmxy"][...] = 0.1


def config1(tables, cfg=None):
    """Single column, veg 7 (grassland), soil 6 (loam), lat 30, namelist defaults."""
    cfg = cfg or ModelConfig()
    s = _base_store(1, 1, cfg)
    a = s.a
    a["ivgtyp"][...] = 7
    a["isltyp"][...] = 6
    a["vegfra"][...] = 60.0
    a["tmn"][...] = 285.0
    a["tsk"][...] = 283.0
    a["tslb"][...] = np.array([283.0, 284.0, 285.0, 285.5], dtype=F)[None, :, None]
    a["smois"][...] = np.array([0.25, 0.27, 0.30, 0.31], dtype=F)[None, :, None]
    a["snow"][...] = 0.0
    a["snowh"][...] = 0.0
    diurnal_forcing(s, 0)
    noahmp_init(s, tables)
    return s


def config2(tables, ni=1024, nj=1024, seed=2, cfg=None):
    """1M synthetic land columns, 4 soil / 0 snow, dynamic_veg off (DVEG=1), opt_run=1."""
    cfg = cfg or ModelConfig(idveg=1)
    r = _rng(seed)
    s = _base_store(ni, nj, cfg)
    a = s.a
    shp = (nj, ni)
    a["ivgtyp"][...] = CONUS_VEG[r.integers(0, len(CONUS_VEG), size=shp)]
    a["isltyp"][...] = r.integers(1, 13, size=shp).astype(np.int32)
    a["vegfra"][...] = r.uniform(20.0, 90.0, size=shp).astype(F)
    a["vegmax"][...] = np.maximum(a["vegfra"], F(90.0))
    a["tmn"][...] = r.uniform(278.0, 292.0, size=shp).astype(F)
    toff = np.clip(r.normal(0.0, 5.0, size=shp), -7.5, 15.0).astype(F)     # keep T > 275 K: no snow
    s.t_offset = toff
    a["tsk"][...] = F(283.0) + toff
    for k in range(s.cfg.nsoil):            # layers below the fourth (NSOIL > 4 builds) continue the profile
        dt_, sm = 0.5 * k, (0.25, 0.27, 0.30, 0.31)[min(k, 3)]
        a["tslb"][:, k, :] = a["tsk"] * F(0.5) + a["tmn"] * F(0.5) + F(dt_)
        a["smois"][:, k, :] = F(sm) + r.uniform(-0.05, 0.05, size=shp).astype(F)
    a["snow"][...] = 0.0
    a["snowh"][...] = 0.0
    diurnal_forcing(s, 0, t_offset=toff)
    noahmp_init(s, tables)
    return s


def config3(tables, ni=4608, nj=1536, seed=3, cfg=None, snow_frac=0.30, urban_frac=0.02,
            glacier_frac=0.01):
    """CONUS-1km-like: ~7.08 M columns, 30 % snow-covered (ISNOW 0..-3), 2 % urban, 1 % glacier."""
    cfg = cfg or ModelConfig()
    r = _rng(seed)
    s = _base_store(ni, nj, cfg)
    a = s.a
    shp = (nj, ni)
    a["ivgtyp"][...] = CONUS_VEG[r.integers(0, len(CONUS_VEG), size=shp)]
    a["isltyp"][...] = r.integers(1, 13, size=shp).astype(np.int32)
    u = r.random(size=shp)
    a["ivgtyp"][u < urban_frac] = cfg.isurban
    gl = (u >= urban_frac) & (u < urban_frac + glacier_frac)
    a["ivgtyp"][gl] = cfg.isice
    a["isltyp"][gl] = 16
    a["vegfra"][...] = r.uniform(20.0, 90.0, size=shp).astype(F)
    a["vegmax"][...] = np.maximum(a["vegfra"], F(90.0))
    a["tmn"][...] = r.uniform(272.0, 290.0, size=shp).astype(F)
    tair = np.clip(r.normal(272.0, 8.0, size=shp), 245.0, 300.0).astype(F)
    has_snow = (r.random(size=shp) < snow_frac) | gl
    tair = np.where(has_snow, np.minimum(tair, F(272.0)), np.maximum(tair, F(274.5))).astype(F)
    s.t_offset = (tair - F(283.0)).astype(F)
    a["tsk"][...] = tair
    swe = r.uniform(5.0, 300.0, size=shp).astype(F)
    rho = r.uniform(100.0, 350.0, size=shp).astype(F)
    a["snow"][...] = np.where(has_snow, swe, F(0.0))
    a["snowh"][...] = np.where(has_snow, swe / rho, F(0.0))
    for k in range(s.cfg.nsoil):            # layers below the fourth (NSOIL > 4 builds) continue the profile
        dt_, sm = 0.5 * k, (0.25, 0.27, 0.30, 0.31)[min(k, 3)]
        a["tslb"][:, k, :] = a["tsk"] * F(0.5) + a["tmn"] * F(0.5) + F(dt_)
        a["smois"][:, k, :] = F(sm) + r.uniform(-0.05, 0.05, size=shp).astype(F)
    diurnal_forcing(s, 0, t_offset=s.t_offset)
    noahmp_init(s, tables)
    return s


def mixed_small(tables, ni=64, nj=8, seed=7, cfg=None, **kw):
    """Small config-3-style tile for parity tests (every branch family present)."""
    return config3(tables, ni=ni, nj=nj, seed=seed, cfg=cfg,
                   snow_frac=kw.get("snow_frac", 0.4), urban_frac=kw.get("urban_frac", 0.05),
                   glacier_frac=kw.get("glacier_frac", 0.05))


def groundwater_fields(store, tables_dict, seed=4, area=1.0e6, stress=0.0, water_frac=0.03, x0=0, y0=0):
    """MMF planes for OPT_RUN=5: FDEPTH~U(50,200), TOPO = smooth random field,
    EQZWT~U(-20,-1), RIVERCOND~U(0,1e-2), RIVERBED=EQZWT-1, PEXP=1, AREA=dx*dx.

    The water table starts around its equilibrium depth with a spread that puts columns in all three
    UPDATEWTD regimes (inside the 2 m soil column, in the layer below it, deep).  ``stress`` > 0 adds a
    pending DEEPRECH of that standard deviation [m] so that multi-layer fills/drains are exercised too
    (with AREA = 1 km2 the lateral and river fluxes alone move only ~1e-5 m of water per call).
    """
    r = _rng(seed)
    store.add_groundwater()
    a = store.a
    nj, ni = store.nj, store.ni
    shp = (nj, ni)
    y, x = np.meshgrid(np.arange(nj, dtype=np.float64) + y0, np.arange(ni, dtype=np.float64) + x0, indexing="ij")   # global cell indices
    topo = 300.0 + 40.0 * np.sin(x / 17.0 + 0.3) * np.cos(y / 23.0) + 15.0 * np.sin((x + 2 * y) / 7.0)
    a["topo"][...] = (topo + r.normal(0.0, 0.5, size=shp)).astype(F)
    a["fdepth"][...] = r.uniform(50.0, 200.0, size=shp).astype(F)
    a["fdepth"][r.random(size=shp) < 0.02] = 0.0                       # gw:239 FDEPTH <= 0 branch
    a["area"][...] = F(area)
    a["eqwtd"][...] = r.uniform(-20.0, -1.0, size=shp).astype(F)
    a["rivercond"][...] = r.uniform(0.0, 1.0e-2, size=shp).astype(F)
    a["riverbed"][...] = a["eqwtd"] - F(1.0)
    a["pexp"][...] = 1.0
    u = r.random(size=shp)
    wtd = np.where(u < 0.4, r.uniform(-2.0, -0.02, size=shp),
                   np.where(u < 0.6, r.uniform(-3.0, -2.0, size=shp), r.uniform(-20.0, -3.0, size=shp)))
    a["zwtxy"][...] = wtd.astype(F)
    smcmax = np.asarray(tables_dict["maxsmc"], dtype=F)[np.clip(a["isltyp"], 1, 19) - 1]
    urban = a["ivgtyp"] == store.cfg.isurban
    smcmax = np.where(urban, F(0.45), smcmax).astype(F)
    for k in range(store.cfg.nsoil):
        sm = np.minimum(a["smois"][:, k, :], smcmax)
        a["smois"][:, k, :] = sm
        a["sh2o"][:, k, :] = np.minimum(a["sh2o"][:, k, :], sm)
        a["smoiseq"][:, k, :] = np.clip(sm * r.uniform(0.7, 1.1, size=shp).astype(F), F(0.02), smcmax * F(0.98))
    a["smcwtdxy"][...] = (smcmax * r.uniform(0.4, 1.0, size=shp)).astype(F)
    a["deeprechxy"][...] = (r.normal(0.0, 1.0e-5 + stress, size=shp)).astype(F)
    a["rechxy"][...] = 0.0
    for n in ("qrf", "qspring", "qslat", "qrfs", "qsprings"):
        a[n][...] = 0.0
    wat = r.random(size=shp) < water_frac
    a["xland"][wat] = 2.0
    return store


ROW_BLOCK = 64


def config3_tile(tables, gx, gy, x0=0, y0=0, nx=None, ny=None, seed=3, cfg=None, groundwater=False, **kw):
    """Cells [x0, x0+nx) x [y0, y0+ny) (0-based) of ONE global gx x gy config-3 grid (config-4 grid with `groundwater`: the MMF
    planes of groundwater_fields on top).  The grid is generated in blocks of ROW_BLOCK full rows, block b from the Philox key
    (seed, b), so a tile is the same cells whatever the decomposition: what N ranks cut (mpp_land_partition_calc, mpp:227-288,
    plus the 1-cell ring of gw:231-252) is what one rank holds.  -> ColumnStore of nx x ny cells with .t_offset."""
    cfg = cfg or ModelConfig()
    nx = gx - x0 if nx is None else nx
    ny = gy - y0 if ny is None else ny
    assert 0 <= x0 and x0 + nx <= gx and 0 <= y0 and y0 + ny <= gy
    out = _base_store(nx, ny, cfg)
    if groundwater:
        out.add_groundwater()
    out.t_offset = np.zeros((ny, nx), dtype=F)
    for b in range(y0 // ROW_BLOCK, (y0 + ny - 1) // ROW_BLOCK + 1):
        r0 = b * ROW_BLOCK
        rows = min(ROW_BLOCK, gy - r0)
        blk = config3(tables, ni=gx, nj=rows, seed=[seed, b], cfg=cfg, **kw)
        if groundwater:
            groundwater_fields(blk, tables, seed=[seed + 1, b], y0=r0)
        lo, hi = max(y0, r0), min(y0 + ny, r0 + rows)
        for k, v in blk.a.items():
            if k != "dzs":
                out.a[k][lo - y0:hi - y0] = v[lo - r0:hi - r0, ..., x0:x0 + nx]
        out.t_offset[lo - y0:hi - y0] = blk.t_offset[lo - r0:hi - r0, x0:x0 + nx]
    return out


SNOW_EDGES = (0.025, 0.05, 0.10, 0.20, 0.25, 0.45)     # layer create / divide / combine depths [m] of SNOW_INIT, DIVIDE, COMBINE


def veg_snow_matrix(tables, cfg=None, seed=21):
    """SURVEY 8c fixture (2): every USGS category (27 columns) x 8 rows = {no snow, 1, 2, 3 snow layers} x {soil 3, soil 9}.
    Category 16 (water) is skipped by noahmplsm, 24 runs the glacier path."""
    cfg = cfg or ModelConfig()
    r = _rng(seed)
    ni, nj = 27, 8
    s = _base_store(ni, nj, cfg)
    a = s.a
    shp = (nj, ni)
    a["ivgtyp"][...] = np.arange(1, 28, dtype=np.int32)[None, :]
    a["isltyp"][...] = np.where(np.arange(nj)[:, None] < 4, 3, 9)
    a["isltyp"][a["ivgtyp"] == cfg.isice] = 16
    a["isltyp"][a["ivgtyp"] == cfg.iswater] = 14
    a["xland"][a["ivgtyp"] == cfg.iswater] = 2.0
    a["vegfra"][...] = r.uniform(30.0, 85.0, size=shp).astype(F)
    a["vegmax"][...] = np.maximum(a["vegfra"], F(90.0))
    depth = np.array([0.0, 0.04, 0.15, 0.60], dtype=F)[np.arange(nj) % 4][:, None] * np.ones(shp, F)
    snowy = depth > 0
    tair = np.where(snowy, F(268.0), F(285.0)) + r.uniform(-2.0, 2.0, size=shp).astype(F)
    tair[a["ivgtyp"] == cfg.isice] = np.minimum(tair[a["ivgtyp"] == cfg.isice], F(266.0))
    s.t_offset = (tair - F(283.0)).astype(F)
    a["tmn"][...] = (tair + F(2.0)).astype(F)
    a["tsk"][...] = tair
    a["snowh"][...] = depth
    a["snow"][...] = depth * F(200.0)
    for k in range(s.cfg.nsoil):            # layers below the fourth (NSOIL > 4 builds) continue the profile
        dt_, sm = 0.5 * k, (0.25, 0.27, 0.30, 0.31)[min(k, 3)]
        a["tslb"][:, k, :] = a["tsk"] * F(0.5) + a["tmn"] * F(0.5) + F(dt_)
        a["smois"][:, k, :] = F(sm) + r.uniform(-0.04, 0.04, size=shp).astype(F)
    diurnal_forcing(s, 0, t_offset=s.t_offset)
    noahmp_init(s, tables)
    return s


def scalar_tile(tables, cfg=None, seed=23):
    """Tile of the uniform-scalar sweep (tests/golden/make_golden_scalars.py): every USGS category (27 columns) x 4 rows of
    snow depth {0, 0.04, 0.15, 0.60 m}; both hemispheres for every category (the sign of XLATIN selects PHENOLOGY's half-year
    shift, lsm:1054-1071) in a checkerboard; soil type by column."""
    cfg = cfg or ModelConfig()
    r = _rng(seed)
    ni, nj = 27, 4
    s = _base_store(ni, nj, cfg)
    a = s.a
    shp = (nj, ni)
    a["ivgtyp"][...] = np.arange(1, 28, dtype=np.int32)[None, :]
    a["isltyp"][...] = (1 + (np.arange(ni) * 5) % 12)[None, :]
    a["isltyp"][a["ivgtyp"] == cfg.isice] = 16
    a["isltyp"][a["ivgtyp"] == cfg.iswater] = 14
    a["xland"][a["ivgtyp"] == cfg.iswater] = 2.0
    south = ((np.arange(nj)[:, None] + np.arange(ni)[None, :]) % 2) == 1
    a["xlatin"][...] = np.where(south, F(-35.0), F(40.0))
    a["vegfra"][...] = r.uniform(30.0, 85.0, size=shp).astype(F)
    a["vegmax"][...] = np.maximum(a["vegfra"], F(90.0))
    depth = np.array([0.0, 0.04, 0.15, 0.60], dtype=F)[:, None] * np.ones(shp, F)
    tair = np.where(depth > 0, F(269.0), F(285.0)) + r.uniform(-2.0, 2.0, size=shp).astype(F)
    tair[a["ivgtyp"] == cfg.isice] = np.minimum(tair[a["ivgtyp"] == cfg.isice], F(266.0))
    s.t_offset = (tair - F(283.0)).astype(F)
    a["tmn"][...] = (tair + F(2.0)).astype(F)
    a["tsk"][...] = tair
    a["snowh"][...] = depth
    a["snow"][...] = depth * F(200.0)
    for k in range(s.cfg.nsoil):            # layers below the fourth (NSOIL > 4 builds) continue the profile
        dt_, sm = 0.5 * k, (0.25, 0.27, 0.30, 0.31)[min(k, 3)]
        a["tslb"][:, k, :] = a["tsk"] * F(0.5) + a["tmn"] * F(0.5) + F(dt_)
        a["smois"][:, k, :] = F(sm) + r.uniform(-0.04, 0.04, size=shp).astype(F)
    diurnal_forcing(s, 0, t_offset=s.t_offset)
    noahmp_init(s, tables)
    return s


def snow_edges(tables, cfg=None, seed=22):
    """SURVEY 8c fixture (4): snow depths one float32 ulp below / at / above each layering threshold, on grass (7),
    evergreen needleleaf (14), barren (19) and glacier (24) columns; rows: cold (accumulating) and near-melting air."""
    cfg = cfg or ModelConfig()
    r = _rng(seed)
    edges = []
    for e in SNOW_EDGES:
        e = F(e)
        edges += [np.nextafter(e, F(0.0)), e, np.nextafter(e, F(1.0))]
    edges = np.array(edges, dtype=F)
    vegs = np.array([7, 14, 19, cfg.isice], dtype=np.int32)
    ni, nj = len(edges), 2 * len(vegs)
    s = _base_store(ni, nj, cfg)
    a = s.a
    shp = (nj, ni)
    a["ivgtyp"][...] = vegs[np.arange(nj) % len(vegs)][:, None]
    a["isltyp"][...] = 6
    a["isltyp"][a["ivgtyp"] == cfg.isice] = 16
    a["vegfra"][...] = 60.0
    a["vegmax"][...] = 90.0
    cold = (np.arange(nj) < len(vegs))[:, None] & np.ones(shp, bool)
    tair = np.where(cold, F(264.0), F(272.5)).astype(F) + r.uniform(-0.5, 0.5, size=shp).astype(F)
    s.t_offset = (tair - F(283.0)).astype(F)
    a["tmn"][...] = F(272.0)
    a["tsk"][...] = tair
    a["snowh"][...] = edges[None, :]
    a["snow"][...] = edges[None, :] * r.uniform(120.0, 300.0, size=shp).astype(F)
    for k in range(s.cfg.nsoil):            # layers below the fourth (NSOIL > 4 builds) continue the profile
        dt_, sm = 0.5 * k, (0.25, 0.27, 0.30, 0.31)[min(k, 3)]
        a["tslb"][:, k, :] = np.minimum(a["tsk"], F(273.0)) + F(dt_)
        a["smois"][:, k, :] = F(sm)
    diurnal_forcing(s, 0, t_offset=s.t_offset)
    noahmp_init(s, tables)
    return s

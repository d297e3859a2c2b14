"""Synthetic-state helper: numpy mirror of NOAHMP_INIT + SNOW_INIT used by noahmp_amd/synth.py to BUILD TEST AND
BENCH INPUTS on machines without a GPU.  It is not the product's cold start and nothing falls back to it: the
product entry is ``noahmp_hip_init`` (noahmp_amd/csrc/noahmp_init.hip, ``Engine.noahmp_init``), which
tests/test_init.py holds bit-identical to the reference; this mirror is checked against the same oracle there.

Reference: phys/module_sf_noahmpdrv.F90:847-1177 (NOAHMP_INIT, ``restart=.false.``,
``iopt_run /= 5`` branch) and :1182-1283 (SNOW_INIT).  float32 arithmetic throughout
(the reference is default REAL).  Inputs that must already be set in the store:
``snow, snowh, tsk, tslb, smois, isltyp, ivgtyp, xice``.
"""
import numpy as np

F = np.float32


def noahmp_init(store, tables, fndsnowh=True):
    cfg = store.cfg
    a = store.a
    ns = cfg.nsoil
    tb = tables if isinstance(tables, dict) else None
    if tb is None:
        from .abi import tables_to_dict
        tb = tables_to_dict(tables)

    snow, snowh, tsk = a["snow"], a["snowh"], a["tsk"]
    if not fndsnowh:                                   # drv:997-1005
        snowh[...] = snow * F(0.005)

    if (a["isltyp"] < 1).any():                        # drv:1010-1020
        raise ValueError("lsminit: out of range value of ISLTYP")

    # ---- soil liquid water SH2O (drv:1032-1069)
    BLIM_UNUSED, HLICE, GRAV, T0 = 5.5, F(3.335e5), F(9.81), F(273.15)
    glac = (a["ivgtyp"] == cfg.isice) & (a["xice"] <= 0.0)
    smois, sh2o, tslb = a["smois"], a["sh2o"], a["tslb"]
    for k in range(ns):
        smois[:, k, :][glac] = 1.0
        sh2o[:, k, :][glac] = 0.0
        tslb[:, k, :][glac] = np.minimum(tslb[:, k, :][glac], F(263.15))
    snow[glac] = np.maximum(snow[glac], F(10.0))
    snowh[glac] = snow[glac] * F(0.01)

    st = a["isltyp"] - 1
    bx = tb["bb"][st].astype(F)
    smcmax = tb["maxsmc"][st].astype(F)
    psisat = tb["satpsi"][st].astype(F)
    ok = (bx > 0) & (smcmax > 0) & (psisat > 0)
    land = ~glac
    for k in range(ns):
        sm = smois[:, k, :]
        sm[land] = np.minimum(sm, smcmax)[land]        # IF (SMOIS > SMCMAX) SMOIS = SMCMAX
        t = tslb[:, k, :]
        frozen = land & ok & (t < F(273.149))
        with np.errstate(all="ignore"):
            base = (HLICE / (GRAV * (-psisat))) * ((t - T0) / t)
            fk = np.power(base.astype(F), (F(-1.0) / bx).astype(F)).astype(F) * smcmax
        fk = np.maximum(fk, F(0.02))
        s = sh2o[:, k, :]
        s[land] = sm[land]
        s[frozen] = np.minimum(fk, sm)[frozen]

    # ---- per-column scalars (drv:1073-1120)
    warm_snow = (snow > 0.0) & (tsk > F(273.15))
    for nm in ("tvxy", "tgxy", "tahxy", "t2mvxy", "t2mbxy"):
        a[nm][...] = np.where(warm_snow, F(273.15), tsk)
    a["canwat"][...] = 0.0
    a["canliqxy"][...] = 0.0
    a["canicexy"][...] = 0.0
    a["eahxy"][...] = 2000.0
    a["cmxy"][...] = 0.0
    a["chxy"][...] = 0.0
    a["fwetxy"][...] = 0.0
    a["sneqvoxy"][...] = 0.0
    a["alboldxy"][...] = 0.65
    a["qsnowxy"][...] = 0.0
    a["wslakexy"][...] = 0.0
    if cfg.iopt_run != 5:
        a["waxy"][...] = 4900.0
        a["wtxy"][...] = 4900.0
        a["zwtxy"][...] = (F(25.0) + F(2.0)) - F(4900.0) / F(1000) / F(0.2)
    else:
        a["waxy"][...] = 0.0
        a["wtxy"][...] = 0.0
    a["lfmassxy"][...] = 50.0
    a["stmassxy"][...] = 50.0
    a["rtmassxy"][...] = 500.0
    a["woodxy"][...] = 500.0
    a["stblcpxy"][...] = 1000.0
    a["fastcpxy"][...] = 1000.0
    a["xsaixy"][...] = 0.1

    snow_init(store)
    return store


def snow_init(store):
    """SNOW_INIT (drv:1182-1283): split SNOWH into up to 3 layers at 0.025/0.05/0.10/0.25/0.45 m."""
    a = store.a
    ns = store.cfg.nsoil
    dzs = np.asarray(store.cfg.dzs, dtype=F)
    zsoil = -np.cumsum(dzs, dtype=F)
    swe, sd, tg = a["snow"], a["snowh"], a["tgxy"]
    nj, ni = sd.shape
    dz = np.zeros((3, nj, ni), dtype=F)          # DZSNO(-2:0)
    isn = np.zeros((nj, ni), dtype=np.int32)

    m1 = (sd >= F(0.025)) & (sd <= F(0.05))
    m2 = (sd > F(0.05)) & (sd <= F(0.10))
    m3 = (sd > F(0.10)) & (sd <= F(0.25))
    m4 = (sd > F(0.25)) & (sd <= F(0.45))
    m5 = sd > F(0.45)
    isn[m1] = -1
    dz[2][m1] = sd[m1]
    isn[m2] = -2
    dz[1][m2] = sd[m2] / F(2.0)
    dz[2][m2] = sd[m2] / F(2.0)
    isn[m3] = -2
    dz[1][m3] = 0.05
    dz[2][m3] = sd[m3] - F(0.05)
    isn[m4] = -3
    dz[0][m4] = 0.05
    dz[1][m4] = F(0.5) * (sd[m4] - F(0.05))
    dz[2][m4] = F(0.5) * (sd[m4] - F(0.05))
    isn[m5] = -3
    dz[0][m5] = 0.05
    dz[1][m5] = 0.20
    dz[2][m5] = (sd[m5] - F(0.20)) - F(0.05)

    a["isnowxy"][...] = isn
    a["tsnoxy"][...] = 0.0
    a["snicexy"][...] = 0.0
    a["snliqxy"][...] = 0.0
    with np.errstate(all="ignore"):
        rho = swe / sd
    for iz in (-2, -1, 0):                      # snow array index iz+2
        act = isn + 1 <= iz
        a["tsnoxy"][:, iz + 2, :][act] = tg[act]
        a["snicexy"][:, iz + 2, :][act] = (F(1.0) * dz[iz + 2] * rho)[act]

    # layer-bottom depths ZSNSOXY(-2:nsoil); inactive snow entries are left untouched by the
    # reference (INTENT(OUT) but never written, drv:1275-1278) -- we keep whatever the store holds.
    dzsoil = np.empty(ns, dtype=F)
    dzsoil[0] = zsoil[0]
    dzsoil[1:] = zsoil[1:] - zsoil[:-1]
    z = a["zsnsoxy"]
    run = np.zeros((nj, ni), dtype=F)
    for iz in (-2, -1, 0):
        act = isn + 1 <= iz
        run = np.where(act, run + (-dz[iz + 2]), run).astype(F)
        z[:, iz + 2, :][act] = run[act]
    for k in range(ns):
        run = (run + dzsoil[k]).astype(F)
        z[:, 3 + k, :] = run
    return store

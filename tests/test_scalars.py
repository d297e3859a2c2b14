"""Sweep over the scalars that are uniform over the grid at the noahmplsm boundary (drv:51-83), against snapshots the COMPILED
REFERENCE produced (tests/golden/make_golden_scalars.py -> golden_scalars.npz):
  DT 600 / 900 / 1800 / 3600 s   -- NITER doubling (lsm:7850-7857; the two "pour" cases trigger it), FACT, accumulators, COMPACT
  DZS (0.05, 0.25, 0.7, 1.5)     -- ZSOIL (drv:392-395): root fractions, tridiagonal coefficients, ZWTEQ
  YR 2000 / 2001 / 2100          -- YEARLEN 366 / 365 / 365 (drv:381-390)
  JULIAN 1 ... 366, both hemispheres for every vegetation category (lsm:1054-1071)
  DZ8W 60 / 20 / 8 m             -- the forcing height
The C restatement, the device source compiled for the host and the GPU (cold start + free run) must reproduce every snapshot bit
for bit."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_store
from noahmp_amd import synth
from golden.make_golden_scalars import CASES, case_config, case_forcing
from test_matrix import _same

FIX = os.path.join(GOLDEN, "golden_scalars.npz")
FORCING = ("coszin", "swdown", "glw", "t3d", "rainbl", "qv3d", "u_phy", "v_phy", "p8w3d", "dz8w")


@pytest.fixture(scope="module")
def golden():
    return np.load(FIX)


def _drive(g, name, init, step, snapshot):
    c = CASES[name]
    cfg = case_config(name)
    ni, nj = 27, 4
    s = init(load_store(g, name + "/raw", ni, nj, cfg))
    want = load_store(g, name + "/init", ni, nj, cfg)
    if isinstance(s.a["tsk"], np.ndarray):
        s.t_offset = g[name + "/t_offset"]
        case_forcing(s, name, 1)
        synth.first_step_fixups(s)
        _same(want, s, name + " cold start")
    forcing = load_store(g, name + "/init", ni, nj, cfg)
    forcing.t_offset = g[name + "/t_offset"]
    for it in range(1, c["nsteps"] + 1):
        yr, jul = case_forcing(forcing, name, it)
        step(s, forcing, it, yr, float(jul))
        if it in c["snap"]:
            _same(load_store(g, "%s/step%02d" % (name, it), ni, nj, cfg), snapshot(s), "%s step %d" % (name, it))


def test_cases_cover_the_scalars():
    dts = {case_config(n).dt for n in CASES}
    assert {600.0, 900.0, 1800.0, 3600.0} <= dts
    assert len({case_config(n).dzs for n in CASES}) >= 2 and len({case_config(n).zlvl for n in CASES}) >= 3
    assert {2000, 2001, 2100} <= {c["yr"] for c in CASES.values()}
    assert min(c["jul"] for c in CASES.values()) == 1.0 and max(c["jul"] for c in CASES.values()) == 366.0


@pytest.mark.parametrize("name", list(CASES))
def test_port_matches_reference_scalars(port, golden, name):
    def init(s):
        rc, _ = port.noahmp_init(s, fndsnowh=True)
        assert rc == 0
        return s

    def step(s, f, it, yr, jul):
        for k in FORCING:
            s.a[k][...] = f.a[k]
        assert port.noahmplsm(s, it, yr, jul).code == 0
    _drive(golden, name, init, step, lambda s: s)


@pytest.mark.parametrize("name", list(CASES))
def test_device_source_on_host_matches_reference_scalars(tables, golden, name):
    from host_emul.emullib import EmulLib
    em = EmulLib()
    em.set_tables(tables[0])

    def init(s):
        rc, _ = em.noahmp_init(s, fndsnowh=True)
        assert rc == 0
        return s

    def step(s, f, it, yr, jul):
        for k in FORCING:
            s.a[k][...] = f.a[k]
        assert em.noahmplsm(s, it, yr, jul).code == 0
    _drive(golden, name, init, step, lambda s: s)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_gpu_matches_reference_scalars(engine, golden, name):
    import torch

    def init(s):
        d = s.to_device("cuda:0")
        engine.noahmp_init(d, fndsnowh=True)
        h = d.to_host()
        h.t_offset = golden[name + "/t_offset"]
        case_forcing(h, name, 1)
        synth.first_step_fixups(h)
        for k in ("eahxy", "tahxy", "chxy", "cmxy"):
            d.a[k].copy_(torch.from_numpy(h.a[k]))
        return d

    def step(d, f, it, yr, jul):
        for k in FORCING:
            d.a[k].copy_(torch.from_numpy(f.a[k]))
        assert engine.noahmplsm(d, it, yr, jul).code == 0
    _drive(golden, name, init, step, lambda d: d.to_host())

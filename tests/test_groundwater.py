"""MMF groundwater: WTABLE_mmf_noahmp, reference phys/module_sf_noahmp_groundwater.F90:14-606.

CPU tests: the C restatement against the compiled reference (bit-exact) and against the committed golden
fixture; the device source compiled for the host against the restatement (bit-exact); decomposition
independence of the halo formulation.  The GPU parity tests are in test_parity_gpu.py.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from noahmp_amd import synth
from noahmp_amd.state import ModelConfig, ColumnStore, GW_EXTRA

GW_OUT = ["smois", "sh2o", "smcwtdxy", "zwtxy", "deeprechxy", "rechxy", "qrf", "qspring", "qslat", "qrfs",
          "qsprings"]


def gw_store(tables, ni=48, nj=40, seed=4, stress=0.0, area=1.0e6):
    cfg = ModelConfig(iopt_run=5)
    s = synth.mixed_small(tables[0], ni=ni, nj=nj, seed=seed, cfg=cfg)
    synth.groundwater_fields(s, tables[1], seed=seed + 100, area=area, stress=stress)
    return s


def assert_same(a, b, names=GW_OUT, what=""):
    for n in names:
        x, y = a.a[n], b.a[n]
        if not np.array_equal(x, y, equal_nan=True):
            bad = np.argwhere(~((x == y) | (np.isnan(x) & np.isnan(y))))
            raise AssertionError("%s %s differs at %d cells, first %s: %r vs %r"
                                 % (what, n, len(bad), bad[0], x[tuple(bad[0])], y[tuple(bad[0])]))


CASES = [dict(stress=0.0, area=1.0e6), dict(stress=0.02, area=1.0e6), dict(stress=0.2, area=2.0e3),
         dict(stress=1.0, area=1.0e6)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "stress%g_area%g" % (c["stress"], c["area"]))
def test_port_matches_reference_bitexact(tables, port, reflib, case):
    reflib.set_tables(tables[0])
    s0 = gw_store(tables, **case)
    a, b = s0.copy(), s0.copy()
    for it in range(4):                       # 4 successive calls: the state feeds back
        reflib.wtable_mmf(a)
        port.wtable_mmf(b)
        assert_same(a, b, what="call %d" % it)
        a.a["deeprechxy"][...] = s0.a["deeprechxy"]      # SFLX would refill DEEPRECH between calls
        b.a["deeprechxy"][...] = s0.a["deeprechxy"]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "stress%g_area%g" % (c["stress"], c["area"]))
def test_device_source_on_host_matches_port(tables, port, case):
    from host_emul.emullib import EmulLib
    em = EmulLib()
    em.set_tables(tables[0])
    s0 = gw_store(tables, **case)
    a, b = s0.copy(), s0.copy()
    for it in range(4):
        sa = port.wtable_mmf(a)
        sb = em.wtable_mmf(b)
        assert sa.n_land == sb.n_land and sa.n_land > 0
        assert_same(a, b, what="call %d" % it)
        a.a["deeprechxy"][...] = s0.a["deeprechxy"]
        b.a["deeprechxy"][...] = s0.a["deeprechxy"]


def test_branch_coverage_of_cases(tables, port):
    """The synthetic cases reach every UPDATEWTD regime (otherwise the parity above proves little)."""
    seen = set()
    for case in CASES:
        s = gw_store(tables, **case)
        before = s.copy()
        port.wtable_mmf(s)
        land = (s.a["xland"] < 1.5) & (s.a["ivgtyp"] != s.cfg.isice) & (s.a["xice"] < s.cfg.xice_thres)
        w0, w1 = before.a["zwtxy"][land], s.a["zwtxy"][land]
        for lo, hi, nm in ((-2.0, 0.0, "soil"), (-3.0, -2.0, "below"), (-1e9, -3.0, "deep")):
            m = (w0 >= lo) & (w0 < hi)
            if (w1[m] > w0[m]).any():
                seen.add(nm + "_up")
            if (w1[m] < w0[m]).any():
                seen.add(nm + "_down")
        if (s.a["qspring"][land] > 0).any():
            seen.add("spring")
        if (np.abs(s.a["smois"] - before.a["smois"]) > 0).any(axis=(1,)).any():
            seen.add("smc_changed")
        if (s.a["qslat"] != 0).any():
            seen.add("qlat")
        if (s.a["qrf"] > 0).any():
            seen.add("qrf")
    assert seen >= {"soil_up", "soil_down", "below_up", "below_down", "deep_up", "deep_down", "spring",
                    "smc_changed", "qlat", "qrf"}, seen


def test_golden_fixture(tables, port):
    """Fixture written by tests/golden/make_golden_gw.py from the compiled reference (oracle/_ref)."""
    z = np.load(os.path.join(GOLDEN, "golden_gw.npz"))
    ni, nj = int(z["ni"]), int(z["nj"])
    s = ColumnStore(ni, nj, ModelConfig(iopt_run=5)).add_groundwater()
    for k in s.a:
        if "in/" + k in z:
            s.a[k][...] = z["in/" + k]
    port.wtable_mmf(s)
    for n in GW_OUT:
        assert np.array_equal(s.a[n], z["out/" + n]), n


def test_halo_tiles_reproduce_single_domain(tables, port):
    """a decomposed domain with a 1-cell ZWTXY halo gives the single-domain answer bit for bit
    (the reference's own MPI run does not: it clamps at tile edges, SURVEY 'reference behaviour caveat')."""
    ni, nj = 48, 40
    s0 = gw_store(tables, ni=ni, nj=nj, stress=0.02)
    whole = s0.copy()
    port.wtable_mmf(whole)
    out = s0.copy()
    from noahmp_amd.partition import partition
    for t in partition(ni, nj, 4):
        its, jts = t["startx"], t["starty"]
        ite, jte = its + t["nx"] - 1, jts + t["ny"] - 1
        ims, ime = max(its - 1, 1), min(ite + 1, ni)
        jms, jme = max(jts - 1, 1), min(jte + 1, nj)
        loc = ColumnStore(ime - ims + 1, jme - jms + 1, s0.cfg).add_groundwater()
        for k, v in s0.a.items():
            if k == "dzs":
                continue
            loc.a[k][...] = v[jms - 1:jme, ..., ims - 1:ime]
        loc.set_index(ids=1, ide=ni, jds=1, jde=nj, ims=ims, ime=ime, jms=jms, jme=jme,
                      its=its, ite=ite, jts=jts, jte=jte)
        port.wtable_mmf(loc)
        for k in GW_OUT:
            out.a[k][jts - 1:jte, ..., its - 1:ite] = loc.a[k][jts - jms:jte - jms + 1, ..., its - ims:ite - ims + 1]
    assert_same(whole, out, what="tiles")


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES[:3], ids=lambda c: "stress%g_area%g" % (c["stress"], c["area"]))
def test_gpu_two_halves_equal_the_whole_call_in_any_column_order(tables, port, engine, case):
    """noahmp_hip_wtable_lateral_async (KCELL / HEAD + QLAT stencil, tile order) + noahmp_hip_wtable_columns_async (the per-column rest,
    columns in ANY order) = noahmp_hip_wtable_mmf = the oracle, bit for bit: the per-column half runs on a randomly permuted copy of
    the store with QLAT permuted the same way."""
    import torch
    from noahmp_amd.abi import FIELD_INFO
    s0 = gw_store(tables, ni=96, nj=50, seed=9, **case)
    want = s0.copy()
    port.wtable_mmf(want)
    whole = s0.to_device("cuda:0")
    engine.wtable_mmf(whole)
    assert_same(want, whole.to_host(), what="whole call")
    # the two halves: stencil on the tile-order store, the rest on a permuted one
    tile = s0.to_device("cuda:0")
    qlat = torch.zeros((s0.nj, s0.ni), dtype=torch.float32, device="cuda:0")
    engine.wtable_lateral_async(tile.wtable_args(), qlat)
    r = np.random.Generator(np.random.Philox(77))
    perm = r.permutation(s0.ni * s0.nj)
    inv = np.empty_like(perm)
    inv[perm] = np.arange(perm.size)

    def shuffle(v, p):
        return (v.transpose(1, 0, 2).reshape(v.shape[1], -1)[:, p].reshape(v.shape[1], v.shape[0], v.shape[2]).transpose(1, 0, 2)
                if v.ndim == 3 else v.reshape(-1)[p].reshape(v.shape))
    sh = s0.copy()
    for k, v in sh.a.items():
        if k != "dzs":
            sh.a[k] = np.ascontiguousarray(shuffle(v, perm))
    dsh = sh.to_device("cuda:0")
    engine.stream_sync()
    q_sh = torch.from_numpy(np.ascontiguousarray(qlat.cpu().numpy().reshape(-1)[perm].reshape(s0.nj, s0.ni))).cuda()
    engine.wtable_columns_async(dsh.wtable_args(), q_sh)
    engine.stream_sync()
    got = dsh.to_host()
    for k in GW_OUT:
        got.a[k] = shuffle(got.a[k], inv)
    assert_same(want, got, what="two halves")
    assert (want.a["qslat"] != 0).any()


"""Asynchronous stepping of device-resident state (noahmp_hip_step_async / noahmp_hip_sync)."""
import ctypes as C

import numpy as np
import pytest

from noahmp_amd import abi, synth
from noahmp_amd.abi import FIELD_INFO

pytestmark = pytest.mark.gpu


def _outs(store):
    return [k for k in store.a if FIELD_INFO[k][2] != "in"]


def test_async_steps_equal_synchronous_steps(engine, tables):
    import torch
    s = synth.mixed_small(tables[1], ni=128, nj=16)
    synth.first_step_fixups(s)
    a, b = s.to_device("cuda:0"), s.to_device("cuda:0")
    forc = []
    for it in range(1, 9):
        synth.diurnal_forcing(s, 8 + it, t_offset=s.t_offset)
        forc.append({k: torch.from_numpy(s.a[k].copy()).cuda() for k in ("coszin", "swdown", "glw", "t3d", "rainbl")})
    n = 0
    for it in range(1, 9):
        a.a.update(forc[it - 1])
        st = engine.noahmplsm(a, it, 2000, 180.0)
        n += st.n_land + st.n_glacier
    args = b.step_args(1, 2000, 180.0)
    for it in range(1, 9):
        for k, v in forc[it - 1].items():
            setattr(args, k, v.data_ptr())
        args.itimestep = it
        engine.noahmplsm_async(args)
    st, step = engine.sync()
    assert st.code == 0 and step == -1 and st.n_land + st.n_glacier == n and st.kernel_ms > 0
    ha, hb = a.to_host(), b.to_host()
    for k in _outs(ha):
        if k in ("coszin", "swdown", "glw", "t3d", "rainbl"):
            continue
        np.testing.assert_array_equal(ha.a[k], hb.a[k], err_msg=k)


def test_async_reports_earliest_failing_step_and_first_column(engine, tables):
    from noahmp_amd.driver import NoahMPFatal
    import torch
    s = synth.mixed_small(tables[1], ni=64, nj=4)
    synth.first_step_fixups(s)
    synth.diurnal_forcing(s, 12, t_offset=s.t_offset)
    d = s.to_device("cuda:0")
    good = d.a["isltyp"]
    bad1 = good.clone(); bad1[3, 40] = 25            # REDPRM: too many input soil types (lsm:9266)
    bad2 = good.clone(); bad2[1, 5] = 25; bad2[3, 40] = 25
    args = d.step_args(1, 2000, 180.0)
    engine.noahmplsm_async(args)                      # step 0: fine
    args.isltyp = bad1.data_ptr()
    engine.noahmplsm_async(args)                      # step 1: one bad column
    args.isltyp = bad2.data_ptr()
    engine.noahmplsm_async(args)                      # step 2: an earlier bad column as well
    with pytest.raises(NoahMPFatal) as e:
        engine.sync()
    assert (e.value.code, e.value.i, e.value.j) == (1, 41, 4) and "step +1" in str(e.value)
    # a synchronous call while asynchronous steps are pending is refused
    engine.noahmplsm_async(d.step_args(1, 2000, 180.0))
    st = abi.Status()
    rc = engine.lib.noahmp_hip_step(C.byref(d.step_args(1, 2000, 180.0)), abi.MEM_DEVICE, None, C.byref(st))
    assert rc == -106
    engine.sync()
    assert engine.sync()[0].n_land == 0               # nothing pending: empty status


def test_sorted_layout_gives_the_same_columns(engine, tables):
    """Engine.sort_store + noahmp_hip_gather_fields: a run on the state sorted by (class, vegetation type), with the
    forcing permuted per step, equals the run in tile order once un-permuted -- bit for bit."""
    import torch
    s = synth.mixed_small(tables[1], ni=96, nj=24, glacier_frac=0.05)
    synth.first_step_fixups(s)
    fkeys = ("coszin", "swdown", "glw", "t3d", "rainbl")
    forc = []
    for it in range(1, 6):
        synth.diurnal_forcing(s, 9 + it, t_offset=s.t_offset)
        forc.append({k: torch.from_numpy(s.a[k].copy()).cuda() for k in fkeys})
    plain, srt = s.to_device("cuda:0"), s.to_device("cuda:0")
    perm = engine.sort_store(srt, tsk_bin=0)
    p = perm.cpu().numpy()
    assert sorted(p.tolist()) == list(range(s.ncol))
    vt = srt.a["ivgtyp"].cpu().numpy().ravel()
    ice = vt == s.cfg.isice
    assert (np.diff(vt[~ice]) >= 0).all() and ice[ice.argmax():].all()         # land by type, then land ice
    work = {k: torch.empty_like(forc[0][k]) for k in fkeys}
    srt.a.update(work)
    g = engine.scatter([work[k] for k in fkeys], [forc[0][k] for k in fkeys], perm, s.ni, s.nj)
    chk = {k: torch.empty_like(forc[0][k]) for k in fkeys}
    g2 = engine.gather([chk[k] for k in fkeys], [forc[2][k] for k in fkeys], perm, s.ni, s.nj)
    g.set_sources([forc[2][k] for k in fkeys]); g(); g2(); torch.cuda.synchronize()
    for k in fkeys:                                   # both permutation kernels agree
        assert torch.equal(work[k], chk[k]), k
    args = srt.step_args(1, 2000, 180.0)
    for it in range(1, 6):
        plain.a.update(forc[it - 1])
        engine.noahmplsm(plain, it, 2000, 180.0)
    for it in range(1, 6):
        g.set_sources([forc[it - 1][k] for k in fkeys])
        g()
        args.itimestep = it
        engine.noahmplsm_async(args)
    st, _ = engine.sync()
    assert st.code == 0
    p2 = engine.sort_store(s.to_device("cuda:0")).cpu().numpy()          # default keys incl. the TSK bin: a permutation too
    assert sorted(p2.tolist()) == list(range(s.ncol))
    hp, hs = plain.to_host(), srt.to_host()
    for k in _outs(hp):
        if k in fkeys:
            continue
        a, b = hp.a[k], hs.a[k]
        if a.ndim == 2:
            np.testing.assert_array_equal(a.ravel()[p], b.ravel(), err_msg=k)
        else:
            nj, nk, ni = a.shape
            np.testing.assert_array_equal(a.transpose(1, 0, 2).reshape(nk, -1)[:, p], b.transpose(1, 0, 2).reshape(nk, -1),
                                          err_msg=k)


def test_class_range_kernels_and_violation(engine, tables):
    """A class-sorted store is advanced by one kernel per class range (land / land ice / skipped): tallies and results equal the
    mixed kernel's, water columns keep their first-step side effects, and a column whose class no longer matches its range is
    reported (NOAHMP_ERR_CLASS_RANGE) instead of being computed by the wrong kernel."""
    import torch
    from noahmp_amd.driver import NoahMPFatal
    s = synth.mixed_small(tables[1], ni=128, nj=16, glacier_frac=0.08, seed=41)
    s["ivgtyp"][0, :9] = s.cfg.iswater
    s["xland"][0, :9] = 2.0
    s["xice"][1, :5] = 1.0                                   # sea ice: skipped, with per-step side effects (drv:436-441)
    synth.first_step_fixups(s)
    synth.diurnal_forcing(s, 13, t_offset=s.t_offset)
    plain, srt = s.to_device("cuda:0"), s.to_device("cuda:0")
    perm = engine.sort_store(srt).cpu().numpy()
    n_land, n_glac = srt.class_ranges
    assert n_glac > 0 and n_land + n_glac == s.ncol - 14
    for it in (1, 2):
        sp = engine.noahmplsm(plain, it, 2000, 180.0)
        ss = engine.noahmplsm(srt, it, 2000, 180.0)           # ranges declared through the store -> three kernels
        assert (ss.n_land, ss.n_glacier, ss.n_skipped) == (sp.n_land, sp.n_glacier, sp.n_skipped) == (n_land, n_glac, 14)
    hp, hs = plain.to_host(), srt.to_host()
    for k in _outs(hp):
        a, b = hp.a[k], hs.a[k]
        a = a.transpose(0, 2, 1).reshape(-1, a.shape[1])[perm] if a.ndim == 3 else a.reshape(-1)[perm]
        b = b.transpose(0, 2, 1).reshape(-1, b.shape[1]) if b.ndim == 3 else b.reshape(-1)
        assert np.array_equal(a, b, equal_nan=True), k
    # a land column turns into land ice without a new sort
    srt.a["ivgtyp"].view(-1)[3] = s.cfg.isice
    with pytest.raises(NoahMPFatal) as e:
        engine.noahmplsm(srt, 3, 2000, 180.0)
    assert e.value.code == 17
    engine.noahmplsm(plain, 3, 2000, 180.0)                    # an undeclared store runs the mixed kernel again


def test_tallies_of_a_long_sync_are_64_bit(engine, tables):
    """More than 2^31 column-steps between two syncs: the int32 members of the status saturate, noahmp_hip_sync_counts is exact
    (bench.py collects once per timed region: 7 M columns x a few hundred steps)."""
    s = synth.config2(tables[1])                       # 1024 x 1024 land columns
    synth.first_step_fixups(s)
    synth.diurnal_forcing(s, 2, t_offset=s.t_offset)   # night: the short kernel
    d = s.to_device("cuda:0")
    args = d.step_args(1, 2000, 180.0)
    ncol = s.ni * s.nj
    nsteps = (1 << 31) // ncol + 8
    for it in range(nsteps):
        args.itimestep = 2 + it
        engine.noahmplsm_async(args)
    st, step = engine.sync()
    assert st.code == 0 and step == -1
    counts = engine.sync_counts()
    assert counts[0] == nsteps * ncol and counts[1] == 0 and counts[2] == 0
    assert counts[0] > (1 << 31) and st.n_land == (1 << 31) - 1
    st1 = engine.noahmplsm(d, 2 + nsteps, 2000, 180.0)   # a synchronous step: both views agree again
    assert st1.n_land == ncol and engine.sync_counts()[0] == ncol

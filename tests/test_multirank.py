"""World-size-2 (and 4) CPU tests of the N>1 path over gloo: tile assignment by the reference's
partition rule, per-rank stepping of its own tile, metric reductions.  No GPU here, so the per-rank
compute stand-in is the oracle (test infrastructure); the engine itself is exercised per rank on the
GPU box by bench.py --gpus N."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from noahmp_amd import synth
from noahmp_amd.abi import FIELD_INFO
from noahmp_amd.partition import partition, nprocs_xy, neighbours


def test_partition_rule_matches_reference_examples():
    # mpp_land_get_nprocsxy (mpp:124-141): most-square factorisation, first best wins
    assert nprocs_xy(8) == (4, 2) and nprocs_xy(4) == (2, 2) and nprocs_xy(2) == (2, 1) and nprocs_xy(7) == (7, 1)
    # the reference's own smoke test: mpp_land_partition(1,101,101,1) on 4 ranks (test/test_mpp_land_partition.F90)
    p = partition(101, 101, 4)
    assert [(r["startx"], r["nx"], r["starty"], r["ny"]) for r in p] == \
        [(1, 51, 1, 51), (52, 50, 1, 51), (1, 51, 52, 50), (52, 50, 52, 50)]
    # tiles cover the domain exactly once
    for (gx, gy, n) in ((4608, 1536, 8), (1024, 1024, 8), (37, 11, 6)):
        cover = np.zeros((gy, gx), dtype=int)
        for r in partition(gx, gy, n):
            cover[r["starty"] - 1:r["starty"] - 1 + r["ny"], r["startx"] - 1:r["startx"] - 1 + r["nx"]] += 1
        assert (cover == 1).all()
    assert neighbours(5, 8) == dict(left=4, right=6, down=1, up=-1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _slice(store, tile):
    from noahmp_amd.state import ColumnStore
    x0, y0 = tile["startx"] - 1, tile["starty"] - 1
    sub = ColumnStore(tile["nx"], tile["ny"], store.cfg)
    for k, v in store.a.items():
        if k == "dzs":
            continue
        sub.a[k][...] = v[y0:y0 + tile["ny"], ..., x0:x0 + tile["nx"]]
    return sub


def _worker(rank, world, port, gx, gy, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from noahmp_amd.parallel import Comm
    from noahmp_amd.tables import load_tables
    from oracle.portlib import PortLib
    comm = Comm(backend="gloo")
    T, tb = load_tables("usgs")
    port_ = PortLib(autobuild=False)
    port_.set_tables(T)
    g = synth.mixed_small(tb, ni=gx, nj=gy, seed=17)        # every rank builds the same global tile
    synth.first_step_fixups(g)
    tile = comm.my_tile(gx, gy)
    mine = _slice(g, tile)
    n_cols = 0
    for it in range(1, 4):
        synth.diurnal_forcing(g, 10 + it, t_offset=g.t_offset)
        for k in ("coszin", "swdown", "glw", "t3d", "rainbl"):
            mine.a[k][...] = _slice(g, tile).a[k]
        st = port_.noahmplsm(mine, it, 2000, 180.0)
        assert st.code == 0
        n_cols += st.n_land + st.n_glacier
    comm.barrier()
    tmax = comm.reduce_max(1.0 + rank)                      # MAX over ranks, as bench.py times a step
    total = comm.reduce_sum(n_cols)
    parts = comm.gather_to_root({k: v for k, v in mine.a.items() if FIELD_INFO[k][2] != "in"})
    if rank == 0:
        q.put((tmax, total, parts, [partition(gx, gy, world)[r] for r in range(world)]))
    comm.close()


def _fallback_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from noahmp_amd.parallel import Comm
    comm = Comm(backend="nccl")                  # no GPU here: RCCL cannot come up
    comm.barrier()
    s = comm.reduce_sum(rank + 1.0)
    if rank == 0:
        q.put((comm.backend, comm.backend_note, s, comm.probe_halo()))
    else:
        comm.probe_halo()
    comm.close()


def _auto_worker(rank, world, port, fail_rank, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      NMP_HALO_AUTO_TRANSPORT="tcp", NMP_HALO_AUTO_TIMEOUT_S="60")
    if fail_rank is not None:
        os.environ["NMP_HALO_AUTO_FAIL_RANK"] = str(fail_rank)
    import torch
    from noahmp_amd.parallel import Comm
    from noahmp_amd.partition import tile_geometry
    comm = Comm(backend="gloo", halo="auto")
    gx, gy = 23, 17
    geo = tile_geometry(gx, gy, world, rank, halo=1)
    y, x = np.meshgrid(np.arange(gy), np.arange(gx), indexing="ij")
    gf = (1000.0 * y + x).astype(np.float32)
    sl = (slice(geo["jms"] - 1, geo["jme"]), slice(geo["ims"] - 1, geo["ime"]))
    f = gf[sl].copy()
    ring = np.ones(f.shape, dtype=bool)
    ring[geo["jts"] - geo["jms"]:geo["jte"] - geo["jms"] + 1, geo["its"] - geo["ims"]:geo["ite"] - geo["ims"] + 1] = False
    f[ring] = np.nan
    t = torch.from_numpy(f)
    comm.exchange_halo([t], geo)                    # whichever mover was agreed on
    q.put((rank, comm.halo, comm.halo_note, bool(np.array_equal(t.numpy(), gf[sl]))))
    comm.close()


@pytest.mark.parametrize("fail_rank", [None, 2])
def test_auto_mover_is_agreed_on_by_all_ranks(fail_rank):
    """`halo="auto"`: every rank starts the engine's own mover and runs the checked probe exchange (here with the socket transport on
    host planes, so that it runs without a GPU); it is used only if EVERY rank succeeded -- one failing rank sends all of them to the
    torch.distributed mover together -- and either way the ring arrives (4 ranks = 2 x 2: every rank has a diagonal neighbour)."""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = _free_port()
    procs = [ctx.Process(target=_auto_worker, args=(r, world, p, fail_rank, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    assert all(ok for _, _, _, ok in res), res
    if fail_rank is None:
        assert all(h == "tcp" and note is None for _, h, note, _ in res), res
    else:
        assert all(h == "torch" and note for _, h, note, _ in res), res
        assert "forced failure" in res[fail_rank][2] and "another rank" in res[0][2]


def test_nccl_that_cannot_initialise_falls_back_to_gloo():
    """A node whose RCCL does not come up (here: no GPU at all) still runs bench.py --gpus N: control plane and ring over gloo."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box where nccl cannot initialise")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = _free_port()
    procs = [ctx.Process(target=_fallback_worker, args=(r, world, p, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    backend, note, s, mover = q.get(timeout=240)
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    assert backend == "gloo" and "initialisation failed" in note and s == 3.0 and "gloo" in mover.lower()


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_cover_domain_and_match_single_process(world, port, tables):
    gx, gy = 48, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, p, gx, gy, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    tmax, total, parts, tiles = q.get(timeout=240)
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    assert tmax == float(world) and total == 3 * gx * gy
    # single-process reference run of the whole domain
    g = synth.mixed_small(tables[1], ni=gx, nj=gy, seed=17)
    synth.first_step_fixups(g)
    for it in range(1, 4):
        synth.diurnal_forcing(g, 10 + it, t_offset=g.t_offset)
        port.noahmplsm(g, it, 2000, 180.0)
    for tile, part in zip(tiles, parts):
        x0, y0 = tile["startx"] - 1, tile["starty"] - 1
        for k, v in part.items():
            np.testing.assert_array_equal(g.a[k][y0:y0 + tile["ny"], ..., x0:x0 + tile["nx"]], v, err_msg=k)


# ------------------------------------------------------------------------------------------------
# the ZWTXY halo exchange of the MMF lateral-flow stencil over torch.distributed (gloo here,
# RCCL on the GPUs).  Parity target: the single-domain sequential oracle.
def _gw_worker(rank, world, port, gx, gy, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    from noahmp_amd.parallel import Comm
    from noahmp_amd.state import ColumnStore
    from noahmp_amd.tables import load_tables
    from oracle.portlib import PortLib
    from test_groundwater import gw_store, GW_OUT
    comm = Comm(backend="gloo")
    tabs = load_tables("usgs")
    port_ = PortLib(autobuild=False)
    port_.set_tables(tabs[0])
    g = gw_store(tabs, ni=gx, nj=gy, stress=0.02)          # every rank builds the same global fields
    geo = comm.my_geometry(gx, gy)
    ims, ime, jms, jme = geo["ims"], geo["ime"], geo["jms"], geo["jme"]
    its, ite, jts, jte = geo["its"], geo["ite"], geo["jts"], geo["jte"]
    loc = ColumnStore(ime - ims + 1, jme - jms + 1, g.cfg).add_groundwater()
    for k, v in g.a.items():
        if k != "dzs":
            loc.a[k][...] = v[jms - 1:jme, ..., ims - 1:ime]
    loc.set_index(**geo)
    # poison the ring: only the exchange may provide it
    ring = np.ones((loc.nj, loc.ni), dtype=bool)
    ring[jts - jms:jte - jms + 1, its - ims:ite - ims + 1] = False
    for k in ("zwtxy", "fdepth", "topo"):
        loc.a[k][ring] = np.nan
    loc.a["isltyp"][ring] = -7
    static = [torch.from_numpy(loc.a[k]) for k in ("fdepth", "topo", "isltyp")]   # views: exchanged in place
    comm.exchange_halo(static, geo)                                               # once (static planes)
    wtd = torch.from_numpy(loc.a["zwtxy"])
    for call in range(3):
        comm.exchange_halo([wtd], geo)                                            # before every call
        port_.wtable_mmf(loc)
        loc.a["deeprechxy"][...] = g.a["deeprechxy"][jms - 1:jme, ims - 1:ime]
    part = {k: loc.a[k][jts - jms:jte - jms + 1, ..., its - ims:ite - ims + 1].copy() for k in GW_OUT}
    parts = comm.gather_to_root((geo, part))
    if rank == 0:
        q.put(parts)
    comm.close()


@pytest.mark.parametrize("world", [2, 4])
def test_groundwater_halo_exchange_matches_single_domain(world, port, tables):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_groundwater import gw_store, GW_OUT
    gx, gy = 37, 26
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = _free_port()
    procs = [ctx.Process(target=_gw_worker, args=(r, world, p, gx, gy, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    parts = q.get(timeout=240)
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    g = gw_store(tables, ni=gx, nj=gy, stress=0.02)
    d0 = g.a["deeprechxy"].copy()
    for call in range(3):
        port.wtable_mmf(g)
        g.a["deeprechxy"][...] = d0
    for geo, part in parts:
        for k in GW_OUT:
            want = g.a[k][geo["jts"] - 1:geo["jte"], ..., geo["its"] - 1:geo["ite"]]
            np.testing.assert_array_equal(want, part[k], err_msg="%s tile its=%d jts=%d" % (k, geo["its"], geo["jts"]))


# ------------------------------------------------------------------------------------------------
# The N > 1 path of bench.py itself: every rank cuts its memory block (tile + ring) from ONE global grid
# (synth.config3_tile), steps it and runs WTABLE_mmf_noahmp after the ZWTXY ring exchange.  Parity target: one rank.
def _cfg4_worker(rank, world, port, gx, gy, nsteps, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    from noahmp_amd.parallel import Comm
    from noahmp_amd.state import ModelConfig
    from noahmp_amd.tables import load_tables
    from oracle.portlib import PortLib
    comm = Comm(backend="gloo")
    T, tb = load_tables("usgs")
    port_ = PortLib(autobuild=False)
    port_.set_tables(T)
    cfg = ModelConfig(iopt_run=5)
    geo = comm.my_geometry(gx, gy)                          # the geometry bench.py's Run uses (partition.tile_geometry, halo 1)
    nx, ny = geo["ime"] - geo["ims"] + 1, geo["jme"] - geo["jms"] + 1
    s = synth.config3_tile(tb, gx, gy, geo["ims"] - 1, geo["jms"] - 1, nx, ny, cfg=cfg, groundwater=True)
    s.set_index(**{k: geo[k] for k in ("ids", "ide", "jds", "jde", "ims", "ime", "jms", "jme", "its", "ite", "jts", "jte")})
    synth.first_step_fixups(s)
    ring = np.ones((ny, nx), dtype=bool)
    ring[geo["jts"] - geo["jms"]:geo["jte"] - geo["jms"] + 1, geo["its"] - geo["ims"]:geo["ite"] - geo["ims"] + 1] = False
    for k in ("zwtxy", "fdepth", "topo"):                   # only the exchange may provide the ring
        s.a[k][ring] = np.nan
    s.a["isltyp"][ring] = -7
    comm.exchange_halo([torch.from_numpy(s.a[k]) for k in ("fdepth", "topo", "isltyp")], geo)
    for it in range(1, nsteps + 1):
        synth.diurnal_forcing(s, (it + 5) % 24, t_offset=s.t_offset)
        assert port_.noahmplsm(s, it, 2000, 180.0).code == 0
        comm.exchange_halo([torch.from_numpy(s.a["zwtxy"])], geo)
        port_.wtable_mmf(s)
    j0, j1 = geo["jts"] - geo["jms"], geo["jte"] - geo["jms"] + 1
    i0, i1 = geo["its"] - geo["ims"], geo["ite"] - geo["ims"] + 1
    part = {k: v[j0:j1, ..., i0:i1].copy() for k, v in s.a.items() if k != "dzs" and (k not in FIELD_INFO or FIELD_INFO[k][2] != "in")}
    parts = comm.gather_to_root((geo, part))
    if rank == 0:
        q.put(parts)
    comm.close()


@pytest.mark.parametrize("world", [2, 4, 8, 9])          # 8 = the north-star 4 x 2 grid; 9 = 3 x 3: a rank with four neighbours
def test_config4_ranks_cut_one_global_grid(world, port, tables):
    gx, gy, nsteps = 70, 130, 3                           # three row blocks of the generator, uneven tiles
    from noahmp_amd.state import ModelConfig
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = _free_port()
    procs = [ctx.Process(target=_cfg4_worker, args=(r, world, p, gx, gy, nsteps, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    parts = q.get(timeout=300)
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    g = synth.config3_tile(tables[1], gx, gy, cfg=ModelConfig(iopt_run=5), groundwater=True)
    synth.first_step_fixups(g)
    for it in range(1, nsteps + 1):
        synth.diurnal_forcing(g, (it + 5) % 24, t_offset=g.t_offset)
        assert port.noahmplsm(g, it, 2000, 180.0).code == 0
        port.wtable_mmf(g)
    assert (g.a["qslat"] != 0).any()
    for geo, part in parts:
        for k, v in part.items():
            want = g.a[k][geo["jts"] - 1:geo["jte"], ..., geo["its"] - 1:geo["ite"]]
            assert np.array_equal(want, v, equal_nan=True), "%s tile its=%d jts=%d" % (k, geo["its"], geo["jts"])


def _run_bench(extra, tmp, tag, backend="gloo"):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("NMP_DIST_BACKEND", None)
    if backend:
        env["NMP_DIST_BACKEND"] = backend                    # "gloo": several ranks on ONE GPU, host-staged ring exchange
    env.pop("WORLD_SIZE", None)
    dump = os.path.join(tmp, tag)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--ni", "96", "--nj", "130", "--steps", "5", "--warmup", "2",
                        "--no-cpu-baseline", "--dump", dump] + extra, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, r.stdout
    return json.loads(line[0]), dump


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["config4", "config3", "config4-cabi", "config4-hostfallback"])
def test_bench_two_ranks_equal_one_rank(workload, tmp_path, monkeypatch):
    """`bench.py --gpus 2` starts its two ranks itself (here both on the one GPU, ring exchange staged through the host) and
    prints n_gpus: 2; the union of the ranks' tiles equals the single-rank run of the same global grid bit for bit -- the
    config-4 groundwater step included (parity target: the single domain)."""
    extra = []
    if workload == "config4-cabi":              # the ring moved by the engine's own C-ABI exchange (socket transport, device planes)
        workload, extra = "config4", ["--halo", "tcp"]
    if workload == "config4-hostfallback":      # the mover probe_halo() falls back to when device send/recv does not work
        workload = "config4"
        monkeypatch.setenv("NMP_HALO_FORCE_HOST", "1")
    one, d1 = _run_bench(["--gpus", "1", "--workload", workload], str(tmp_path), "one")
    two, d2 = _run_bench(["--gpus", "2", "--workload", workload] + extra, str(tmp_path), "two")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    if workload == "config4":
        assert two["groundwater"]["calls"] == 5 and "ring exchange" in two["config"]["parallelism"]
        assert ("noahmp_hip_exchange_halo" in two["config"]["parallelism"]) == bool(extra)
        assert two["groundwater"]["halo_mover"]
        if os.environ.get("NMP_HALO_FORCE_HOST") == "1":
            assert "staged through the host" in two["groundwater"]["halo_mover"]
    whole = np.load(d1 + ".rank0.npz")
    seen = 0
    for r in range(2):
        part = np.load(d2 + ".rank%d.npz" % r)
        its, ite, jts, jte = part["geom"]
        for k in part.files:
            if k == "geom":
                continue
            want = whole[k][jts - 1:jte, ..., its - 1:ite]
            assert np.array_equal(want, part[k], equal_nan=True), "%s rank %d" % (k, r)
        seen += (ite - its + 1) * (jte - jts + 1)
    assert seen == 96 * 130


@pytest.mark.gpu
def test_bench_ranks_agree_when_rccl_refuses(tmp_path, tmp_path_factory):
    """`bench.py --gpus 2` with the default backend (nccl = RCCL) on a box whose ranks share one GPU: RCCL refuses a communicator
    with a duplicate device on every rank; the ranks learn of it over the gloo control plane, switch TOGETHER to host-staged edges
    and the run completes, says so (`distributed.note`) and still equals the single-rank run bit for bit.  (On a box with two GPUs the
    same command brings RCCL up and moves the ring with the engine's RCCL mover or torch's send/recv -- then that is what is checked.)"""
    whole = _one_rank_dump("config4", tmp_path_factory)
    res, dn = _run_bench(["--gpus", "2", "--workload", "config4"], str(tmp_path), "nccl2", backend=None)
    d = res["distributed"]
    assert d["halo_requested"] == "auto" and d["backend"] in ("gloo", "nccl")
    if d["backend"] == "gloo":
        assert "initialisation failed" in d["note"] and d["halo"] == "torch"
    else:
        assert d["halo"] in ("rccl", "torch")
    for r in range(2):
        part = np.load(dn + ".rank%d.npz" % r)
        its, ite, jts, jte = part["geom"]
        for k in part.files:
            if k != "geom":
                assert np.array_equal(whole[k][jts - 1:jte, ..., its - 1:ite], part[k], equal_nan=True), "%s rank %d" % (k, r)


_ONE_RANK = {}


def _one_rank_dump(workload, tmp_path_factory):
    """the single-rank run of the global grid, once per workload and test session"""
    if workload not in _ONE_RANK:
        d = str(tmp_path_factory.mktemp("one_" + workload))
        one, dump = _run_bench(["--gpus", "1", "--workload", workload], d, "one")
        assert one["n_gpus"] == 1
        _ONE_RANK[workload] = np.load(dump + ".rank0.npz")
    return _ONE_RANK[workload]


@pytest.mark.gpu
@pytest.mark.parametrize("world,workload,halo", [(8, "config4", "torch"), (8, "config4", "tcp"), (9, "config4", "torch"), (9, "config4", "tcp"),
                                                 (8, "config3", "torch"), (8, "config5", "torch")])
def test_bench_north_star_tiling_equals_one_rank(world, workload, halo, tmp_path, tmp_path_factory):
    """The north-star decomposition on the HIP path: `bench.py --gpus 8` = the 4 x 2 rank grid of mpp_land_get_nprocsxy
    (mpp:124-141), `--gpus 9` = 3 x 3, whose centre rank has four edge neighbours and four diagonal ones (the corner cells of the
    9-point LATERALFLOW stencil, gw:264-286, arrive through the two-phase order of mpp_land_comlr_real / comub_real, mpp:344-369,
    603-613).  All ranks share the one GPU of the box (NMP_DIST_BACKEND=gloo: the ring is staged through the host, or moved by the
    engine's own C-ABI exchange over its socket transport); every rank runs the device kernels on its own tile + ring.  Every INOUT /
    OUT array of every tile equals the single-rank run of the same global grid bit for bit after 2 + 5 steps -- the config-4
    groundwater step (ring exchange every step) included; config 3 / config 5: the collective-free split of the same rank grid."""
    gx, gy = (96, 130) if workload != "config5" else (120, 66)
    common = ["--workload", workload] + (["--ni", str(gx), "--nj", str(gy)] if workload == "config5" else [])
    if workload == "config5":
        one, d1 = _run_bench(["--gpus", "1"] + common, str(tmp_path), "one5")
        whole = np.load(d1 + ".rank0.npz")
    else:
        whole = _one_rank_dump(workload, tmp_path_factory)
    res, dn = _run_bench(["--gpus", str(world)] + common + (["--halo", halo] if workload == "config4" else []), str(tmp_path), "many")
    assert res["n_gpus"] == world and res["scaling"] == "strong"
    # the N > 1 line is self-contained: the N = 1 point of the SAME workload (rank 0, whole grid, same steps) and the efficiency against it
    n1 = res["n1_reference"]
    assert n1["steps"] == res["steps"] and n1["value"] > 0 and ("--workload %s" % workload) in n1["workload"]
    assert abs(res["strong_scaling_efficiency"] - res["value"] / (world * n1["value"])) < 1e-12
    lo, hi = res["kernel_ms_per_step_min_max_over_ranks"]
    assert 0 < lo <= hi
    if workload == "config4":
        assert res["groundwater"]["calls"] == 5 and "ring exchange" in res["config"]["parallelism"]
        assert ("noahmp_hip_exchange_halo" in res["config"]["parallelism"]) == (halo == "tcp")
        assert res["halo_mover"] == res["groundwater"]["halo_mover"] and res["halo_ms_per_call"] > 0 and n1["groundwater_calls"] == 5
    else:
        assert res["halo_ms_per_call"] is None and "no exchange" in res["halo_mover"]
    from noahmp_amd.partition import tile_geometry, neighbours
    seen, most_nb = 0, 0
    for r in range(world):
        part = np.load(dn + ".rank%d.npz" % r)
        its, ite, jts, jte = part["geom"]
        g = tile_geometry(gx, gy, world, r, halo=0)
        assert (its, ite, jts, jte) == (g["its"], g["ite"], g["jts"], g["jte"])
        most_nb = max(most_nb, sum(1 for v in neighbours(r, world).values() if v >= 0))
        for k in part.files:
            if k == "geom":
                continue
            want = whole[k][jts - 1:jte, ..., its - 1:ite]
            assert np.array_equal(want, part[k], equal_nan=True), "%s rank %d of %d" % (k, r, world)
        seen += (ite - its + 1) * (jte - jts + 1)
    assert seen == gx * gy
    assert most_nb == (4 if world == 9 else 3)              # 4 x 2: an inner rank has three edge neighbours; 3 x 3: the centre has four


# ------------------------------------------------------------------------------------------------
# The C-ABI halo exchange (noahmp_hip_halo_init / noahmp_hip_exchange_halo, socket transport) on host planes: no torch.distributed,
# no gloo, no GPU -- what a Fortran / MPI caller binds.  Parity target: the ring cells equal the neighbours' tile cells of ONE
# global field, corners included (mpp_land_comlr_real + comub_real flag 99, mpp:344-369, 603-642).
def _cabi_worker(rank, world, port, gx, gy, q):
    import ctypes as C
    from noahmp_amd import abi
    from noahmp_amd.partition import tile_geometry
    lib = abi.load_library()
    rc = lib.noahmp_hip_halo_init(rank, world, b"127.0.0.1", port, abi.HALO_TCP)
    assert rc == 0, lib.noahmp_hip_last_error().decode()
    geo = tile_geometry(gx, gy, world, rank, halo=1)
    y, x = np.meshgrid(np.arange(gy), np.arange(gx), indexing="ij")
    gf = (1000.0 * y + x).astype(np.float32)                 # global fields: the value names the cell
    gi = (7 * y + 3 * x).astype(np.int32)
    sl = (slice(geo["jms"] - 1, geo["jme"]), slice(geo["ims"] - 1, geo["ime"]))
    f, i = gf[sl].copy(), gi[sl].copy()
    ring = np.ones(f.shape, dtype=bool)
    ring[geo["jts"] - geo["jms"]:geo["jte"] - geo["jms"] + 1, geo["its"] - geo["ims"]:geo["ite"] - geo["ims"] + 1] = False
    f[ring] = np.nan
    i[ring] = -1
    idx = (C.c_int32 * 8)(*[geo[k] for k in ("ims", "ime", "jms", "jme", "its", "ite", "jts", "jte")])
    for _ in range(3):                                       # repeated calls reuse links and staging buffers
        ptrs = (C.c_void_p * 2)(f.ctypes.data, i.ctypes.data)
        rc = lib.noahmp_hip_exchange_halo(2, ptrs, idx, abi.MEM_HOST, None)
        assert rc == 0, lib.noahmp_hip_last_error().decode()
    ok = bool(np.array_equal(f, gf[sl]) and np.array_equal(i, gi[sl]))
    lib.noahmp_hip_halo_finalize()
    q.put((rank, ok, int(ring.sum())))


@pytest.mark.parametrize("world", [2, 4, 6, 8, 9])       # 8 = 4 x 2 (north star), 9 = 3 x 3 (the centre rank has four neighbours)
def test_cabi_halo_exchange_on_host_planes(world):
    gx, gy = 41, 29
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = _free_port()
    procs = [ctx.Process(target=_cabi_worker, args=(r, world, p, gx, gy, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(n > 0 for _, _, n in res)


def _cabi_dead_rank_worker(rank, world, port, q):
    import time
    from noahmp_amd import abi
    os.environ["NMP_HALO_TIMEOUT_S"] = "4"
    lib = abi.load_library()
    t0 = time.time()
    rc = lib.noahmp_hip_halo_init(rank, world, b"127.0.0.1", port, abi.HALO_TCP)
    q.put((rank, rc, time.time() - t0, lib.noahmp_hip_last_error().decode()))


@pytest.mark.parametrize("dead", [2, 0])
def test_cabi_halo_init_ends_when_a_rank_never_arrives(dead):
    """A rank that dies before the rendezvous (here: is never started) ends every other rank's noahmp_hip_halo_init with an error
    within the time limit (NMP_HALO_TIMEOUT_S) -- accept() and the table transfer have deadlines, nothing hangs, nothing leaks --
    whether the missing rank is a neighbour (2) or the master itself (0)."""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = _free_port()
    procs = [ctx.Process(target=_cabi_dead_rank_worker, args=(r, world, p, q)) for r in range(world) if r != dead]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=120) for _ in procs]
    for pr in procs:
        pr.join(30)
        assert pr.exitcode == 0
    assert all(rc != 0 for _, rc, _, _ in res), res
    assert all(t < 40.0 for _, _, t, _ in res), res
    assert all(msg for _, _, _, msg in res), res


@pytest.mark.gpu
def test_bench_config4_sorted_equals_tile_order(tmp_path):
    """Config 4 on the sorted layout (groundwater planes returned to (i,j) order around every WTABLE_mmf_noahmp call,
    noahmp_hip_sorted_exchange) gives the bits of the tile-order run: column step, groundwater step and accumulators."""
    a, da = _run_bench(["--gpus", "1", "--workload", "config4"], str(tmp_path), "sorted")
    b, db = _run_bench(["--gpus", "1", "--workload", "config4", "--no-sort"], str(tmp_path), "tile")
    assert "sort" in a and "sort" not in b
    x, y = np.load(da + ".rank0.npz"), np.load(db + ".rank0.npz")
    assert set(x.files) == set(y.files) and "qslat" in x.files
    for k in x.files:
        assert np.array_equal(x[k], y[k], equal_nan=True), k
    assert (x["qslat"] != 0).any() and (x["isnowxy"] < 0).any()


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["config4", "config3", "config4-dt900", "config3-prefetch", "config4-prefetch"])
def test_bench_run_equals_the_oracle(workload, tmp_path, port, tables):
    """The run bench.py times (small grid: column step on the sorted layout with the per-step forcing permutation; config 4: plus
    WTABLE_mmf_noahmp every STEPWTD steps on planes returned to (i,j) order) against the oracle advancing the same global grid in
    tile order with the same forcing sequence -- every INOUT / OUT array bit for bit after 7 steps.  config4-dt900: DT = 900 s, so
    STEPWTD = nint(30 min / DT) = 2 (hdrv:247-248, 420) and the sub-hourly DT of the column step (NITER, FACT, accumulators)."""
    from noahmp_amd.state import ModelConfig
    import bench
    gx, gy = 96, 130
    dt = 900.0 if workload.endswith("dt900") else 3600.0
    extra = ["--prefetch"] if workload.endswith("prefetch") else []      # step n + 1's forcing permuted on a second stream beside step n's kernel
    workload = workload.split("-")[0]
    res, dump = _run_bench(["--gpus", "1", "--workload", workload, "--dt", str(dt)] + extra, str(tmp_path), "gpu")
    lateral = workload == "config4"
    stepwtd = max(int(30.0 * 60.0 / dt + 0.5), 1)
    if lateral:
        assert res["groundwater"]["stepwtd"] == stepwtd and res["groundwater"]["calls"] == len([i for i in range(3, 8) if i % stepwtd == 0])
    g = synth.config3_tile(tables[1], gx, gy, cfg=ModelConfig(iopt_run=5 if lateral else 1, dt=dt), groundwater=lateral)
    synth.first_step_fixups(g)
    for it in range(1, 8):                                   # 2 warm-up + 5 timed steps, hours as bench.py's Run.step
        synth.diurnal_forcing(g, bench.forcing_hour(it, dt), t_offset=g.t_offset)
        assert port.noahmplsm(g, it, 2000, 180.0).code == 0
        if lateral and it % stepwtd == 0:
            port.wtable_mmf(g)
    got = np.load(dump + ".rank0.npz")
    names = [k for k in got.files if k != "geom"]
    assert "tslb" in names and "isnowxy" in names and (not lateral or "qslat" in names)
    for k in names:
        assert np.array_equal(g.a[k], got[k], equal_nan=True), k


@pytest.mark.gpu
def test_bench_config5_ranks_and_oracle(tmp_path, port, tables):
    """`bench.py --workload config5` (BASELINE configs[4]: tile of the global lat/lon grid, cold start on the device, forcing
    interpolation -> CALC_DECLIN forcing preparation -> column step per hour, sorted layout): two ranks on the one GPU give the bits
    of one rank, and the single-rank run equals the oracle advancing the same grid through the same chain from the same raw state
    (forcing records evaluated by the same device code, tile order) -- every INOUT / OUT array after 2 + 5 steps."""
    import torch
    from noahmp_amd import synth5
    gx, gy = 120, 66
    common = ["--workload", "config5", "--ni", str(gx), "--nj", str(gy)]
    one, d1 = _run_bench(["--gpus", "1"] + common, str(tmp_path), "one5")
    two, d2 = _run_bench(["--gpus", "2"] + common + ["--prefetch"], str(tmp_path), "two5")      # (the ranks: forcing chain three steps ahead on other streams)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and "no collective" in two["config"]["parallelism"]
    assert "three steps ahead" in two["config"]["workload"] and "three steps ahead" not in one["config"]["workload"]
    assert "configs[4]" in one["config"]["workload"] and one["cold_start_s"] > 0
    whole = np.load(d1 + ".rank0.npz")
    seen = 0
    for r in range(2):
        part = np.load(d2 + ".rank%d.npz" % r)
        its, ite, jts, jte = part["geom"]
        for k in part.files:
            if k != "geom":
                assert np.array_equal(whole[k][jts - 1:jte, ..., its - 1:ite], part[k], equal_nan=True), "%s rank %d" % (k, r)
        seen += (ite - its + 1) * (jte - jts + 1)
    assert seen == gx * gy
    # the oracle through the same chain
    raw, lon, static = synth5.config5_tile(gx, gy)
    rc, _ = port.noahmp_init(raw, fndsnowh=True)
    assert rc == 0
    dev = torch.device("cuda", 0)
    recs = synth5.Records(torch.from_numpy(raw.a["xlatin"]).to(dev), torch.from_numpy(lon).to(dev),
                          {k: torch.from_numpy(v).to(dev) for k, v in static.items()})
    host = lambda rec: {k: (v.cpu().numpy().copy() if v is not None else None) for k, v in rec.items()}
    rain = np.zeros((gy, gx), np.float32)
    for n in range(7):
        ri, k = divmod(n, synth5.RECORD_HOURS)
        port.forcing_interpolate(raw, host(recs.at(ri)), host(recs.at(ri + 1)) if k else None, 3600 * k, 3600 * synth5.RECORD_HOURS, rain)
        iday, ihour = synth5.step_time(n)
        jul = port.forcing_prep(raw, lon, rain, iday, ihour, first_step=(n == 0))
        assert port.noahmplsm(raw, n + 1, 2000, jul).code == 0
    names = [k for k in whole.files if k != "geom"]
    assert "tslb" in names and (whole["isnowxy"] < 0).any() and (raw.a["ivgtyp"] == 24).any()
    for k in names:
        assert np.array_equal(raw.a[k], whole[k], equal_nan=True), k


@pytest.mark.gpu
def test_rccl_plumbing_selftest_on_one_gpu():
    """The RCCL transport of noahmp_hip_exchange_halo needs one GPU per rank; what one GPU can check is its plumbing: librccl
    resolves (dlopen, beside torch's copy), ncclCommInitRank with a unique id, ncclSend / ncclRecv in one group on the engine's
    stream (to self) return the data."""
    from noahmp_amd import abi
    lib = abi.load_library()
    rc = lib.noahmp_hip_halo_selftest_rccl(25000)
    assert rc == 0, lib.noahmp_hip_last_error().decode()

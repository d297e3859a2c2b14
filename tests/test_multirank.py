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

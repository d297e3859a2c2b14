"""Tile decomposition: one MPI rank (here: one process) per GPU, the reference's block rule.

Mirrors mpp/module_mpp_land.F90: ``mpp_land_get_nprocsxy`` (:124-141, most-square factorisation,
first best wins) and ``mpp_land_partition_calc`` (:227-288: base = n/np, the first mod(n,np) ranks
along each axis get one extra cell, contiguous blocks, rank = iprocy*nprocx + iprocx).
Columns are independent for every option except opt_run=5, so no data-path collective exists;
the only exchange the physics can need is the 1-cell ZWTXY ring of LATERALFLOW.
"""


def nprocs_xy(nproc):
    best = nproc
    nx, ny = nproc, 1
    for j in range(1, nproc + 1):
        if nproc % j == 0:
            i = nproc // j
            if abs(i - j) < best:
                best = abs(i - j)
                nx, ny = i, j
    return nx, ny


def partition(global_nx, global_ny, nproc):
    """-> list over ranks of dict(startx, starty, nx, ny) with 1-based Fortran starts."""
    npx, npy = nprocs_xy(nproc)
    base_nx, base_ny = global_nx // npx, global_ny // npy
    out = []
    for rank in range(nproc):
        ipx, ipy = rank % npx, rank // npx
        nx = base_nx + 1 if ipx <= (global_nx % npx) - 1 else base_nx
        ny = base_ny + 1 if ipy <= (global_ny % npy) - 1 else base_ny
        out.append(dict(nx=nx, ny=ny, ipx=ipx, ipy=ipy))
    for r in out:
        r["startx"] = 1 + sum(o["nx"] for o in out if o["ipy"] == 0 and o["ipx"] < r["ipx"])
        r["starty"] = 1 + sum(o["ny"] for o in out if o["ipx"] == 0 and o["ipy"] < r["ipy"])
    return out


def neighbours(rank, nproc):
    """left/right/down/up rank ids (or -1), mpp:93-107."""
    npx, npy = nprocs_xy(nproc)
    ipx, ipy = rank % npx, rank // npx
    return dict(left=rank - 1 if ipx > 0 else -1, right=rank + 1 if ipx < npx - 1 else -1,
                down=rank - npx if ipy > 0 else -1, up=rank + npx if ipy < npy - 1 else -1)


# the eight ring neighbours in the order the one-phase exchange visits them (edge types first; inside a type the lower-ranked
# peer first, so that blocking pairwise transfers follow one global order of the links and cannot wait in a cycle)
DIRS8 = (("left", -1, 0), ("right", 1, 0), ("down", 0, -1), ("up", 0, 1),
         ("down_left", -1, -1), ("up_right", 1, 1), ("down_right", 1, -1), ("up_left", -1, 1))


def neighbours8(rank, nproc):
    """The four edge neighbours of mpp:93-107 plus the four diagonal ranks whose corner cell the 9-point LATERALFLOW stencil reads
    (gw:264-286): name -> rank id or -1, in DIRS8 order."""
    npx, npy = nprocs_xy(nproc)
    ipx, ipy = rank % npx, rank // npx
    out = {}
    for name, dx, dy in DIRS8:
        x, y = ipx + dx, ipy + dy
        out[name] = y * npx + x if (0 <= x < npx and 0 <= y < npy) else -1
    return out


def tile_geometry(global_nx, global_ny, nproc, rank, halo=1):
    """WRF-style index block of one rank: domain ids..jde = the global grid, tile its..jte = the rank's
    block, memory ims..jme = the tile plus `halo` cells towards every side that has a neighbour (so the
    LATERALFLOW ring of gw:231-234 is addressable).  All 1-based inclusive."""
    t = partition(global_nx, global_ny, nproc)[rank]
    its, jts = t["startx"], t["starty"]
    ite, jte = its + t["nx"] - 1, jts + t["ny"] - 1
    return dict(ids=1, ide=global_nx, jds=1, jde=global_ny, kds=1, kde=2,
                ims=max(its - halo, 1), ime=min(ite + halo, global_nx),
                jms=max(jts - halo, 1), jme=min(jte + halo, global_ny), kms=1, kme=2,
                its=its, ite=ite, jts=jts, jte=jte, kts=1, kte=1)

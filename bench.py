#!/usr/bin/env python3
"""Headline benchmark: column-steps/s of the Noah-MP column engine on MI355X.

A "step" is one noahmplsm call (reference drv:11) over one batch of synthetic columns that is already resident in HBM.

N = 1 workload (default) = BASELINE.json configs[2]: the CONUS-1-km-like grid, 4608 x 1536 = 7 077 888 columns, 4 soil /
up to 3 snow layers (30 % of the columns carry snow: ISNOW 0..-3), 2 % urban, 1 % land ice, the reference's namelist options;
state device-resident and sorted by (class, vegetation type, snow-layer count, skin-temperature bin) on the device, re-sorted
when snow layers appear or vanish, forcing delivered in tile order and permuted into the sorted working set every step --
sort, staleness checks and permutation all INSIDE the timed region.

N > 1 workload (default) = BASELINE.json configs[3]: the SAME grid with OPT_RUN = 5 cut into N tiles by the reference's
mpp_land_partition_calc rule (mpp:227-288; 4 x 2 at N = 8), one process per GPU; every STEPWTD steps WTABLE_mmf_noahmp
(gw:14) runs after the 1-cell ZWTXY ring has been exchanged between neighbouring ranks (RCCL send/recv over xGMI) -- the only
data-path exchange.  Total work is fixed: STRONG scaling.  `--workload config4 --gpus 1` gives the N = 1 point of that curve,
`--workload config3 --gpus N` the collective-free split of the N = 1 workload, `--workload config2` round 1's 1 M-column case,
`--workload config5 [--gpus N]` BASELINE.json configs[4]: the global 0.1-degree grid cut into N tiles, cold start on the device, then
forcing interpolation -> forcing preparation with CALC_DECLIN's zenith angle -> column step per hour (no collective; weak in
nothing: the grid is fixed, STRONG scaling).  `--dt 900` runs config 4 with STEPWTD = 2 (hdrv:247-248).

`python bench.py --gpus N` without a torchrun environment starts the N ranks itself (children are started before anything
touches the GPU).  Prints ONE JSON line on rank 0 (metric contract of the driver) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# Set-up copies of whole 7 M-column arrays go through torch's pageable copies: keep the HIP runtime from page-locking such buffers in place
# (its cached mappings of heap memory fault later on this stack: tests/conftest.py, profiles/r05_experiments.md section 3).  Nothing inside a
# timed region copies from or to pageable memory except the first host_path_reference leg, which says so.
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")
# torch's and the engine's streams share the runtime's hardware queues (4 by default, in-order each): with 8 the run's stream, the engine's
# second stream (land-ice / skipped-cell kernels beside the land kernel) and the optional prefetch streams never sit behind one another
# (profiles/r05_experiments.md section 4; no effect on the default config-3 / 4 / 5 runs, 0.2 ms per step in one --prefetch mapping)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_COLSTEP = 824          # 87 reads + 119 writes x 4 B at the noahmplsm ABI
GW_BYTES_PER_CELL = 216              # gw_head_kernel 24 B + gw_column_kernel 192 B per WTABLE_mmf_noahmp call
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: 8 TB/s spec
SIMDS, CLOCK_GHZ = 1024, 2.4         # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz; a wave64 VALU op issues over 2 cycles
FKEYS = ("coszin", "swdown", "glw", "t3d", "rainbl")     # what the diurnal forcing changes from hour to hour


def forcing_hour(it, dt=3600.0):
    """Local hour (0..23) of the synthetic diurnal forcing at 1-based step `it`: step 1 is 06:00."""
    return int((it - 1) * dt / 3600.0 + 6.0) % 24


# ------------------------------------------------------------------------------------------------ CPU leg
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cpus():
    """What this process may use of the host: CPUs in its affinity mask, the cgroup CPU quota (cpu.max), physical cores, SMT."""
    info = {"logical_cpus": os.cpu_count() or 1}
    try:
        info["affinity_cpus"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        info["affinity_cpus"] = info["logical_cpus"]
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    info["cgroup_cpu_quota"] = quota
    cores = set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    info["physical_cores"] = len(cores) or None
    usable = info["affinity_cpus"]
    if quota:
        usable = max(1, min(usable, int(quota + 0.5)))
    info["usable_cpus"] = usable
    return info


def cpu_baseline(workload, ncol=16384, nsteps=36, budget_s=60.0):
    """Time the CPU path on this box's host cores: the compiled reference (oracle/_ref, -O2) when its .so travelled with the
    repo, else the C restatement.  The reference has no threads, so the node-level figure is P independent processes,
    each on its own `ncol`-column sample of the bench workload (same generator, same class / snow mix), `nsteps` steps (three passes over
    a day of every second hour: ~1.4 s per process, ~20 CPU-seconds at the largest process count),
    started together behind a barrier; rate = P x ncol x nsteps / (last end - first start).  P sweeps 1, 2, 4, ... up to the CPUs
    this process may use (affinity mask and cgroup quota, not os.cpu_count()); the best aggregate is reported with the process
    count that gave it, and the whole sweep beside it so that the scaling over P can be read (bounded: `budget_s` of wall clock)."""
    import multiprocessing as mp
    from oracle import reflib
    kind = "reference" if reflib.available("O2") else "port"
    hw = host_cpus()
    pmax = max(1, min(hw["usable_cpus"], 256))
    counts = sorted({p for p in (1, 2, 4, 8, 16, 32, 48, 64, 96, 128, 192, 256) if p < pmax} | {pmax})
    ctx = mp.get_context("fork")
    sweep = []
    t_begin = time.perf_counter()
    for P in counts:
        if sweep and time.perf_counter() - t_begin > budget_s:
            break
        barrier = ctx.Barrier(P)
        q = ctx.Queue()
        procs = [ctx.Process(target=_cpu_worker, args=(kind, workload, ncol, nsteps, r, barrier, q)) for r in range(P)]
        for pr in procs:
            pr.start()
        res = [q.get(timeout=600) for _ in range(P)]
        for pr in procs:
            pr.join()
        wall = max(r[1] for r in res) - min(r[0] for r in res)
        per_proc = sorted(ncol * nsteps / r[2] for r in res)
        sweep.append({"processes": P, "value": P * ncol * nsteps / wall, "per_process_median": per_proc[len(per_proc) // 2],
                      "per_process_min": per_proc[0], "wall_s": wall})
    best = max(sweep, key=lambda e: e["value"])
    one = sweep[0]
    what = {"config2": "config-2", "config3": "config-3 (30 % snow, 2 % urban, 1 % land ice)",
            "config4": "config-4 (config-3 mix, OPT_RUN=5; column step only)"}[workload]
    # why P processes do not give P times one process: fewer usable CPUs than logical ones (quota / affinity), two hardware threads
    # per core (the per-process rate halves once P exceeds the physical cores), lower clocks with all cores busy
    return {"value": best["value"], "unit": "column-steps/s", "cores": best["processes"], "kind": kind,
            "cpu_model": cpu_model(), "host": hw,
            "single_process": one["value"], "per_process_at_best": best["per_process_median"],
            "speedup_over_single_process": best["value"] / one["value"],
            "sweep": sweep,
            "sample": "%d processes (best of the sweep %s; CPUs usable by this process: %d of %d logical, %s physical cores, cgroup quota %s) "
                      "x %d columns x %d steps (every second hour of the diurnal cycle) of the %s workload (%s, float32), the noahmplsm call loop"
                      % (best["processes"], [e["processes"] for e in sweep], hw["usable_cpus"], hw["logical_cpus"], hw["physical_cores"],
                         hw["cgroup_cpu_quota"], ncol, nsteps, what,
                         "reference Fortran flang -O2" if kind == "reference" else "C restatement gcc -O2")}


def _cpu_worker(kind, workload, ncol, nsteps, r, barrier, q):
    from noahmp_amd import synth
    from noahmp_amd.state import ModelConfig
    from noahmp_amd.tables import load_tables
    T, tb = load_tables("usgs")
    if workload == "config2":
        s = synth.config2(tb, ni=ncol // 8, nj=8, seed=100 + r, cfg=ModelConfig(idveg=1))
    else:
        s = synth.config3(tb, ni=ncol // 8, nj=8, seed=100 + r, cfg=ModelConfig(iopt_run=5 if workload == "config4" else 1))
        if workload == "config4":
            synth.groundwater_fields(s, tb, seed=200 + r)
    synth.first_step_fixups(s)
    if kind == "reference":
        from oracle.reflib import RefLib
        lib = RefLib("O2")
    else:
        from oracle.portlib import PortLib
        lib = PortLib(autobuild=False)
    lib.set_tables(T)
    forcing = []
    for it in range(1, nsteps + 1):                                          # forcing prep is not timed
        synth.diurnal_forcing(s, (2 * it + 4) % 24, t_offset=s.t_offset)    # every second hour: a whole day per 12 steps
        forcing.append({k: s.a[k].copy() for k in FKEYS})
    barrier.wait()
    t0 = time.perf_counter()
    calls = 0.0
    for it in range(1, nsteps + 1):
        for k in FKEYS:
            s.a[k][...] = forcing[it - 1][k]
        c0 = time.perf_counter()
        lib.noahmplsm(s, it, 2000, 180.0)                                    # the noahmplsm call
        calls += time.perf_counter() - c0
    t1 = time.perf_counter()
    q.put((t0, t1, calls))


# ------------------------------------------------------------------------------------------------ N ranks without torchrun
def self_launch(args, argv):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as children -- this process never touches the GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    # wait for all ranks; if one dies the others would wait for it in a collective for ever: end them (exact PIDs)
    rc, alive = 0, list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            r = p.poll()
            if r is None:
                continue
            alive.remove(p)
            rc = max(rc, abs(r))
        if rc and alive:
            time.sleep(5.0)                    # let the others report their own error first
            for p in alive:
                if p.poll() is None:
                    p.terminate()
            for p in alive:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
    return rc


# ------------------------------------------------------------------------------------------------ workloads
GW_ONLY_OUT = ("qrf", "qspring", "qslat", "qrfs", "qsprings")
GW_LATERAL = ("zwtxy", "fdepth", "topo", "isltyp", "xland", "xice", "ivgtyp", "area")     # what the QLAT stencil reads, in (i,j) order


class Run:
    """One rank's share of a workload: the constructor builds the device-resident state, step(it) enqueues one timestep,
    collect() waits and returns the tallies.

    Sorted layout (default): the rank's tile lives in HBM sorted by (class, vegetation type, snow-layer count, TSK bin); forcing
    arrives in tile order and is permuted per step.  With OPT_RUN = 5 (config 4) the groundwater planes additionally live in
    a tile-order memory block that carries the 1-cell ring: around every WTABLE_mmf_noahmp call the six planes it shares with the
    column step return to (i,j) order, the ZWTXY ring is exchanged, the stencil runs, and the planes go back to sorted order.
    Round 3: only the QLAT stencil needs the (i,j) neighbourhood, so WTABLE_mmf_noahmp runs in two halves
    (noahmp_hip_wtable_lateral_async on the tile-order ZWTXY, noahmp_hip_wtable_columns_async on the sorted store): ONE plane travels
    to (i,j) order (ZWTXY) and ONE back (QLAT) per call instead of twelve each way."""

    def __init__(self, args, workload, comm, eng, tb, dev):
        import numpy as np
        import torch
        from noahmp_amd import synth
        from noahmp_amd.partition import tile_geometry
        from noahmp_amd.state import ColumnStore, DeviceColumnStore, ModelConfig, GW_ALIAS, GW_EXTRA
        self.torch, self.args, self.workload, self.comm, self.eng, self.dev = torch, args, workload, comm, eng, dev
        self.lateral = workload == "config4"
        self.sorted = not args.no_sort
        gx, gy = args.ni, args.nj
        if workload == "config2":
            cfg = ModelConfig(idveg=1, dt=args.dt)
        else:
            cfg = ModelConfig(**dict(dict(iopt_run=5 if self.lateral else 1, idveg=args.dveg, dt=args.dt), **(getattr(args, "opts", None) or {})))
        self.cfg = cfg
        geom = tile_geometry(gx, gy, comm.world, comm.rank, halo=1 if self.lateral else 0)
        self.geom = geom
        nx, ny = geom["ime"] - geom["ims"] + 1, geom["jme"] - geom["jms"] + 1
        if workload == "config2":
            assert comm.world == 1, "config2 is the single-GPU case of round 1"
            s = synth.config2(tb, ni=gx, nj=gy, seed=2, cfg=cfg)
        else:       # this rank's memory block (tile + ring) cut from ONE global grid
            s = synth.config3_tile(tb, gx, gy, geom["ims"] - 1, geom["jms"] - 1, nx, ny, cfg=cfg, groundwater=self.lateral)
        idx_keys = ("ids", "ide", "jds", "jde", "ims", "ime", "jms", "jme", "its", "ite", "jts", "jte")
        s.set_index(**{k: geom[k] for k in idx_keys})
        synth.first_step_fixups(s)
        self.i_off, self.j_off = geom["its"] - geom["ims"], geom["jts"] - geom["jms"]
        nti, ntj = geom["ite"] - geom["its"] + 1, geom["jte"] - geom["jts"] + 1
        self.tile_cells = nti * ntj
        self.gw = None
        self.tsk_bin = None
        self.block_forcing = None
        if self.lateral and self.sorted:
            # forcing arrives shaped like the rank's memory block (tile + ring), as the groundwater planes are: its permutation into
            # the sorted working set then shares a launch with the return of the groundwater planes to sorted order
            self.block_forcing = []
            for h in range(24):
                synth.diurnal_forcing(s, h, t_offset=s.t_offset)
                self.block_forcing.append({k: torch.from_numpy(s.a[k].copy()).to(dev) for k in FKEYS})
            # the groundwater planes stay in a tile-order block with the ring; the column state is the tile without it
            gwb = DeviceColumnStore.__new__(DeviceColumnStore)
            gwb.ni, gwb.nj, gwb.cfg, gwb.device, gwb.idx = s.ni, s.nj, cfg, torch.device(dev), dict(s.idx)
            gwb.a = {k: torch.from_numpy(s.a[k]).to(dev) for k in GW_LATERAL}       # what the stencil half reads (tile + ring, (i,j) order)
            gwb.a["qlat"] = torch.zeros((s.nj, s.ni), dtype=torch.float32, device=dev)
            gwb.a["dzs"] = s.a["dzs"].copy()
            self.gw = gwb
            inner = ColumnStore(nti, ntj, cfg).add_groundwater()                    # the sorted store carries every per-column MMF plane
            inner.a["qlat"] = np.zeros((ntj, nti), dtype=np.float32)
            for k in inner.a:
                if k not in ("dzs", "qlat"):
                    inner.a[k][...] = s.a[k][self.j_off:self.j_off + ntj, ..., self.i_off:self.i_off + nti]
            inner.t_offset = s.t_offset[self.j_off:self.j_off + ntj, self.i_off:self.i_off + nti].copy()
            inner.set_index(**dict({k: geom[k] for k in idx_keys}, ims=geom["its"], ime=geom["ite"], jms=geom["jts"], jme=geom["jte"]))
            s = inner
        self.ni, self.nj = s.ni, s.nj
        # 24 hourly forcing sets in tile order, resident in HBM (as a driver would have staged them)
        self.forcing = []
        for h in range(24):
            synth.diurnal_forcing(s, h, t_offset=s.t_offset)
            self.forcing.append(self.block_forcing[h] if self.block_forcing else {k: torch.from_numpy(s.a[k].copy()).to(dev) for k in FKEYS})
        self.d = d = s.to_device(dev)
        self.stepwtd = max(int(cfg.wtddt * 60.0 / cfg.dt + 0.5), 1)                 # hdrv:1227 NINT
        self.ts = torch.cuda.Stream(device=dev)                                     # every kernel and exchange of the run
        self.sp = self.ts.cuda_stream
        self.reset_counters()
        halo_store = self.gw if self.gw is not None else d
        self.halo_mover = None
        if self.lateral:
            self.halo_mover = comm.probe_halo()                                      # agree on a working mover before the first exchange
            self.wargs = self._lateral_args() if self.gw is not None else halo_store.wtable_args()
            with torch.cuda.stream(self.ts):                                         # static planes of the stencil: once
                comm.exchange_halo([halo_store.a["fdepth"], halo_store.a["topo"]], geom)
                comm.exchange_halo([halo_store.a["isltyp"]], geom)
            self.ts.synchronize()
        if self.sorted:
            self.tsk_bin = args.tsk_bin if args.tsk_bin is not None else (4.0 if self.lateral else 1.0)
            self.sort_kw = dict(tsk_bin=self.tsk_bin, allow_lateral=True, snow_first=args.snow_first, veg=not args.no_veg_key,
                                snow=not args.no_snow_key, tair=args.tair_key)
            self.perm = eng.sort_store(d, **self.sort_kw)
            self._bind_sorted()
        else:
            self.sargs = []
            for h in range(24):                                                      # a step just points the block at the hour's set
                d.a.update(self.forcing[h])
                self.sargs.append(d.step_args(1, 2000, 180.0))

    def _lateral_args(self):
        """noahmp_wtable_args of the (i,j)-order block for the stencil half: only GW_LATERAL is read; the other members point at ZWTXY."""
        from noahmp_amd.abi import WtableArgs
        from noahmp_amd.abi_spec import WTABLE_FIELDS
        from noahmp_amd.state import GW_ALIAS
        gw, cfg = self.gw, self.cfg
        w = WtableArgs()
        scal = dict(nsoil=cfg.nsoil, xice_threshold=cfg.xice_thres, isice=cfg.isice, wtddt=cfg.wtddt, isurban=cfg.isurban)
        scal.update(gw.index())
        for n, k, lev, io, ln in WTABLE_FIELDS:
            if k in ("pf", "pi"):
                name = GW_ALIAS.get(n, n)
                setattr(w, n, gw.ptr(name if name in gw.a else "zwtxy"))
            else:
                setattr(w, n, scal[n])
        return w

    def _bind_sorted(self):
        d, eng = self.d, self.eng
        self.forcing_ready_for = None                         # a forcing set permuted ahead dies with the old column order
        if self.gw is not None:
            self.wargs_col = d.wtable_args()                  # the per-column half works on the sorted store itself
        self.work = {k: d.a[k] for k in FKEYS}
        src0 = [self.work[k] for k in FKEYS] if self.block_forcing else [self.forcing[0][k] for k in FKEYS]   # (plan only; sources are set per step)
        # T3D has two levels in memory (HRLDAS passes kms:kme = 1:2) and noahmplsm reads level 1: only that level is permuted
        self.lvl1 = tuple(i for i, k in enumerate(FKEYS) if self.work[k].dim() == 3)
        self.scat = eng.scatter([self.work[k] for k in FKEYS], src0, self.perm, self.ni, self.nj, first_level_only=self.lvl1)
        self.sarg = d.step_args(1, 2000, 180.0)
        # Forcing prefetch: a second forcing working set, so that the permutation of step n + 1's
        # forcing runs on a second stream BESIDE step n's column kernel (the kernel is issue-bound at ~21 % of HBM, the permutation
        # memory-bound) instead of in front of it.  Two event waits per step order the two streams (read-after-write on the set the
        # kernel is about to read, write-after-read on the set the previous kernel read).
        self.work2 = None
        self.prefetched = None
        if self.args.prefetch:
            torch = self.torch
            self.work2 = ({k: self.work[k] for k in FKEYS}, {k: self.work[k].clone() for k in FKEYS})
            sarg_b = d.step_args(1, 2000, 180.0)
            for k in FKEYS:
                setattr(sarg_b, k, self.work2[1][k].data_ptr())
            self.sargs2 = (self.sarg, sarg_b)
            if not hasattr(self, "ts2"):
                self.ts2 = torch.cuda.Stream(device=self.dev)
                self.ev_scat = (torch.cuda.Event(), torch.cuda.Event())
                self.ev_kern = (torch.cuda.Event(), torch.cuda.Event())
            torch.cuda.current_stream().synchronize()

    def _permute_forcing(self, h, b, stream):
        """hour h's forcing (tile order) -> forcing working set b of the sorted store, enqueued on `stream`"""
        dst = [(self.work2[b] if self.work2 is not None else self.work)[k] for k in FKEYS]
        if self.gw is not None:
            # config 4: the forcing arrives shaped like the rank's memory block (tile + ring), as the groundwater planes are
            self.scat.exchange(dst, [self.forcing[h][k] for k in FKEYS], False, self.gw.ni, self.i_off, self.j_off, stream, first_level_only=self.lvl1)
        else:
            self.scat.set_dests(dst)
            self.scat.set_sources([self.forcing[h][k] for k in FKEYS])
            self.scat(stream)

    def step(self, it):
        h = forcing_hour(it, self.cfg.dt)
        if not hasattr(self, "first_hour"):
            self.first_hour = h
        self.step_hours.append(h)
        self.steps_since_sort = getattr(self, "steps_since_sort", 0) + 1
        if self.sorted and self.work2 is not None:
            b = it & 1
            if self.prefetched != it:                     # the first step, or the first one after a re-sort: permute in front of the kernel
                self._permute_forcing(h, b, self.sp)
            else:
                self.ts.wait_event(self.ev_scat[b])       # this step's forcing has arrived in set b
            sa = self.sargs2[b]
            sa.itimestep = it
            self.eng.noahmplsm_async(sa, self.sp)
            self.ev_kern[b].record(self.ts)
            h1 = forcing_hour(it + 1, self.cfg.dt)        # the next step's forcing into the other set, beside this step's kernel
            self.ts2.wait_event(self.ev_kern[1 - b])      # (the previous step's kernel read that set)
            self._permute_forcing(h1, 1 - b, self.ts2.cuda_stream)
            self.ev_scat[1 - b].record(self.ts2)
            self.prefetched = it + 1
        elif self.sorted and self.gw is not None:
            if getattr(self, "forcing_ready_for", None) != it:       # (else: it travelled with the previous step's QLAT plane, groundwater())
                self._permute_forcing(h, 0, self.sp)
            self.sarg.itimestep = it
            self.eng.noahmplsm_async(self.sarg, self.sp)
            self.next_it = it + 1
        elif self.sorted:
            self._permute_forcing(h, 0, self.sp)
            self.sarg.itimestep = it
            self.eng.noahmplsm_async(self.sarg, self.sp)
        else:
            sa = self.sargs[h]
            sa.itimestep = it
            self.eng.noahmplsm_async(sa, self.sp)
        if self.lateral and it % self.stepwtd == 0:
            self.groundwater()
        if self.sorted and self.args.resort_every and it % self.args.resort_every == 0:
            self.maybe_resort()

    def groundwater(self):
        """WTABLE_mmf_noahmp (gw:14) after the ZWTXY ring exchange (gw:231-252), enqueued on the run's stream."""
        torch = self.torch
        halo_store = self.gw if self.gw is not None else self.d
        if self.gw is not None:                                                      # ZWTXY: sorted -> (i,j) order, into the ring-carrying block
            self.scat.exchange([self.d.a["zwtxy"]], [self.gw.a["zwtxy"]], True, self.gw.ni, self.i_off, self.j_off, self.sp)
        if self.comm.world > 1:                # (one rank: nothing to exchange)
            with torch.cuda.stream(self.ts):
                # the exchange is timed on every 8th call only: an event is a packet of its own between two kernels (~5 us each on this
                # chip, profiles/r06_experiments.md section 1d) -- two of them per step would cost an 8-rank tile 2 % of its step
                timed = self.gw_calls % 8 == 0
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                self.comm.exchange_halo([halo_store.a["zwtxy"]], self.geom)          # ZWTXY ring before every call
                if timed:
                    e1.record()
                    self.halo_events.append((e0, e1))
        if self.gw is not None:
            self.eng.wtable_lateral_async(self.wargs, self.gw.a["qlat"], self.sp)     # KCELL / HEAD + QLAT stencil, (i,j) order, one launch
            # QLAT -> sorted order -- and with it, in the same launch, the NEXT step's forcing (same direction, same plan; the column kernel
            # that read the forcing working set has finished): one launch boundary less per step (round 6)
            planes_s, planes_t, lvl1 = [self.d.a["qlat"]], [self.gw.a["qlat"]], ()
            if self.work2 is None and getattr(self, "next_it", None) is not None:
                hn = forcing_hour(self.next_it, self.cfg.dt)
                planes_s += [self.work[k] for k in FKEYS]
                planes_t += [self.forcing[hn][k] for k in FKEYS]
                lvl1 = tuple(i + 1 for i in self.lvl1)
                self.forcing_ready_for = self.next_it
            self.scat.exchange(planes_s, planes_t, False, self.gw.ni, self.i_off, self.j_off, self.sp, first_level_only=lvl1)
            self.eng.wtable_columns_async(self.wargs_col, self.d.a["qlat"], self.sp)  # everything else, on the sorted store
        else:
            self.eng.wtable_mmf_async(self.wargs, self.sp)
        self.gw_calls += 1

    def flush(self):
        pass

    def collect(self):
        self.flush()
        st, bad = self.eng.sync()                       # waits for the pending steps; tallies and kernel times summed over them
        self.kernel_ms += st.kernel_ms
        cm, _ = self.eng.sync_timing()
        for c in range(3):
            self.class_ms[c] += cm[c]
        per_step = self.eng.sync_step_timing()            # land (or mixed) kernel of every step since the last sync, by forcing hour
        for h, ms in zip(self.step_hours[len(self.step_hours) - len(per_step):], per_step):
            self.hour_ms.setdefault(h, []).append(ms)
        n_land, n_glacier, _ = self.eng.sync_counts()    # 64-bit: a sync over hundreds of steps of 7 M columns passes 2^31 (Status is int32)
        self.n_adv += n_land + n_glacier
        self.n_land += n_land
        return st

    def maybe_resort(self):
        """Every `resort_every` steps: how many columns left the bucket they were sorted into (snow layers appeared or
        vanished)?  Above the threshold the state is sorted again on the device -- all of it inside the timed region.  The count is
        enqueued on the run's stream and read at the NEXT check: a check that waited for it drained the stream (~0.4 ms of idle GPU while
        the host refills the queue); a re-sort, which is rare, still does."""
        cost_due = self.args.cost_key and self.args.cost_resort_every and self.steps_since_sort >= self.args.cost_resort_every
        if self.stale_result() > self.args.resort_frac * self.d.ncol or cost_due:
            self.resort()
        self.eng.sort_staleness_async(self.d, self.sp)
        self.stale_pending = True

    def resort(self, it=None):
        """Sort the device-resident state again (drains the stream).  With --cost-key the key carries a bucket of every land column's own
        trip counts in the step just run (set_option record_cost + NOAHMP_SORT_COST)."""
        self.collect()
        self.ts.synchronize()
        if getattr(self, "work2", None) is not None:
            self.ts2.synchronize()                        # a prefetched forcing set dies with the old column order
        if self.args.cost_key:
            self.sort_kw["cost"] = True
        self.perm = self.eng.sort_store(self.d, **self.sort_kw)
        self._bind_sorted()
        self.resorts += 1
        self.steps_since_sort = 0

    def stale_result(self):
        """The count the previous check enqueued (one interval ago: it has long arrived, nothing drains); 0 if there was none."""
        if not getattr(self, "stale_pending", False):
            return 0
        self.stale_pending = False
        stale = self.eng.sort_staleness_result(wait=True)
        self.stale_seen.append(stale)
        return stale

    def dump(self, path):
        """The rank's INOUT / OUT arrays in tile order without the ring (tests compare decompositions with them)."""
        import numpy as np
        from noahmp_amd.abi import FIELD_INFO
        self.ts.synchronize()
        h = self.d.to_host()
        g = self.geom
        j0, j1 = g["jts"] - g["jms"], g["jte"] - g["jms"] + 1
        i0, i1 = g["its"] - g["ims"], g["ite"] - g["ims"] + 1
        inv = None
        if self.sorted:
            p = self.perm.cpu().numpy().astype(np.int64)
            inv = np.empty_like(p)
            inv[p] = np.arange(p.size)
        out = {"geom": np.array([g["its"], g["ite"], g["jts"], g["jte"]])}
        for k, v in h.a.items():
            if k not in FIELD_INFO or FIELD_INFO[k][2] == "in" or k == "dzs":
                continue
            if inv is not None:                      # sorted position -> tile order
                v = (v.transpose(1, 0, 2).reshape(v.shape[1], -1)[:, inv].reshape(v.shape[1], v.shape[0], v.shape[2]).transpose(1, 0, 2)
                     if v.ndim == 3 else v.reshape(-1)[inv].reshape(v.shape))
            out[k] = v if self.gw is not None else v[j0:j1, ..., i0:i1]      # the sorted config-4 state is the tile without the ring
        if self.lateral:
            for k in GW_ONLY_OUT:
                v = self.d.a[k].cpu().numpy()
                out[k] = v.reshape(-1)[inv].reshape(v.shape) if self.gw is not None else v[j0:j1, i0:i1]
        np.savez(path, **out)

    def reset_counters(self):
        self.kernel_ms, self.class_ms, self.n_adv, self.n_land, self.resorts, self.stale_seen = 0.0, [0.0, 0.0, 0.0], 0, 0, 0, []
        self.halo_events, self.gw_calls = [], 0
        self.step_hours = []
        # forcing hour -> land-kernel ms of the steps at that hour: timed steps in hour_ms, warm-up steps (without the run's very first
        # launch, which pays the code upload) in hour_ms_warm -- used only for hours the timed window does not reach
        warm = getattr(self, "hour_ms", None)
        if warm is not None and not hasattr(self, "hour_ms_warm"):
            first = getattr(self, "first_hour", None)
            if first in warm and warm[first]:
                warm[first] = warm[first][1:]
            self.hour_ms_warm = warm
        self.hour_ms = {}


class Run5:
    """One rank's share of BASELINE configs[4]: its tile of the global lat/lon grid (mpp_land_partition_calc,
    mpp:227-288; no ring -- the columns are independent), cold start on the device (NOAHMP_INIT + SNOW_INIT, drv:847), then per
    hourly step the chain the HRLDAS driver runs on the host (hdrv:331-415): temporal interpolation between 3-hourly forcing
    records (netcdf_io:1369-1403), forcing preparation with CALC_DECLIN's zenith angle per cell (hdrv:336-354, 813-863) and the
    column step -- all on device-resident arrays in the sorted layout, nothing returns to the host.  No collective."""

    def __init__(self, args, comm, eng, tb, dev):
        import torch
        from noahmp_amd import synth5
        from noahmp_amd.partition import tile_geometry
        from noahmp_amd.state import ModelConfig
        self.torch, self.args, self.comm, self.eng, self.dev, self.synth5 = torch, args, comm, eng, dev, synth5
        self.workload, self.lateral, self.sorted = "config5", False, not args.no_sort
        cfg = ModelConfig(idveg=1)
        self.cfg = cfg
        gx, gy = args.ni, args.nj
        geom = tile_geometry(gx, gy, comm.world, comm.rank, halo=0)
        self.geom = geom
        nx, ny = geom["ite"] - geom["its"] + 1, geom["jte"] - geom["jts"] + 1
        raw, lon, static = synth5.config5_tile(gx, gy, geom["its"] - 1, geom["jts"] - 1, nx, ny, cfg=cfg, smooth=int(getattr(args, "config5_smooth", 0) or 0))
        self.ni, self.nj, self.tile_cells = nx, ny, nx * ny
        self.raw_host = (raw, lon, static) if args.dump else None
        self.d = d = raw.to_device(dev)
        self.ts = torch.cuda.Stream(device=dev)
        self.sp = self.ts.cuda_stream
        self.stepwtd, self.tsk_bin, self.halo_mover, self.gw = 0, None, None, None
        t0 = time.perf_counter()
        eng.noahmp_init(d, fndsnowh=True)                                   # cold start on the device
        self.cold_start_s = time.perf_counter() - t0
        self.lon_t = torch.from_numpy(lon).to(dev).reshape(-1)
        self.static_t = {k: torch.from_numpy(v).to(dev).reshape(-1) for k, v in static.items()}
        self.perm = None
        self.band = None
        if self.sorted:
            self.tsk_bin = args.tsk_bin if args.tsk_bin is not None else 1.0
            self.sort_kw = dict(tsk_bin=self.tsk_bin)
            if args.lon_band > 0:
                # sub-key of the column order: the longitude band (hours of local solar time), so that the columns of a wavefront are in
                # day or in night together -- the class / vegetation / snow keys put columns of all longitudes side by side
                d.a["lonband"] = ((torch.from_numpy(lon).to(dev) + 180.0) / float(args.lon_band)).floor().clamp_(0, 31).to(torch.int32).contiguous()
                self.sort_kw["band"] = self.band = "lonband"
            self.perm = eng.sort_store(d, **self.sort_kw)
        self._bind()
        self.rain = torch.zeros((ny, nx), dtype=torch.float32, device=dev)
        self.rec_a = self.rec_b = None
        self.rec_idx = None
        self.reset_counters()

    def _bind(self):
        """what lives outside the store but in its column order: longitude and the static fields of the forcing records"""
        if self.perm is not None:
            pl = self.perm.long()
            self.lon_d = self.lon_t[pl].reshape(self.nj, self.ni).contiguous()
            st = {k: v[pl].reshape(self.nj, self.ni).contiguous() for k, v in self.static_t.items()}
        else:
            self.lon_d = self.lon_t.reshape(self.nj, self.ni)
            st = {k: v.reshape(self.nj, self.ni) for k, v in self.static_t.items()}
        self.recs = self.synth5.Records(self.d.a["xlatin"], self.lon_d, st)
        self.torch.cuda.current_stream().synchronize()      # built on torch's current stream, read on the run's stream (a re-sort inside a run)

    F5 = ("t3d", "qv3d", "u_phy", "v_phy", "p8w3d", "glw", "swdown", "rainbl", "dz8w", "coszin")    # what the forcing chain writes every step

    DEPTH = 3          # how many steps ahead the forcing chain runs (one more working set than that)

    def _forcing_sets(self):
        """DEPTH + 1 working sets of the forcing arrays (views of the store that share its state arrays), so that the forcing chain of step
        n + DEPTH -- record evaluation, interpolation, preparation -- is enqueued on a second stream while step n's column kernel runs.
        Depth 1 was not enough: beside the land kernel the chain's kernels only get a share of the wave slots that come free, so the last of
        them ends ~70 us after the land kernel -- and the ~45 small launches of a record evaluation (every third step) were still running,
        one after the other on an idle GPU, 0.55 ms after it (kernel trace, profiles/r05_experiments.md section 4).  Three steps of slack
        let all of that overlap later kernels instead of standing in front of the next one."""
        import copy
        torch = self.torch
        self.dv, self.rain2 = [self.d], [self.rain]
        for _ in range(self.DEPTH):
            d2 = copy.copy(self.d)
            d2.a = dict(self.d.a)
            for k in self.F5:
                d2.a[k] = self.d.a[k].clone()
            self.dv.append(d2)
            self.rain2.append(torch.zeros_like(self.rain))
        self.jul2 = [0.0] * (self.DEPTH + 1)
        self.prefetched_to = None                 # chains are in their working sets for steps < prefetched_to
        self.rec_idx = None
        self.rec_staged = None                    # staged records are in the column order they were evaluated in
        if not hasattr(self, "ts2"):
            # One stream PER WORKING SET.  On this runtime a stream that waits for an event of another stream waits for everything that
            # stream has queued by then, not for the event's own position: with ONE prefetch stream every land kernel waited for the chain
            # enqueued during the previous step (which, starved beside that step's land kernel, ends ~90 us after it) however many steps
            # ahead the chain ran (kernel traces, profiles/r05_experiments.md section 4).  (Normal priority: while a HIGH-priority queue
            # has work pending the command processor starts nothing new from the run's queue.)
            self.tsk = [torch.cuda.Stream(device=self.dev) for _ in range(self.DEPTH + 1)]
            self.ts2 = self.tsk[0]
            self.ev_forc = [torch.cuda.Event() for _ in range(self.DEPTH + 1)]
            self.ev_kern = [torch.cuda.Event() for _ in range(self.DEPTH + 1)]
        torch.cuda.current_stream().synchronize()

    def stage_records(self, first_it, nsteps):
        """Evaluate the 3-hourly forcing records the steps first_it .. first_it + nsteps - 1 (+ the prefetch depth) interpolate between,
        before a timed region: they stand for forcing files a driver has read and uploaded, resident in HBM like config 3's hourly sets.
        (Their evaluation is ~45 torch elementwise launches of 6.5 M elements each per record -- a workload GENERATOR, not the product.)
        A re-sort drops them (they are in the store's column order); records that are not staged are evaluated when first needed."""
        s5 = self.synth5
        lo = (first_it - 1) // s5.RECORD_HOURS
        hi = (first_it - 1 + nsteps + self.DEPTH) // s5.RECORD_HOURS + 1
        with self.torch.cuda.stream(self.ts2 if hasattr(self, "ts2") else self.ts):
            self.rec_staged = {ri: self.recs.at(ri) for ri in range(lo, hi + 1)}
        self.torch.cuda.synchronize()

    def _record(self, ri):
        st = getattr(self, "rec_staged", None)
        return st[ri] if st and ri in st else self.recs.at(ri)

    def _chain(self, n, b, stream):
        """the forcing of 0-based step n into working set b, enqueued on `stream` (a torch stream); steps come in increasing order"""
        s5, eng = self.synth5, self.eng
        ri, k = divmod(n, s5.RECORD_HOURS)
        with self.torch.cuda.stream(stream):                                # record evaluation (torch) and the engine's kernels share one stream
            if self.rec_idx != ri:
                self.rec_a = self.rec_b if (self.rec_idx is not None and self.rec_idx == ri - 1) else self._record(ri)
                self.rec_b = self._record(ri + 1)
                self.rec_idx = ri
            for rec in (self.rec_a, self.rec_b):                            # (allocated on one stream, read on this one)
                for v in rec.values():
                    if v is not None and hasattr(v, "record_stream"):
                        v.record_stream(stream)
            iday, ihour = s5.step_time(n)              # interpolation + preparation in one launch (noahmp_hip_forcing_interpolate_prep)
            self.jul2[b] = eng.forcing_interpolate_prep(self.dv[b], self.rec_a, self.rec_b if k else None, 3600 * k, 3600 * s5.RECORD_HOURS,
                                                        self.rain2[b], self.lon_d, iday, ihour, first_step=(n == 0),
                                                        stream=stream.cuda_stream, wait=False)

    def step(self, it):
        n = it - 1
        if not hasattr(self, "dv"):
            self._forcing_sets()
        ihour = self.synth5.step_time(n)[1]
        if not hasattr(self, "first_hour"):
            self.first_hour = ihour
        self.step_hours.append(ihour)
        if not self.args.prefetch:                                          # the chain in front of its kernel, one stream (default)
            self._chain(n, 0, self.ts)
            self.eng.noahmplsm_async(self.dv[0].step_args(it, 2000, self.jul2[0]), stream=self.sp)
        else:
            nb = self.DEPTH + 1
            if self.prefetched_to is None or self.prefetched_to <= it:      # the first step, or the first one after a re-sort: fill the pipeline
                # (step 1's chain writes the first-step guesses into STATE arrays: it must be complete before anything else starts)
                for j in range(it, it + self.DEPTH):
                    if j > it:
                        self.tsk[j % nb].wait_stream(self.tsk[(j - 1) % nb])   # the chains share the record tensors and the first-step writes
                    self._chain(j - 1, j % nb, self.tsk[j % nb])
                    self.ev_forc[j % nb].record(self.tsk[j % nb])
                self.prefetched_to = it + self.DEPTH
            b = it % nb
            self.ts.wait_event(self.ev_forc[b])                             # this step's forcing has arrived in set b
            self.eng.noahmplsm_async(self.dv[b].step_args(it, 2000, self.jul2[b]), stream=self.sp)
            self.ev_kern[b].record(self.ts)
            j = self.prefetched_to                                          # the chain of step j = it + DEPTH into the set step it - 1 read
            sj = self.tsk[j % nb]
            sj.wait_event(self.ev_kern[j % nb])
            sj.wait_stream(self.tsk[(j - 1) % nb])                          # chains run one after the other (they share the record tensors)
            self._chain(j - 1, j % nb, sj)
            self.ev_forc[j % nb].record(sj)
            self.prefetched_to = j + 1
        self.steps_since_sort = getattr(self, "steps_since_sort", 0) + 1
        if self.sorted and self.args.resort_every and it % self.args.resort_every == 0:
            self.maybe_resort(it)

    def maybe_resort(self, it):
        cost_due = self.args.cost_key and self.args.cost_resort_every and self.steps_since_sort >= self.args.cost_resort_every
        if self.stale_result() > self.args.resort_frac * self.d.ncol or cost_due:
            self.resort(it)
        self.eng.sort_staleness_async(self.d, self.sp)
        self.stale_pending = True

    def resort(self, it):
        """`it` = the step just enqueued (the forcing records of the next one are evaluated again in the new column order)"""
        self.collect()
        self.ts.synchronize()
        for q in getattr(self, "tsk", []):
            q.synchronize()
        if self.args.cost_key:
            self.sort_kw["cost"] = True
        self.perm = self.eng.sort_store(self.d, **self.sort_kw)
        self._bind()
        self._forcing_sets()                                             # records and a prefetched set die with the old column order
        self.resorts += 1
        self.steps_since_sort = 0

    stale_result = Run.stale_result
    collect = Run.collect
    reset_counters = Run.reset_counters

    def flush(self):
        pass

    def dump(self, path):
        """The rank's INOUT / OUT arrays in tile order (tests compare decompositions and the oracle with them)."""
        import numpy as np
        from noahmp_amd.abi import FIELD_INFO
        self.ts.synchronize()
        h = self.d.to_host()
        g = self.geom
        inv = None
        if self.perm is not None:
            p = self.perm.cpu().numpy().astype(np.int64)
            inv = np.empty_like(p)
            inv[p] = np.arange(p.size)
        out = {"geom": np.array([g["its"], g["ite"], g["jts"], g["jte"]])}
        for k, v in h.a.items():
            if k not in FIELD_INFO or FIELD_INFO[k][2] == "in" or k == "dzs":
                continue
            if inv is not None:
                v = (v.transpose(1, 0, 2).reshape(v.shape[1], -1)[:, inv].reshape(v.shape[1], v.shape[0], v.shape[2]).transpose(1, 0, 2)
                     if v.ndim == 3 else v.reshape(-1)[inv].reshape(v.shape))
            out[k] = v
        np.savez(path, **out)


WORKLOAD_TEXT = {
    "config2": "BASELINE configs[1]: %(cols)d synthetic land columns (%(ni)dx%(nj)d tile), 4 soil / 0 snow layers, DVEG=1 (dynamic_veg off), "
               "opt_run=1",
    "config3": "BASELINE configs[2]: CONUS-1-km-like grid %(ni)dx%(nj)d = %(cols)d columns, 4 soil / up to 3 snow layers (30 %% snow-covered, "
               "ISNOW 0..-3), 2 %% urban, 1 %% land ice, reference namelist options (DVEG=%(dveg)d, opt_run=1)",
    "config4": "BASELINE configs[3]: the config-3 grid %(ni)dx%(nj)d = %(cols)d columns with OPT_RUN=5 cut into %(world)d tile(s) by "
               "mpp_land_partition_calc, WTABLE_mmf_noahmp every %(stepwtd)d step(s) after the ZWTXY ring exchange (its planes return to "
               "(i,j) order around the call)",
    "config5": "BASELINE configs[4]: global %(ni)dx%(nj)d lat/lon grid = %(cols)d cells (3 %% open water, polar land ice, 1 %% urban, snow where "
               "cold) cut into %(world)d tile(s) by mpp_land_partition_calc; cold start on the device (NOAHMP_INIT), then per hourly step "
               "forcing interpolation between 3-hourly records, forcing preparation with CALC_DECLIN's zenith angle per cell and the "
               "column step (DVEG=1, opt_run=1), all device-resident; a %(steps)d-step leg of the 30-day spin-up",
}


# SIMD cycles a wave64 VALU instruction occupies the vector ALU with two or more waves per SIMD, measured on this chip
# (tools/micro/valu_issue.hip, profiles/r04_valu_issue.txt): full-rate float32 / integer ~2.15, compares / selects / min / max / the
# division helpers / lane moves / conversions / packed float32 ~4.1, float64 arithmetic ~4.4, v_rcp / v_sqrt float32 8.2
VALU_CYCLES = {"full": 2.15, "half": 4.1, "f64": 4.4, "cvt": 4.1, "trans": 8.2}


def valu_roofline(pm, dv, dom_ms, source):
    """Share of the kernel's duration in which the SIMDs' vector ALUs are occupied, under three prices per instruction: the guide's
    2 cycles for everything (lower bound), this chip's measured price list applied to the PMC instruction classes (float64, conversions,
    transcendentals counted; the rest split full-rate / half-rate by the kernel's static mix, `half_rate_share_of_rest`), and 4
    cycles for everything (upper bound = round 3's figure)."""
    n = pm["SQ_INSTS_VALU"]
    simd_cycles = SIMDS * CLOCK_GHZ * 1e9 * dom_ms * 1e-3
    f64 = sum(pm.get(k, 0.0) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64"))
    cvt, trans = pm.get("SQ_INSTS_VALU_CVT", 0.0), pm.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
    rest = max(n - f64 - cvt - trans, 0.0)
    half_share = dv.get("half_rate_share_of_rest", 0.40)      # static mix of the land kernel (tools/isa_stats.py): selects, compares, division helpers, lane moves, packed, min / max
    priced = (f64 * VALU_CYCLES["f64"] + cvt * VALU_CYCLES["cvt"] + trans * VALU_CYCLES["trans"] +
              rest * (half_share * VALU_CYCLES["half"] + (1.0 - half_share) * VALU_CYCLES["full"]))
    out = {"bound": "valu", "wave_insts_per_launch": n, "insts_per_column_step_wave": dv.get("valu_insts_per_column_step"),
           "lane_utilisation": dv.get("lane_utilisation"), "simds": SIMDS, "clock_ghz": CLOCK_GHZ,
           "frac": priced / simd_cycles, "frac_if_every_instruction_took_2_cycles": n * 2.0 / simd_cycles,
           "frac_if_every_instruction_took_4_cycles": n * 4.0 / simd_cycles,
           "cycles_per_instruction_class": VALU_CYCLES, "half_rate_share_of_rest": half_share,
           "instruction_classes_per_launch": {"float64": f64, "conversions": cvt, "transcendental_f32": trans, "other": rest},
           "source": source,
           "note": "share of the kernel's duration in which the SIMDs' vector ALUs are occupied: instructions of the launch (PMC) priced by the "
                   "measured SIMD cycles per instruction class (profiles/r04_valu_issue.txt) / (1024 SIMDs x 2.4 GHz x kernel time).  The "
                   "kernel holds two waves per SIMD and a wave by itself issues at most one VALU instruction per ~4.2 cycles, so its waves are "
                   "bound by their own serial instruction streams and s_waitcnt stalls, not by ALU throughput (profiles/r04_experiments.md)"}
    if pm.get("SQ_ACTIVE_INST_VALU"):
        out["wave_cadence_cycles_per_instruction"] = 4.0 * pm["SQ_ACTIVE_INST_VALU"] / n
    return out


def timed_leg(run, steps, warmup, barrier):
    """warm-up, then `steps` timed steps between two barriers; -> seconds of the timed region"""
    it = 0
    for _ in range(warmup):
        it += 1
        run.step(it)
    if run.args.cost_key and run.sorted and warmup:
        run.resort(it)
    run.collect()
    if hasattr(run, "stage_records") and not run.args.no_stage_records:
        run.stage_records(it + 1, steps)
    run.reset_counters()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        it += 1
        run.step(it)
    run.collect()
    barrier()
    return time.perf_counter() - t0


# Option sets of the `options_reference` legs: none of them has an ahead-of-time kernel, so they run through the kernels
# noahmp_jit.hip specialises with hiprtc (loaded from the in-tree cache build() warms), the last leg through the generic kernel
# (run-time options) that serves a call when hiprtc is not available
OPTION_LEGS = (("namelist options (the headline's ahead-of-time kernel) in the window of these legs", {}, True),
               ("DVEG=2 (dynamic vegetation + CARBON)", dict(idveg=2), True),
               ("OPT_SFC=2 (Chen97 surface layer)", dict(iopt_sfc=2), True),
               ("OPT_FRZ=2 x OPT_INF=2 (Koren99 supercooled water and frozen-soil permeability)", dict(iopt_frz=2, iopt_inf=2), True),
               ("OPT_RUN=3 (Schaake96 free drainage)", dict(iopt_run=3), True),
               ("namelist options through the GENERIC kernel (options as run-time values: fixed_option_kernels = jit_option_kernels = 0)", {}, False))


def options_legs(args, comm, eng, tb, dev, barrier, torch):
    """SURVEY 8d config 3 "full option sweep run as separate launches": the headline grid under other option sets."""
    out = []
    steps, warmup = min(args.steps, 24), min(args.warmup, 3)
    for label, opts, specialised in OPTION_LEGS:
        a = argparse.Namespace(**vars(args))
        a.opts, a.dump = opts, None
        prev = {}
        if not specialised:
            prev = {k: eng.set_option(k, 0) for k in ("fixed_option_kernels", "jit_option_kernels")}
        try:
            r = Run(a, "config3", comm, eng, tb, dev)
            dt = timed_leg(r, steps, warmup, barrier)
            K = steps
            out.append({"options": label, "kernel": ("ahead-of-time specialised" if not opts else "run-time specialised (hiprtc)") if specialised else "generic",
                        "value": r.n_adv / dt, "unit": "column-steps/s", "ms_per_step": dt / K * 1e3, "steps": K,
                        "land_kernel_ms": r.class_ms[0] / K, "columns_per_launch": int(r.n_adv / K),
                        "roofline_frac": ALG_BYTES_PER_COLSTEP * (r.n_adv / K) / (r.class_ms[0] / K * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "first_forcing_hour": forcing_hour(warmup + 1, args.dt)})
            del r
        finally:
            for k, v in prev.items():
                eng.set_option(k, v)
        torch.cuda.empty_cache()
    # The price of bit-exactness: the same kernel with ocml's float32 routines instead of the restated reference libm
    # (-DNMP_EXACT_LIBM=0, noahmp_amd/csrc/variants/lib_ocml.so, built by __graft_entry__.build()).  NOT bit-identical to the reference
    # (statistics: profiles/r05_parity_ocml.md); never the default.
    ocml = os.path.join(ROOT, "noahmp_amd", "csrc", "variants", "lib_ocml.so")
    if os.path.exists(ocml) and not os.environ.get("NMP_LIB"):
        from noahmp_amd.driver import Engine
        from noahmp_amd.tables import load_tables
        eng2 = Engine(load_tables("usgs")[0], device=dev.index, lib_path=ocml)
        eng2.set_veg_order(getattr(args, "veg_order_list", None))
        try:
            a = argparse.Namespace(**vars(args))
            a.opts, a.dump = {}, None
            r = Run(a, "config3", comm, eng2, tb, dev)
            dt = timed_leg(r, steps, warmup, barrier)
            out.append({"options": "namelist options, ocml libm instead of the reference's (-DNMP_EXACT_LIBM=0; results NOT bit-identical, "
                                   "profiles/r05_parity_ocml.md)", "kernel": "ahead-of-time specialised, variants/lib_ocml.so",
                        "value": r.n_adv / dt, "unit": "column-steps/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
                        "land_kernel_ms": r.class_ms[0] / steps, "columns_per_launch": int(r.n_adv / steps),
                        "roofline_frac": ALG_BYTES_PER_COLSTEP * (r.n_adv / steps) / (r.class_ms[0] / steps * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "first_forcing_hour": forcing_hour(warmup + 1, args.dt)})
            del r
        finally:
            eng2.finalize()
        torch.cuda.empty_cache()
    return out


def host_path_leg(args, eng, tb, torch):
    """SURVEY 8d "end-to-end incl. H2D/D2H": the headline grid through noahmp_hip_step(NOAHMP_MEM_HOST) -- what the unedited call
    site hdrv:386-415 gets, host arrays in, host arrays out.  (1) the plain path: every array staged up, INOUT + OUT staged back,
    every call; (2) the resident path the Fortran shim can switch on (resident_state + lazy_download + static_inputs +
    deferred_status, arrays page-locked): the state stays in the device mirrors, a call uploads the forcing and returns when the
    caller may overwrite its forcing arrays; one fetch at the end (an output / restart time).  The loop writes the next hour's
    forcing into the SAME host arrays between calls, as the driver's reader does (that host copy is inside the wall time).
    Tile order, mixed-class kernel: the caller owns the column order on this path."""
    import numpy as np
    from noahmp_amd import synth
    from noahmp_amd.state import ModelConfig
    cfg = ModelConfig(idveg=args.dveg, dt=args.dt)
    s = synth.config3_tile(tb, args.ni, args.nj, cfg=cfg)
    synth.first_step_fixups(s)
    hours = {}
    for h in range(6, 18):
        synth.diurnal_forcing(s, h, t_offset=s.t_offset)
        hours[h] = {k: s.a[k].copy() for k in FKEYS}
    res = {"workload": "the headline grid %dx%d through noahmp_hip_step(NOAHMP_MEM_HOST): host arrays in, host arrays out, tile order "
                       "(mixed-class kernel), forcing rewritten in the caller's arrays between calls" % (args.ni, args.nj), "unit": "column-steps/s"}

    def write_forcing(h):
        for k in FKEYS:
            np.copyto(s.a[k], hours[h][k])

    call_s = [0.0]                                # time inside noahmp_hip_step alone (the rest of a loop pass is this caller's forcing rewrite)

    def loop(n, first_it):
        km, adv = 0.0, 0
        call_s[0] = 0.0
        t0 = time.perf_counter()
        for i in range(n):
            write_forcing(6 + (first_it + i - 1) % 12)
            tc = time.perf_counter()
            st = eng.noahmplsm(s, first_it + i, 2000, 180.0)
            call_s[0] += time.perf_counter() - tc
            km += st.kernel_ms
            adv += st.n_land + st.n_glacier
        return time.perf_counter() - t0, km, adv

    tw = time.perf_counter()
    write_forcing(6)
    res["host_forcing_write_ms"] = (time.perf_counter() - tw) * 1e3
    # (1) plain: stage everything both ways (pageable arrays, single shot)
    prev = {"host_chunks": eng.set_option("host_chunks", 0)}
    eng.noahmplsm(s, 1, 2000, 180.0)
    dt, km, adv = loop(2, 2)
    res["stage_everything"] = {"value": adv / dt, "ms_per_step": dt / 2 * 1e3, "kernel_ms": km / 2, "steps": 2,
                               "bytes_up_per_step": int(sum(v.nbytes for k, v in s.a.items() if k != "dzs")),
                               "note": "pageable host arrays, every array H2D and INOUT + OUT D2H per call"}
    # (1b) page-locked arrays + row chunks (upload | kernel | download overlapped), every array both ways, every call
    eng.set_option("host_chunks", prev["host_chunks"])
    prev["pin_host_arrays"] = eng.set_option("pin_host_arrays", 1)
    prev["trust_out_mirror"] = eng.set_option("trust_out_mirror", 0)
    for it in (4, 5):                             # second sighting of every array: registered
        eng.noahmplsm(s, it, 2000, 180.0)
    dt, km, adv = loop(3, 6)
    res["pinned_row_chunks"] = {"value": adv / dt, "ms_per_step": dt / 3 * 1e3, "kernel_ms": km / 3, "steps": 3,
                                "host_chunks": (lambda v: "chosen by the engine from the tile size (3..8: %d here)" % max(3, min(8, (s.ncol + 600000) // 1200000))
                                                if v in (-1, -2) else v)(eng.set_option("host_chunks", prev["host_chunks"])),
                                "page_locked_arrays": int(eng.lib.noahmp_hip_debug_live_host_registrations()),
                                "note": "caller arrays page-locked in place, the tile advanced in row chunks (H2D | kernel | D2H on three "
                                        "streams); every array H2D and INOUT + OUT D2H per call"}
    # (1c) what the generated Fortran shim sets by itself (module_sf_noahmpdrv_hip.F90: pin_host_arrays = 1, trust_out_mirror = 1): the 119
    # OUT / INOUT words still come back every call, but pure OUT arrays are not uploaded again (the HRLDAS driver only reads them, for
    # output: hdrv:453-558) -- the unedited drop-in
    eng.set_option("trust_out_mirror", 1)
    eng.noahmplsm(s, 9, 2000, 180.0)
    dt, km, adv = loop(3, 10)
    res["shim_default"] = {"value": adv / dt, "ms_per_step": dt / 3 * 1e3, "kernel_ms": km / 3, "steps": 3,
                           "note": "the Fortran shim's own defaults: the above + set_option(trust_out_mirror, 1): OUT arrays are not re-uploaded"}
    eng.set_option("trust_out_mirror", prev["trust_out_mirror"])
    # (2) resident state behind the same call
    opts = dict(pin_host_arrays=1, resident_state=1, lazy_download=1, static_inputs=1, deferred_status=1)
    for k, v in opts.items():
        old = eng.set_option(k, v)
        prev.setdefault(k, old)                   # (pin_host_arrays: the value from before (1b))
    try:
        for it in (13, 14, 15):                   # state rebuilt, arrays registered (second sighting), both IN buffers filled
            eng.noahmplsm(s, it, 2000, 180.0)
        n = 12
        t0 = time.perf_counter()
        dt, km, adv = loop(n, 16)
        tf = time.perf_counter()
        eng.fetch()                               # waits for the last step, brings INOUT + OUT arrays back
        t1 = time.perf_counter()
        res["resident"] = {"value": s.ncol * n / (t1 - t0), "ms_per_step": (tf - t0) / n * 1e3, "fetch_ms": (t1 - tf) * 1e3, "steps": n,
                           "kernel_ms": km / n, "engine_ms_per_call": call_s[0] / n * 1e3,
                           "engine_column_steps_per_s": s.ncol * n / call_s[0],
                           "ms_per_step_with_one_fetch_per_%d_steps" % n: (t1 - t0) / n * 1e3,
                           "options": sorted(opts), "note": "value = all cells of the tile x steps / wall time incl. the final fetch (the "
                           "deferred status of a call reports the PREVIOUS step, so tallies lag by one); ms_per_step = this caller's rewrite of "
                           "its five forcing arrays (host_forcing_write_ms) + engine_ms_per_call, the time inside noahmp_hip_step: the upload of "
                           "the call's forcing (PCIe), under which the previous step's kernel runs"}
        # (3) the same with "resident_sorted": the engine keeps a second, sorted set of mirrors and runs the class-range kernels on it
        prev["resident_sorted"] = eng.set_option("resident_sorted", 1)
        km = []
        for it in (30, 31, 32):                   # the state is sorted at the first of these calls
            write_forcing(6 + (it - 1) % 12)
            eng.noahmplsm(s, it, 2000, 180.0)
        t0 = time.perf_counter()
        call_s[0] = 0.0
        for i in range(n):
            write_forcing(6 + (33 + i - 1) % 12)
            tc = time.perf_counter()
            km.append(eng.noahmplsm(s, 33 + i, 2000, 180.0).kernel_ms)
            call_s[0] += time.perf_counter() - tc
        tf = time.perf_counter()
        eng.fetch()
        t1 = time.perf_counter()
        res["resident_sorted"] = {"value": s.ncol * n / (t1 - t0), "ms_per_step": (tf - t0) / n * 1e3, "fetch_ms": (t1 - tf) * 1e3, "steps": n,
                                  "kernel_ms": sum(km[1:]) / max(len(km) - 1, 1), "engine_ms_per_call": call_s[0] / n * 1e3,
                                  "engine_column_steps_per_s": s.ncol * n / call_s[0],
                                  "note": "resident + set_option(resident_sorted, 1): host arrays in tile order, device mirrors sorted by (class, "
                                          "vegetation type, snow-layer count, TSK bin); kernel_ms = the class-range kernels of a call (under deferred "
                                          "status a call reports the previous step)"}
    finally:
        for k in ("resident_sorted", "deferred_status", "static_inputs", "lazy_download", "resident_state", "trust_out_mirror", "pin_host_arrays", "host_chunks"):
            if k in prev:
                eng.set_option(k, prev[k])
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--workload", choices=("config2", "config3", "config4", "config5"), default=None)
    ap.add_argument("--dt", type=float, default=3600.0, help="model time step [s]; 900 makes config 4 call WTABLE_mmf_noahmp every 2nd step "
                    "(STEPWTD = nint(WTDDT*60/DT), hdrv:247-248); the synthetic forcing advances with it")
    ap.add_argument("--no-config5-reference", action="store_true",
                    help="N = 1, default workload: skip the short config-5 leg (BASELINE configs[4]) reported beside the headline")
    ap.add_argument("--ni", type=int, default=None)
    ap.add_argument("--nj", type=int, default=None)
    ap.add_argument("--dveg", type=int, default=3)
    ap.add_argument("--tsk-bin", type=float, default=None,
                    help="skin-temperature bin of the sort key [K], 0 = off; default 1 K, 4 K for config 4 (coarser bins = longer runs in "
                         "the permutations around WTABLE_mmf_noahmp: 5.40 -> 5.19 ms/step, column kernel +0.4 %%)")
    ap.add_argument("--resort-every", type=int, default=24, help="steps between staleness checks of the sorted layout (0 = never)")
    ap.add_argument("--resort-frac", type=float, default=0.10,
                    help="re-sort when this share of the columns left their bucket (measured: 11 %% stale columns cost the land kernel 0.8 %%, "
                         "a re-sort 1.7 ms -- profiles/r02_experiments.md)")
    ap.add_argument("--cost-key", action="store_true",
                    help="sorted layout: the steps record every land column's trip counts (canopy iterations, STOMATA bisection steps); the state "
                         "is sorted again at the end of the warm-up with a bucket of them in the key (a wavefront runs as long as its slowest lane)")
    ap.add_argument("--cost-resort-every", type=int, default=0,
                    help="with --cost-key: sort again (inside the timed region) whenever this many steps have passed since the last sort, "
                         "checked at the --resort-every cadence (0 = only at the end of the warm-up and when stale)")
    ap.add_argument("--lon-band", type=float, default=15.0,
                    help="config 5: width [degrees] of the longitude bands of the sort key (0 = no band key); 15 = one hour of local solar time")
    ap.add_argument("--config5-smooth", type=int, nargs="?", const=1, default=0,
                    help="config 5: 1 = the per-column forcing factors (cloud, humidity, pressure, wind, rain timing) are spatially smooth random "
                         "fields (synth5.smooth_uniform: correlation length a few hundred km) instead of i.i.d. draws per cell; same marginals, "
                         "same state.  2 = the state's per-cell noise (soil category, temperature scatter, vegetation fraction, soil moisture, "
                         "snow) is smooth too.  The i.i.d. generator stays the quoted config-5 number (the conservative one)")
    ap.add_argument("--no-sort", action="store_true")
    ap.add_argument("--no-stage-records", action="store_true",
                    help="config 5: evaluate the synthetic 3-hourly forcing records inside the timed region, when a step first needs them "
                         "(rounds 1-4), instead of before it")
    ap.add_argument("--prefetch", action="store_true",
                    help="enqueue the forcing work of LATER steps on other streams beside the running column kernel: configs 2 / 3 / 4 the "
                         "permutation of step n + 1's forcing (two working sets), config 5 the interpolation + preparation three steps ahead (four "
                         "working sets, one stream each).  Measured in round 5: no gain for configs 3 / 4 (two land waves per SIMD hold the whole "
                         "register file: a wave of another kernel only ever takes the place of one), 0-2 %% for config 5 depending on how the "
                         "runtime maps streams to hardware queues -- off by default")
    ap.add_argument("--veg-order", default="canopy-first",
                    help="order of the vegetation categories inside the sorted land range: `canopy-first` (default: categories with a canopy, then "
                         "the bare ones -- urban, barren, LAI + SAI = 0 -- whose waves cost 0.57x as much: cheap waves at the tail of the launch), "
                         "`numeric` (the categories' own numbers, rounds 2-5), or a comma-separated list (the rest follow in numeric order)")
    ap.add_argument("--snow-first", action="store_true", help="sort key: snow-layer count above vegetation type")
    ap.add_argument("--tair-key", action="store_true", help="temperature bins of the sort key from the air temperature instead of TSK")
    ap.add_argument("--no-veg-key", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-snow-key", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--halo", choices=("auto", "torch", "rccl", "tcp"), default=os.environ.get("NMP_HALO", "auto"),
                    help="who moves the groundwater ring: the engine's C-ABI exchange noahmp_hip_exchange_halo with its RCCL or socket "
                         "transport, or torch.distributed send/recv (RCCL under the nccl backend); auto = the C-ABI RCCL mover when it "
                         "starts and passes a checked probe exchange on every rank, else torch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-options-reference", action="store_true",
                    help="N = 1, default workload: skip the legs under other option sets (run-time specialised kernels, generic kernel)")
    ap.add_argument("--no-host-path-reference", action="store_true",
                    help="N = 1, default workload: skip the PCIe-inclusive leg through noahmp_hip_step(NOAHMP_MEM_HOST)")
    ap.add_argument("--no-n1-reference", action="store_true",
                    help="N > 1: skip rank 0's own run of the same workload on the whole grid (n1_reference / strong_scaling_efficiency)")
    ap.add_argument("--no-scaling-reference", action="store_true",
                    help="N = 1, default workload: skip the short config-4 run that gives the N = 1 point of the --gpus N curve")
    ap.add_argument("--dump", default=None, help="write every rank's tile (tile order, without the ring) to DUMP.rank<r>.npz after the run")
    ap.add_argument("--cpu-baseline-only", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_baseline_only:              # child process: never touches the GPU
        print("CPU_BASELINE " + json.dumps(cpu_baseline(args.cpu_baseline_only)))
        return 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, sys.argv[1:])

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    workload = args.workload or ("config3" if world == 1 else "config4")
    if args.ni is None:
        args.ni, args.nj = (1024, 1024) if workload == "config2" else ((3600, 1800) if workload == "config5" else (4608, 1536))

    from noahmp_amd.tables import load_tables
    T, tb = load_tables("usgs")
    import torch
    from noahmp_amd.driver import Engine
    from noahmp_amd.parallel import Comm
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise RuntimeError("bench.py: no GPU visible (the engine has no CPU path)")
    dev_index = local_rank % ndev            # several ranks on one GPU: only with NMP_DIST_BACKEND=gloo (1-GPU check of the N>1 path)
    torch.cuda.set_device(dev_index)
    comm = Comm(device_index=dev_index, halo=args.halo)      # one process per GPU; "nccl" = RCCL when world > 1
    dev = torch.device("cuda", dev_index)
    eng = Engine(T, device=dev_index, lib_path=os.environ.get("NMP_LIB"))
    if os.environ.get("NMP_BLOCK"):
        eng.set_option("block", int(os.environ["NMP_BLOCK"]))
    args.veg_order_list = None
    if args.veg_order == "canopy-first":
        from noahmp_amd.tables import canopy_first_order
        args.veg_order_list = canopy_first_order(tb)
    elif args.veg_order and args.veg_order != "numeric":
        args.veg_order_list = [int(v) for v in args.veg_order.split(",")]
    eng.set_veg_order(args.veg_order_list)

    def barrier():
        comm.barrier()
        torch.cuda.synchronize()

    # N > 1: the N = 1 point of the strong-scaling curve on the SAME workload, inside this very run -- rank 0 advances the whole grid on
    # its GPU (same steps, same warm-up) while the other ranks wait at a barrier; the line then carries n1_reference and
    # strong_scaling_efficiency = value / (N x n1_reference.value) and needs no second run to be read.  (The default N = 1 run is the
    # headline workload, config 3; the default N > 1 workload is config 4, ~13 % more work per column.)
    n1_ref = None
    if world > 1 and not args.no_n1_reference:
        if rank == 0:
            t1 = time.perf_counter()
            solo = Comm.solo(dev_index)
            r1 = Run5(args, solo, eng, tb, dev) if workload == "config5" else Run(args, workload, solo, eng, tb, dev)
            dt1 = timed_leg(r1, args.steps, args.warmup, lambda: torch.cuda.synchronize())
            n1_ref = {"workload": "the same workload (`--workload %s`, %d x %d) on ONE GPU: rank 0 of this run, before the distributed region"
                                  % (workload, args.ni, args.nj),
                      "value": r1.n_adv / dt1, "unit": "column-steps/s", "ms_per_step": dt1 / args.steps * 1e3, "steps": args.steps,
                      "warmup": args.warmup, "column_kernels_ms_per_step": r1.kernel_ms / args.steps, "groundwater_calls": r1.gw_calls,
                      "wall_s_incl_setup": None}
            del r1
            torch.cuda.empty_cache()
            n1_ref["wall_s_incl_setup"] = time.perf_counter() - t1
        comm.barrier()

    t_setup = time.perf_counter()
    run = Run5(args, comm, eng, tb, dev) if workload == "config5" else Run(args, workload, comm, eng, tb, dev)
    t_setup = time.perf_counter() - t_setup

    if args.cost_key and eng.set_option("record_cost", 1) < 0:
        raise RuntimeError("--cost-key needs a library built with -DNMP_COST_RECORD (tools/build_variants.py cost=-DNMP_COST_RECORD; NMP_LIB=...)")
    it = 0
    for _ in range(args.warmup):
        it += 1
        run.step(it)
    if args.cost_key and run.sorted and args.warmup:
        run.resort(it)                                  # outside the timed region, like the first sort
    run.collect()
    if hasattr(run, "stage_records") and not args.no_stage_records:
        run.stage_records(it + 1, args.steps)           # config 5: the window's forcing records resident in HBM, as config 3's hourly sets are
    run.reset_counters()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        it += 1
        run.step(it)
    t_enqueued = time.perf_counter() - t0               # the host's side of the timed steps (the stream runs behind it)
    run.collect()
    barrier()
    dt_local = time.perf_counter() - t0
    run.stale_result()                                  # the last check's count, for the report only
    dt = comm.reduce_max(dt_local)                      # MAX over ranks
    n_adv_all = comm.reduce_sum(run.n_adv)              # column-steps advanced by the whole job (land + land ice, not the skips)
    halo_ms = sum(e0.elapsed_time(e1) for e0, e1 in run.halo_events) / max(len(run.halo_events), 1) * run.gw_calls    # (every 8th call is timed)
    halo_ms_max = comm.reduce_max(halo_ms)
    kernel_ms_max = comm.reduce_max(run.kernel_ms)
    kernel_ms_min = comm.reduce_min(run.kernel_ms)

    if args.dump:
        run.dump(args.dump + ".rank%d.npz" % rank)
    if rank == 0 and os.environ.get("NMP_PHASE_PROF"):
        # profiling library only (-DNMP_PHASE_TIMERS, tools/phase_prof.py): shares of the kernel's time by phase, to stderr
        import ctypes as C
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
        from phase_prof import NAMES
        ticks = (C.c_ulonglong * 32)()
        eng.lib.noahmp_hip_debug_phase_ticks(ticks, 32)
        tot = float(sum(ticks)) or 1.0
        for ph in sorted(NAMES, key=lambda ph: -ticks[ph]):
            print("phase %-42s %5.1f %%" % (NAMES[ph], 100.0 * ticks[ph] / tot), file=sys.stderr)

    if rank == 0 and os.environ.get("NMP_COST_SPREAD") and run.sorted:
        # analysis only (tools/cost_spread.py): the recorded trip counts per wavefront under the current column order, at a few hours
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import cost_spread
        eng.set_option("record_cost", 1)
        for _ in range(4):
            for _ in range(int(os.environ.get("NMP_COST_SPREAD_STRIDE", "5"))):
                it += 1
                run.step(it)
            run.collect()
            cost_spread.report(eng, run, run.step_hours[-1], run.d.class_ranges[0])
        if not args.cost_key:
            eng.set_option("record_cost", 0)

    # The default N > 1 workload is config 4 (the same grid with the groundwater exchange).  So that a scaling curve over
    # N = 1, 2, 4, 8 has its N = 1 point on the SAME workload, the default N = 1 run measures it too, after the headline (a second,
    # separately timed region of the same length; reported beside the headline, never as `value`).
    scaling_ref = config2_ref = config5_ref = config5_smooth_ref = options_ref = host_ref = None
    if world == 1 and workload == "config3" and args.workload is None and not args.no_scaling_reference:
        summary = dict(tsk_bin=run.tsk_bin, class_ms=list(run.class_ms), n_land=run.n_land, n_adv=run.n_adv, resorts=run.resorts, stale=list(run.stale_seen),
                       sorted=run.sorted, lateral=run.lateral, tile_cells=run.tile_cells, stepwtd=run.stepwtd, kernel_ms=run.kernel_ms,
                       hour_ms=dict(run.hour_ms), hour_ms_warm=dict(getattr(run, "hour_ms_warm", {})), step_hours=list(run.step_hours))
        del run
        torch.cuda.empty_cache()
        r4 = Run(args, "config4", comm, eng, tb, dev)
        dt4 = timed_leg(r4, args.steps, args.warmup, barrier)
        scaling_ref = {"workload": "BASELINE configs[3] on one GPU (`--workload config4`): the N = 1 point of the --gpus N strong-scaling curve",
                       "value": r4.n_adv / dt4, "unit": "column-steps/s", "ms_per_step": dt4 / args.steps * 1e3, "steps": args.steps,
                       "groundwater_calls": r4.gw_calls, "column_kernels_ms_per_step": r4.kernel_ms / args.steps}
        del r4
        torch.cuda.empty_cache()
        # BASELINE configs[1]: 1 M synthetic land columns, 0 snow layers, DVEG = 1 (round 1's headline; no number on any later build until round 6)
        a2 = argparse.Namespace(**vars(args))
        a2.ni, a2.nj, a2.dump = 1024, 1024, None
        r2 = Run(a2, "config2", comm, eng, tb, dev)
        dt2 = timed_leg(r2, args.steps, args.warmup, barrier)
        K2 = args.steps
        config2_ref = {"workload": WORKLOAD_TEXT["config2"] % dict(cols=1024 * 1024, ni=1024, nj=1024) + " (`--workload config2`), sorted on the device, "
                                   "forcing permutation inside the timed region",
                       "value": r2.n_adv / dt2, "unit": "column-steps/s", "ms_per_step": dt2 / K2 * 1e3, "steps": K2,
                       "land_kernel_ms": r2.class_ms[0] / K2, "columns_per_launch": int(r2.n_adv / K2),
                       "roofline_frac": ALG_BYTES_PER_COLSTEP * (r2.n_adv / K2) / (r2.class_ms[0] / K2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "first_forcing_hour": forcing_hour(args.warmup + 1, args.dt)}
        del r2
        torch.cuda.empty_cache()
        if not args.no_config5_reference:
            # BASELINE configs[4] on one GPU: a short leg of the 30-day spin-up (cold start outside the timed region, reported)
            a5 = argparse.Namespace(**vars(args))
            a5.ni, a5.nj, a5.dump = 3600, 1800, None
            r5 = Run5(a5, comm, eng, tb, dev)
            dt5 = timed_leg(r5, args.steps, args.warmup, barrier)
            config5_ref = {"workload": WORKLOAD_TEXT["config5"] % dict(cols=3600 * 1800, ni=3600, nj=1800, world=1, steps=args.steps) +
                           " (`--workload config5`)",
                           "value": r5.n_adv / dt5, "unit": "column-steps/s", "ms_per_step": dt5 / args.steps * 1e3, "steps": args.steps,
                           "columns_advanced_per_step": r5.n_adv // args.steps, "column_kernels_ms_per_step": r5.kernel_ms / args.steps,
                           "land_kernel_ms": r5.class_ms[0] / args.steps,          # (the one column kernel of a step: land + land-ice + skipped ranges)
                           "cold_start_s": r5.cold_start_s}
            del r5
            torch.cuda.empty_cache()
            # the same leg with spatially smooth forcing factors: how much of config 5's divergence is white noise in the generator
            a5.config5_smooth = 1
            r5 = Run5(a5, comm, eng, tb, dev)
            dt5 = timed_leg(r5, args.steps, args.warmup, barrier)
            config5_smooth_ref = {"workload": "config5_reference with spatially smooth forcing factors (`--workload config5 --config5-smooth`: cloud, humidity, "
                                              "pressure, wind and rain timing are smooth random fields with a correlation length of a few hundred km "
                                              "instead of i.i.d. draws per cell; same marginal distributions, same state).  The i.i.d. leg stays the quoted "
                                              "config-5 number",
                                  "value": r5.n_adv / dt5, "unit": "column-steps/s", "ms_per_step": dt5 / args.steps * 1e3, "steps": args.steps,
                                  "columns_advanced_per_step": r5.n_adv // args.steps, "column_kernels_ms_per_step": r5.kernel_ms / args.steps,
                                  "land_kernel_ms": r5.class_ms[0] / args.steps}
            del r5
            torch.cuda.empty_cache()
        if not args.no_options_reference:
            options_ref = options_legs(args, comm, eng, tb, dev, barrier, torch)
        if not args.no_host_path_reference:
            host_ref = host_path_leg(args, eng, tb, torch)
            torch.cuda.empty_cache()

        class _R:            # what the report below needs of the headline run
            pass
        run = _R()
        run.class_ms, run.n_land, run.n_adv, run.resorts, run.stale_seen = summary["class_ms"], summary["n_land"], summary["n_adv"], summary["resorts"], summary["stale"]
        run.sorted, run.lateral, run.tile_cells, run.stepwtd, run.kernel_ms = summary["sorted"], summary["lateral"], summary["tile_cells"], summary["stepwtd"], summary["kernel_ms"]
        run.gw_calls = 0
        run.tsk_bin = summary["tsk_bin"]
        run.hour_ms, run.hour_ms_warm, run.step_hours = summary["hour_ms"], summary["hour_ms_warm"], summary["step_hours"]

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU leg: after the timed GPU region, in a child process that never initialises the GPU
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", workload], capture_output=True, text=True)
        for line in r.stdout.splitlines():
            if line.startswith("CPU_BASELINE "):
                cpu = json.loads(line[len("CPU_BASELINE "):])

    if rank == 0:
        K = args.steps
        prefetching = workload in ("config2", "config3", "config4") and not args.no_sort and args.prefetch
        prefetch5 = workload == "config5" and args.prefetch
        value = n_adv_all / dt
        # dominant kernel: the land range of the sorted layout (the mixed kernel of a tile-order run); its own event pair per step
        # Round 6: ONE kernel per step either way -- a sorted run's land, land-ice and skipped ranges are workgroup ranges of one launch
        # (noahmp_ranges_kernel), so the launch advances land AND land-ice columns; its own start / stop events ride on the dispatch
        dom_ms = run.class_ms[0] / K
        dom_cols = run.n_adv / K                                       # columns one launch of that kernel advances
        achieved = ALG_BYTES_PER_COLSTEP * dom_cols / (dom_ms * 1e-3) / 1e9
        traffic, valu, traffic_source = None, None, None
        # PMC counters cannot be collected inside this run (rocprofv3 --pmc passes are separate runs of this very command,
        # tools/run_profile.sh -> tools/collect_profile.py): the newest committed summary is quoted, and only when workload and
        # columns per launch are those of the profile
        for tag in ("r06", "r05", "r04", "r03", "r02"):
            tpath = os.path.join(ROOT, "profiles", "%s_traffic.json" % tag)
            if not os.path.exists(tpath):
                continue
            try:
                prof = json.load(open(tpath))
                if prof.get("workload") == workload and prof.get("columns_per_launch") == int(dom_cols) and world == 1:
                    traffic = prof.get("hbm_bytes_per_launch")
                    traffic_source = "profiles/%s_traffic.json (separate rocprofv3 --pmc passes of this command; not measured in this run)" % tag
                    pm = prof.get("pmc_mean_per_launch") or {}
                    dv = prof.get("derived") or {}
                    if pm.get("SQ_INSTS_VALU"):
                        valu = valu_roofline(pm, dv, dom_ms, traffic_source)
                    break
            except Exception:
                traffic, valu, traffic_source = None, None, None
        # the land kernel by forcing hour: day (COSZ > 0: hours 7..17 of the synthetic cycle) and night steps cost differently, and a
        # timed window shorter than a day samples them unevenly -- the 24-hour mean takes each hour from the timed steps, and from the
        # warm-up steps only the hours the timed window does not reach
        hours = {h: sum(v) / len(v) for h, v in getattr(run, "hour_ms_warm", {}).items() if v}
        hours.update({h: sum(v) / len(v) for h, v in getattr(run, "hour_ms", {}).items() if v})        # timed steps win
        day = [hours[h] for h in hours if 6 < h < 18] if workload != "config5" else []
        night = [hours[h] for h in hours if not 6 < h < 18] if workload != "config5" else []
        ms_24h = sum(hours.values()) / 24.0 if len(hours) == 24 else None
        # Round 6: the roofline is quoted from the 24-HOUR-WEIGHTED kernel mean whenever every hour of the cycle was sampled -- a window that is
        # not a multiple of 24 steps holds day and night steps in another proportion than a day does (--steps 20 --warmup 5: 7 day + 13
        # night steps = 35 % day, a day has 11 of 24 = 46 %), which moved the headline fraction by 4 % between window lengths.  The raw window
        # figures stay beside it (window_*); value / ms_per_step are wall-clock figures of the window, value_24h_weighted replaces the
        # window's kernel mean by the 24-hour one inside the wall-clock step (everything else of a step does not depend on the hour).
        window_ms, window_achieved = dom_ms, achieved
        timed_hours = [h for h in getattr(run, "step_hours", [])][-K:] if getattr(run, "step_hours", None) else []
        n_day = len([h for h in timed_hours if 6 < h < 18])
        if ms_24h:
            dom_ms = ms_24h
            achieved = ALG_BYTES_PER_COLSTEP * dom_cols / (dom_ms * 1e-3) / 1e9
        ms_step_24h = (dt / K * 1e3 - window_ms + ms_24h) if ms_24h else None
        desc = WORKLOAD_TEXT[workload] % dict(cols=args.ni * args.nj, ni=args.ni, nj=args.nj, dveg=args.dveg, world=world,
                                              stepwtd=run.stepwtd, steps=K)
        if workload == "config5":
            desc += ("; sorted on the device by (class, vegetation type, snow-layer count, %s%g-K skin-temperature bin)"
                     % (("%g-degree longitude band, " % args.lon_band) if getattr(run, "band", None) else "", run.tsk_bin)
                     if run.sorted else "; tile order")
            if prefetch5:
                desc += "; the forcing chain (interpolation + preparation) runs three steps ahead on a second stream beside the column kernels (four forcing working sets)"
            if getattr(args, "config5_smooth", False):
                desc += ("; forcing factors (cloud, humidity, pressure, wind, rain timing) spatially smooth (--config5-smooth) instead of i.i.d. per cell"
                         + ("; soil category, temperature scatter, vegetation fraction, soil moisture and snow smooth too" if int(args.config5_smooth) >= 2 else ""))
            desc += ("; the 3-hourly forcing records of the timed window are resident in HBM when it starts (evaluated before it, as config 3's hourly sets are)"
                     if not args.no_stage_records else "; the synthetic 3-hourly forcing records are evaluated inside the timed region (torch elementwise kernels)")
        elif run.sorted:
            desc += ("; state resident in HBM, sorted on the device by (class, vegetation type, snow-layer count, %g-K skin-temperature "
                     "bin); inside the timed region: the per-step permutation of the forcing (which arrives in tile order%s) and a staleness "
                     "check every %d steps (a re-sort follows above %g %% stale columns: %d happened).  Forcing: the SURVEY 8d diurnal cycle, "
                     "time step %g s, SPATIALLY UNIFORM in zenith angle / short wave / rain (the whole grid is in day or night together: no "
                     "terminator, no precipitation fronts; per-column air-temperature offsets only) -- config 5's lat/lon-dependent zenith "
                     "angle costs ~25 %% more per column (config5_reference)"
                     % (run.tsk_bin, "; step n + 1's runs on a second stream beside step n's column kernel, two forcing working sets" if prefetching else "",
                        args.resort_every, args.resort_frac * 100, run.resorts, args.dt))
            if timed_hours and workload != "config5":
                desc += ("; the timed window holds %d day + %d night steps (a day of this forcing: 11 + 13); roofline figures are the 24-hour-weighted "
                         "kernel mean%s" % (n_day, len(timed_hours) - n_day, "" if ms_24h else " -- NOT available here (not every hour sampled): window mean"))
        else:
            desc += "; state resident in HBM, diurnal forcing (spatially uniform zenith angle / short wave / rain), time step %g s" % args.dt
        out = {
            "metric": "column-steps/sec", "value": value, "unit": "column-steps/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3, "host_enqueue_ms_per_step": t_enqueued / K * 1e3, "higher_is_better": True, "scaling": "strong",
            "value_24h_weighted": (n_adv_all / K / (ms_step_24h * 1e-3)) if ms_step_24h else None, "ms_per_step_24h_weighted": ms_step_24h,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "grid": [args.ni, args.nj], "columns_per_gpu": run.tile_cells,
                       "parallelism": ("1 GPU" if world == 1 else "%d tiles (mpp_land_partition_calc), one rank per GPU%s"
                                       % (world, (", one-phase ZWTXY ring exchange (%s)" % {"torch": "torch.distributed send/recv", "rccl": "noahmp_hip_exchange_halo, RCCL transport",
                                                                          "tcp": "noahmp_hip_exchange_halo, socket transport"}[comm.halo])
                                          if run.lateral else ", no collective"))},
            "timed_region_s": dt, "setup_s": t_setup,
            "achieved_hbm_gbs": achieved,          # the second half of BASELINE.json's metric: algorithmic GB/s of the dominant kernel (= roofline.achieved)
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": ("noahmp_ranges_kernel (the sorted layout's land + land-ice + skipped ranges in one launch)" if run.sorted
                                    else "noahmp_column_kernel (mixed tile)"),
                         "kernel_ms_avg": dom_ms, "columns_per_launch": int(dom_cols),
                         "weighting": ("24-hour-weighted: mean over the 24 forcing hours of the per-hour kernel means (timed steps; warm-up steps only "
                                       "for hours the timed window does not reach)") if ms_24h else "mean over the timed window",
                         "window_kernel_ms_avg": window_ms, "window_achieved": window_achieved, "window_frac": window_achieved / HBM_PEAK_GBS,
                         "window_day_steps": n_day if timed_hours else None, "window_night_steps": (len(timed_hours) - n_day) if timed_hours else None,
                         "kernel_ms_day": (sum(day) / len(day)) if day else None, "kernel_ms_night": (sum(night) / len(night)) if night else None,
                         "kernel_ms_24h_mean": ms_24h, "hours_sampled": len(hours),
                         "frac_24h_mean": (ALG_BYTES_PER_COLSTEP * dom_cols / (ms_24h * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms_24h else None,
                         "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": ALG_BYTES_PER_COLSTEP * int(dom_cols), "valu": valu,
                         "note": "824 B/column-step x columns of the launch / HIP-event time of that kernel (its own event pair per step on the "
                                 "stream it runs on, mean over the timed steps); the kernel is bound by the serial instruction streams of its "
                                 "two waves per SIMD and their stalls, not by HBM (SURVEY 8d) -- see `valu`"},
            "column_kernels_ms_per_step": {"land_or_mixed": run.class_ms[0] / K, "land_ice": run.class_ms[1] / K,
                                           "skipped": run.class_ms[2] / K, "all_max_over_ranks": kernel_ms_max / K},
            "kernel_only_column_steps_per_s": n_adv_all / (kernel_ms_max * 1e-3) if kernel_ms_max else None,
        }
        if run.sorted:
            out["sort"] = {"resorts_in_timed_region": run.resorts, "stale_columns_seen": run.stale_seen}
        if run.lateral:
            out["groundwater"] = {"calls": run.gw_calls, "stepwtd": run.stepwtd, "halo_mover": getattr(run, "halo_mover", None),
                                  "halo_probe_per_rank": comm.probe_results,
                                  "halo_exchange_us_per_call_max_over_ranks": (halo_ms_max / run.gw_calls * 1e3) if run.gw_calls else None,
                                  "algorithmic_bytes_per_cell_per_call": GW_BYTES_PER_CELL}
        if world > 1:
            out["kernel_ms_per_step_min_max_over_ranks"] = [kernel_ms_min / K, kernel_ms_max / K]
            out["halo_mover"] = getattr(run, "halo_mover", None) if run.lateral else "none (no exchange in this workload)"
            out["halo_ms_per_call"] = (halo_ms_max / run.gw_calls) if (run.lateral and run.gw_calls) else None
            if n1_ref is not None:
                out["n1_reference"] = n1_ref
                out["strong_scaling_efficiency"] = value / (world * n1_ref["value"])
            out["distributed"] = {"backend": comm.backend, "note": getattr(comm, "backend_note", None), "halo_requested": args.halo,
                                  "halo": comm.halo, "halo_note": comm.halo_note}
        if scaling_ref is not None:
            out["scaling_reference"] = scaling_ref
        if config2_ref is not None:
            out["config2_reference"] = config2_ref
        if config5_ref is not None:
            out["config5_reference"] = config5_ref
        if config5_smooth_ref is not None:
            out["config5_smooth_reference"] = config5_smooth_ref
        if options_ref is not None:
            out["options_reference"] = options_ref
        if host_ref is not None:
            out["host_path_reference"] = host_ref
        if workload == "config5":
            out["cold_start_s"] = run.cold_start_s
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    comm.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())

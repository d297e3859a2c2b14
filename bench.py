#!/usr/bin/env python3
"""Headline benchmark: column-steps/s of the Noah-MP column engine on MI355X.

A "step" is one noahmplsm call (reference drv:11) over one batch of synthetic land columns that is
already resident in HBM.  N=1 workload = BASELINE.json configs[1]: 1 048 576 synthetic land
columns, 4 soil / 0 snow layers, dynamic_veg off (DVEG=1), namelist-default physics options.
N>1: one process per GPU, every rank advances its own tile of the same size (weak scaling, the
columns are independent: no data-path collective); value = all ranks' columns x K / max-rank time.

Prints ONE JSON line on rank 0 (metric contract of the driver) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_COLSTEP = 824          # 87 reads + 119 writes x 4 B at the noahmplsm ABI
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(tables_struct, tb, ncol=32768, nsteps=24):
    """Time the CPU path on this box's host cores: the compiled reference (oracle/_ref, -O2) when
    its .so travelled with the repo, else the C restatement.  One process per core, each on its own
    `ncol`-column sample of the bench workload, `nsteps` hourly steps (bounded: ~10-30 s of CPU)."""
    import multiprocessing as mp
    from oracle import reflib
    kind = "reference" if reflib.available("O2") else "port"
    cores = max(1, min(os.cpu_count() or 1, 64))
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(kind, ncol, nsteps, r) for r in range(cores)])
    wall = max(r[0] for r in res)
    single = res[0][1]
    total = cores * ncol * nsteps
    return {"value": total / wall, "unit": "column-steps/s", "cores": cores, "kind": kind,
            "single_core": single,
            "sample": "%d procs x %d columns x %d hourly steps of the config-2 workload (%s, float32)"
                      % (cores, ncol, nsteps, "reference Fortran flang -O2" if kind == "reference" else "C restatement gcc -O2")}


def _cpu_worker(arg):
    kind, ncol, nsteps, r = arg
    import numpy as np  # noqa: F401
    from noahmp_amd import synth
    from noahmp_amd.tables import load_tables
    T, tb = load_tables("usgs")
    s = synth.config2(tb, ni=ncol // 8, nj=8, seed=100 + r)
    synth.first_step_fixups(s)
    if kind == "reference":
        from oracle.reflib import RefLib
        lib = RefLib("O2")
        lib.set_tables(T)
        step = lambda it: lib.noahmplsm(s, it, 2000, 180.0)          # noqa: E731
    else:
        from oracle.portlib import PortLib
        lib = PortLib(autobuild=False)
        lib.set_tables(T)
        step = lambda it: lib.noahmplsm(s, it, 2000, 180.0)          # noqa: E731
    dt = 0.0
    for it in range(1, nsteps + 1):
        synth.diurnal_forcing(s, (it + 5) % 24, t_offset=s.t_offset)     # forcing prep is not timed
        t0 = time.perf_counter()
        step(it)                                                          # the noahmplsm call only
        dt += time.perf_counter() - t0
    return dt, ncol * nsteps / dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--ni", type=int, default=1024)
    ap.add_argument("--nj", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_baseline_only:              # child process: never touches the GPU
        from noahmp_amd.tables import load_tables
        T, tb = load_tables("usgs")
        print("CPU_BASELINE " + json.dumps(cpu_baseline(T, tb)))
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    from noahmp_amd.tables import load_tables
    T, tb = load_tables("usgs")

    cpu = None
    import torch
    from noahmp_amd import synth
    from noahmp_amd.driver import Engine
    from noahmp_amd.state import ModelConfig

    from noahmp_amd.parallel import Comm
    torch.cuda.set_device(local_rank)
    comm = Comm()                                           # one process per GPU; "nccl" = RCCL when world > 1

    eng = Engine(T, device=local_rank, lib_path=os.environ.get("NMP_LIB"))
    if os.environ.get("NMP_BLOCK"):
        eng.set_option("block", int(os.environ["NMP_BLOCK"]))
    cfg = ModelConfig(idveg=1)                              # "dynamic_veg off", config 2
    s = synth.config2(tb, ni=args.ni, nj=args.nj, seed=2 + rank, cfg=cfg)
    synth.first_step_fixups(s)
    # 24 hourly forcing sets, resident in HBM; a step just points the argument block at the hour's set
    fkeys = ("coszin", "swdown", "glw", "t3d", "rainbl")
    forcing = []
    for h in range(24):
        synth.diurnal_forcing(s, h, t_offset=s.t_offset)
        forcing.append({k: torch.from_numpy(s.a[k].copy()).cuda(local_rank) for k in fkeys})
    d = s.to_device("cuda:%d" % local_rank)
    ncol = s.ncol

    # Layout in HBM: the device-resident state is kept sorted by (class, vegetation type, 1-K skin-temperature bin) so that a
    # wavefront holds columns that take the same branches.  Forcing arrives in tile order (as a driver would deliver it)
    # and is permuted into the sorted working set every step, INSIDE the timed region.
    perm = eng.sort_store(d)
    work = {k: torch.empty_like(forcing[0][k]) for k in fkeys}
    d.a.update(work)
    gather = eng.scatter([work[k] for k in fkeys], [forcing[0][k] for k in fkeys], perm, s.ni, s.nj)
    # One argument block, built once; a step swaps the forcing record, permutes it and enqueues the kernel
    # (noahmp_hip_step_async: device-resident state, nothing to wait for until output is due).
    sargs = d.step_args(1, 2000, 180.0)

    def step(it):
        gather.set_sources([forcing[(it + 5) % 24][k] for k in fkeys])
        gather()
        sargs.itimestep = it
        eng.noahmplsm_async(sargs)

    it = 0
    for _ in range(args.warmup):
        it += 1
        step(it)
    st, _ = eng.sync()

    def barrier():
        comm.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        it += 1
        step(it)
    st, bad_step = eng.sync()                   # waits for the K steps; tallies and device time summed over them
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms = st.kernel_ms
    n_land = st.n_land
    dt = comm.reduce_max(dt)                    # MAX over ranks
    n_land_all = comm.reduce_sum(n_land)        # columns advanced by the whole job

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU leg: after the timed GPU region, in a child process that never initialises the GPU
        # (the 64 forked workers would otherwise disturb the GPU timing and are not fork-safe here)
        import subprocess
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"],
                           capture_output=True, text=True)
        for line in r.stdout.splitlines():
            if line.startswith("CPU_BASELINE "):
                cpu = json.loads(line[len("CPU_BASELINE "):])

    if rank == 0:
        value = n_land_all / dt
        k_avg_ms = kernel_ms / args.steps
        achieved = ALG_BYTES_PER_COLSTEP * (n_land / args.steps) / (k_avg_ms * 1e-3) / 1e9
        traffic = None
        valu = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath):
            try:
                prof = json.load(open(tpath))
                # PMC counters come from a separate rocprofv3 --pmc pass of this very workload (tools/run_profile.sh)
                same = prof.get("columns_per_launch") == ncol
                traffic = prof.get("hbm_bytes_per_launch") if same else None
                dv = prof.get("derived") if same else None
                if dv:      # what actually binds this kernel (same PMC passes): VALU issue, 2 waves per SIMD
                    valu = {"insts_per_column_step": dv["valu_insts_per_column_step"], "lane_utilisation": dv["lane_utilisation"],
                            "simd_issue_utilisation": 2.0 * dv["valu_active_share_of_wave_cycles"], "waves_per_simd": 2}
            except Exception:
                traffic = None
        out = {
            "metric": "column-steps/sec", "value": value, "unit": "column-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d synthetic land columns per GPU (%dx%d tile), "
                                   "4 soil / 0 snow layers, DVEG=1 (dynamic_veg off), opt_run=1, hourly "
                                   "diurnal forcing, state resident in HBM sorted by (vegetation type, skin-temperature bin), forcing permuted "
                                   "per step inside the timed region" % (ncol, args.ni, args.nj),
                       "columns_per_gpu": ncol, "parallelism": "columns split %d-way, no collective" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "noahmp_column_kernel", "kernel_ms_avg": k_avg_ms,
                         "algorithmic_bytes_per_launch": ALG_BYTES_PER_COLSTEP * ncol, "valu": valu,
                         "note": "824 B/column-step x columns / HIP-event kernel time; the kernel is VALU-issue and "
                                 "divergence bound (23 k VALU instructions per column-step wave, 90 % lane utilisation, SIMD issue 92 % busy: "
                                 "profiles/r01_profile.md), not HBM bound (SURVEY 8d)"},
            "kernel_only_column_steps_per_s": (n_land / args.steps) / (k_avg_ms * 1e-3) * world,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    comm.close()


if __name__ == "__main__":
    main()

"""One process per GPU (= one MPI rank per GPU, mpp/module_mpp_land.F90) over torch.distributed.

Backend "nccl" is RCCL on ROCm (xGMI inside a node); "gloo" is used by the CPU tests.
The column physics needs no data-path collective: these helpers only carry the tile
assignment, the timing barrier and the metric reductions.  The one physics exchange is the 1-cell
ring LATERALFLOW reads (gw:231-252): ``exchange_halo`` below, ZWTXY before every groundwater call and
the static FDEPTH / TOPO / ISLTYP planes once.
"""
import os
import sys

from .partition import partition, neighbours, tile_geometry


class Comm:
    def __init__(self, backend=None, device_index=None, halo="torch", halo_port=None):
        """device_index: the GPU of this rank (default LOCAL_RANK); several ranks may share one GPU under gloo only
        (a 1-GPU box exercising the N>1 code path).  halo: who moves the LATERALFLOW ring -- "torch" (torch.distributed
        send/recv: RCCL for device planes under the nccl backend), or the engine's own C-ABI exchange noahmp_hip_exchange_halo
        ("rccl": ncclSend / ncclRecv issued by the library; "tcp": its socket transport), which is what a Fortran / MPI caller
        binds."""
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.device_index = self.local_rank if device_index is None else int(device_index)
        self.dist = None
        self.backend = None
        self.backend_note = None
        self.halo = halo
        self.halo_lib = None
        self._halo_plans = {}
        if halo != "torch" and self.world > 1:
            from . import abi
            self.halo_lib = abi.load_library()
            # bind the engine to THIS rank's GPU before anything of it touches a device: halo_init (RCCL transport) creates the
            # engine's stream and the communicator on the current device, which would be GPU 0 for every rank otherwise
            if halo == "rccl" or self.halo_lib.noahmp_hip_device_count() > 0:
                rc = self.halo_lib.noahmp_hip_set_device(self.device_index)
                if rc:
                    raise RuntimeError("noahmp_hip_set_device(%d): %s" % (self.device_index, self.halo_lib.noahmp_hip_last_error().decode()))
            port = halo_port or int(os.environ.get("NMP_HALO_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1))
            rc = self.halo_lib.noahmp_hip_halo_init(self.rank, self.world, os.environ.get("MASTER_ADDR", "127.0.0.1").encode(), port,
                                                    abi.HALO_RCCL if halo == "rccl" else abi.HALO_TCP)
            if rc:
                raise RuntimeError("noahmp_hip_halo_init: rc=%d %s" % (rc, self.halo_lib.noahmp_hip_last_error().decode()))
        if self.world > 1:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = backend or os.environ.get("NMP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            import datetime
            # a collective that never completes ends the job after 5 minutes instead of hanging a GPU box
            tmo = datetime.timedelta(minutes=5)
            self.backend_note = None
            try:
                kw = {}
                if backend == "nccl":
                    torch.cuda.set_device(self.device_index)
                    kw["device_id"] = torch.device("cuda", self.device_index)       # eager communicator: a broken RCCL shows here
                dist.init_process_group(backend, rank=self.rank, world_size=self.world, timeout=tmo, **kw)
            except Exception as e:                                   # noqa: BLE001
                if backend != "nccl":
                    raise
                # RCCL could not be brought up (every rank sees the same node): the control plane and the ring move to gloo -- timing
                # barrier, reductions and host-staged edges -- so that the run still completes and says so (`backend_note`)
                self.backend_note = "nccl (RCCL) initialisation failed on rank %d: %s; running over gloo" % (self.rank, str(e).splitlines()[0][:200])
                print("noahmp_amd.parallel: " + self.backend_note, file=sys.stderr, flush=True)
                try:
                    if dist.is_initialized():
                        dist.destroy_process_group()
                except Exception:                                    # noqa: BLE001
                    pass
                backend = "gloo"
                port2 = int(os.environ.get("MASTER_PORT", "29500")) + 23       # a fresh store: the first one may be half alive
                dist.init_process_group("gloo", init_method="tcp://%s:%d" % (os.environ["MASTER_ADDR"], port2), rank=self.rank,
                                        world_size=self.world, timeout=tmo)
            self.dist = dist
            self.backend = backend
            # control-plane group on the host, created while every rank is still healthy: probe_halo() agrees over it, so a rank
            # whose device send/recv has just failed does not have to use the process group that failed
            self.ctrl_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(minutes=5)) if backend == "nccl" else None
        self.probe_results = None
        self.p2p_group = None          # set by probe_halo() when device send/recv does not work: host-staged edges over a gloo group
        self.p2p_host = False

    # ---- tile assignment (mpp_land_partition_calc, mpp:227-288)
    def my_tile(self, global_nx, global_ny):
        return partition(global_nx, global_ny, self.world)[self.rank]

    def my_neighbours(self):
        return neighbours(self.rank, self.world)

    def my_geometry(self, global_nx, global_ny, halo=1):
        return tile_geometry(global_nx, global_ny, self.world, self.rank, halo)

    # ---- the one physics exchange
    def exchange_halo(self, planes, geom):
        """Fill the 1-cell ring of 2-D planes shaped (jme-jms+1, ime-ims+1) from the neighbouring ranks.

        Two phases, so that corners arrive without diagonal messages (the stencil has diagonal terms,
        gw:264-286): (1) left/right over the tile's rows, (2) down/up over full memory rows, which by
        then carry the columns received in (1) -- the order mpp_land_comlr_real / comub_real use
        (mpp:344-369, 603-613).  Planes are torch tensors: HBM tensors travel over RCCL send/recv (xGMI,
        GPU-direct), CPU tensors over gloo.  <= 4 messages of a tile edge each per plane: latency-bound, so
        all planes of a phase go out as one batch.
        """
        if self.halo_lib is not None:
            return self._exchange_cabi(planes, geom)
        if not self.dist:
            return
        key = (tuple(int(p.data_ptr()) for p in planes), tuple(sorted(geom.items())))
        plan = self._halo_plans.get(key)
        if plan is None:
            plan = self._halo_plans[key] = self._build_plan(planes, geom)
        for legs in plan:                                    # phase 1: left / right, phase 2: down / up
            self._run_phase(legs)

    def probe_halo(self):
        """One tiny send/recv round between ring neighbours with the mover exchange_halo will use, checked and AGREED on by all
        ranks (one all_reduce) -- called once at set-up by bench.py.  If device send/recv raises or returns wrong data on any rank
        (or NMP_HALO_FORCE_HOST=1), every rank switches to edges staged through the host over a gloo group: slower, but the run
        completes and says so (`halo_mover`).  Returns the mover's name."""
        if not self.dist or self.halo_lib is not None:
            return "noahmp_hip_exchange_halo" if self.halo_lib is not None else "none"
        import torch
        dist = self.dist
        dev = self._dev()
        bad = 1.0 if os.environ.get("NMP_HALO_FORCE_HOST") == "1" else 0.0
        if not bad and self.backend == "nccl":
            try:
                send = torch.full((256,), float(self.rank), device=dev)
                recv = torch.full((256,), -1.0, device=dev)
                nxt, prv = (self.rank + 1) % self.world, (self.rank - 1) % self.world
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, send, nxt), dist.P2POp(dist.irecv, recv, prv)]):
                    w.wait()
                torch.cuda.synchronize()
                if not bool((recv == float(prv)).all()):
                    bad = 1.0
            except Exception as e:                                   # noqa: BLE001  (any failure of the device mover)
                print("noahmp_amd.parallel: device send/recv failed on rank %d (%s)" % (self.rank, e), file=sys.stderr, flush=True)
                bad = 1.0
        ctrl = getattr(self, "ctrl_group", None)
        flag = torch.tensor([bad], dtype=torch.float32, device="cpu" if ctrl is not None else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=ctrl)     # nccl runs: over the gloo control group (host tensor)
        per_rank = [None] * self.world
        dist.all_gather_object(per_rank, "ok" if not bad else "device send/recv failed or forced off", group=ctrl)
        self.probe_results = per_rank
        if float(flag.item()) > 0.0:
            self.p2p_group = ctrl if ctrl is not None else dist.new_group(backend="gloo")    # (collective: every rank creates it)
            self.p2p_host = True
            self._halo_plans = {}
            return "torch.distributed over gloo, edges staged through the host (device send/recv unavailable)"
        return "torch.distributed send/recv (%s)" % ("RCCL" if self.backend == "nccl" else self.backend)

    def _build_plan(self, planes, geom):
        """Views, persistent staging buffers and P2P descriptors of the two phases for these planes: the per-call work is then one
        copy per edge, one batch_isend_irecv per phase and one copy per received edge (the exchange is latency-bound; at 8 ranks
        a step of the config-4 run is under a millisecond, so the Python side must not rebuild anything per call)."""
        dist = self.dist
        nb = self.my_neighbours()
        i0, i1 = geom["its"] - geom["ims"], geom["ite"] - geom["ims"]
        j0, j1 = geom["jts"] - geom["jms"], geom["jte"] - geom["jms"]
        rows = slice(j0, j1 + 1)
        full = slice(None)
        phases = [[(nb["left"], (rows, i0), (rows, i0 - 1)), (nb["right"], (rows, i1), (rows, i1 + 1))],
                  [(nb["down"], (j0, full), (j0 - 1, full)), (nb["up"], (j1, full), (j1 + 1, full))]]
        plan = []
        for legs in phases:
            edges, ops = [], []
            for peer, send_ix, recv_ix in legs:
                if peer < 0:
                    continue
                for p in planes:
                    sview, rview = p[send_ix], p[recv_ix]
                    host = (self.backend == "gloo" or self.p2p_host) and p.is_cuda   # gloo has no device send/recv: stage through the host
                    sbuf = sview.new_empty(sview.shape, device="cpu" if host else p.device)
                    rbuf = sbuf.new_empty(sbuf.shape)
                    ops.append(dist.P2POp(dist.isend, sbuf, peer, group=self.p2p_group))
                    ops.append(dist.P2POp(dist.irecv, rbuf, peer, group=self.p2p_group))
                    edges.append((sview, sbuf, rview, rbuf))
            plan.append((edges, ops))
        return plan

    def _run_phase(self, phase):
        edges, ops = phase
        if not ops:
            return
        for sview, sbuf, rview, rbuf in edges:
            sbuf.copy_(sview)
        for w in self.dist.batch_isend_irecv(ops):
            w.wait()                                             # RCCL: orders the current stream after the transfer, no host wait
        for sview, sbuf, rview, rbuf in edges:
            rview.copy_(rbuf, non_blocking=True)

    def _exchange_cabi(self, planes, geom):
        """The same two-phase exchange done by the engine library (noahmp_hip_exchange_halo): device planes on torch's current
        stream, host planes (numpy arrays / CPU tensors) over its socket transport."""
        import ctypes as C
        import numpy as np
        from . import abi
        n = len(planes)
        ptrs = (C.c_void_p * n)(*[(p.ctypes.data if isinstance(p, np.ndarray) else p.data_ptr()) for p in planes])
        idx = (C.c_int32 * 8)(*[geom[k] for k in ("ims", "ime", "jms", "jme", "its", "ite", "jts", "jte")])
        cuda = (not isinstance(planes[0], np.ndarray)) and planes[0].is_cuda
        stream = None
        if cuda:
            import torch
            stream = torch.cuda.current_stream().cuda_stream
        rc = self.halo_lib.noahmp_hip_exchange_halo(n, ptrs, idx, abi.MEM_DEVICE if cuda else abi.MEM_HOST, stream)
        if rc:
            raise RuntimeError("noahmp_hip_exchange_halo: rc=%d %s" % (rc, self.halo_lib.noahmp_hip_last_error().decode()))

    # ---- timing / metric plumbing
    def _dev(self):
        import torch
        return torch.device("cuda", self.device_index) if (self.dist and self.backend == "nccl") else torch.device("cpu")

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def reduce_max(self, x):
        if not self.dist:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def reduce_sum(self, x):
        if not self.dist:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather_to_root(self, array):
        """numpy array -> list of arrays on rank 0 (None elsewhere); test helper for tile reassembly."""
        if not self.dist:
            return [array]
        out = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(array, out, dst=0)
        return out

    def close(self):
        if self.halo_lib is not None:
            self.halo_lib.noahmp_hip_halo_finalize()
            self.halo_lib = None
        if self.dist:
            self.dist.destroy_process_group()

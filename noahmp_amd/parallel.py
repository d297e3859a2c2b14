"""One process per GPU (= one MPI rank per GPU, mpp/module_mpp_land.F90) over torch.distributed.

The column physics needs no data-path collective: these helpers carry the tile assignment, the timing barrier,
the metric reductions and the one physics exchange -- the 1-cell ring LATERALFLOW reads (gw:231-252): ``exchange_halo``, ZWTXY
before every groundwater call and the static FDEPTH / TOPO / ISLTYP planes once.

Process groups.  The CONTROL plane (barrier, reductions, agreement between ranks) is always a gloo group: it comes up on any
node, and every decision about the device transports below is agreed over it BEFORE any rank acts on it (a rank that decides
alone to leave a process group strands the others in it).  The DATA plane for device tensors is RCCL ("nccl" on ROCm, xGMI inside a
node), brought up second and tested with one all-reduce; when any rank fails, all ranks run the ring over gloo (host-staged edges).

Ring movers (``halo``): "rccl" / "tcp" -- the engine's own C-ABI exchange noahmp_hip_exchange_halo (noahmp_halo.hip; what a
Fortran / MPI caller binds) with its RCCL or socket transport; "torch" -- torch.distributed
batch_isend_irecv (RCCL send/recv for device planes under nccl, host-staged under gloo); "auto" (default) -- the C-ABI RCCL mover
when RCCL is up and its start + a checked probe exchange succeed on EVERY rank within a time limit, else "torch".
All movers fill the ring in ONE phase: the four tile edges and the four corner cells travel to the eight neighbours at once --
the cells mpp_land_comlr_real followed by mpp_land_comub_real (flag 99, mpp:344-369, 603-613) deliver in two dependent phases.
"""
import os
import sys

from .partition import partition, neighbours, neighbours8, tile_geometry, nprocs_xy, DIRS8


def _edge_slices(geom, dx, dy):
    """(send rows, send cols), (recv rows, recv cols) of the memory block for the neighbour at (dx, dy): the tile cells that
    touch that side go out, the ring cells on that side come in."""
    i0, i1 = geom["its"] - geom["ims"], geom["ite"] - geom["ims"]
    j0, j1 = geom["jts"] - geom["jms"], geom["jte"] - geom["jms"]
    srow = slice(j0, j1 + 1) if dy == 0 else (slice(j0, j0 + 1) if dy < 0 else slice(j1, j1 + 1))
    scol = slice(i0, i1 + 1) if dx == 0 else (slice(i0, i0 + 1) if dx < 0 else slice(i1, i1 + 1))
    rrow = slice(j0, j1 + 1) if dy == 0 else (slice(j0 - 1, j0) if dy < 0 else slice(j1 + 1, j1 + 2))
    rcol = slice(i0, i1 + 1) if dx == 0 else (slice(i0 - 1, i0) if dx < 0 else slice(i1 + 1, i1 + 2))
    return (srow, scol), (rrow, rcol)


class Comm:
    def __init__(self, backend=None, device_index=None, halo="auto", halo_port=None):
        """device_index: the GPU of this rank (default LOCAL_RANK); several ranks may share one GPU under gloo only
        (a 1-GPU box exercising the N>1 code path).  halo: who moves the LATERALFLOW ring (module docstring)."""
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.device_index = self.local_rank if device_index is None else int(device_index)
        self.dist = None
        self.backend = None
        self.backend_note = None
        self.dev_group = None          # RCCL group for device tensors (None: everything over the gloo control group)
        self.halo_requested = halo
        self.halo = halo if halo != "auto" else "torch"
        self.halo_note = None
        self.halo_lib = None
        self._halo_plans = {}
        self.probe_results = None
        self.p2p_host = False          # device planes staged through the host (gloo data plane, or after a failed probe)
        if self.world == 1:
            return
        import datetime
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        want = backend or os.environ.get("NMP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        self._tmo = datetime.timedelta(minutes=5)         # a collective that never completes ends the job instead of hanging a GPU box
        dist.init_process_group("gloo", rank=self.rank, world_size=self.world, timeout=self._tmo)     # control plane
        self.dist = dist
        self.backend = "gloo"
        if want == "nccl":
            # Three steps, each AGREED over the control plane before the next starts, so that no rank is left alone inside a collective:
            # (1) bind the GPU; (2) create the RCCL group (new_group is itself collective over the control plane, it does not touch RCCL yet);
            # (3) one all-reduce, asynchronous and polled against a deadline -- the call that creates the communicator and the one a broken
            # RCCL hangs in.  A failure that is an exception on some rank: all ranks run over gloo.  A failure that is a TIMEOUT: that
            # collective may still be pending on the device, nothing sane can follow in this process -- all ranks leave together, non-zero.
            import time
            err, pg, hung = None, None, False
            try:
                torch.cuda.set_device(self.device_index)
            except Exception as e:                                # noqa: BLE001
                err = "set_device(%d): %s" % (self.device_index, str(e).splitlines()[0][:160] if str(e) else repr(e))
            if not self._agree_any(err is not None):
                try:
                    pg = dist.new_group(backend="nccl", timeout=self._tmo)
                    t = torch.ones(1, device=torch.device("cuda", self.device_index))
                    work = dist.all_reduce(t, group=pg, async_op=True)
                    limit = time.monotonic() + float(os.environ.get("NMP_RCCL_PROBE_TIMEOUT_S", "120"))
                    while not work.is_completed():
                        if time.monotonic() > limit:
                            hung = True
                            raise TimeoutError("the probe all-reduce over RCCL did not complete within its deadline")
                        time.sleep(0.005)
                    work.wait()
                    torch.cuda.synchronize()
                    if int(t.item()) != self.world:
                        err = "all-reduce over RCCL returned %r" % t.item()
                except Exception as e:                            # noqa: BLE001
                    err = str(e).splitlines()[0][:200] if str(e) else repr(e)
                    pg = None
            if self._agree_any(hung):
                print("noahmp_amd.parallel: rank %d: an RCCL collective hung on %s; all ranks stop (set NMP_DIST_BACKEND=gloo to run "
                      "without RCCL)" % (self.rank, "this rank" if hung else "another rank"), file=sys.stderr, flush=True)
                os._exit(3)          # a pending collective cannot be cancelled; destructors would wait for it
            bad = self._agree_any(err is not None)                # every rank learns whether ANY rank failed, then all act alike
            if bad:
                self.backend_note = ("nccl (RCCL) initialisation failed on %s: %s; running over gloo"
                                     % ("rank %d" % self.rank if err else "another rank", err or "see its message"))
                print("noahmp_amd.parallel: " + self.backend_note, file=sys.stderr, flush=True)
            else:
                self.backend, self.dev_group = "nccl", pg
        self.p2p_host = self.backend == "gloo"
        if halo in ("rccl", "tcp"):
            self.halo_lib, self.halo = self._start_cabi(halo, halo_port), halo      # explicit request: a failure is an error
        elif halo == "auto" and os.environ.get("NMP_HALO_AUTO", "1") != "0" and (
                self.backend == "nccl" or os.environ.get("NMP_HALO_AUTO_TRANSPORT") == "tcp"):
            self._try_cabi_rccl(halo_port)

    @classmethod
    def solo(cls, device_index=0):
        """A one-rank communicator whatever the environment says: rank 0 of a `--gpus N` run uses it to time the SAME workload on the
        whole grid by itself (bench.py's n1_reference) before the distributed region starts."""
        c = cls.__new__(cls)
        c.rank, c.local_rank, c.world, c.device_index = 0, 0, 1, int(device_index)
        c.dist = c.backend = c.backend_note = c.dev_group = c.halo_note = c.halo_lib = c.probe_results = None
        c.halo_requested = c.halo = "torch"
        c._halo_plans = {}
        c.p2p_host = False
        return c

    # ---- agreement over the control plane
    def _agree_any(self, flag):
        import torch
        t = torch.tensor([1.0 if flag else 0.0])
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return bool(t.item() > 0.0)

    # ---- the engine's own exchange (C-ABI)
    def _halo_port(self, halo_port):
        return halo_port or int(os.environ.get("NMP_HALO_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1))

    def _start_cabi(self, halo, halo_port):
        from . import abi
        lib = abi.load_library()
        # bind the engine to THIS rank's GPU before anything of it touches a device: halo_init (RCCL transport) creates the
        # engine's stream and the communicator on the current device, which would be GPU 0 for every rank otherwise
        if halo == "rccl" or lib.noahmp_hip_device_count() > 0:
            rc = lib.noahmp_hip_set_device(self.device_index)
            if rc:
                raise RuntimeError("noahmp_hip_set_device(%d): %s" % (self.device_index, lib.noahmp_hip_last_error().decode()))
        rc = lib.noahmp_hip_halo_init(self.rank, self.world, os.environ.get("MASTER_ADDR", "127.0.0.1").encode(), self._halo_port(halo_port),
                                      abi.HALO_RCCL if halo == "rccl" else abi.HALO_TCP)
        if rc:
            raise RuntimeError("noahmp_hip_halo_init: rc=%d %s" % (rc, lib.noahmp_hip_last_error().decode()))
        return lib

    def _probe_cabi(self, lib, device):
        """One checked exchange with the C-ABI mover on a virtual 3 x 3-cells-per-rank grid whose values name their cell: every
        ring cell (edges and corners, all eight directions) must come back as the neighbour's tile cell."""
        import ctypes as C
        import numpy as np
        from . import abi
        npx, npy = nprocs_xy(self.world)
        gx, gy = 3 * npx, 3 * npy
        geo = tile_geometry(gx, gy, self.world, self.rank, halo=1)
        y, x = np.meshgrid(np.arange(gy), np.arange(gx), indexing="ij")
        gf = (100.0 * y + x + 1.0).astype(np.float32)
        sl = (slice(geo["jms"] - 1, geo["jme"]), slice(geo["ims"] - 1, geo["ime"]))
        f = gf[sl].copy()
        ring = np.ones(f.shape, dtype=bool)
        ring[geo["jts"] - geo["jms"]:geo["jte"] - geo["jms"] + 1, geo["its"] - geo["ims"]:geo["ite"] - geo["ims"] + 1] = False
        f[ring] = -1.0
        idx = (C.c_int32 * 8)(*[geo[k] for k in ("ims", "ime", "jms", "jme", "its", "ite", "jts", "jte")])
        if device:
            import torch
            with torch.cuda.device(self.device_index):        # (a helper thread starts with device 0 as its current device)
                t = torch.from_numpy(f).to(torch.device("cuda", self.device_index))
                ptrs = (C.c_void_p * 1)(t.data_ptr())
                rc = lib.noahmp_hip_exchange_halo(1, ptrs, idx, abi.MEM_DEVICE, torch.cuda.current_stream(self.device_index).cuda_stream)
                torch.cuda.synchronize(self.device_index)
                f = t.cpu().numpy()
        else:
            ptrs = (C.c_void_p * 1)(f.ctypes.data)
            rc = lib.noahmp_hip_exchange_halo(1, ptrs, idx, abi.MEM_HOST, None)
        if rc:
            return "noahmp_hip_exchange_halo: rc=%d %s" % (rc, lib.noahmp_hip_last_error().decode())
        return None if np.array_equal(f, gf[sl]) else "probe exchange returned wrong ring cells"

    def _try_cabi_rccl(self, halo_port):
        """halo="auto" with RCCL up: start the engine's RCCL mover and run one checked probe exchange in a helper thread with a time
        limit (a communicator that never forms must not park the job), then AGREE: it is used only if every rank succeeded."""
        import threading
        # NMP_HALO_AUTO_TRANSPORT=tcp (tests): the same start / probe / agreement with the socket transport on host planes, so that the
        # selection logic runs on a CPU-only box; NMP_HALO_AUTO_FAIL_RANK=r makes rank r report a failed probe
        transport = "tcp" if os.environ.get("NMP_HALO_AUTO_TRANSPORT") == "tcp" else "rccl"
        limit = float(os.environ.get("NMP_HALO_AUTO_TIMEOUT_S", "90"))
        os.environ.setdefault("NMP_HALO_TIMEOUT_S", str(int(max(limit - 30.0, 20.0))))     # the rendezvous' own deadlines end first
        box = {}

        def work():                      # touches only `box`: a helper that answers after the time limit must not change the mover
            try:
                box["lib"] = self._start_cabi(transport, halo_port)
                box["err"] = self._probe_cabi(box["lib"], device=(transport == "rccl"))
                if os.environ.get("NMP_HALO_AUTO_FAIL_RANK") == str(self.rank):
                    box["err"] = "forced failure (NMP_HALO_AUTO_FAIL_RANK)"
            except Exception as e:                                # noqa: BLE001
                box["err"] = str(e).splitlines()[0][:200] if str(e) else repr(e)

        th = threading.Thread(target=work, daemon=True)
        th.start()
        th.join(limit)
        late = th.is_alive()
        if self._agree_any(late):
            # A helper that has not answered is still INSIDE the library (rendezvous, communicator creation) and the engine's globals are not
            # thread-safe: nothing of the library may be called from this process any more.  All ranks learn of it and leave together.
            print("noahmp_amd.parallel: rank %d: the engine's RCCL mover gave no answer within %g s on %s; all ranks stop (--halo torch "
                  "skips it)" % (self.rank, limit, "this rank" if late else "another rank"), file=sys.stderr, flush=True)
            os._exit(3)
        err = box.get("err")
        if self._agree_any(err is not None):
            self.halo_note = ("C-ABI RCCL mover not used (%s): torch.distributed send/recv moves the ring"
                              % (("rank %d: %s" % (self.rank, err)) if err else "another rank failed"))
            print("noahmp_amd.parallel: " + self.halo_note, file=sys.stderr, flush=True)
            if box.get("lib") is not None:
                box["lib"].noahmp_hip_halo_finalize()
            self.halo_lib, self.halo = None, "torch"
        else:
            self.halo_lib, self.halo = box["lib"], transport
            # hipSetDevice is per thread: the helper bound ITS thread; bind the caller's too before it uses the library
            if transport == "rccl" or self.halo_lib.noahmp_hip_device_count() > 0:
                self.halo_lib.noahmp_hip_set_device(self.device_index)

    # ---- tile assignment (mpp_land_partition_calc, mpp:227-288)
    def my_tile(self, global_nx, global_ny):
        return partition(global_nx, global_ny, self.world)[self.rank]

    def my_neighbours(self):
        return neighbours(self.rank, self.world)

    def my_geometry(self, global_nx, global_ny, halo=1):
        return tile_geometry(global_nx, global_ny, self.world, self.rank, halo)

    # ---- the one physics exchange
    def exchange_halo(self, planes, geom):
        """Fill the 1-cell ring of 2-D planes shaped (jme-jms+1, ime-ims+1) from the neighbouring ranks, corners included (the
        stencil has diagonal terms, gw:264-286) -- one phase: <= 8 messages per call, all planes of a call sharing them
        (~25 KB per edge neighbour at the config-4 grid on 8 ranks: latency-bound).  Planes are torch tensors: HBM tensors travel
        over RCCL send/recv (xGMI, GPU-direct), CPU tensors over gloo."""
        if self.halo_lib is not None:
            return self._exchange_cabi(planes, geom)
        if not self.dist:
            return
        key = (tuple(int(p.data_ptr()) for p in planes), tuple(sorted(geom.items())))
        plan = self._halo_plans.get(key)
        if plan is None:
            plan = self._halo_plans[key] = self._build_plan(planes, geom)
        self._run_plan(plan)

    def probe_halo(self):
        """Once at set-up (bench.py): one tiny send/recv round between ring neighbours with the mover exchange_halo will use,
        checked and AGREED on by all ranks.  If device send/recv raises or returns wrong data on any rank (or
        NMP_HALO_FORCE_HOST=1), every rank switches to edges staged through the host over the gloo group: slower, but the run
        completes and says so.  Returns the mover's name."""
        if not self.dist:
            return "none"
        if self.halo_lib is not None:
            return "noahmp_hip_exchange_halo (%s transport, one phase)" % ("RCCL" if self.halo == "rccl" else "socket")
        import torch
        dist = self.dist
        bad = os.environ.get("NMP_HALO_FORCE_HOST") == "1"
        if not bad and self.dev_group is not None:
            try:
                dev = torch.device("cuda", self.device_index)
                send = torch.full((256,), float(self.rank), device=dev)
                recv = torch.full((256,), -1.0, device=dev)
                nxt, prv = (self.rank + 1) % self.world, (self.rank - 1) % self.world
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, send, nxt, group=self.dev_group),
                                                 dist.P2POp(dist.irecv, recv, prv, group=self.dev_group)]):
                    w.wait()
                torch.cuda.synchronize()
                if not bool((recv == float(prv)).all()):
                    bad = True
            except Exception as e:                                   # noqa: BLE001  (any failure of the device mover)
                print("noahmp_amd.parallel: device send/recv failed on rank %d (%s)" % (self.rank, e), file=sys.stderr, flush=True)
                bad = True
        mine = "ok" if not bad else "device send/recv failed or forced off"
        any_bad = self._agree_any(bad)
        per_rank = [None] * self.world
        dist.all_gather_object(per_rank, mine)
        self.probe_results = per_rank
        if any_bad or self.dev_group is None:
            self.p2p_host = True
            self._halo_plans = {}
            if self.backend == "nccl" or any_bad:
                return "torch.distributed over gloo, edges staged through the host (device send/recv unavailable)"
            return "torch.distributed send/recv (gloo, one phase)"
        return "torch.distributed send/recv (RCCL, one phase)"

    def _build_plan(self, planes, geom):
        """Views, persistent staging buffers and P2P descriptors for these planes: the per-call work is then one copy per outgoing
        edge, ONE batch_isend_irecv and one copy per received edge (the exchange is latency-bound; at 8 ranks a step of the
        config-4 run is about half a millisecond, so the Python side must not rebuild anything per call)."""
        dist = self.dist
        nb = neighbours8(self.rank, self.world)
        group = None if self.p2p_host else self.dev_group
        edges, ops = [], []
        for name, dx, dy in DIRS8:
            peer = nb[name]
            if peer < 0:
                continue
            send_ix, recv_ix = _edge_slices(geom, dx, dy)
            for p in planes:
                sview, rview = p[send_ix], p[recv_ix]
                host = self.p2p_host and p.is_cuda              # gloo has no device send/recv: stage through the host
                sbuf = sview.new_empty(sview.shape, device="cpu" if host else p.device)
                rbuf = sbuf.new_empty(sbuf.shape)
                ops.append(dist.P2POp(dist.isend, sbuf, peer, group=group))
                ops.append(dist.P2POp(dist.irecv, rbuf, peer, group=group))
                edges.append((sview, sbuf, rview, rbuf))
        return edges, ops

    def _run_plan(self, plan):
        edges, ops = plan
        if not ops:
            return
        for sview, sbuf, rview, rbuf in edges:
            sbuf.copy_(sview)
        for w in self.dist.batch_isend_irecv(ops):
            w.wait()                                             # RCCL: orders the current stream after the transfer, no host wait
        for sview, sbuf, rview, rbuf in edges:
            rview.copy_(rbuf, non_blocking=True)

    def _exchange_cabi(self, planes, geom):
        """The same exchange done by the engine library (noahmp_hip_exchange_halo): device planes on torch's current
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

    # ---- timing / metric plumbing (control plane: host tensors over gloo)
    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def reduce_max(self, x):
        if not self.dist:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def reduce_min(self, x):
        return -self.reduce_max(-float(x))

    def reduce_sum(self, x):
        if not self.dist:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
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

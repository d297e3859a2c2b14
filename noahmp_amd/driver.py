"""Host-side mirror of the reference's operator interface for the hot path.

``noahmplsm(store, itimestep, yr, julian)`` has the meaning of
``module_sf_noahmpdrv :: noahmplsm`` (reference drv:11): one call advances every land column
of the tile by one timestep.  It packs the caller's arrays into the C-ABI block and calls the
HIP engine -- nothing else.  A fatal column raises :class:`NoahMPFatal` (the reference STOPs via
wrf_error_fatal, util/module_wrf_utilities.F:12-24).
"""
import ctypes as C
import os

from . import abi
from .state import DeviceColumnStore


class NoahMPFatal(RuntimeError):
    def __init__(self, code, i, j, msg):
        super().__init__("Noah-MP fatal %d (%s) at I=%d J=%d: %s" % (code, abi.error_name(code), i, j, msg))
        self.code, self.i, self.j = code, i, j


class Engine:
    """Thin handle on libnoahmp_hip.so (one per process == one per GPU/MPI rank)."""

    def __init__(self, tables, device=None, lib_path=None):
        self.lib = abi.load_library(lib_path)
        if self.lib.noahmp_hip_device_count() < 1:
            raise RuntimeError("Noah-MP HIP engine: no GPU visible and there is no CPU fallback")
        if device is not None:
            rc = self.lib.noahmp_hip_set_device(int(device))
            if rc:
                raise RuntimeError(self.lib.noahmp_hip_last_error().decode())
        self.tables = tables
        rc = self.lib.noahmp_hip_set_tables(C.byref(tables))
        if rc:
            raise RuntimeError("noahmp_hip_set_tables: " + self.lib.noahmp_hip_last_error().decode())
        self.last_status = abi.Status()
        if os.environ.get("NMP_JIT") in ("0", "1"):          # run-time specialisation for every option set (noahmp_jit.hip)
            self.set_option("jit_option_kernels", int(os.environ["NMP_JIT"]))

    @property
    def exact_libm(self):
        """True if the library was built with the reference libm's algorithms (bit-identical results)."""
        return self.lib.noahmp_hip_set_option(b"exact_libm", -1) == 1

    def set_option(self, key, value):
        return self.lib.noahmp_hip_set_option(key.encode(), int(value))

    def _apply_ranges(self, ranges):
        """Declare (or withdraw) the class ranges of the store about to be stepped (set_option sorted_*_columns)."""
        if ranges != getattr(self.lib, "_ranges_set", None):          # the engine is one per process: remember it on the library
            n0, n1 = ranges if ranges else (-1, -1)
            self.lib.noahmp_hip_set_option(b"sorted_land_columns", int(n0))
            self.lib.noahmp_hip_set_option(b"sorted_glacier_columns", int(n1))
            self.lib._ranges_set = ranges

    def fetch(self):
        """Bring the host arrays of a resident run ("resident_state" + "lazy_download") up to date."""
        rc = self.lib.noahmp_hip_fetch(None)
        if rc:
            raise RuntimeError("noahmp_hip_fetch: rc=%d %s" % (rc, self.lib.noahmp_hip_last_error().decode()))

    def noahmplsm(self, store, itimestep, yr, julian, stream=None, check=True):
        a = store.step_args(itimestep, yr, julian)
        mem = abi.MEM_DEVICE if isinstance(store, DeviceColumnStore) else abi.MEM_HOST
        self._apply_ranges(a._ranges if mem == abi.MEM_DEVICE else None)
        st = abi.Status()
        rc = self.lib.noahmp_hip_step(C.byref(a), mem, stream, C.byref(st))
        self.last_status = st
        if rc < 0:
            raise RuntimeError("noahmp_hip_step: " + self.lib.noahmp_hip_last_error().decode())
        if rc > 0 and check:
            raise NoahMPFatal(rc, st.i, st.j, self.lib.noahmp_hip_error_string(rc).decode())
        return st

    def noahmp_init(self, store, fndsnowh=True, stream=None, check=True):
        """Cold start on the device: NOAHMP_INIT's per-column part + SNOW_INIT (reference drv:988-1283), in place.
        ide+1 / jde+1 are passed as the reference driver does (hdrv:291; the routine loops to min(ite, ide-1))."""
        a = store.step_args(1, 2000, 1.0)
        a.ide += 1
        a.jde += 1
        mem = abi.MEM_DEVICE if isinstance(store, DeviceColumnStore) else abi.MEM_HOST
        st = abi.Status()
        rc = self.lib.noahmp_hip_init(C.byref(a), store.cfg.iswater, 1 if fndsnowh else 0, mem, stream, C.byref(st))
        self.last_status = st
        if rc < 0:
            raise RuntimeError("noahmp_hip_init: " + self.lib.noahmp_hip_last_error().decode())
        if rc > 0 and check:
            raise NoahMPFatal(rc, st.i, st.j, "lsminit: out of range value of ISLTYP (drv:1018)")
        return st

    def wtable_mmf(self, store, stream=None):
        """WTABLE_mmf_noahmp (reference gw:14): lateral groundwater flow + water-table update, in place.
        A decomposed domain must have exchanged the ZWTXY halo first (parallel.exchange_halo)."""
        w = store.wtable_args()
        mem = abi.MEM_DEVICE if isinstance(store, DeviceColumnStore) else abi.MEM_HOST
        st = abi.Status()
        rc = self.lib.noahmp_hip_wtable_mmf(C.byref(w), mem, stream, C.byref(st))
        self.last_status = st
        if rc:
            raise RuntimeError("noahmp_hip_wtable_mmf: rc=%d %s" % (rc, self.lib.noahmp_hip_last_error().decode()))
        return st

    def wtable_mmf_async(self, wargs, stream=None):
        """Enqueue one WTABLE_mmf_noahmp call described by a prepared WtableArgs block (store.wtable_args(), device pointers)."""
        rc = self.lib.noahmp_hip_wtable_mmf_async(C.byref(wargs), stream)
        if rc:
            raise RuntimeError("noahmp_hip_wtable_mmf_async: rc=%d %s" % (rc, self.lib.noahmp_hip_last_error().decode()))

    def wtable_lateral_async(self, wargs, qlat, stream=None):
        """First half of WTABLE_mmf_noahmp for a sorted run: KCELL / HEAD + the QLAT stencil on the TILE-order planes of `wargs`
        (wtd, fdepth, topo, isltyp with the ring; xland, xice, ivgtyp, area) -> `qlat` (device tensor shaped like the memory block)."""
        rc = self.lib.noahmp_hip_wtable_lateral_async(C.byref(wargs), qlat.data_ptr(), stream)
        if rc:
            raise RuntimeError("noahmp_hip_wtable_lateral_async: rc=%d %s" % (rc, self.lib.noahmp_hip_last_error().decode()))

    def wtable_columns_async(self, wargs, qlat, stream=None):
        """Second half: river flux, deep recharge, UPDATEWTD and the accumulators on a block whose columns are in any order, QLAT
        taken from `qlat` (same order)."""
        rc = self.lib.noahmp_hip_wtable_columns_async(C.byref(wargs), qlat.data_ptr(), stream)
        if rc:
            raise RuntimeError("noahmp_hip_wtable_columns_async: rc=%d %s" % (rc, self.lib.noahmp_hip_last_error().decode()))

    def forcing_prep(self, store, lon, rain_rate, iday, ihour, iminute=0, isecond=0, scale_vegfra=False, stream=None,
                     first_step=False, wait=True):
        """Device-resident forcing preparation (reference hdrv:336-354 + CALC_DECLIN): `lon` and `rain_rate` are
        device tensors shaped like a 2-D field; level 1 of t3d/qv3d/u_phy/v_phy/p8w3d holds the new forcing.
        first_step adds the driver's itime = 1 guesses of EAH / TAH / CH / CM (hdrv:374-384).  Returns JULIAN."""
        assert isinstance(store, DeviceColumnStore), "forcing_prep works on device-resident arrays"
        a = store.step_args(1, 2000, 1.0)
        jul = C.c_float(0)
        st = abi.Status()
        rc = self.lib.noahmp_hip_forcing_prep(C.byref(a), lon.data_ptr(), rain_rate.data_ptr(), iday, ihour, iminute,
                                              isecond, store.cfg.zlvl, (1 if scale_vegfra else 0) | (2 if first_step else 0), C.byref(jul),
                                              abi.MEM_DEVICE, stream, C.byref(st) if wait else None)
        self.last_status = st
        if rc:
            raise RuntimeError("noahmp_hip_forcing_prep: rc=%d %s" % (rc, self.lib.noahmp_hip_last_error().decode()))
        return jul.value

    def forcing_interpolate(self, store, rec_a, rec_b, idts, idts2, rain_rate, stream=None, wait=True):
        """Device-resident hrldas_input_interpolate / hrldas_input_copy (netcdf_io:1351-1403): `rec_a`, `rec_b` are
        dicts of device tensors keyed t q u v p lw sw pcp [fpar lai] (rec_b None = take rec_a as it is); `rain_rate`
        receives RAINBL_tmp.  Follow with forcing_prep(scale_vegfra=True)."""
        assert isinstance(store, DeviceColumnStore), "forcing_interpolate works on device-resident arrays"
        a = store.step_args(1, 2000, 1.0)

        def rec(d):
            r = abi.ForcingRecord()
            for n in abi.FORCING_RECORD_FIELDS:
                if d.get(n) is not None:
                    setattr(r, n, d[n].data_ptr())
            return r
        ra = rec(rec_a)
        rb = rec(rec_b) if rec_b is not None else None
        st = abi.Status()
        rc = self.lib.noahmp_hip_forcing_interpolate(C.byref(a), C.byref(ra), C.byref(rb) if rb is not None else None,
                                                     idts, idts2, rain_rate.data_ptr(), abi.MEM_DEVICE, stream,
                                                     C.byref(st) if wait else None)
        self.last_status = st
        if rc:
            raise RuntimeError("noahmp_hip_forcing_interpolate: rc=%d %s" % (rc, self.lib.noahmp_hip_last_error().decode()))
        return st

    def forcing_interpolate_prep(self, store, rec_a, rec_b, idts, idts2, rain_rate, lon, iday, ihour, iminute=0, isecond=0,
                                 scale_vegfra=False, first_step=False, stream=None, wait=True):
        """forcing_interpolate followed by forcing_prep in one launch (noahmp_hip_forcing_interpolate_prep): same arguments, same
        results.  Returns JULIAN."""
        assert isinstance(store, DeviceColumnStore), "forcing_interpolate_prep works on device-resident arrays"
        a = store.step_args(1, 2000, 1.0)

        def rec(d):
            r = abi.ForcingRecord()
            for n in abi.FORCING_RECORD_FIELDS:
                if d.get(n) is not None:
                    setattr(r, n, d[n].data_ptr())
            return r
        ra = rec(rec_a)
        rb = rec(rec_b) if rec_b is not None else None
        jul = C.c_float(0)
        st = abi.Status()
        rc = self.lib.noahmp_hip_forcing_interpolate_prep(C.byref(a), C.byref(ra), C.byref(rb) if rb is not None else None, idts, idts2,
                                                          rain_rate.data_ptr(), lon.data_ptr(), iday, ihour, iminute, isecond, store.cfg.zlvl,
                                                          (1 if scale_vegfra else 0) | (2 if first_step else 0), C.byref(jul), abi.MEM_DEVICE,
                                                          stream, C.byref(st) if wait else None)
        self.last_status = st
        if rc:
            raise RuntimeError("noahmp_hip_forcing_interpolate_prep: rc=%d %s" % (rc, self.lib.noahmp_hip_last_error().decode()))
        return jul.value

    # ---- sorted device-resident layout
    class Gather:
        """A prepared noahmp_hip_gather_fields call: dst tensors <- src tensors through a column permutation."""

        def __init__(self, lib, dst, src, perm, ni, nj):
            n = len(dst)
            self.lib, self.n, self.ni, self.nj, self.perm = lib, n, ni, nj, perm
            self.keep = (dst, src)
            self.dst = (C.c_void_p * n)(*[t.data_ptr() for t in dst])
            self.src = (C.c_void_p * n)(*[t.data_ptr() for t in src])
            self.nlev = (C.c_int * n)(*[(t.shape[1] if t.dim() == 3 else 1) for t in dst])

        def set_sources(self, src):
            for i, t in enumerate(src):
                self.src[i] = t.data_ptr()

        def set_dests(self, dst):
            """other destination tensors of the same shapes (e.g. the second of two forcing working sets)"""
            for i, t in enumerate(dst):
                self.dst[i] = t.data_ptr()

        def __call__(self, stream=None):
            rc = self.lib.noahmp_hip_gather_fields(self.n, self.dst, self.src, self.nlev, self.perm.data_ptr(), self.ni,
                                                   self.nj, stream)
            if rc:
                raise RuntimeError("noahmp_hip_gather_fields: rc=%d" % rc)

    def stream_sync(self, stream=None):
        """Wait for `stream` (None: the engine's own stream)."""
        rc = self.lib.noahmp_hip_stream_sync(stream)
        if rc:
            raise RuntimeError("noahmp_hip_stream_sync: " + self.lib.noahmp_hip_last_error().decode())

    def set_veg_order(self, first=None, ncat=64):
        """Order of the vegetation categories inside the land range of later sorts (noahmp_hip_sort_set_veg_order): `first` = categories
        in the order their columns should RUN (e.g. the expensive ones first, so that cheap waves form the tail of a launch); categories
        it does not name follow in numeric order.  None: the categories' own numbers."""
        if not first:
            rc = self.lib.noahmp_hip_sort_set_veg_order(None, 0)
        else:
            seq = [int(v) for v in first] + [v for v in range(ncat) if v not in set(int(x) for x in first)]
            rank = (C.c_int32 * ncat)()
            for place, v in enumerate(seq[:ncat]):
                rank[v] = place
            rc = self.lib.noahmp_hip_sort_set_veg_order(rank, ncat)
        if rc:
            raise RuntimeError("noahmp_hip_sort_set_veg_order: " + self.lib.noahmp_hip_last_error().decode())

    def sort_store(self, store, tsk_bin=1.0, veg=True, snow=True, snow_first=False, allow_lateral=False, tair=False, band=None, cost=False):
        """Reorder a DeviceColumnStore in place so that columns with equal (class, vegetation type, snow-layer count,
        skin-temperature bin) are adjacent, and return the permutation as an int32 device tensor: sorted position p holds the
        column that sits at linear tile index perm[p] of the ORIGINAL tile order (a second call on an already sorted
        store re-sorts it and returns the composed permutation).  Key, stable radix sort and the permutation of every
        array run on the device (noahmp_hip_sort_columns / noahmp_hip_permute_step_arrays).  Columns are independent
        (every option except the MMF lateral flow), so this only changes which lane computes which column; wavefronts
        then hold columns that take the same branches: class and vegetation type select code paths and are static, the
        snow-layer count bounds the layer loops and changes slowly (see sort_staleness), the skin temperature (`tsk_bin`
        K wide bins, 0 = off) is a cheap proxy for the stability / freezing regime a column is in.  Forcing that arrives
        in tile order goes through `scatter`.  band: name of an int32 device plane of the store (values 0..31, static; it is
        permuted with the state) used as a sub-key between the snow-layer count and the temperature bin (noahmp_hip_sort_set_band: an argument of the next sort / staleness call, consumed by it),
        e.g. the 15-degree longitude band of a lat/lon grid: a wavefront's columns then share their local solar time."""
        import numpy as np
        import torch
        assert isinstance(store, DeviceColumnStore)
        # the stencil of WTABLE_mmf_noahmp needs the (i,j) neighbourhood: a sorted OPT_RUN=5 store must return its planes to
        # tile order around every groundwater call (Engine.wtable_mmf_sorted)
        assert store.cfg.iopt_run != 5 or allow_lateral, "OPT_RUN=5 needs the (i,j) order for WTABLE_mmf_noahmp"
        n = store.ncol
        flags = ((abi.SORT_VEG if veg else 0) | (abi.SORT_SNOW if snow else 0) | (abi.SORT_SNOW_FIRST if snow_first else 0) |
                 (abi.SORT_TAIR if tair else 0) |        # tair: temperature bins from the forcing air temperature instead of TSK
                 (abi.SORT_COST if cost else 0))         # cost: bucket of the columns' own trip counts in the last step (set_option record_cost)
        a = store.step_args(1, 2000, 1.0)
        perm = torch.empty(n, dtype=torch.int32, device=store.device)
        keys = torch.empty(n, dtype=torch.int32, device=store.device)
        counts = (C.c_int64 * 3)()
        torch.cuda.current_stream().synchronize()          # tensors written by torch kernels are read on the engine's stream
        self.lib.noahmp_hip_sort_set_band(store.a[band].data_ptr() if band else None)
        rc = self.lib.noahmp_hip_sort_columns(C.byref(a), flags, int(round(tsk_bin * 1000)) if tsk_bin else 0,
                                              perm.data_ptr(), keys.data_ptr(), counts, None)
        if rc:
            raise RuntimeError("noahmp_hip_sort_columns: rc=%d %s" % (rc, self.lib.noahmp_hip_last_error().decode()))
        # The sorted block's arrays are carved out of ONE allocation, packed back to back (256-byte aligned, each start one odd multiple of
        # 256 B further): separately allocated planes all start on 2 MiB boundaries, so the words of one column sit at the same offset in
        # 2 MiB-aligned regions of all 115 arrays -- the column kernel's wavefront then asks for 206 lines whose low address bits agree.
        # A/B on one box (round 6): column kernel 3.235 / 3.234 -> 3.215 / 3.212 / 3.212 ms (-0.7 %) whatever the shift (256 B .. 70 KB).
        pad = int(os.environ.get("NMP_POOL_SHIFT", "4352"))         # 17 x 256 B; 0 = separate allocations (rounds 2-5)
        if pad:
            ten = [(k, v) for k, v in store.a.items() if not isinstance(v, np.ndarray)]
            tot = sum(((v.numel() * v.element_size() + 255) // 256) * 256 + pad for k, v in ten) + 4096
            pool = torch.empty(tot, dtype=torch.uint8, device=store.device)
            new, off = {}, 0
            for k, v in ten:
                off += pad
                nb = v.numel() * v.element_size()
                new[k] = pool[off:off + nb].view(v.dtype).view(v.shape)
                off += ((nb + 255) // 256) * 256
            for k, v in store.a.items():
                if isinstance(v, np.ndarray):
                    new[k] = v
        else:
            new = {k: (v if isinstance(v, np.ndarray) else torch.empty_like(v)) for k, v in store.a.items()}
        old = store.a
        store.a = new
        b = store.step_args(1, 2000, 1.0)
        rc = self.lib.noahmp_hip_permute_step_arrays(C.byref(a), C.byref(b), perm.data_ptr(), None)
        if rc:
            store.a = old
            raise RuntimeError("noahmp_hip_permute_step_arrays: rc=%d %s" % (rc, self.lib.noahmp_hip_last_error().decode()))
        extra = [k for k in old if k not in abi.FIELD_INFO and not isinstance(old[k], np.ndarray)]    # e.g. the MMF planes
        for i in range(0, len(extra), 32):
            chunk = extra[i:i + 32]
            Engine.Gather(self.lib, [new[k] for k in chunk], [old[k] for k in chunk], perm, store.ni, store.nj)()
        prev = getattr(store, "sort_perm", None)
        if prev is not None:                                 # already sorted: compose with the earlier permutation
            total = torch.empty_like(perm)
            Engine.Gather(self.lib, [total], [prev], perm, store.ni, store.nj)()
            perm = total
        self.stream_sync()
        # classes are contiguous now: land, land ice, skipped -- each range gets its own kernel (noahmp_engine.hip, launch_any)
        store.class_ranges = (int(counts[0]), int(counts[1]))
        store.sort_perm, store.sort_keys, store.sort_flags, store.sort_band = perm, keys, flags, band
        return perm

    def sort_staleness(self, store):
        """Number of columns of a sorted store whose (class, vegetation type, snow-layer count) no longer is what they were
        sorted by -- snow layers that appeared or vanished (lsm:7044, 7110, 7177, 7294-7343).  Waits for the engine's stream."""
        a = store.step_args(1, 2000, 1.0)
        changed = C.c_int64(0)
        band = getattr(store, "sort_band", None)
        self.lib.noahmp_hip_sort_set_band(store.a[band].data_ptr() if band else None)      # the key the store was sorted by
        rc = self.lib.noahmp_hip_sort_staleness(C.byref(a), store.sort_flags, store.sort_keys.data_ptr(), C.byref(changed), None)
        if rc:
            raise RuntimeError("noahmp_hip_sort_staleness: rc=%d %s" % (rc, self.lib.noahmp_hip_last_error().decode()))
        return int(changed.value)

    def sort_staleness_async(self, store, stream=None):
        """sort_staleness without the wait: enqueued on `stream`, read later with sort_staleness_result (a check that drains the
        stream idles the GPU while the host refills its queue)."""
        a = store.step_args(1, 2000, 1.0)
        band = getattr(store, "sort_band", None)
        self.lib.noahmp_hip_sort_set_band(store.a[band].data_ptr() if band else None)
        rc = self.lib.noahmp_hip_sort_staleness_async(C.byref(a), store.sort_flags, store.sort_keys.data_ptr(), stream)
        if rc:
            raise RuntimeError("noahmp_hip_sort_staleness_async: rc=%d %s" % (rc, self.lib.noahmp_hip_last_error().decode()))

    def sort_staleness_result(self, wait=True):
        """The count sort_staleness_async asked for; None if it has not arrived and wait is False."""
        changed = C.c_int64(0)
        rc = self.lib.noahmp_hip_sort_staleness_result(C.byref(changed), 1 if wait else 0)
        if rc == 1:
            return None
        if rc:
            raise RuntimeError("noahmp_hip_sort_staleness_result: rc=%d %s" % (rc, self.lib.noahmp_hip_last_error().decode()))
        return int(changed.value)

    def gather(self, dst, src, perm, ni, nj):
        return Engine.Gather(self.lib, dst, src, perm, ni, nj)

    class Scatter(Gather):
        """The same permutation through noahmp_hip_scatter_fields (chunked, LDS-staged, sorted write order): the form
        to use for records that arrive every step."""

        def __init__(self, lib, dst, src, perm, ni, nj):
            Engine.Gather.__init__(self, lib, dst, src, perm, ni, nj)
            self.set_perm(perm)

        def set_perm(self, perm):
            """(Re)build the plan for `perm` on the device (noahmp_hip_scatter_plan); waits for it."""
            import torch
            n = perm.numel()
            self.perm = perm
            self.order = torch.empty(n, dtype=torch.int16, device=perm.device)
            self.dpos = torch.empty(n, dtype=torch.int32, device=perm.device)
            rc = self.lib.noahmp_hip_scatter_plan(perm.data_ptr(), self.ni, self.nj, self.order.data_ptr(), self.dpos.data_ptr(), None)
            if rc == 0:
                rc = self.lib.noahmp_hip_stream_sync(None)
            if rc:
                raise RuntimeError("noahmp_hip_scatter_plan: rc=%d" % rc)

        def __call__(self, stream=None):
            rc = self.lib.noahmp_hip_scatter_fields(self.n, self.dst, self.src, self.nlev, self.order.data_ptr(),
                                                    self.dpos.data_ptr(), self.ni, self.nj, stream)
            if rc:
                raise RuntimeError("noahmp_hip_scatter_fields: rc=%d" % rc)

        def exchange(self, sorted_planes, tile_planes, to_tile, ni_mem=None, i_off=0, j_off=0, stream=None, first_level_only=()):
            """Move planes between the sorted store and TILE-order planes (possibly the interior of a memory block that carries
            the LATERALFLOW ring: rows ni_mem long, tile origin at (i_off, j_off)) with this permutation's plan
            (noahmp_hip_sorted_exchange).  to_tile: sorted -> tile order, else tile order -> sorted.  first_level_only: as in scatter()."""
            n = len(sorted_planes)
            sp = (C.c_void_p * n)(*[t.data_ptr() for t in sorted_planes])
            tp = (C.c_void_p * n)(*[t.data_ptr() for t in tile_planes])
            nl = (C.c_int * n)(*[(t.shape[1] if t.dim() == 3 else 1) for t in sorted_planes])
            for i in first_level_only:
                if nl[i] > 1:
                    nl[i] = -nl[i]
            rc = self.lib.noahmp_hip_sorted_exchange(n, sp, tp, nl, self.order.data_ptr(), self.dpos.data_ptr(), self.ni, self.nj,
                                                     ni_mem or self.ni, i_off, j_off, 1 if to_tile else 0, stream)
            if rc:
                raise RuntimeError("noahmp_hip_sorted_exchange: rc=%d" % rc)

    def scatter(self, dst, src, perm, ni, nj, first_level_only=()):
        """first_level_only: indices of level arrays of which only the first level is moved -- T3D / QV3D / U_PHY / V_PHY / DZ8W only
        (the column kernel reads their level 1, drv:451-459; the driver's level-2 copies, hdrv:336-344, need not travel).  Never
        P8W3D: noahmplsm reads its levels 1 AND 2 (SFCPRS = (P8W3D(kts+1) + P8W3D(kts)) * 0.5, drv:463), so a P8W3D passed this way
        would keep a stale level-2 pressure."""
        sc = Engine.Scatter(self.lib, dst, src, perm, ni, nj)
        for i in first_level_only:
            if sc.nlev[i] > 1:
                sc.nlev[i] = -sc.nlev[i]
        return sc

    def groundwater_init(self, store, stream=None):
        """GROUNDWATER_INIT + EQSMOISTURE (reference drv:1286-1522): equilibrium soil moisture, deep-layer moisture
        and water-table adjustment for OPT_RUN=5, in place.  ide+1 / jde+1 as NOAHMP_INIT receives them (hdrv:291)."""
        w = store.wtable_args()
        w.ide += 1
        w.jde += 1
        mem = abi.MEM_DEVICE if isinstance(store, DeviceColumnStore) else abi.MEM_HOST
        st = abi.Status()
        rc = self.lib.noahmp_hip_groundwater_init(C.byref(w), store.cfg.iswater, mem, stream, C.byref(st))
        self.last_status = st
        if rc:
            raise RuntimeError("noahmp_hip_groundwater_init: rc=%d %s" % (rc, self.lib.noahmp_hip_last_error().decode()))
        return st

    # ---- asynchronous stepping (device-resident state only)
    def noahmplsm_async(self, args, stream=None):
        """Enqueue one step described by a prepared StepArgs block (store.step_args(...), device pointers)."""
        self._apply_ranges(getattr(args, "_ranges", None))
        rc = self.lib.noahmp_hip_step_async(C.byref(args), stream)
        if rc:
            raise RuntimeError("noahmp_hip_step_async: rc=%d %s" % (rc, self.lib.noahmp_hip_last_error().decode()))

    def sync(self, check=True):
        """Wait for the pending asynchronous steps; returns (Status, ordinal of the failing step or -1)."""
        st = abi.Status()
        step = C.c_int(-1)
        rc = self.lib.noahmp_hip_sync(C.byref(st), C.byref(step))
        self.last_status = st
        if rc < 0:
            raise RuntimeError("noahmp_hip_sync: " + self.lib.noahmp_hip_last_error().decode())
        if rc > 0 and check:
            raise NoahMPFatal(rc, st.i, st.j, "step +%d: %s" % (step.value, self.lib.noahmp_hip_error_string(rc).decode()))
        return st, step.value

    def sync_timing(self):
        """(land-or-mixed, land-ice, skipped) kernel ms summed over the steps of the last sync(), and their number."""
        out = (C.c_float * 3)()
        n = self.lib.noahmp_hip_sync_timing(out, 3)
        return [float(x) for x in out], n

    def sync_counts(self):
        """(land, land-ice, skipped) columns summed over the steps of the last sync(), as 64-bit integers (Status carries int32)."""
        out = (C.c_int64 * 3)()
        self.lib.noahmp_hip_sync_counts(out, 3)
        return int(out[0]), int(out[1]), int(out[2])

    def sync_step_timing(self):
        """Land (or mixed) kernel ms of every step of the last sync(), in step order."""
        n = self.lib.noahmp_hip_sync_step_timing(None, 0)
        out = (C.c_float * max(n, 1))()
        self.lib.noahmp_hip_sync_step_timing(out, n)
        return [float(out[i]) for i in range(n)]

    def jit_cache_info(self):
        """(cache directory, option sets compiled by this process, loaded from the disk cache, fallen back to the generic kernel)."""
        c = (C.c_int32 * 3)()
        d = self.lib.noahmp_hip_jit_cache_info(c)
        return (d.decode() if d else ""), int(c[0]), int(c[1]), int(c[2])

    def finalize(self):
        self.lib.noahmp_hip_finalize()

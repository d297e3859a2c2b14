"""TEST INFRASTRUCTURE (oracle) -- ctypes loader for oracle/_ref/libnoahmp_ref*.so.

That library is the UNMODIFIED reference Fortran (compiled by oracle/Makefile from
/root/reference) behind the same ``noahmp_step_args`` block as the HIP engine.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os

from noahmp_amd.abi import StepArgs, Tables, WtableArgs, tables_to_dict

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")
# table text files: read from the reference tree in the dev container; on the GPU box the
# reference library is used with tables injected from the committed fixture instead.
REF_RUN_DIR = "/root/reference/run"


def available(opt="O0"):
    return os.path.exists(_path(opt))


def _path(opt):
    return os.path.join(REF_DIR, "libnoahmp_ref.so" if opt == "O0" else "libnoahmp_ref_%s.so" % opt)


class RefLib:
    def __init__(self, opt="O0"):
        p = _path(opt)
        if not os.path.exists(p):
            raise RuntimeError("%s missing: run `make -C oracle ref` in the dev container" % p)
        self.lib = C.CDLL(p)
        self.lib.ref_read_tables.argtypes = [C.c_int]
        self.lib.ref_noahmp_init.argtypes = [C.POINTER(StepArgs), C.c_int, C.c_int]
        self.lib.ref_noahmplsm.argtypes = [C.POINTER(StepArgs)]
        self.lib.ref_get_tables.argtypes = [C.POINTER(Tables)]
        self.lib.ref_set_tables.argtypes = [C.POINTER(Tables)]
        self.lib.ref_wtable_mmf.argtypes = [C.POINTER(WtableArgs)]
        self.lib.ref_noahmp_init_mmf.argtypes = [C.POINTER(StepArgs), C.POINTER(WtableArgs), C.c_int, C.c_int,
                                                 C.c_float, C.c_float, C.c_float]
        self.tables_loaded = False

    def read_tables(self, run_dir=REF_RUN_DIR, modis=False):
        """read_mp_veg_parameters + SOIL_VEG_GEN_PARM on the .TBL files in run_dir."""
        cwd = os.getcwd()
        os.chdir(run_dir)
        try:
            self.lib.ref_read_tables(1 if modis else 0)
        finally:
            os.chdir(cwd)
        self.tables_loaded = True

    def set_tables(self, tables):
        """Inject a table image into the reference's module arrays (used on the GPU box,
        where the .TBL text files do not exist; values come from the committed fixture that
        get_tables() produced in the dev container)."""
        self.lib.ref_set_tables(C.byref(tables))
        self.tables_loaded = True

    def get_tables(self, isurban=1):
        t = Tables()
        self.lib.ref_get_tables(C.byref(t))
        t.isurban = isurban
        return t

    def get_tables_dict(self, isurban=1):
        return tables_to_dict(self.get_tables(isurban))

    def noahmp_init(self, store, fndsnowh=True, run_dir=REF_RUN_DIR):
        """NOAHMP_INIT re-reads the .TBL files itself (drv:979-987), so it needs run_dir."""
        a = store.step_args(1, 2000, 1.0)
        cwd = os.getcwd()
        os.chdir(run_dir)
        try:
            self.lib.ref_noahmp_init(C.byref(a), store.cfg.iswater, 1 if fndsnowh else 0)
        finally:
            os.chdir(cwd)
        self.tables_loaded = True

    def noahmplsm(self, store, itimestep, yr, julian):
        assert self.tables_loaded
        a = store.step_args(itimestep, yr, julian)
        self.lib.ref_noahmplsm(C.byref(a))

    def wtable_mmf(self, store):
        """WTABLE_mmf_noahmp of the compiled reference (gw:14) on the store's arrays."""
        assert self.tables_loaded
        w = store.wtable_args()
        self.lib.ref_wtable_mmf(C.byref(w))

    def noahmp_init_mmf(self, store, fndsnowh=True, run_dir=REF_RUN_DIR):
        """NOAHMP_INIT with OPT_RUN=5 and the optional MMF arguments: standard cold start + GROUNDWATER_INIT.
        AREAXY is overwritten with DX*DY (drv:1117)."""
        a = store.step_args(1, 2000, 1.0)
        w = store.wtable_args()
        cwd = os.getcwd()
        os.chdir(run_dir)
        try:
            self.lib.ref_noahmp_init_mmf(C.byref(a), C.byref(w), store.cfg.iswater, 1 if fndsnowh else 0,
                                         store.cfg.dx, store.cfg.dx, store.cfg.dt)
        finally:
            os.chdir(cwd)
        self.tables_loaded = True

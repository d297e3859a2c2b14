"""TEST INFRASTRUCTURE (oracle) -- ctypes loader for the C restatement (oracle/_build).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

from noahmp_amd.abi import StepArgs, Tables, Status, WtableArgs, ForcingRecord, FORCING_RECORD_FIELDS

_HERE = os.path.dirname(os.path.abspath(__file__))
PORT_PATH = os.path.join(_HERE, "_build", "libnoahmp_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "port"])


class PortLib:
    def __init__(self, autobuild=True):
        if autobuild:
            build()
        self.lib = C.CDLL(PORT_PATH)
        self.lib.nmp_oracle_set_tables.argtypes = [C.POINTER(Tables)]
        self.lib.nmp_oracle_step.argtypes = [C.POINTER(StepArgs), C.POINTER(Status)]
        self.lib.nmp_oracle_init.argtypes = [C.POINTER(StepArgs), C.c_int, C.c_int, C.POINTER(Status)]
        self.lib.nmp_oracle_groundwater_init.argtypes = [C.POINTER(WtableArgs), C.c_int, C.POINTER(Status)]
        self.lib.nmp_oracle_forcing_prep.argtypes = [C.POINTER(StepArgs), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                     C.c_int, C.c_float, C.c_int, C.POINTER(C.c_float)]
        self.lib.nmp_oracle_forcing_interpolate.argtypes = [C.POINTER(StepArgs), C.POINTER(ForcingRecord),
                                                            C.POINTER(ForcingRecord), C.c_int, C.c_int, C.c_void_p]
        self.lib.nmp_oracle_wtable_mmf.argtypes = [C.POINTER(WtableArgs), C.POINTER(Status)]

    def set_tables(self, tables):
        self.lib.nmp_oracle_set_tables(C.byref(tables))

    def noahmplsm(self, store, itimestep, yr, julian):
        a = store.step_args(itimestep, yr, julian)
        st = Status()
        self.lib.nmp_oracle_step(C.byref(a), C.byref(st))
        return st

    def wtable_mmf(self, store):
        w = store.wtable_args()
        st = Status()
        rc = self.lib.nmp_oracle_wtable_mmf(C.byref(w), C.byref(st))
        assert rc == 0, rc
        return st

    def noahmp_init(self, store, fndsnowh=True):
        """NOAHMP_INIT + SNOW_INIT (drv:847-1283); ide+1 / jde+1 as hdrv:291 passes them."""
        a = store.step_args(1, 2000, 1.0)
        a.ide += 1
        a.jde += 1
        st = Status()
        rc = self.lib.nmp_oracle_init(C.byref(a), store.cfg.iswater, 1 if fndsnowh else 0, C.byref(st))
        return rc, st

    def groundwater_init(self, store):
        """GROUNDWATER_INIT (drv:1286-1522); ide+1 / jde+1 as NOAHMP_INIT receives them (hdrv:291)."""
        w = store.wtable_args()
        w.ide += 1
        w.jde += 1
        st = Status()
        rc = self.lib.nmp_oracle_groundwater_init(C.byref(w), store.cfg.iswater, C.byref(st))
        assert rc == 0, rc
        return st

    def forcing_prep(self, store, lon, rain_rate, iday, ihour, iminute=0, isecond=0, scale_vegfra=False, first_step=False):
        a = store.step_args(1, 2000, 1.0)
        jul = C.c_float(0)
        self.lib.nmp_oracle_forcing_prep(C.byref(a), lon.ctypes.data, rain_rate.ctypes.data, iday, ihour, iminute, isecond,
                                         store.cfg.zlvl, (1 if scale_vegfra else 0) | (2 if first_step else 0), C.byref(jul))
        return jul.value

    def forcing_interpolate(self, store, rec_a, rec_b, idts, idts2, rain_rate):
        a = store.step_args(1, 2000, 1.0)
        ra, rb = _record(rec_a), (_record(rec_b) if rec_b is not None else None)
        return self.lib.nmp_oracle_forcing_interpolate(C.byref(a), C.byref(ra), C.byref(rb) if rb is not None else None,
                                                       idts, idts2, rain_rate.ctypes.data)


def _record(d):
    r = ForcingRecord()
    for n in FORCING_RECORD_FIELDS:
        if d.get(n) is not None:
            setattr(r, n, d[n].ctypes.data)
    return r

"""TEST INFRASTRUCTURE: loader for the host-compiled emulation of the device source."""
import ctypes as C
import os
import subprocess

from noahmp_amd.abi import StepArgs, Tables, Status, WtableArgs, ForcingRecord, FORCING_RECORD_FIELDS

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(_HERE))
LIB = os.path.join(_HERE, "libnmp_emul.so")


def build(nsoil=4):
    """nsoil != 4: the device source compiled with -DNOAHMP_NSOIL=n (a library of its own beside the default one)."""
    lib = LIB if nsoil == 4 else LIB.replace(".so", "_nsoil%d.so" % nsoil)
    src = os.path.join(_HERE, "emul.hip")
    csrc = os.path.join(ROOT, "noahmp_amd", "csrc")
    deps = [src, os.path.join(ROOT, "include", "noahmp_hip.h")] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp")]
    if os.path.exists(lib) and all(os.path.getmtime(d) <= os.path.getmtime(lib) for d in deps):
        return lib
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-fPIC", "-shared", "-std=c++17",
                           "-ffp-contract=off", "-mfma", "-Wno-unused-value", "-DNOAHMP_NSOIL=%d" % nsoil,
                           "-I" + os.path.join(ROOT, "include"), "-I" + csrc, src, "-o", lib])
    return lib


class EmulLib:
    def __init__(self, nsoil=4):
        self.lib = C.CDLL(build(nsoil))
        self.lib.emul_set_tables.argtypes = [C.POINTER(Tables)]
        self.lib.emul_step.argtypes = [C.POINTER(StepArgs), C.POINTER(Status)]
        self.lib.emul_init.argtypes = [C.POINTER(StepArgs), C.c_int, C.c_int, C.POINTER(Status)]
        self.lib.emul_groundwater_init.argtypes = [C.POINTER(WtableArgs), C.c_int, C.POINTER(Status)]
        self.lib.emul_forcing_interpolate.argtypes = [C.POINTER(StepArgs), C.POINTER(ForcingRecord), C.POINTER(ForcingRecord),
                                                      C.c_int, C.c_int, C.c_void_p]
        self.lib.emul_forcing_prep.argtypes = [C.POINTER(StepArgs), C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                               C.c_float, C.c_int]
        self.lib.emul_wtable_mmf.argtypes = [C.POINTER(WtableArgs), C.POINTER(Status)]

    def set_tables(self, t):
        self.lib.emul_set_tables(C.byref(t))

    def noahmplsm(self, store, itimestep, yr, julian):
        a = store.step_args(itimestep, yr, julian)
        st = Status()
        self.lib.emul_step(C.byref(a), C.byref(st))
        return st

    def wtable_mmf(self, store):
        w = store.wtable_args()
        st = Status()
        rc = self.lib.emul_wtable_mmf(C.byref(w), C.byref(st))
        assert rc == 0, rc
        return st

    def noahmp_init(self, store, fndsnowh=True):
        a = store.step_args(1, 2000, 1.0)
        a.ide += 1
        a.jde += 1
        st = Status()
        rc = self.lib.emul_init(C.byref(a), store.cfg.iswater, 1 if fndsnowh else 0, C.byref(st))
        return rc, st

    def groundwater_init(self, store):
        w = store.wtable_args()
        w.ide += 1
        w.jde += 1
        st = Status()
        rc = self.lib.emul_groundwater_init(C.byref(w), store.cfg.iswater, C.byref(st))
        assert rc == 0, rc
        return st

    def forcing_interpolate(self, store, rec_a, rec_b, idts, idts2, rain_rate):
        a = store.step_args(1, 2000, 1.0)

        def rec(d):
            r = ForcingRecord()
            for n in FORCING_RECORD_FIELDS:
                if d.get(n) is not None:
                    setattr(r, n, d[n].ctypes.data)
            return r
        ra, rb = rec(rec_a), (rec(rec_b) if rec_b is not None else None)
        return self.lib.emul_forcing_interpolate(C.byref(a), C.byref(ra), C.byref(rb) if rb is not None else None,
                                                 idts, idts2, rain_rate.ctypes.data)

    def forcing_prep(self, store, lon, rain_rate, hour_utc, sin_declin, cos_declin, scale_vegfra=False, first_step=False):
        a = store.step_args(1, 2000, 1.0)
        return self.lib.emul_forcing_prep(C.byref(a), lon.ctypes.data, rain_rate.ctypes.data, hour_utc, sin_declin,
                                          cos_declin, store.cfg.zlvl, (1 if scale_vegfra else 0) | (2 if first_step else 0))

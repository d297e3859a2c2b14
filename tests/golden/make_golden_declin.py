"""Write tests/golden/golden_declin.npz by RUNNING THE REFERENCE'S CALC_DECLIN (driver/module_hrldas_noahmp_driver.F90:813-863; compiled
unmodified by `make -C oracle declin` into oracle/_ref/libnoahmp_declin_ref.so): COSZ and JULIAN over a latitude / longitude grid at
dates and times of day across a leap and a common year (both branches of the JULIAN >= 80 test, the wrap of the local time).
Dev container only.     python tests/golden/make_golden_declin.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "oracle", "_ref", "libnoahmp_declin_ref.so")
# (year, month, day, hour, minute, second)
WHEN = [(2000, 1, 1, 0, 0, 0), (2000, 2, 14, 13, 30, 0), (2000, 3, 20, 23, 59, 59), (2000, 3, 21, 0, 0, 1), (2000, 6, 20, 12, 0, 0),
        (2001, 3, 21, 6, 15, 30), (2001, 3, 22, 6, 15, 30), (2001, 7, 19, 6, 15, 30), (2001, 12, 31, 18, 45, 12), (2004, 2, 29, 9, 0, 0),
        (2100, 12, 31, 23, 0, 0)]


def grid(seed=17, n=4096):
    r = np.random.Generator(np.random.Philox(seed))
    lat = np.concatenate([r.uniform(-90.0, 90.0, n - 6), [-90.0, 90.0, 0.0, 23.5, -23.5, 66.5]]).astype(np.float32)
    lon = np.concatenate([r.uniform(-180.0, 180.0, n - 6), [-180.0, 180.0, 0.0, 179.99, -0.01, 15.0]]).astype(np.float32)
    return lat, lon


def ref_lib():
    lib = C.CDLL(LIB)
    lib.ref_calc_declin.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    return lib


def run_reference(lib, when, lat, lon):
    cosz = np.zeros_like(lat)
    jul = C.c_float(0)
    lib.ref_calc_declin(*when, lat.size, lat.ctypes.data, lon.ctypes.data, cosz.ctypes.data, C.byref(jul))
    return cosz, np.float32(jul.value)


def main():
    lib = ref_lib()
    lat, lon = grid()
    out = {"lat": lat, "lon": lon, "when": np.array(WHEN, dtype=np.int32)}
    for i, w in enumerate(WHEN):
        out["cosz%02d" % i], out["julian%02d" % i] = run_reference(lib, w, lat, lon)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_declin.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

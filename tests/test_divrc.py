"""The exact-division helpers of the device source (noahmp_amd/csrc/nmp_dev_common.hpp): RN32(x / y) == (float)((double)x * r)
for r within 2^-50 of 1/y.  CPU: every float32 numerator (stride NMP_DIV_STRIDE, default: exhaustive for DT = 3600 and HVAP, every
61st bit pattern for the other constant divisors of the physics headers) against the IEEE division, 0 mismatches outside the
documented exception (exact ties below the normal range, |x/c| < 2^-126).  GPU: rc64 over all 2^32 divisors -- relative error
<= 2^-52, zeros / infinities / NaNs as 1/y -- and three numerators per divisor through div_rc against the device's own division."""
import ctypes as C
import os
import re
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "noahmp_amd", "csrc")
SRC = os.path.join(HERE, "host_emul", "div_check.hip")
LIB = os.path.join(HERE, "host_emul", "libdiv_check.so")
# the divisors behind NMP_RCC(...) in the physics headers, and the uniform ones of Urc at the namelist values
CONSTANTS = {"DT=3600": 3600.0, "HVAP": 2.5104E06, "HSUB": 2.8440E06, "HFUS": 0.3336E06, "DENICE": 917.0, "DENH2O": 1000.0, "2.59": 2.59,
             "1.87E5": 1.87E5, "1.56E5": 1.56E5, "ETA0": 0.8e6, "3": 3.0, "6": 6.0, "ROUS": 0.2, "100": 100.0, "10": 10.0,
             "E-1": 2.71828 - 1.0, "0.622*HSUB": None, "0.622*HVAP": None, "DT=600": 600.0, "DT=900": 900.0, "DT*HFUS": None,
             "DZ(1)": 0.1, "DZ(2)*1000": None, "ZSOIL(1)-ZSOIL(3)": None}


def build():
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("nmp_dev_common.hpp", "nmp_libm.hpp")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-std=c++17",
                               "-ffp-contract=off", "-mfma", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), SRC, "-o", LIB, "-lpthread"])


def _lib():
    build()
    try:
        import torch  # noqa: F401  (map torch's HIP runtime first, noahmp_amd/abi.py::load_library)
    except ImportError:
        pass
    lib = C.CDLL(LIB)
    lib.div_check_host.argtypes = [C.c_float, C.c_uint32, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_uint32)]
    lib.div_check_gpu.restype = C.c_long
    lib.div_check_gpu.argtypes = [C.POINTER(C.c_double)]
    return lib


def _value(name):
    import numpy as np
    F = np.float32
    v = CONSTANTS[name]
    if v is not None:
        return float(F(v))
    return float({"0.622*HSUB": F(0.622) * F(2.8440E06), "0.622*HVAP": F(0.622) * F(2.5104E06), "DT*HFUS": F(3600.0) * F(0.3336E06),
                  "DZ(2)*1000": F(0.3) * F(1000.0), "ZSOIL(1)-ZSOIL(3)": F(-0.1) - (F(-0.6) + (F(-0.3) + F(-0.1)))}[name])


def test_every_constant_divisor_of_the_headers_is_listed():
    """NMP_RCC(...) arguments in the physics headers: each one has an entry above (so that a new constant gets its sweep)."""
    known = {"2.59f", "HVAP", "HSUB", "1.87E5f", "1.56E5f", "DENICE", "DENH2O", "ETA0", "3.f", "6.f", "1000.f", "ROUS", "HFUS", "100.f", "10.0f",
             "2.71828f - 1.0f", "0.622f * HSUB", "0.622f * HVAP"}
    seen = set()
    for f in os.listdir(CSRC):
        if f.startswith("nmp_dev_") and f.endswith(".hpp") and f != "nmp_dev_common.hpp":
            seen |= set(re.findall(r"NMP_RCC\(([^()]*)\)", open(os.path.join(CSRC, f)).read()))
    assert seen and seen <= known, seen - known


@pytest.mark.parametrize("name", list(CONSTANTS))
def test_div_rc_equals_ieee_division_for_every_numerator(name):
    lib = _lib()
    c = _value(name)
    stride = int(os.environ.get("NMP_DIV_STRIDE", "1" if name in ("DT=3600", "HVAP") else "61"))
    out = (C.c_long * 2)()
    fb = C.c_uint32(0)
    lib.div_check_host(c, stride, 8, out, C.byref(fb))
    assert out[0] == 0, "x / %r: %d mismatches in the normal range, first numerator bits 0x%08x" % (c, out[0], fb.value)
    # out[1]: exact ties below 2^-126 rounded the other way (|x| < |c| 2^-126): the documented exception, reported only


@pytest.mark.gpu
def test_rc64_on_the_device_over_all_divisors():
    lib = _lib()
    e = C.c_double(0)
    bad = lib.div_check_gpu(C.byref(e))
    assert bad == 0, "%d special-value or quotient mismatches" % bad
    assert 0 < e.value <= 2.0 ** -52, e.value

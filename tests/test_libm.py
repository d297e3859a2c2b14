"""noahmp_amd/csrc/nmp_libm.hpp against the live libm of the machine, bit for bit.

The reference's EXP / LOG / ** / LOG10 / ATAN / TANH / TAN / ACOS / COS resolve to glibc's float32 routines; the device runs
restatements of the same algorithms.  CPU: the host compilation of that header over a stride of the whole
2^32 argument space (the exhaustive run -- NMP_LIBM_STRIDE=1, ~1 min on 8 cores -- gives 0 mismatches for every routine on
a host whose libm runs its FMA build).  GPU: the device code itself over 2^24 arguments per routine.

expf is the one routine whose bits depend on the host: glibc picks __expf_fma or __expf_sse2 at load time and the two
differ at exactly two arguments (oracle/nmp_pin_expf.c, tools/expf_variants.c).  The pin is __expf_fma; the oracle and
the device evaluate it on any host, and the tests below know which build the live libm is.
"""
import ctypes as C
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "host_emul", "libm_check.hip")
LIB = os.path.join(HERE, "host_emul", "liblibm_check.so")
NAMES = ["expf", "logf", "log10f", "atanf", "tanhf", "expm1f", "powf", "acosf", "tanf", "cosf", "sinf", "atanf_ge1", "logfN", "expfN", "powfN"]
UNARY = [0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12, 13]      # 6 = powf (binary)


def build():
    """Compile the checker library (no dlopen: it links the system HIP runtime, which must not be mapped before
    PyTorch's own copy in a process that later uses the GPU through torch)."""
    csrc = os.path.join(ROOT, "noahmp_amd", "csrc")
    deps = [SRC, os.path.join(csrc, "nmp_libm.hpp"), os.path.join(csrc, "nmp_libm_tables.inc")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-std=c++17",
                               "-ffp-contract=off", "-mfma", "-I" + csrc, SRC, "-o", LIB, "-lpthread"])


def _lib():
    build()
    try:
        import torch  # noqa: F401  (see noahmp_amd/abi.py::load_library: map torch's HIP runtime first)
    except ImportError:
        pass
    lib = C.CDLL(LIB)
    lib.libm_check_unary.restype = C.c_long
    lib.libm_check_unary.argtypes = [C.c_int, C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]
    lib.libm_check_pow.restype = C.c_long
    lib.libm_check_pow.argtypes = [C.c_int, C.c_long, C.c_int, C.POINTER(C.c_uint32)]
    lib.libm_gpu_check.restype = C.c_long
    lib.libm_gpu_check.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_long, C.POINTER(C.c_uint32)]
    return lib


@pytest.mark.parametrize("fn", UNARY, ids=[NAMES[i] for i in UNARY])
def test_host_build_matches_libm(fn):
    lib = _lib()
    fb = C.c_uint32(0)
    stride = int(os.environ.get("NMP_LIBM_STRIDE", "61"))          # 7e7 arguments per routine by default
    n = lib.libm_check_unary(fn, stride, 8, C.byref(fb))
    allowed = 0 if (fn not in (0, 13) or _host_expf_is_pinned()) else 2      # an SSE2-build host differs at the two known arguments
    assert n <= allowed, "%s: %d mismatches, first at bits 0x%08x" % (NAMES[fn], n, fb.value)


EXPF_DISCRIMINATING = ((0x4202422f, 0x56fc9f1c, 0x56fc9f1b), (0xc27c65d9, 0x11fa2993, 0x11fa2992))   # x, __expf_fma, __expf_sse2


def _oracle():
    from oracle.portlib import PortLib
    lib = PortLib(autobuild=True).lib
    lib.nmp_pin_expf.restype = C.c_float
    lib.nmp_pin_expf.argtypes = [C.c_float]
    return lib


def _host_expf_is_pinned():
    return bool(_oracle().nmp_pin_expf_host_variant_is_pinned())


def _f(bits):
    import struct
    return struct.unpack("<f", struct.pack("<I", bits))[0]


def _bits(x):
    import struct
    return struct.unpack("<I", struct.pack("<f", x))[0]


def test_oracle_expf_is_the_pinned_build():
    """The checker's EXP (oracle/nmp_pin_expf.c) returns __expf_fma's bits at the two arguments where glibc's builds differ,
    whatever the host, and equals the live libm elsewhere (numpy float32 exp goes to its own SIMD code: ctypes on libm)."""
    lib = _oracle()
    for x, fma, sse2 in EXPF_DISCRIMINATING:
        assert _bits(lib.nmp_pin_expf(_f(x))) == fma
    libm = C.CDLL("libm.so.6")
    libm.expf.restype = C.c_float
    libm.expf.argtypes = [C.c_float]
    live = {x: _bits(libm.expf(_f(x))) for x, _, _ in EXPF_DISCRIMINATING}
    assert all(live[x] in (fma, sse2) for x, fma, sse2 in EXPF_DISCRIMINATING), live
    assert _host_expf_is_pinned() == all(live[x] == fma for x, fma, _ in EXPF_DISCRIMINATING)
    import numpy as np
    r = np.random.Generator(np.random.Philox(11))
    xs = np.concatenate([r.uniform(-104, 89, 200000), r.normal(0, 3, 100000), [0.0, -0.0, 88.72, -103.9, 1e-30, -1e-30]]).astype(np.float32)
    skip = {x for x, _, _ in EXPF_DISCRIMINATING}
    for x in xs.tolist():
        if _bits(x) in skip:
            continue
        assert _bits(lib.nmp_pin_expf(x)) == _bits(libm.expf(x)), x
    for x in (float("inf"), float("-inf")):
        assert _bits(lib.nmp_pin_expf(x)) == _bits(libm.expf(x))
    assert lib.nmp_pin_expf(float("nan")) != lib.nmp_pin_expf(float("nan"))


def test_device_source_expf_is_the_pinned_build():
    """The device header's expf_ compiled for the host: __expf_fma's bits at the discriminating arguments."""
    lib = _lib()
    lib.libm_eval_unary.restype = C.c_uint32
    lib.libm_eval_unary.argtypes = [C.c_int, C.c_uint32]
    for x, fma, sse2 in EXPF_DISCRIMINATING:
        assert lib.libm_eval_unary(0, x) == fma


def test_host_powf_matches_libm():
    lib = _lib()
    xy = (C.c_uint32 * 2)()
    # specials x specials, model range, random bits, the reference's constant exponents over every 37th positive float
    for mode, n in ((2, 0), (1, 40_000_000), (0, 40_000_000), (3, 37)):
        bad = lib.libm_check_pow(mode, n, 8, xy)
        assert bad == 0, "powf mode %d: %d mismatches, first x=0x%08x y=0x%08x" % (mode, bad, xy[0], xy[1])


def test_tables_regenerate_identically(tmp_path):
    """tools/gen_libm_tables.py on this image's libm reproduces the committed constants."""
    inc = os.path.join(ROOT, "noahmp_amd", "csrc", "nmp_libm_tables.inc")
    before = open(inc).read()
    subprocess.check_call(["python", os.path.join(ROOT, "tools", "gen_libm_tables.py")], stdout=subprocess.DEVNULL)
    assert open(inc).read() == before


@pytest.mark.gpu
@pytest.mark.parametrize("fn", range(15), ids=NAMES)
def test_device_code_matches_libm(fn):
    lib = _lib()
    fb = C.c_uint32(0)
    n = 1 << 24
    if fn in (6, 14):
        bad = lib.libm_gpu_check(fn, 0x00800000, 127, n, C.byref(fb))     # positive normal bases, y in (-32, 32)
    else:
        bad = lib.libm_gpu_check(fn, 12345, 256, n, C.byref(fb))          # every 256th bit pattern, all of 2^32
    assert bad >= 0, "HIP error"
    allowed = 0 if (fn not in (0, 13) or _host_expf_is_pinned()) else 1
    assert bad <= allowed, "%s on the GPU: %d mismatches, first at bits 0x%08x" % (NAMES[fn], bad, fb.value)


@pytest.mark.gpu
def test_device_expf_is_the_pinned_build():
    """The device code at the two arguments where glibc's expf builds differ: __expf_fma's bits."""
    lib = _lib()
    lib.libm_gpu_eval_unary.restype = C.c_long
    lib.libm_gpu_eval_unary.argtypes = [C.c_int, C.c_uint32]
    for x, fma, sse2 in EXPF_DISCRIMINATING:
        assert lib.libm_gpu_eval_unary(0, x) == fma

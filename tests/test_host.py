"""CPU tests of the host side: ABI layout, exported symbols, cold-start init, partition rule."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from noahmp_amd import abi, synth
from noahmp_amd.abi_spec import STEP_FIELDS, TABLE_FIELDS
from noahmp_amd.state import ColumnStore, ModelConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_is_in_sync_with_spec():
    """include/noahmp_hip.h is generated from abi_spec.py; every field appears in order."""
    txt = open(os.path.join(ROOT, "include", "noahmp_hip.h")).read()
    pos = -1
    body = txt[txt.index("typedef struct noahmp_step_args"):txt.index("} noahmp_step_args;")]
    for n, k, lev, io, ln in STEP_FIELDS:
        p = body.find(" %s;" % n)
        assert p > pos, n
        pos = p
    body = txt[txt.index("typedef struct noahmp_tables"):txt.index("} noahmp_tables;")]
    for n, k, s, src in TABLE_FIELDS:
        assert (" %s;" % n in body) or (" %s[" % n in body), n


def test_ctypes_layout_matches_compiled_header(tmp_path):
    """sizeof/offsetof of the ctypes mirrors == what gcc makes of the C header."""
    src = tmp_path / "sz.c"
    probes = ["coszin", "dzs", "isnowxy", "chb2xy", "kte"]
    tprobes = ["saim", "nrotbl", "albsat", "eg"]
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "noahmp_hip.h"\nint main(){'
                   'printf("%zu %zu", sizeof(noahmp_step_args), sizeof(noahmp_tables));'
                   + "".join('printf(" %%zu", offsetof(noahmp_step_args,%s));' % p for p in probes)
                   + "".join('printf(" %%zu", offsetof(noahmp_tables,%s));' % p for p in tprobes)
                   + 'return 0;}')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out[0] == C.sizeof(abi.StepArgs) and out[1] == C.sizeof(abi.Tables)
    for p, o in zip(probes, out[2:]):
        assert getattr(abi.StepArgs, p).offset == o, p
    for p, o in zip(tprobes, out[2 + len(probes):]):
        assert getattr(abi.Tables, p).offset == o, p


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads and exports every entry point of include/noahmp_hip.h (no compute)."""
    if not os.path.exists(abi.LIB_PATH):
        from noahmp_amd import build
        build.build()
    hdr = open(os.path.join(ROOT, "include", "noahmp_hip.h")).read()
    import re
    declared = set(re.findall(r"\b(noahmp_hip_\w+)\s*\(", hdr))
    assert declared == set(abi.EXPORTED_SYMBOLS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", abi.LIB_PATH]).decode()
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert declared <= exported, declared - exported
    lib = abi.load_library()
    assert lib.noahmp_hip_abi_version() == 1
    assert lib.noahmp_hip_sizeof_step_args() == C.sizeof(abi.StepArgs)
    assert lib.noahmp_hip_error_string(7).decode().startswith("Water budget")


def test_index_width_gate():
    """The option-specialised kernels address with 32-bit byte offsets: a call whose widest array (max(7, atmospheric levels) levels)
    reaches 4 GiB is served by the generic kernels (host arithmetic only; no device call)."""
    lib = abi.load_library()
    w = lib.noahmp_hip_index_width
    assert w(4608, 1536, 2) == 32 and w(3600, 1800, 2) == 32
    assert w(153391689, 1, 2) == 32 and w(153391690, 1, 2) == 64          # 28 B per cell over 7 levels: 2^32 / 28
    assert w(16384, 16384, 2) == 64
    assert w(4608, 1536, 50) == 32 and w(21474836, 1, 50) == 32 and w(21474837, 1, 50) == 64   # 200 B per cell


def _device_code_objects(path):
    """The gfx950 ELF images inside a HIP fat binary (clang offload bundles, one per translation unit)."""
    import struct
    blob = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = [], blob.find(magic)
    while pos >= 0:
        n, = struct.unpack_from("<Q", blob, pos + 24)
        q = pos + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(magic, pos + 1)
    return out


def test_no_function_is_called_inside_our_kernels(tmp_path):
    """Everything in the column / groundwater / forcing / init kernels is inline: a real call (s_swappc_b64 to a noinline routine)
    inside the column kernel once corrupted a live value of its caller (profiles/r03_experiments.md section 3d).  Checked on the
    ISA of the built library; rocprim's own sort kernels are not ours."""
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not found")
    from noahmp_amd import build as b
    images = _device_code_objects(b.LIB)
    assert len(images) >= 8, len(images)                    # one per translation unit with device code
    kernels_seen = 0
    for n, img in enumerate(images):
        f = tmp_path / ("co%d.elf" % n)
        f.write_bytes(img)
        fn, calls = None, {}
        for line in subprocess.run([objdump, "-d", str(f)], capture_output=True, text=True, check=True).stdout.splitlines():
            if line.endswith(">:"):
                fn = line.split("<", 1)[1][:-2]
            elif "s_swappc_b64" in line or "s_call_b64" in line:
                calls[fn] = calls.get(fn, 0) + 1
            if fn and "noahmp_column_kernel" in fn:
                kernels_seen += 1
        ours = {k: v for k, v in calls.items() if "rocprim" not in k}
        assert not ours, ours
    assert kernels_seen > 0


def test_engine_fails_loudly_without_gpu(tables):
    """No CPU fallback: without a device the engine refuses to run."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from noahmp_amd.driver import Engine
    with pytest.raises(RuntimeError):
        Engine(tables[0])


def test_tables_roundtrip(tables):
    T, d = tables
    d2 = abi.tables_to_dict(T)
    for k, v in d.items():
        np.testing.assert_array_equal(np.asarray(v, dtype=np.float32 if np.asarray(v).dtype.kind == "f" else None),
                                      np.asarray(d2[k]))
    assert d["laim"].shape == (27, 12) and d["nrotbl"].shape == (50,)
    assert T.laim[6][6] == d["laim"][6, 6]          # Fortran LAIM(7,7) == C laim[6][6]


def test_cold_start_matches_reference_init(reflib, tables):
    """noahmp_amd.init (NOAHMP_INIT + SNOW_INIT mirror) vs the reference's own NOAHMP_INIT."""
    import noahmp_amd.init as ini
    captured = []
    orig = ini.noahmp_init

    def spy(store, tb, fndsnowh=True):
        captured.append(store.copy())
        return orig(store, tb, fndsnowh)
    synth.noahmp_init = spy
    try:
        mine = synth.mixed_small(tables[1], ni=64, nj=8, seed=5)
    finally:
        synth.noahmp_init = orig
    theirs = captured[0]
    reflib.noahmp_init(theirs)
    for k in mine.a:
        if k == "sh2o":     # powf differs by <= 1 ulp between numpy and the Fortran runtime
            np.testing.assert_allclose(mine.a[k], theirs.a[k], rtol=3e-7, atol=0)
        else:
            np.testing.assert_array_equal(mine.a[k], theirs.a[k], err_msg=k)
    assert set(np.unique(mine["isnowxy"])) == {0, -1, -2, -3}


def test_synth_is_seeded(tables):
    a = synth.mixed_small(tables[1], ni=16, nj=2, seed=3)
    b = synth.mixed_small(tables[1], ni=16, nj=2, seed=3)
    for k in a.a:
        np.testing.assert_array_equal(a.a[k], b.a[k])


def test_store_layout_is_fortran_image():
    s = ColumnStore(5, 3, ModelConfig())
    assert s["tslb"].shape == (3, 4, 5) and s["zsnsoxy"].shape == (3, 7, 5) and s["tsk"].shape == (3, 5)
    a = s.step_args(1, 2000, 1.0)
    assert a.ime - a.ims + 1 == 5 and a.jme - a.jms + 1 == 3 and a.kme == 2
    # element (i=2,k=3,j=1) (Fortran, 1-based) sits at ((j-1)*nk + (k-1))*ni + (i-1)
    s["tslb"][0, 2, 1] = 42.0
    flat = s["tslb"].ravel()
    assert flat[((1 - 1) * 4 + (3 - 1)) * 5 + (2 - 1)] == 42.0


def test_runtime_specialisation_compiles_without_gpu_and_fills_the_disk_cache(tmp_path):
    """noahmp_jit.hip: the column kernel compiled by hiprtc for an option set that has no ahead-of-time kernel (the compile step
    needs no GPU; loading and launching are covered by the GPU tests).  The code object lands in the on-disk cache, keyed by
    the options and the source hash; a second process finds it there instead of compiling."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        "from noahmp_amd import abi\n"
        "lib = abi.load_library()\n"
        "opts = (C.c_int32 * 12)(2, 2, 2, 3, 1, 2, 2, 1, 1, 3, 1, 2)\n"
        "log = C.create_string_buffer(4096)\n"
        "rc = lib.noahmp_hip_jit_compile_check(opts, log, 4096)\n"
        "c = (C.c_int32 * 3)()\n"
        "d = lib.noahmp_hip_jit_cache_info(c)\n"
        "print('RESULT', rc, log.value.decode()[:200].replace('\\n', ' '), '|', d.decode(), list(c))\n")
    env = dict(os.environ, NOAHMP_HIP_CACHE_DIR=str(tmp_path), PYTHONPATH=ROOT)
    outs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
        assert line, r.stdout + r.stderr
        outs.append(line[0])
    assert outs[0].startswith("RESULT 0 compiled ") and outs[0].endswith("[1, 0, 0]"), outs[0]
    assert outs[1].startswith("RESULT 0 cached ") and outs[1].endswith("[0, 1, 0]"), outs[1]
    files = os.listdir(str(tmp_path))
    assert len(files) == 1 and files[0].startswith("nmp_gfx950_o2_2_2_3_1_2_2_1_1_3_1_2_") and files[0].endswith(".hsaco")
    assert str(tmp_path) in outs[0]


def test_staleness_result_without_a_pending_count_is_an_error():
    """noahmp_hip_sort_staleness_result before any noahmp_hip_sort_staleness_async: -105, no GPU touched."""
    import ctypes as C
    from noahmp_amd import abi
    lib = abi.load_library()
    changed = C.c_int64(7)
    assert lib.noahmp_hip_sort_staleness_result(C.byref(changed), 0) == -105
    assert lib.noahmp_hip_sort_staleness_result(None, 1) == -105
    assert b"no pending count" in lib.noahmp_hip_last_error()


@pytest.mark.parametrize("threads", [1, 4])
def test_copy_pool_of_the_staging_path(tmp_path, threads):
    """noahmp_amd/csrc/nmp_copy_pool.hpp (the copy threads that fill / drain the engine's page-locked bounce buffers, nmp_stage.hpp) on the
    CPU: exact copies at sizes around the threading threshold, stop + restart, and a process that never stops a pool it never destroys still
    exits (round 6's first GPU run hung in a static destructor: pthread_cond_destroy waits for parked workers)."""
    import subprocess
    exe = str(tmp_path / "copy_pool_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "noahmp_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cxx", "copy_pool_check.cpp"), "-o", exe])
    out = subprocess.run([exe], env=dict(os.environ, NMP_COPY_THREADS=str(threads)), capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "copy pool ok" in out.stdout, out.stdout + out.stderr

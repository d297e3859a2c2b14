"""Fill the on-disk cache of run-time specialised column kernels (noahmp_jit.hip) ahead of time: hiprtc compiles without a GPU,
so a build step can pay the 3-10 s per option set once instead of the first run.  `python -m noahmp_amd.jit_warm` warms the
option sets below (the reference's namelist alternatives one at a time, lsm:9352-9388, and the mixes the test-suite runs);
`warm([ModelConfig(...), ...])` any others.  Sets that have an ahead-of-time kernel are skipped."""
import ctypes as C
import os
import subprocess
import sys

from .state import ModelConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORDER = ("idveg", "iopt_crs", "iopt_btr", "iopt_run", "iopt_sfc", "iopt_frz", "iopt_inf", "iopt_rad", "iopt_alb", "iopt_snf",
         "iopt_tbot", "iopt_stc")
AOT = {(1, 1), (3, 1), (3, 5), (4, 1), (4, 3)}          # (DVEG, RUN) with the other ten at the namelist values: noahmp_engine_d*_r*.hip
SINGLE = [dict(idveg=2), dict(idveg=5), dict(iopt_crs=2), dict(iopt_btr=2), dict(iopt_btr=3), dict(iopt_run=2), dict(iopt_run=3),
          dict(iopt_run=4), dict(iopt_sfc=2), dict(iopt_frz=2), dict(iopt_inf=2), dict(iopt_rad=1), dict(iopt_rad=2),
          dict(iopt_alb=1), dict(iopt_snf=2), dict(iopt_snf=3), dict(iopt_tbot=1), dict(iopt_stc=2)]
MIXES = [dict(idveg=2, iopt_run=3, iopt_stc=2, iopt_sfc=2, iopt_frz=2),
         dict(iopt_rad=1, iopt_alb=1, iopt_snf=3, iopt_tbot=1, idveg=5, iopt_crs=2, iopt_btr=2, iopt_inf=2),
         dict(idveg=2, iopt_run=3, iopt_stc=2), dict(iopt_sfc=2, iopt_crs=2, iopt_btr=2, iopt_frz=2, iopt_inf=2),
         dict(iopt_rad=1, iopt_alb=1, iopt_snf=3, iopt_tbot=1, idveg=5), dict(idveg=2, iopt_run=3, iopt_stc=2, iopt_frz=2),
         dict(iopt_btr=2, iopt_crs=2, idveg=5), dict(idveg=2, iopt_run=5), dict(idveg=1, iopt_run=5),
         dict(iopt_frz=2, iopt_inf=2), dict(idveg=5, iopt_crs=1, iopt_btr=3, iopt_run=3, iopt_frz=2, iopt_rad=1, iopt_alb=1)]        # bench.py's options_reference legs: DVEG 2, SFC 2, RUN 3 (above) and this mix


def options12(cfg):
    return tuple(int(getattr(cfg, k)) for k in ORDER)


def has_aot_kernel(o):
    return o[1:3] == (1, 1) and o[4:] == (1, 1, 1, 3, 2, 1, 2, 1) and (o[0], o[3]) in AOT


def compile_one(o):
    from . import abi
    lib = abi.load_library()
    log = C.create_string_buffer(4096)
    rc = lib.noahmp_hip_jit_compile_check((C.c_int32 * 12)(*o), log, 4096)
    return rc, log.value.decode()


def warm(cfgs=None, jobs=None, verbose=False):
    sets = []
    for c in (cfgs if cfgs is not None else [ModelConfig(**kw) for kw in SINGLE + MIXES]):
        o = options12(c)
        if not has_aot_kernel(o) and o not in sets:
            sets.append(o)
    jobs = jobs or max(1, min(6, (os.cpu_count() or 2) - 1))
    running, failed = [], 0
    todo = list(sets)
    while todo or running:
        while todo and len(running) < jobs:                # one process per option set: hiprtc compilations run in parallel
            o = todo.pop(0)
            p = subprocess.Popen([sys.executable, "-m", "noahmp_amd.jit_warm", "--one", ",".join(map(str, o))], cwd=ROOT,
                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            running.append((o, p))
        o, p = running.pop(0)
        out = p.communicate()[0]
        if p.returncode:
            failed += 1
            print("jit_warm: option set %s FAILED\n%s" % (o, out[-600:]))
        elif verbose:
            print("jit_warm:", o, out.strip().splitlines()[-1] if out.strip() else "")
    _drop_stale()
    return len(sets), failed


def _drop_stale():
    """Remove code objects of earlier source versions from the cache directory (their hash no longer matches)."""
    from . import abi
    lib = abi.load_library()
    d = lib.noahmp_hip_jit_cache_info(None)
    h = "%016x" % lib.noahmp_hip_jit_source_hash()
    if not d or h == "0" * 16:
        return
    d = d.decode()
    if os.path.realpath(d) != os.path.realpath(os.path.join(ROOT, "noahmp_amd", "csrc", "jit_cache")):
        return          # a shared directory ($NOAHMP_HIP_CACHE_DIR, ~/.cache) may hold the objects of other library versions: leave it alone
    for f in os.listdir(d):
        if f.startswith("nmp_gfx950_") and f.endswith(".hsaco") and not f.endswith("_" + h + ".hsaco"):
            try:
                os.unlink(os.path.join(d, f))
            except OSError:
                pass


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--one":
        rc, msg = compile_one(tuple(int(x) for x in sys.argv[2].split(",")))
        print(msg[:300])
        sys.exit(rc)
    n, bad = warm(verbose=True)
    print("jit_warm: %d option sets, %d failed" % (n, bad))
    sys.exit(1 if bad else 0)

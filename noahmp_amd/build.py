"""Build the in-tree HIP engine for gfx950 (cross-compiles without a GPU).

Every translation unit is compiled to its own object (in parallel, rebuilt only when it or a header changed) and the
objects are linked into noahmp_amd/csrc/libnoahmp_hip.so."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(CSRC, "libnoahmp_hip.so")
SOURCES = ["noahmp_engine.hip", "noahmp_groundwater.hip", "noahmp_init.hip", "noahmp_forcing.hip", "noahmp_engine_d1_r1.hip",
           "noahmp_engine_d3_r1.hip", "noahmp_engine_d3_r5.hip", "noahmp_engine_d4_r1.hip", "noahmp_engine_d4_r3.hip",
           "noahmp_jit.hip", "noahmp_sort.hip", "noahmp_halo.hip", "noahmp_stage.hip"]


def _headers():
    return [f for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc"))] + ["../../include/noahmp_hip.h"]


# -ffp-contract=off: keep the reference's a*b+c rounding (no FMA contraction); no fast-math.
# -amdgpu-sched-strategy=max-ilp: the column kernel is bound by its waves' own instruction streams; this scheduler fills the gfx950 hazard
# slots (v_cmp -> v_cndmask, v_div_scale -> v_div_fmas) with independent work instead of s_nop: 584 -> 376 static s_nop, land kernel -0.45 %
# (A/B, profiles/r05_experiments.md section 6).  Scheduling only: same instructions, same results.  noahmp_jit.hip passes the same flag.
# -fno-slp-vectorize: packed float32 (v_pk_*) is no throughput lever on this chip (a v_pk_mul_f32 occupies the SIMD as long as two v_mul_f32) and
# pairing the operands costs moves and registers: without the SLP vectorizer the land kernel has 242 fewer static v_mov, 231 instead of 243
# VGPRs and runs 0.6 % faster (A/B twice on one box: 3.210 / 3.215 -> 3.193 / 3.193 ms).  Same IEEE operations, same results.
# -instcombine-max-copied-from-constant-users: the kernel-argument block (KArgs, 1.6 KB, passed by value) is read straight from the
# kernel-argument segment by scalar loads only if InstCombine can prove that its private copy is never written -- and it gives up after 300
# users of that copy.  The one-launch class-range kernel (noahmp_ranges_kernel: land + land-ice + skipped bodies, round 6) has more: without
# the flag it starts by copying all 1 640 bytes into scratch memory and reads every array pointer back from there (ISA: 22 scratch stores,
# 309 scratch loads, private_segment_fixed_size 1640); with it 0, and the registers of the land kernel alone (231).  noahmp_jit.hip passes it too.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp",
         "-mllvm", "-instcombine-max-copied-from-constant-users=100000",
         "-Wno-unused-value", "-I" + os.path.join(_HERE, "..", "include")]


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in deps)


def _flags_stamp(objdir):
    p = os.path.join(objdir, "flags.txt")
    return open(p).read() if os.path.exists(p) else None


def needs_build():
    """The default library is missing, older than a source / header, or was last built with other flags than the default ones (e.g. an
    NSOIL experiment that overwrote it: every nsoil = 4 call would fail with NOAHMP_ERR_NSOIL_UNSUPPORTED until a forced rebuild)."""
    return _stale(LIB, _sources() + _headers()) or _flags_stamp(OBJ) != " ".join(FLAGS)


def build(force=False, verbose=False, extra_flags=(), lib=None, jobs=None):
    """extra_flags / lib: experiment builds (e.g. -DNMP_TRUNC=3 into another .so); they get their own object directory."""
    nsoil_env = os.environ.get("NMP_NSOIL")
    if nsoil_env and lib is None and not any(f.startswith("-DNOAHMP_NSOIL") for f in extra_flags) and int(nsoil_env) != 4:
        # another layer count is a library of its own (variants/lib_nsoil<n>.so; NMP_LIB=... or Engine(lib_path=...) loads it): the default
        # library stays the 4-layer build every BASELINE config uses
        lib = os.path.join(CSRC, "variants", "lib_nsoil%d.so" % int(nsoil_env))
    lib = lib or LIB
    if not force and not extra_flags and not nsoil_env and not needs_build():
        return lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = OBJ if lib == LIB else lib + ".obj"
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    flags = FLAGS + list(extra_flags)
    if nsoil_env and int(nsoil_env) != 4 and not any(f.startswith("-DNOAHMP_NSOIL") for f in flags):
        flags.append("-DNOAHMP_NSOIL=%d" % int(nsoil_env))       # soil layers: a build-time choice (include/noahmp_hip.h); default 4
    stamp = os.path.join(objdir, "flags.txt")
    same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(flags)
    hdrs = _headers()

    def one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or not same_flags or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(jobs or min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(one, _sources()))
    with open(stamp, "w") as f:
        f.write(" ".join(flags))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib, "-lhiprtc", "-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    build(force=True, verbose=True)

"""Build the in-tree HIP engine for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "libnoahmp_hip.so")
SOURCES = ["noahmp_engine.hip", "noahmp_groundwater.hip", "noahmp_init.hip", "noahmp_forcing.hip", "noahmp_engine_d1_r1.hip", "noahmp_engine_d3_r1.hip", "noahmp_engine_d3_r5.hip", "noahmp_engine_d4_r1.hip", "noahmp_engine_d4_r3.hip", "noahmp_jit.hip"]
def _headers():
    return [f for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc"))] + ["../../include/noahmp_hip.h"]
# -ffp-contract=off: keep the reference's a*b+c rounding (no FMA contraction); no fast-math.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
         "-Wno-unused-value", "-I" + os.path.join(_HERE, "..", "include")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + _headers())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB, "-lhiprtc", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)

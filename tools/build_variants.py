"""Experiment builds of the engine (other launch bounds / workgroup size / layer storage, -D knobs) into
noahmp_amd/csrc/variants/lib_<tag>.so; run them with NMP_LIB=... bench.py.  usage: build_variants.py tag=flags ..."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from noahmp_amd import build  # noqa: E402

for spec in sys.argv[1:]:
    tag, flags = spec.split("=", 1)
    lib = os.path.join(build.CSRC, "variants", "lib_%s.so" % tag)
    build.build(force=False, extra_flags=flags.split(), lib=lib, jobs=4)
    print("built", lib, flush=True)

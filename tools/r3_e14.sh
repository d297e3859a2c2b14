#!/bin/bash
# round-3: A/B of the libraries given (if any), then the phase shares of the profiling build (-DNMP_PHASE_TIMERS) on the headline workload.
# Build the profiling library first, in the dev container:
#   python -c "from noahmp_amd import build as b; import os; b.build(extra_flags=['-DNMP_PHASE_TIMERS'], lib=os.path.join(b.CSRC, 'variants', 'lib_prof.so'))"
O=gpurun_out/r3_e14; mkdir -p $O
[ $# -gt 0 ] && bash tools/r3_ab.sh "$@"
NMP_PHASE_PROF=1 NMP_LIB=noahmp_amd/csrc/variants/lib_prof.so timeout 600 python bench.py --steps 24 --warmup 2 --no-cpu-baseline --no-scaling-reference > $O/prof.json 2> $O/prof.err
grep "^phase" $O/prof.err
grep -v "^phase" $O/prof.err | tail -3

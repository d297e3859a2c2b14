#!/bin/bash
# Build the truncated profiling variants (-DNMP_TRUNC=1..9) of the engine into noahmp_amd/csrc/variants/ (dev container),
# then on the GPU box:  for n in 1..9: NMP_LIB=.../lib_trunc$n.so python tools/steps_run.py 8
# The difference between consecutive kernel times is the cost of a phase.
cd "$(dirname "$0")/../noahmp_amd/csrc" || exit 1
mkdir -p variants
B="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -ffp-contract=off -Wno-unused-value -I../../include -DNMP_NO_FIXED_KERNELS noahmp_engine.hip noahmp_groundwater.hip noahmp_init.hip noahmp_forcing.hip"
for grp in "1 2 3" "4 5 6" "7 8 9"; do
  for n in $grp; do $B -DNMP_TRUNC=$n -o variants/lib_trunc$n.so 2>&1 | grep -E " error|error:" & done
  wait
done
ls -la variants/lib_trunc*.so

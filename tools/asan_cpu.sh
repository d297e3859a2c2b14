#!/bin/bash
# Host-side AddressSanitizer pass on the CPU (device ASan / host ASan beside the HSA runtime are not available on this pool): the
# library's host code instrumented, the gloo-free halo tests (2 ... 9 ranks over sockets; a rank that never arrives), the run-time compilation + disk cache test.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); CS=$R/noahmp_amd/csrc; W=/tmp/nmp_asan; mkdir -p $W
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
for f in noahmp_engine noahmp_forcing noahmp_groundwater noahmp_init noahmp_halo noahmp_jit noahmp_sort noahmp_stage; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -g -fPIC -std=c++17 -ffp-contract=off -Wno-unused-value -Wno-option-ignored \
      -fsanitize=address -fno-omit-frame-pointer -I$R/include -c $CS/$f.hip -o $W/$f.o &
done; wait
python -c "from noahmp_amd import build; build.build()"
cp $CS/libnoahmp_hip.so $W/orig.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -shared-libasan $W/*.o $CS/_obj/noahmp_engine_d*.o -o $CS/libnoahmp_hip.so -lhiprtc -ldl -lpthread
trap 'cp $W/orig.so $CS/libnoahmp_hip.so' EXIT
cd $R
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$RT python -m pytest tests/test_multirank.py tests/test_host.py -x -q -m "not gpu" -k "cabi or runtime or symbols or abi"

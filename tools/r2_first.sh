#!/bin/bash
# round-2 first GPU pass: sort path tests, small benches (1 rank, 2 ranks over gloo), then the full-size headline
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export NMP_DIST_BACKEND=gloo
O=gpurun_out/r2_first; mkdir -p $O
timeout 900 python -m pytest tests/test_async.py tests/test_restart.py -x -q -m gpu > $O/pytest_sort.log 2>&1; echo "pytest rc=$?" >> $O/pytest_sort.log
timeout 300 python bench.py --ni 512 --nj 256 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_small.log 2>&1; echo "rc=$?" >> $O/bench_small.log
timeout 300 python bench.py --ni 512 --nj 256 --steps 8 --warmup 2 --workload config4 --no-cpu-baseline > $O/bench_small_c4.log 2>&1; echo "rc=$?" >> $O/bench_small_c4.log
timeout 600 python bench.py --gpus 2 --ni 512 --nj 256 --steps 8 --warmup 2 > $O/bench_small_2r.log 2>&1; echo "rc=$?" >> $O/bench_small_2r.log
timeout 900 python bench.py > $O/bench_full.log 2>&1; echo "rc=$?" >> $O/bench_full.log
timeout 600 python bench.py --workload config4 --no-cpu-baseline > $O/bench_full_c4.log 2>&1; echo "rc=$?" >> $O/bench_full_c4.log
timeout 600 python bench.py --no-sort --no-cpu-baseline > $O/bench_full_nosort.log 2>&1; echo "rc=$?" >> $O/bench_full_nosort.log
tail -3 $O/*.log

#!/bin/bash
# what the driver runs at round end: GPU tests, smoke, the bench line
O=gpurun_out/r2_final; mkdir -p $O
( time timeout 3000 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -n 8 $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -n 3 $O/smoke.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.log 2>&1; tail -n 6 $O/bench_driver.log | cut -c1-1500

#!/bin/bash
O=gpurun_out/r2_tests2; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -n 12 $O/pytest_gpu.log
timeout 600 python tools/overhead.py > $O/overhead.log 2>&1; tail -n 22 $O/overhead.log

#!/bin/bash
# round-2 GPU pass: the new tests
O=gpurun_out/r2_tests; mkdir -p $O
timeout 1500 python -m pytest tests/test_sort_gpu.py tests/test_multirank.py tests/test_libm.py -x -q -m gpu > $O/pytest_new.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new.log
tail -n 25 $O/pytest_new.log

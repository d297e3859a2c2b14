#!/bin/bash
# round-3 first GPU pass: division micro-benchmark + the scalar-sweep parity tests
O=gpurun_out/r3_first; mkdir -p $O
timeout 120 tools/micro/div_known.bin > $O/div_known.txt 2>&1; cat $O/div_known.txt
timeout 1500 python -m pytest tests/test_scalars.py tests/test_fuzz.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 8 $O/pytest.log

#!/bin/bash
# round 4, first GPU call: VALU issue micro-benchmark, the new multi-rank / config-5 tests, a headline bench line
O=gpurun_out/r4_first; mkdir -p $O
./tools/micro/valu_issue.bin > $O/valu_issue.txt 2>&1; echo "micro rc=$?"
timeout 1500 python -m pytest tests/test_multirank.py tests/test_config5.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.json

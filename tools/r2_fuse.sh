#!/bin/bash
timeout 900 python -m pytest tests/test_multirank.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for args in "--ni 1152 --nj 768 --workload config4 --steps 96 --warmup 12" "--workload config4"; do
  python bench.py $args --no-cpu-baseline --no-scaling-reference 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$args', '%.4g' % j['value'], '%.3f' % j['ms_per_step'], j['column_kernels_ms_per_step']['all_max_over_ranks'])"
done

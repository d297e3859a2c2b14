#!/bin/bash
for b in 1 2 4 8; do
  python bench.py --workload config4 --tsk-bin $b --no-cpu-baseline --no-scaling-reference 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('config4 tsk-bin $b', '%.4g' % j['value'], '%.3f ms/step' % j['ms_per_step'], 'kernels %.3f' % j['column_kernels_ms_per_step']['all_max_over_ranks'])"
done
for b in 2 4; do
  python bench.py --tsk-bin $b --no-cpu-baseline --no-scaling-reference 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('config3 tsk-bin $b', '%.4g' % j['value'], '%.3f ms/step' % j['ms_per_step'], 'kernels %.3f' % j['column_kernels_ms_per_step']['all_max_over_ranks'])"
done

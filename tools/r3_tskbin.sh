#!/bin/bash
# round-3: skin-temperature bin width of the sort key on the headline workload
for b in 0.25 0.5 1 2 4; do
  python bench.py --steps 48 --warmup 6 --no-cpu-baseline --no-scaling-reference --tsk-bin $b 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('tsk_bin $b: value %.4g ms/step %.3f land kernel %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms_avg']))"
done

#!/bin/bash
# single-GPU timing of the config-4 step at the tile sizes of N = 1, 2, 4, 8 ranks (what each rank of a --gpus N run advances; no exchange partner)
for t in "4608 1536 1" "2304 1536 2" "2304 768 4" "1152 768 8"; do
  set -- $t
  python bench.py --ni $1 --nj $2 --workload config4 --steps 96 --warmup 12 --no-cpu-baseline --no-scaling-reference 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('tile of N=$3 ($1 x $2): %.3f ms/step, %.4g column-steps/s per GPU, column kernels %.3f ms' % (j['ms_per_step'], j['value'], j['column_kernels_ms_per_step']['all_max_over_ranks']))"
done

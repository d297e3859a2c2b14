#!/bin/bash
# round-3: the chunked permutation with large chunks -- tests, config-3 / config-4 bench, the N = 8 tile size
O=gpurun_out/r3_scatter; mkdir -p $O
timeout 1500 python -m pytest tests/test_sort_gpu.py tests/test_multirank.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 5 $O/pytest.log
for w in config3 config4; do
  timeout 600 python bench.py --workload $w --steps 48 --warmup 6 --no-cpu-baseline --no-scaling-reference > $O/$w.json 2> $O/$w.err
done
timeout 600 python bench.py --workload config4 --ni 1152 --nj 768 --steps 96 --warmup 6 --no-cpu-baseline > $O/tile8.json 2> $O/tile8.err
python - <<'PY'
import json
for f in ("config3", "config4", "tile8"):
    try:
        d = json.loads(open("gpurun_out/r3_scatter/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value %.4g ms/step %.3f column kernels %.3f" % (d["value"], d["ms_per_step"], d["column_kernels_ms_per_step"]["all_max_over_ranks"]))
    except Exception as e:
        print(f, "failed", e)
PY

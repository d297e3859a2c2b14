#!/bin/bash
# round-3: forcing index (the step reads the forcing in (i,j) order) vs the per-step permutation, configs 3 and 4, alternated
O=gpurun_out/r3_fidx; mkdir -p $O
for rep in 1 2; do
for w in config3 config4; do
for f in "" "--permute-forcing"; do
  python bench.py --workload $w --steps 48 --warmup 6 --no-cpu-baseline --no-scaling-reference $f 2> $O/err.log | tail -1 > $O/b.json || tail -5 $O/err.log
  python - $O/b.json "$w $f" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read()); r=d["roofline"]
print("%-28s value %.4g ms/step %.3f land kernel ms %.3f" % (sys.argv[2], d["value"], d["ms_per_step"], r["kernel_ms_avg"]))
PY
done; done; done
timeout 900 python -m pytest tests/test_multirank.py -m gpu -x -q 2>&1 | tail -3

#!/bin/bash
# round-3 A/B on one box: headline bench (no CPU leg) with each library given on the command line, twice, interleaved
O=gpurun_out/r3_ab; mkdir -p $O
for rep in 1 2; do
for lib in "$@"; do
  tag=$(basename $lib .so)
  NMP_LIB=$lib timeout 600 python bench.py --steps 48 --warmup 6 --no-cpu-baseline --no-scaling-reference > $O/$tag.$rep.json 2> $O/$tag.$rep.err
  python - $O/$tag.$rep.json $tag <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-28s value %.4g ms/step %.3f land kernel ms %.3f frac %.4f" % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"]))
PY
done; done

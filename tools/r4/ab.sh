#!/bin/bash
# A/B on one box: headline bench (no reference legs) with each library given on the command line, twice, interleaved
# usage: tools/r4/ab.sh [-a "extra bench args"] lib1.so lib2.so ...
EXTRA=""
if [ "$1" == "-a" ]; then EXTRA="$2"; shift 2; fi
O=gpurun_out/r4_ab; mkdir -p $O
for rep in 1 2; do
for lib in "$@"; do
  tag=$(basename $lib .so)
  NMP_LIB=$lib timeout 600 python bench.py --steps 48 --warmup 6 --no-cpu-baseline --no-scaling-reference $EXTRA > $O/$tag.$rep.json 2> $O/$tag.$rep.err
  python - $O/$tag.$rep.json $tag <<'PY'
import json, sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print("%-28s value %.4g ms/step %.3f land kernel ms %.3f (day %.3f night %.3f) frac %.4f" % (sys.argv[2], d["value"], d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_day"] or 0, r["kernel_ms_night"] or 0, r["frac"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done; done

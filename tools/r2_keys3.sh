#!/bin/bash
O=gpurun_out/r2_keys3; mkdir -p $O
run() {
  tag=$1; shift
  timeout 600 python bench.py --no-cpu-baseline "$@" > $O/$tag.log 2>&1
  python - "$O/$tag.log" "$tag" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        j = json.loads(ln)
        print("%-22s value %.4g  ms/step %.3f  land kernel %.3f ms  sort %s" % (sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["kernel_ms_avg"], j.get("sort")))
        break
else:
    print(sys.argv[2], "FAILED"); print(open(sys.argv[1]).read()[-800:])
PY
}
NMP_SORT_INPLACE=1 run inplace_resort6 --resort-every 6 --resort-frac 0
run resort6_nosnow --resort-every 6 --resort-frac 0 --no-snow-key
run resort6_notsk --resort-every 6 --resort-frac 0 --tsk-bin 0
run resort18 --resort-every 18 --resort-frac 0

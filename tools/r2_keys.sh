#!/bin/bash
# sort-key and re-sort cadence experiments on the config-3 bench, then the truncated-phase builds (profiles/r02_experiments.md)
O=gpurun_out/r2_keys; mkdir -p $O
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
run default
run resort6 --resort-every 6
run resort12 --resort-every 12
run resort0 --resort-every 0
run resort24_5pct --resort-frac 0.05
run tsk0 --tsk-bin 0
run tsk05 --tsk-bin 0.5
run tsk2 --tsk-bin 2
run tsk4 --tsk-bin 4
run snowfirst --snow-first
run nosnowkey --no-snow-key
run novegkey --no-veg-key
for n in 1 2 3 4 5 6 7 8 9; do
  v=noahmp_amd/csrc/variants/lib_trunc$n.so
  [ -f $v ] && NMP_LIB=$v run trunc$n --resort-every 0 --steps 24
done

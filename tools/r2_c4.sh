#!/bin/bash
O=gpurun_out/r2_c4; mkdir -p $O
timeout 1200 python -m pytest tests/test_multirank.py tests/test_async.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 15 $O/pytest.log
for w in "config4" "config4 --no-sort" "config3"; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline > $O/bench_$(echo $w | tr -d ' -').log 2>&1
  python - "$O/bench_$(echo $w | tr -d ' -').log" "$w" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        j = json.loads(ln)
        print("%-20s value %.4g  ms/step %.3f  kernels %s  gw %s sort %s" % (sys.argv[2], j["value"], j["ms_per_step"], j["column_kernels_ms_per_step"], j.get("groundwater"), j.get("sort")))
        break
else:
    print(sys.argv[2], "FAILED"); print(open(sys.argv[1]).read()[-1500:])
PY
done
timeout 900 python bench.py --gpus 2 --workload config4 > $O/bench_2r.log 2>&1; tail -n 3 $O/bench_2r.log | cut -c1-600

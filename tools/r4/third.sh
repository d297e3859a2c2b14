#!/bin/bash
O=gpurun_out/r4_third; mkdir -p $O
timeout 900 python -m pytest tests/test_sort_gpu.py tests/test_config5.py tests/test_parity_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for band in 0 15 15 0; do
  timeout 600 python bench.py --workload config5 --steps 48 --warmup 6 --no-cpu-baseline --lon-band $band > $O/c5_band$band.json 2> $O/c5.err
  python - $O/c5_band$band.json $band <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("config5 lon-band %s: ms/step %.3f value %.4g land kernel %.3f land-ice %.3f" % (sys.argv[2], d["ms_per_step"], d["value"], d["column_kernels_ms_per_step"]["land_or_mixed"], d["column_kernels_ms_per_step"]["land_ice"]))
PY
done

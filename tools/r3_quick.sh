#!/bin/bash
# round-3 quick GPU pass: selected parity tests + the headline bench without the CPU leg
O=gpurun_out/r3_quick; mkdir -p $O
timeout 900 python -m pytest tests/test_scalars.py tests/test_matrix.py tests/test_fuzz.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 4 $O/pytest.log
timeout 600 python bench.py --steps 48 --warmup 6 --no-cpu-baseline --no-scaling-reference > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_quick/bench.json').read().strip().splitlines()[-1])
print("value %.4g ms/step %.3f land kernel ms %.3f frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"]))
PY

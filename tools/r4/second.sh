#!/bin/bash
O=gpurun_out/r4_second; mkdir -p $O
./tools/micro/valu_issue.bin > $O/valu_issue.txt 2>&1; echo "micro rc=$?"
./tools/r4/ab.sh noahmp_amd/csrc/variants/lib_nolaunder.so noahmp_amd/csrc/variants/lib_launder.so
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 3000 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_second/bench.json').read().strip().splitlines()[-1])
print(json.dumps(d.get('options_reference'), indent=1)); print(json.dumps(d.get('host_path_reference'), indent=1))
print(d['value'], d['ms_per_step'])
PY

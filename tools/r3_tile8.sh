#!/bin/bash
# round-3: kernel trace of the config-4 step at the tile size of one rank of an 8-GPU run (1152 x 768)
O=$PWD/gpurun_out/r3_tile8; rm -rf $O; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o t8 --output-format csv -- python3 $R/bench.py --ni 1152 --nj 768 --workload config4 --steps 96 --warmup 12 --no-cpu-baseline --no-scaling-reference > $O/bench.log 2>&1
tail -1 $O/bench.log | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms/step', j['ms_per_step'], 'column kernels', j['column_kernels_ms_per_step'])"
python3 - $O <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print("%-70s calls %5s avg %9.1f us total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2_small; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/t -o b --output-format csv -- python3 $R/bench.py --ni 1152 --nj 768 --workload config4 --steps 96 --warmup 12 --no-cpu-baseline --no-scaling-reference > $O/log.txt 2>&1
python3 - "$O/t" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*kernel_stats.csv") + glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:14]:
    print("%-90s calls %5s avg %9.1f us total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY

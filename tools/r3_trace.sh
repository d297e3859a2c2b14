#!/bin/bash
# round-3: kernel trace of one workload (no counters)
R=${GRAFT_REPO_ROOT:-$(pwd)}; W=${1:-config4}; shift
O=$R/gpurun_out/r3_trace_$W; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o t --output-format csv -- python3 $R/bench.py --workload $W --steps 48 --warmup 6 --no-cpu-baseline --no-scaling-reference "$@" > $O/log.txt 2>&1
f=$(ls $O/*kernel_stats.csv $O/*/*kernel_stats.csv 2>/dev/null | head -1)
python3 - $f <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print("%-90s calls %5s avg %9.1f us  total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY

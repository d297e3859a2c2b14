#!/bin/bash
# round-3: duration of the forcing permutation kernel in the headline run (rocprofv3 kernel trace)
O=$PWD/gpurun_out/r3_scat_time; rm -rf $O; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o b --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-scaling-reference > $O/bench.log 2>&1
python3 - $O <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:5]:
    print("%-70s calls %5s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2_gw; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/gw -o gw --output-format csv -- python3 $R/tools/gw_check.py perf > $O/gw_trace.log 2>&1
cat $O/gw/*kernel_stats.csv | head -4
cd $R && timeout 600 python -m pytest tests/test_groundwater.py tests/test_multirank.py -x -q -m gpu 2>&1 | tail -3

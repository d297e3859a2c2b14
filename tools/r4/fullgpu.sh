#!/bin/bash
O=gpurun_out/r4_fullgpu; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q --durations=15 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -25 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log

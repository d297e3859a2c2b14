#!/bin/bash
# round-3: the whole GPU suite + smoke
O=gpurun_out/r3_fullgpu; mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 6 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 2 $O/smoke.log

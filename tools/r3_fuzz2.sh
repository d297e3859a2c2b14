#!/bin/bash
# round-3, second randomised GPU-vs-oracle run: option sets the first run (tools/r3_fuzz.sh) does not hold, most of them served by run-time
# compiled kernels, 8 seeds x 16 384 columns x 3 steps each, per-seed scalars
O=gpurun_out/r3_fuzz2; mkdir -p $O; : > $O/fuzz.log
for o in "scalars=1 idveg=4 iopt_run=3 iopt_inf=1 iopt_frz=2" "scalars=1 iopt_run=4 iopt_inf=2 iopt_frz=2 iopt_btr=2" \
         "scalars=1 idveg=5 iopt_crs=2 iopt_sfc=2 iopt_stc=2" "scalars=1 idveg=1 iopt_alb=1 iopt_snf=2 iopt_tbot=1 iopt_rad=1" \
         "scalars=1 idveg=2 iopt_run=2 iopt_btr=3 iopt_rad=2" "scalars=1 idveg=3 iopt_run=5 iopt_frz=2 iopt_inf=2 iopt_snf=3"; do
  timeout 1500 python tools/fuzz_parity.py gpu 8 16384 $o 2>&1 | grep "^gpu\|DIFFER\|Error\|Traceback" | head -4 | tee -a $O/fuzz.log
done

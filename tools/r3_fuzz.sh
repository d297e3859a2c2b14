#!/bin/bash
# round-3 randomised GPU-vs-oracle run: option sets x seeds, with the namelist scalars and with per-seed DT / DZS / YR / JULIAN / ZLVL
O=gpurun_out/r3_fuzz; mkdir -p $O; : > $O/fuzz.log
for o in "" "scalars=1" "idveg=2 iopt_run=3 iopt_stc=2 iopt_sfc=2 iopt_frz=2 scalars=1" "iopt_rad=1 iopt_alb=1 iopt_snf=3 iopt_tbot=1 idveg=5 iopt_crs=2 iopt_btr=2 iopt_inf=2" "iopt_run=5 idveg=3 scalars=1" "iopt_run=2 iopt_btr=3 iopt_rad=2 iopt_snf=2 scalars=1" "idveg=4 iopt_run=3" "scalars=1 idveg=4 iopt_run=3 iopt_inf=1 iopt_frz=2"; do
  timeout 1200 python tools/fuzz_parity.py gpu 6 16384 $o 2>&1 | grep "^gpu\|DIFFER\|Error\|Traceback" | head -4 | tee -a $O/fuzz.log
done
timeout 900 python tools/config5_run.py 720 360 96 4096 2>&1 | tail -3 | cut -c1-400 | tee -a $O/fuzz.log

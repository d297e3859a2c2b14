python bench.py 2>&1 | tail -1 | cut -c1-200
for o in "" "idveg=2 iopt_run=3 iopt_stc=2 iopt_sfc=2 iopt_frz=2" "iopt_rad=1 iopt_alb=1 iopt_snf=3 iopt_tbot=1 idveg=5 iopt_crs=2 iopt_btr=2 iopt_inf=2" "iopt_run=5 idveg=3" "iopt_run=2 iopt_btr=3 iopt_rad=2 iopt_snf=2"; do timeout 900 python tools/fuzz_parity.py gpu 3 16384 $o 2>&1 | grep "^gpu\|DIFFER\|Error" | head -4; done

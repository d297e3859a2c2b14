for o in "idveg=2 iopt_run=3 iopt_stc=2 iopt_sfc=2 iopt_frz=2" "iopt_rad=1 iopt_alb=1 iopt_snf=3 iopt_tbot=1 idveg=5 iopt_crs=2 iopt_btr=2 iopt_inf=2" "iopt_run=2 iopt_btr=3 iopt_rad=2 iopt_snf=2" "idveg=4 iopt_run=4" "idveg=3"; do
  timeout 600 python tools/config5_run.py 720 360 96 8192 $o 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['options'], d['sample_bit_identical'], d['checkpoints'], '%.3g' % d['column_steps_per_s'])"
done

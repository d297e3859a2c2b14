#!/bin/bash
# round-3: truncated builds (-DNMP_TRUNC=n: the column step stops after phase n) of the specialised land kernel on config 3, sorted layout
O=gpurun_out/r3_trunc; mkdir -p $O
for n in 1 2 3 4 5 6 7 8; do
  NMP_LIB=noahmp_amd/csrc/variants/lib_trunc$n.so timeout 600 python bench.py --steps 24 --warmup 2 --no-cpu-baseline --no-scaling-reference --resort-every 0 > $O/t$n.json 2> $O/t$n.err
  python - $O/t$n.json $n <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print("trunc %s: land kernel ms %.3f (day %.3f night %.3f)" % (sys.argv[2], r["kernel_ms_avg"], r["kernel_ms_day"] or 0, r["kernel_ms_night"] or 0))
except Exception as e:
    print("trunc", sys.argv[2], "failed", e)
PY
done
tools/r3_ab.sh noahmp_amd/csrc/variants/lib_e5.so noahmp_amd/csrc/variants/lib_e6.so

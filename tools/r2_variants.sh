#!/bin/bash
# occupancy / storage variants of the specialised column kernel on the config-3 bench (profiles/r02_experiments.md)
O=gpurun_out/r2_variants; mkdir -p $O
for v in "" $(ls noahmp_amd/csrc/variants/lib_*.so 2>/dev/null); do
  tag=$(basename "${v:-default}" .so)
  NMP_LIB=$v timeout 600 python bench.py --no-cpu-baseline --no-scaling-reference --steps 24 --warmup 6 --resort-every 0 > $O/$tag.log 2>&1
  python - "$O/$tag.log" "$tag" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        j = json.loads(ln)
        print("%-12s value %.4g  ms/step %.3f  land kernel %.3f ms  frac %.4f" % (sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["kernel_ms_avg"], j["roofline"]["frac"]))
        break
else:
    print(sys.argv[2], "FAILED"); print(open(sys.argv[1]).read()[-800:])
PY
done

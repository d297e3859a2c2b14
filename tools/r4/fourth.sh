#!/bin/bash
# round 4: (1) config-4 step at the N = 8 tile size (1152 x 768) with workgroups of 256 / 128 / 64 threads; (2) longitude-band width
# sweep on config 5; (3) PMC pass of config 5 with and without the band key (lane utilisation, VALU per column-step)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4_fourth; mkdir -p $O
cd $R
for lib in "" noahmp_amd/csrc/variants/lib_block128.so noahmp_amd/csrc/variants/lib_block64.so ""; do
  tag=$(basename "${lib:-default}" .so)
  NMP_LIB=${lib:+$R/$lib} timeout 600 python bench.py --ni 1152 --nj 768 --workload config4 --steps 96 --warmup 12 --no-cpu-baseline --no-scaling-reference > $O/t8_$tag.json 2> $O/t8_$tag.err
  python - $O/t8_$tag.json $tag <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("tile8 %-10s ms/step %.4f column kernels %s" % (sys.argv[2], d["ms_per_step"], d["column_kernels_ms_per_step"]))
PY
done
for lib in "" noahmp_amd/csrc/variants/lib_block128.so; do
  tag=$(basename "${lib:-default}" .so)
  NMP_LIB=${lib:+$R/$lib} timeout 600 python bench.py --steps 48 --warmup 6 --no-cpu-baseline --no-scaling-reference > $O/c3_$tag.json 2> $O/c3_$tag.err
  python - $O/c3_$tag.json $tag <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("config3 %-10s ms/step %.4f land kernel %.4f" % (sys.argv[2], d["ms_per_step"], d["roofline"]["kernel_ms_avg"]))
PY
done
for band in 30 22.5 15 11.25; do
  timeout 600 python bench.py --workload config5 --steps 48 --warmup 6 --no-cpu-baseline --lon-band $band > $O/c5_band$band.json 2> $O/c5.err
  python - $O/c5_band$band.json $band <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("config5 lon-band %s: ms/step %.3f value %.4g land kernel %.3f land-ice %.3f" % (sys.argv[2], d["ms_per_step"], d["value"], d["column_kernels_ms_per_step"]["land_or_mixed"], d["column_kernels_ms_per_step"]["land_ice"]))
PY
done
cd /tmp && export TMPDIR=/tmp
for band in 0 15; do
  rm -rf $O/pmc5_$band
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc5_$band -o bench --output-format csv -- python3 $R/bench.py --workload config5 --no-cpu-baseline --steps 24 --warmup 2 --lon-band $band > $O/pmc5_$band.log 2>&1
  python3 - $O/pmc5_$band $band <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/*_counter_collection.csv") + glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "noahmp_column_kernel<256, true, 1>" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
w = m["SQ_WAVES"]
print("config5 band %s: launches %d waves %.0f VALU/wave %.0f SALU/wave %.0f lane util %.3f wave cycles x4 / wave %.0f WAIT_ANY share %.3f WAIT_INST_ANY share %.3f" % (
    sys.argv[2], len(acc["SQ_WAVES"]), w, m["SQ_INSTS_VALU"] / w, m["SQ_INSTS_SALU"] / w, m["SQ_THREAD_CYCLES_VALU"] / (64 * m["SQ_ACTIVE_INST_VALU"]),
    4 * m["SQ_WAVE_CYCLES"] / w, m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]))
import json
json.dump({"band_degrees": float(sys.argv[2]), "launches": len(acc["SQ_WAVES"]), "pmc_mean_per_launch": m}, open(sys.argv[1] + ".json", "w"), indent=1)
PY
done

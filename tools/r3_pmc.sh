#!/bin/bash
# round-3: SQ counters of the land kernel for each library given (separate --pmc pass each; no trace)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  export NMP_LIB=$R/$lib
  rm -rf $O/$tag
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $O/$tag -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-scaling-reference --steps 24 --warmup 2 > $O/$tag.log 2>&1
  python3 - $O/$tag $tag <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/*_counter_collection.csv") + glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "noahmp_column_kernel<256, true, 1>" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
w = m["SQ_WAVES"]
print("%-12s launches %d  VALU/wave %.0f  SALU/wave %.0f  ACTIVE_INST_VALU %.4g (x4 cyc/inst %.2f)  lane util %.3f  WAVE_CYCLES %.4g  WAIT_INST_ANY share %.3f" % (
    sys.argv[2], len(acc["SQ_WAVES"]), m["SQ_INSTS_VALU"] / w, m["SQ_INSTS_SALU"] / w, m["SQ_ACTIVE_INST_VALU"],
    4 * m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"], m["SQ_THREAD_CYCLES_VALU"] / (64 * m["SQ_ACTIVE_INST_VALU"]) if m.get("SQ_THREAD_CYCLES_VALU") else 0,
    m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]))
PY
done

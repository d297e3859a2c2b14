#!/bin/bash
# round-3: instruction-fetch / instruction-class counters of the land kernel (separate --pmc passes; no trace)
R=${GRAFT_REPO_ROOT:-$(pwd)}
lib=$1; tag=$(basename $lib .so)
O=$R/gpurun_out/r3_pmc2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export NMP_LIB=$R/$lib
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  i=$((i+1)); rm -rf $O/$tag.$i
  rocprofv3 --pmc $set -d $O/$tag.$i -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-scaling-reference --steps 24 --warmup 2 > $O/$tag.$i.log 2>&1
  python3 - $O/$tag.$i <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/*_counter_collection.csv") + glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")
if not fs:
    print("no counters in", sys.argv[1]); sys.exit(0)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "noahmp_column_kernel<256, true, 1>" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("  ".join("%s %.4g" % (k, sum(v) / len(v)) for k, v in sorted(acc.items())))
PY
done

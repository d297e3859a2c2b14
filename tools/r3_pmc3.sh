#!/bin/bash
# round-3: memory-latency counters of the land kernel (in-flight levels / instruction counts)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_pmc3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_WAVES SQ_INSTS_FLAT"; do
  i=$((i+1)); rm -rf $O/p$i
  rocprofv3 --pmc $set -d $O/p$i -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-scaling-reference --steps 24 --warmup 0 > $O/p$i.log 2>&1
  python3 - $O/p$i <<'PY'
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

#!/bin/bash
# GPU experiments of the Noah-MP HIP engine, one parameterised script (run ON THE GPU BOX from the repo root):
#
#   gpurun --timeout 1800 -- ./tools/experiments.sh <command> [args]
#
# Every number quoted in profiles/r0N_experiments.md comes from one of these commands (tools/README.md maps the per-round script names
# those files cite onto them).  Output goes under gpurun_out/exp_<command>/ (scratch); copy what is to be judged into profiles/.
# Counters are collected in passes of their own (--pmc without any trace), as MI355X_MICROARCH.md prescribes.
#
#   tests [pytest args]            GPU test suite (default: all of tests/ -m gpu) + __graft_entry__.smoke()
#   bench [bench.py args]          one bench line, summarised
#   ab [-a "bench args"] LIB...    headline bench with each library, twice, interleaved (A/B on one box)
#   variants [bench args]          the default library and every noahmp_amd/csrc/variants/lib_*.so once
#   tiles                          config-4 step at the tile sizes of N = 1, 2, 4, 8 ranks (one GPU, no exchange partner)
#   tile8 [LIB]                    kernel trace of the config-4 step at the N = 8 tile size (1152 x 768)
#   trace WORKLOAD [bench args]    rocprofv3 kernel trace of one workload (top kernels)
#   pmc SET [LIB] [bench args]     counters of the land kernel: SET = sq | inst | mem  (see pmc_sets below)
#   pmc5 [--smooth] [BAND...]      config 5 (--smooth: spatially smooth forcing factors): lane utilisation / instructions per wave of the land kernel for longitude-band widths
#   band [WIDTH...]                config 5 bench for longitude-band widths (degrees; 0 = no band key)
#   phase [LIB...]                 phase shares of the profiling build (-DNMP_PHASE_TIMERS: variants/lib_prof.so), after an optional A/B
#                                  (round 5's `k2` command and the -DNMP_TRUNC=n builds went with their code in round 6: git show 5bb0a94:tools/experiments.sh)
#   faulthunt [N [- [ENV=VALUE...]]]   the whole -m gpu suite N times (default 10), each in its own process, with the abort shim;
#                                  GPU_PINNED_MIN_XFER_SIZE=128 restores the runtime default under which 7 of 10 runs died (round 5)
#   cost [WORKLOAD...]             the cost sub-key of the column order (bench.py --cost-key: trip counts recorded by the step before the sort),
#                                  A/B per workload (default config3 config5): off | on after the warm-up | on + a re-sort every 6 / 12 steps
#   micro NAME                     run tools/micro/NAME.bin (built in the dev container: hipcc --offload-arch=gfx950 -O3 NAME.hip -o NAME.bin)
#   fuzz [SEEDS [COLUMNS]]         randomised GPU-vs-oracle runs over option sets (tools/fuzz_parity.py) + a config-5 chain
#   fuzzopts [NSETS [SEED]]        the same over NSETS random option sets (every OPT_* drawn from its supported range; hiprtc kernels)
#   fuzzopts5 [NSETS [SEED]]       the sorted config-5 chain (class-range kernels) under random option sets, sample vs the oracle
#   spread5                        config 5 under the i.i.d. and the spatially smooth forcing-factor generator: trip-count spread per wavefront (cost-record build)
#   vegcost [usgs|modis] [ni nj] [dveg]   land-kernel time per vegetation category (input of noahmp_hip_sort_set_veg_order)
#   stage [THREADS...]             pageable arrays through the engine's bounce buffers by copy threads + a 10-process fault hunt without GPU_PINNED_MIN_XFER_SIZE
#   profile TAG                    the evidence for profiles/: plain bench, kernel traces (config 3 / 4 / 5, groundwater), FETCH / WRITE /
#                                  SQ passes; then in the dev container: python tools/collect_profile.py TAG
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
CMD=$1; shift
O=$R/gpurun_out/exp_$CMD; mkdir -p "$O"
LAND='noahmp_ranges_kernel<64>'       # the step's column kernel (land + land-ice + skipped ranges in one launch since round 6)

summarise() {      # summarise FILE TAG: one line of a bench JSON
  python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
    r, c = d["roofline"], d["column_kernels_ms_per_step"]
    print("%-24s value %.4g  ms/step %.4f  land kernel ms %.4f (day %s night %s)  land ice %.3f  frac %.4f" % (
        sys.argv[2], d["value"], d["ms_per_step"], r["kernel_ms_avg"], "%.3f" % r["kernel_ms_day"] if r.get("kernel_ms_day") else "-",
        "%.3f" % r["kernel_ms_night"] if r.get("kernel_ms_night") else "-", c["land_ice"], r["frac"]))
except Exception as e:                                  # noqa: BLE001
    print(sys.argv[2], "FAILED", e)
    print(open(sys.argv[1]).read()[-600:])
PY
}

top_kernels() {    # top_kernels DIR [N]
  python3 - "$1" "${2:-12}" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f:
    print("no kernel_stats.csv under", sys.argv[1]); sys.exit(0)
for r in list(csv.DictReader(open(f[0])))[:int(sys.argv[2])]:
    print("%-90s calls %5s avg %9.1f us  total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
}

land_counters() {  # land_counters DIR TAG: means per launch of the land kernel's counters + derived figures
  python3 - "$1" "$2" "$LAND" <<'PY'
import collections, csv, glob, json, sys
fs = glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True)
if not fs:
    print("no counters in", sys.argv[1]); sys.exit(0)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if sys.argv[3] in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print(sys.argv[2], " ".join("%s %.5g" % kv for kv in sorted(m.items())))
w = m.get("SQ_WAVES")
if w and "SQ_INSTS_VALU" in m:
    line = "   launches %d  waves %.0f  VALU/wave %.0f" % (len(acc["SQ_WAVES"]), w, m["SQ_INSTS_VALU"] / w)
    if "SQ_INSTS_SALU" in m: line += "  SALU/wave %.0f" % (m["SQ_INSTS_SALU"] / w)
    if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m: line += "  lane utilisation %.3f" % (m["SQ_THREAD_CYCLES_VALU"] / (64 * m["SQ_ACTIVE_INST_VALU"]))
    if "SQ_WAVE_CYCLES" in m:
        line += "  cycles resident per wave %.0f" % (4 * m["SQ_WAVE_CYCLES"] / w)
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            if k in m: line += "  %s share %.3f" % (k, m[k] / m["SQ_WAVE_CYCLES"])
    print(line)
json.dump({"tag": sys.argv[2], "launches": len(next(iter(acc.values()))), "pmc_mean_per_launch": m}, open(sys.argv[1] + ".json", "w"), indent=1)
PY
}

pmc_sets() {
  case $1 in
    sq)   echo "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" ;;
    inst) echo "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES" ;;
    mem)  echo "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAVES" ;;
    icache) echo "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAVES" ;;
    *) echo "" ;;
  esac
}

QUIET="--no-cpu-baseline --no-scaling-reference"

case $CMD in
tests)
  if [ $# -eq 0 ]; then set -- tests; fi
  timeout ${NMP_TEST_TIMEOUT:-900} python -m pytest "$@" -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
  ;;
bench)
  timeout 1200 python bench.py "$@" > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
  summarise $O/bench.json bench; tail -3 $O/bench.err
  ;;
ab)
  EXTRA=""
  if [ "$1" == "-a" ]; then EXTRA="$2"; shift 2; fi
  for rep in 1 2; do for lib in "$@"; do
    tag=$(basename $lib .so)
    NMP_LIB=$R/$lib timeout 600 python bench.py --steps 48 --warmup 6 $QUIET $EXTRA > $O/$tag.$rep.json 2> $O/$tag.$rep.err
    summarise $O/$tag.$rep.json $tag
  done; done
  ;;
variants)
  for v in "" $(ls noahmp_amd/csrc/variants/lib_*.so 2>/dev/null); do
    tag=$(basename "${v:-default}" .so)
    NMP_LIB=${v:+$R/$v} timeout 600 python bench.py $QUIET --steps 24 --warmup 6 --resort-every 0 "$@" > $O/$tag.json 2> $O/$tag.err
    summarise $O/$tag.json $tag
  done
  ;;
tiles)
  for t in "4608 1536 1" "2304 1536 2" "2304 768 4" "1152 768 8"; do
    set -- $t
    python bench.py --ni $1 --nj $2 --workload config4 --steps 96 --warmup 12 $QUIET > $O/n$3.json 2> $O/n$3.err
    summarise $O/n$3.json "tile of N=$3 ($1 x $2)"
  done
  ;;
tile8)
  lib=$1
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/trace
  NMP_LIB=${lib:+$R/$lib} rocprofv3 --kernel-trace --stats -d $O/trace -o t8 --output-format csv -- python3 $R/bench.py --ni 1152 --nj 768 --workload config4 --steps 96 --warmup 12 $QUIET > $O/bench.log 2>&1
  summarise $O/bench.log tile8; top_kernels $O/trace 14
  ;;
trace)
  W=${1:-config4}; shift
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/$W
  rocprofv3 --kernel-trace --stats -d $O/$W -o t --output-format csv -- python3 $R/bench.py --workload $W --steps 48 --warmup 6 $QUIET "$@" > $O/$W.log 2>&1
  summarise $O/$W.log $W; top_kernels $O/$W 12
  ;;
pmc)
  set_name=$1; shift
  lib=""; if [ -n "$1" ] && [ -f "$R/$1" ]; then lib=$1; shift; fi
  counters=$(pmc_sets $set_name)
  [ -z "$counters" ] && { echo "unknown counter set $set_name"; exit 1; }
  tag=${set_name}_$(basename "${lib:-default}" .so)
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/$tag
  NMP_LIB=${lib:+$R/$lib} rocprofv3 --pmc $counters -d $O/$tag -o bench --output-format csv -- python3 $R/bench.py $QUIET --steps 24 --warmup 2 "$@" > $O/$tag.log 2>&1
  land_counters $O/$tag $tag
  ;;
pmc5)
  SM=""; TAG=""
  if [ "$1" == "--smooth" ]; then SM="--config5-smooth=1"; TAG="smooth_"; shift; fi
  if [ "$1" == "--smooth2" ]; then SM="--config5-smooth=2"; TAG="smooth2_"; shift; fi
  if [ $# -eq 0 ]; then set -- 0 15; fi
  cd /tmp && export TMPDIR=/tmp
  for band in "$@"; do
    rm -rf $O/${TAG}band$band
    rocprofv3 --pmc $(pmc_sets sq) -d $O/${TAG}band$band -o bench --output-format csv -- python3 $R/bench.py --workload config5 $SM --no-cpu-baseline --steps 24 --warmup 2 --lon-band $band > $O/${TAG}band$band.log 2>&1
    land_counters $O/${TAG}band$band "config5 ${TAG}lon-band $band"
  done
  ;;
band)
  if [ $# -eq 0 ]; then set -- 0 30 22.5 15 11.25 0; fi
  for band in "$@"; do
    timeout 600 python bench.py --workload config5 --steps 48 --warmup 6 --no-cpu-baseline --lon-band $band > $O/band$band.json 2> $O/band$band.err
    summarise $O/band$band.json "config5 lon-band $band"
  done
  ;;
phase)
  # build first (dev container): python tools/build_variants.py prof=-DNMP_PHASE_TIMERS
  [ $# -gt 0 ] && "$0" ab "$@"
  NMP_PHASE_PROF=1 NMP_LIB=$R/noahmp_amd/csrc/variants/lib_prof.so timeout 600 python bench.py --steps 24 --warmup 2 $QUIET > $O/prof.json 2> $O/prof.err
  grep "^phase" $O/prof.err; grep -v "^phase" $O/prof.err | tail -3
  ;;
faulthunt)
  n=${1:-10}; bad=0
  # (round 5: the sort tests' copies are pageable again by default; "pageable" is kept as a no-op second argument)
  shift; shift
  for kv in "$@"; do export "$kv"; echo "faulthunt: $kv"; done          # e.g. GPU_PINNED_MIN_XFER_SIZE=100000 (MiB: never pin a pageable buffer in place)
  O=$O/$(echo "run$*" | tr -c 'A-Za-z0-9=_\n' '_'); mkdir -p $O
  for i in $(seq 1 $n); do
    ABORT_SHIM_OUT=$O/abort$i.txt LD_PRELOAD=$R/tools/dbg/libabort_shim.so timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/run$i.log 2>&1
    rc=$?
    [ $rc -ne 0 ] && bad=$((bad+1))
    echo "run $i rc=$rc $(tail -1 $O/run$i.log | cut -c1-120)"
    [ -f $O/abort$i.txt ] && head -30 $O/abort$i.txt
  done
  echo "faulthunt: $bad of $n runs failed (GPU_PINNED_MIN_XFER_SIZE=${GPU_PINNED_MIN_XFER_SIZE:-conftest default})"
  ;;
cost)
  # build first (dev container): python tools/build_variants.py cost=-DNMP_COST_RECORD
  export NMP_LIB=$R/noahmp_amd/csrc/variants/lib_cost.so
  if [ $# -eq 0 ]; then set -- config3 config5; fi
  for w in "$@"; do
    for v in "" "--cost-key" "--cost-key --cost-resort-every 12 --resort-every 12" "--cost-key --cost-resort-every 6 --resort-every 6" ""; do
      tag=$(echo "$w$v" | tr -d ' ' | tr -s '-' '_')
      timeout 900 python bench.py --workload $w --steps 48 --warmup 6 --no-cpu-baseline $v > $O/$tag.json 2> $O/$tag.err
      summarise $O/$tag.json "$w $v"
    done
  done
  ;;
micro)
  timeout 300 ./tools/micro/$1.bin > $O/$1.txt 2>&1; echo "$1 rc=$?"; cat $O/$1.txt
  ;;
fuzz)
  seeds=${1:-6}; cols=${2:-16384}; : > $O/fuzz.log
  for o in "" "scalars=1" "idveg=2 iopt_run=3 iopt_stc=2 iopt_sfc=2 iopt_frz=2 scalars=1" \
           "iopt_rad=1 iopt_alb=1 iopt_snf=3 iopt_tbot=1 idveg=5 iopt_crs=2 iopt_btr=2 iopt_inf=2" "iopt_run=5 idveg=3 scalars=1" \
           "iopt_run=2 iopt_btr=3 iopt_rad=2 iopt_snf=2 scalars=1" "idveg=4 iopt_run=3 iopt_inf=1 iopt_frz=2 scalars=1"; do
    timeout 1200 python tools/fuzz_parity.py gpu $seeds $cols $o 2>&1 | grep "^gpu\|DIFFER\|Error\|Traceback" | head -4 | tee -a $O/fuzz.log
  done
  timeout 900 python tools/config5_run.py 720 360 96 4096 2>&1 | tail -3 | cut -c1-400 | tee -a $O/fuzz.log
  # the config-5 chain (cold start -> interpolate -> prepare -> step, sorted with the longitude band) under other option sets
  for o in "idveg=2 iopt_run=3 iopt_stc=2 iopt_sfc=2 iopt_frz=2" "iopt_rad=1 iopt_alb=1 iopt_snf=3 iopt_tbot=1 idveg=5 iopt_crs=2 iopt_btr=2 iopt_inf=2" \
           "iopt_run=2 iopt_btr=3 iopt_rad=2 iopt_snf=2" "idveg=4 iopt_run=4"; do
    timeout 600 python tools/config5_run.py 720 360 96 8192 $o 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['options'], d['sample_bit_identical'], d['checkpoints'], '%.3g' % d['column_steps_per_s'])" | tee -a $O/fuzz.log
  done
  ;;
fuzzopts)         # fuzzopts [NSETS [SEED [SEEDS COLUMNS [key=value ...]]]]: random OPTION SETS (every OPT_* drawn from its supported range), each
                  # through hiprtc; further keys go to fuzz_parity.py (modis=1, steps=48)
  nsets=${1:-16}; seed=${2:-404}; seeds=${3:-3}; cols=${4:-8192}; shift 4 2>/dev/null; extra="$*"; : > $O/fuzzopts.log
  python3 - $nsets $seed > $O/sets.txt <<'PY'
import sys
import numpy as np
r = np.random.Generator(np.random.Philox(int(sys.argv[2])))
rng = dict(idveg=(1, 5), iopt_crs=(1, 2), iopt_btr=(1, 3), iopt_run=(1, 5), iopt_sfc=(1, 2), iopt_frz=(1, 2), iopt_inf=(1, 2), iopt_rad=(1, 3),
           iopt_alb=(1, 2), iopt_snf=(1, 3), iopt_tbot=(1, 2), iopt_stc=(1, 2))
for n in range(int(sys.argv[1])):
    print(" ".join("%s=%d" % (k, r.integers(lo, hi + 1)) for k, (lo, hi) in rng.items()))
PY
  while read o; do
    timeout 900 python tools/fuzz_parity.py gpu $seeds $cols $o scalars=1 $extra 2>&1 | grep "^gpu\|DIFFER\|Error\|Traceback" | head -4 | tee -a $O/fuzzopts.log
  done < $O/sets.txt
  ;;
fuzzopts5)        # fuzzopts5 [NSETS [SEED]]: the config-5 chain (cold start -> interpolate -> prepare -> step; sorted, class-range kernels through
                  # hiprtc) under random option sets (OPT_RUN 1..4: the global grid carries no MMF planes), sample vs the oracle
  nsets=${1:-8}; seed=${2:-515}; : > $O/fuzzopts5.log
  python3 - $nsets $seed > $O/sets.txt <<'PY'
import sys
import numpy as np
r = np.random.Generator(np.random.Philox(int(sys.argv[2])))
rng = dict(idveg=(1, 5), iopt_crs=(1, 2), iopt_btr=(1, 3), iopt_run=(1, 4), iopt_sfc=(1, 2), iopt_frz=(1, 2), iopt_inf=(1, 2), iopt_rad=(1, 3),
           iopt_alb=(1, 2), iopt_snf=(1, 3), iopt_tbot=(1, 2), iopt_stc=(1, 2))
for n in range(int(sys.argv[1])):
    print(" ".join("%s=%d" % (k, r.integers(lo, hi + 1)) for k, (lo, hi) in rng.items()))
PY
  while read o; do
    timeout 900 python tools/config5_run.py 720 360 72 8192 $o 2>&1 | tail -1 | python3 -c "import sys,json
try:
    d=json.loads(sys.stdin.read())
    if d.get('device_fatal'): print(d['options'], 'STOPPED BY THE MODEL', d['device_fatal'], 'oracle stops at the same column, step and code:', d['oracle_same_fatal'])
    else: print(d['options'], 'sample_bit_identical', d['sample_bit_identical'], d['checkpoints'], 'status_max', d.get('device_status_max'))
except Exception as e: print('FAILED', '$o', e)" | tee -a $O/fuzzopts5.log
  done < $O/sets.txt
  ;;
spread5)          # round 6: canopy-loop lane use of config 5 under the two forcing generators (i.i.d. factors | spatially smooth factors), same column order
  # build first (dev container): python tools/build_variants.py cost=-DNMP_COST_RECORD
  for v in "" "--config5-smooth=1" "--config5-smooth=2"; do
    NMP_COST_SPREAD=1 NMP_LIB=$R/noahmp_amd/csrc/variants/lib_cost.so timeout 900 python bench.py --workload config5 $v --steps 24 --warmup 6 --no-cpu-baseline > $O/spread$v.json 2> $O/spread$v.err
    summarise $O/spread$v.json "config5 $v (cost-record build)"
    grep "^COSTSPREAD" $O/spread$v.err | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l[len('COSTSPREAD '):]); c = d['orders'][0]
    print('   hour %2d: canopy iterations mean %.2f / wave-max mean %.2f  canopy-loop lane use %.3f  bisection lane use %.3f  waves without canopy %.3f' % (
        d['hour'], c['canopy_iterations_mean'], c['canopy_iterations_wave_max_mean'], c['canopy_loop_lane_use'], c['bisection_lane_use'], c['waves_without_canopy']))"
  done
  ;;
vegcost)          # land-kernel cost per vegetation category (tools/veg_cost.py [usgs|modis] [ni nj] [dveg]) -> gpurun_out/exp_vegcost/
  timeout 1500 python tools/veg_cost.py "$@" > $O/veg_cost_${1:-usgs}_d${4:-3}.json 2> $O/veg_cost_${1:-usgs}_d${4:-3}.err; echo "rc=$?"
  cat $O/veg_cost_${1:-usgs}_d${4:-3}.err | tail -30; tail -c 600 $O/veg_cost_${1:-usgs}_d${4:-3}.json
  ;;
stage)            # pageable host arrays through the engine's bounce buffers: copy threads at 1 M columns, the 7 M-column figure, then the churn fault hunt
  NMP_STAGE_NI=1024 NMP_STAGE_NJ=1024 timeout 600 python tools/stage_exp.py ${@:-1 4 8 16} 2>&1 | grep "^copy threads" | tee $O/stage.log
  timeout 900 python tools/stage_exp.py 8 2>&1 | grep "^copy threads" | tee -a $O/stage.log
  bad=0
  for i in $(seq 1 ${NMP_CHURN_PROCS:-5}); do
    env -u GPU_PINNED_MIN_XFER_SIZE NMP_STAGE_CHILD=1 NMP_STAGE_CHURN=30 NMP_STAGE_NI=1024 NMP_STAGE_NJ=1024 timeout 300 python tools/stage_exp.py > $O/churn$i.log 2>&1 || bad=$((bad+1))
    tail -1 $O/churn$i.log | cut -c1-200
  done
  echo "stage churn: $bad of ${NMP_CHURN_PROCS:-5} processes failed"
  ;;
profile)
  TAG=${1:-r05}
  P=$R/gpurun_out/prof; rm -rf $P; mkdir -p $P
  python3 $R/bench.py --no-cpu-baseline 2> $P/bench_plain.err | tail -1 > $P/bench_plain.json
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d $P/trace -o bench --output-format csv -- python3 $R/bench.py $QUIET > $P/bench_trace.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$(echo $c | tr 'A-Z' 'a-z' | cut -d_ -f1)
    rocprofv3 --pmc $c -d $P/$d -o bench --output-format csv -- python3 $R/bench.py $QUIET --steps 24 --warmup 0 > $P/bench_$d.log 2>&1
  done
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $P/sq -o bench --output-format csv -- python3 $R/bench.py $QUIET --steps 24 --warmup 0 > $P/bench_sq.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY -d $P/sq2 -o bench --output-format csv -- python3 $R/bench.py $QUIET --steps 24 --warmup 0 > $P/bench_sq2.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_WAIT_INST_LDS -d $P/sq3 -o bench --output-format csv -- python3 $R/bench.py $QUIET --steps 24 --warmup 0 > $P/bench_sq3.log 2>&1
  # config 4 (plane moves around WTABLE_mmf_noahmp, groundwater kernels inside a run) and config 5: kernel traces; config 5: counters
  rocprofv3 --kernel-trace --stats -d $P/trace4 -o bench --output-format csv -- python3 $R/bench.py --workload config4 --no-cpu-baseline > $P/bench_trace4.log 2>&1
  rocprofv3 --kernel-trace --stats -d $P/trace5 -o bench --output-format csv -- python3 $R/bench.py --workload config5 --no-cpu-baseline > $P/bench_trace5.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $P/fetch5 -o bench --output-format csv -- python3 $R/bench.py --workload config5 --no-cpu-baseline --steps 24 --warmup 0 > $P/bench_fetch5.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $P/write5 -o bench --output-format csv -- python3 $R/bench.py --workload config5 --no-cpu-baseline --steps 24 --warmup 0 > $P/bench_write5.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $P/sq5 -o bench --output-format csv -- python3 $R/bench.py --workload config5 --no-cpu-baseline --steps 24 --warmup 0 > $P/bench_sq5.log 2>&1
  # MMF groundwater kernels at the config-4 grid
  rocprofv3 --kernel-trace --stats -d $P/gw -o gw --output-format csv -- python3 $R/tools/gw_check.py perf > $P/gw_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $P/gw_fetch -o gw --output-format csv -- python3 $R/tools/gw_check.py perf > $P/gw_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $P/gw_write -o gw --output-format csv -- python3 $R/tools/gw_check.py perf > $P/gw_write.log 2>&1
  ls -R $P | head -80
  ;;
*)
  sed -n 2,32p "$0"
  ;;
esac

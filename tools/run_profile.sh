#!/bin/bash
# Collect the rocprofv3 evidence for profiles/: run ON THE GPU BOX from the repo root, e.g.
#   gpurun --timeout 1500 -- 'bash tools/run_profile.sh'
# then, back in the dev container:  python tools/collect_profile.py r03
# Counters are collected in their own passes (never together with a trace), as the guide prescribes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --no-cpu-baseline 2> $O/bench_plain.err | tail -1 > $O/bench_plain.json
rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-scaling-reference > $O/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-scaling-reference --steps 24 --warmup 0 > $O/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-scaling-reference --steps 24 --warmup 0 > $O/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $O/sq -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-scaling-reference --steps 24 --warmup 0 > $O/bench_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY -d $O/sq2 -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-scaling-reference --steps 24 --warmup 0 > $O/bench_sq2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_WAIT_INST_LDS -d $O/sq3 -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-scaling-reference --steps 24 --warmup 0 > $O/bench_sq3.log 2>&1
# config 4 (the plane exchanges around WTABLE_mmf_noahmp, the groundwater kernels inside a run) and config 5: kernel traces
rocprofv3 --kernel-trace --stats -d $O/trace4 -o bench --output-format csv -- python3 $R/bench.py --workload config4 --no-cpu-baseline > $O/bench_trace4.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace5 -o bench --output-format csv -- python3 $R/bench.py --workload config5 --no-cpu-baseline > $O/bench_trace5.log 2>&1
# MMF groundwater kernels at the config-4 grid
rocprofv3 --kernel-trace --stats -d $O/gw -o gw --output-format csv -- python3 $R/tools/gw_check.py perf > $O/gw_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/gw_fetch -o gw --output-format csv -- python3 $R/tools/gw_check.py perf > $O/gw_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/gw_write -o gw --output-format csv -- python3 $R/tools/gw_check.py perf > $O/gw_write.log 2>&1
ls -R $O | head -60

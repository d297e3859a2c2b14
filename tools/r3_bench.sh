#!/bin/bash
# round-3: the new bench tests + the default bench line (as the driver runs it) + the config-5 workload
O=gpurun_out/r3_bench; mkdir -p $O
timeout 1500 python -m pytest tests/test_multirank.py tests/test_divrc.py tests/test_async.py tests/test_host.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 15 $O/pytest.log
(time timeout 900 python bench.py --steps 20 --warmup 5) > $O/bench_default.json 2> $O/bench_default.err; tail -n 4 $O/bench_default.err
timeout 900 python bench.py --workload config5 --steps 48 --warmup 6 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
python - <<'PY'
import json
for f in ("bench_default", "bench_c5"):
    try:
        d = json.loads(open("gpurun_out/r3_bench/%s.json" % f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no json", e); continue
    r = d["roofline"]
    print(f, "value %.4g ms/step %.3f kernel ms %.3f day %s night %s 24h %s frac %.4f frac24 %s" % (d["value"], d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_day"], r["kernel_ms_night"], r["kernel_ms_24h_mean"], r["frac"], r["frac_24h_mean"]))
    for k in ("scaling_reference", "config5_reference"):
        if k in d: print("  ", k, "%.4g" % d[k]["value"], "ms/step %.3f" % d[k]["ms_per_step"])
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]; print("   cpu", "%.4g" % c["value"], c["cores"], c["host"], "x%.1f" % c["speedup_over_single_process"]); [print("     ", e) for e in c["sweep"]]
PY

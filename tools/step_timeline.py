"""Timeline of a bench step from a rocprofv3 kernel trace (`*_kernel_trace.csv`): for the steady-state steps (between consecutive launches
of the land kernel) the mean duration of every kernel, the idle gaps between consecutive kernels, and what runs beside the land kernel.
usage: step_timeline.py TRACE.csv [land-kernel substring]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
land = sys.argv[2] if len(sys.argv) > 2 else "noahmp_ranges_kernel"
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if land in r[2]]
if len(starts) < 12:
    sys.exit("too few land-kernel launches")
lo, hi = starts[len(starts) // 4], starts[-len(starts) // 4]            # the middle half of the run
sel = [s for s in starts if lo <= s <= hi]
period, dur, gap, cnt = [], defaultdict(float), 0.0, 0
idle_after = defaultdict(float)
for a, b in zip(sel[:-1], sel[1:]):
    step = rows[a:b]
    period.append(rows[b][0] - rows[a][0])
    busy_end = step[0][0]
    for s, e, name, q in step:
        short = name.replace("void ", "").replace("(anonymous namespace)::", "")
        short = short[:short.index("(")] if "(" in short else short
        short = short[:70]
        dur[short] += e - s
        if s > busy_end:
            idle_after[prev] += s - busy_end
            gap += s - busy_end
        if e > busy_end:
            busy_end, prev = e, short
    nxt = rows[b][0]
    if nxt > busy_end:
        idle_after[prev] += nxt - busy_end
        gap += nxt - busy_end
    cnt += 1
print("steps analysed: %d   mean step period %.1f us   idle (no kernel running) %.1f us per step" % (cnt, sum(period) / cnt / 1e3, gap / cnt / 1e3))
for k, v in sorted(dur.items(), key=lambda kv: -kv[1]):
    print("  %-62s %8.1f us per step   idle after it %6.1f us" % (k, v / cnt / 1e3, idle_after.get(k, 0.0) / cnt / 1e3))

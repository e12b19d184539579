"""How much of a step does the device spend with 0 / 1 / 2 / ... kernels in flight?

    python tools/trace_concurrency.py <kernel_trace.csv> <profile_step log with PROFILE_STEPS_BEGIN/END> <steps> [out.json]

From a rocprofv3 --kernel-trace table of tools/profile_step.py: inside the timed window, a sweep over the dispatches' begin / end stamps
gives the wall time, the sum of the kernel durations, the time with k kernels executing at once, and the idle gaps.  `sum / wall` near 1
with little idle time means the step is a serial chain of kernels (shorter or fewer kernels is the lever); a large idle share means
dispatch / dependency latency; `>= 2 in flight` says how much the stream lanes overlap."""
import csv
import json
import re
import sys

trace, log, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
out = sys.argv[4] if len(sys.argv) > 4 else None
text = open(log).read()
b = int(re.search(r"PROFILE_STEPS_BEGIN \d+ (\d+)", text).group(1))
e = int(re.search(r"PROFILE_STEPS_END \d+ (\d+)", text).group(1))
ev = []
n = 0
total = 0
with open(trace) as f:
    for r in csv.DictReader(f):
        s, t = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s < b or s > e:
            continue
        ev.append((s, 1))
        ev.append((t, -1))
        n += 1
        total += t - s
ev.sort()
level, last = 0, ev[0][0]
at = {}
gaps = []
for ts, d in ev:
    if ts > last:
        at[level] = at.get(level, 0) + ts - last
        if level == 0:
            gaps.append(ts - last)
    level += d
    last = ts
wall = ev[-1][0] - ev[0][0]
gaps.sort()
res = {"steps": steps, "launches_per_step": n / steps, "wall_ms_per_step": wall / 1e6 / steps, "sum_kernel_ms_per_step": total / 1e6 / steps,
       "ms_per_step_with_k_kernels_in_flight": {str(k): v / 1e6 / steps for k, v in sorted(at.items())},
       "idle_gaps_per_step": len(gaps) / steps, "median_idle_gap_us": gaps[len(gaps) // 2] / 1e3 if gaps else 0.0,
       "p90_idle_gap_us": gaps[int(len(gaps) * 0.9)] / 1e3 if gaps else 0.0,
       "host_window_ms_per_step": (e - b) / 1e6 / steps}
print(json.dumps(res, indent=1))
if out:
    with open(out, "w") as f:
        json.dump(res, f, indent=1)

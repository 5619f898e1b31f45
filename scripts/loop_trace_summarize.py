"""rocprofv3 kernel trace of scripts/perf_loop_q1.py -> how much of the loop's time has the render and the matcher in flight TOGETHER.

    python scripts/loop_trace_summarize.py <kernel_trace.csv> [out.json]

The timed loop is the last group of kernels behind an idle gap of >= 20 ms.  Reported per query (window between the first and the last
ray-generation launch of the loop, divided by the queries in it): wall, summed kernel time, time with a NeRF kernel in flight, time with a
matcher kernel in flight, time with BOTH, time with nothing in flight."""
import csv
import json
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
groups, cur = [], []
for s, e, k in rows:
    if cur and s - max(x[1] for x in cur[-8:]) > 20e6:
        groups.append(cur)
        cur = []
    cur.append((s, e, k))
groups.append(cur)
loop = max(groups[-2:], key=len) if len(groups) > 1 else groups[-1]
starts = [s for s, e, k in loop if "raygen" in k]
t0, t1, nq = starts[2], starts[-3], len(starts) - 5  # a window of whole queries away from fill and drain
win = [(max(s, t0), min(e, t1), k) for s, e, k in loop if e > t0 and s < t1]
is_nerf = lambda k: any(t in k for t in ("nerf_fwd", "raygen", "far_fallback", "sample_coarse", "resample", "unnormalize", "distribution_elementwise"))
ev = []
for s, e, k in win:
    c = 0 if is_nerf(k) else 1
    ev.append((s, 1, c))
    ev.append((e, -1, c))
ev.sort()
live = [0, 0]
acc = dict(nerf_only=0, matcher_only=0, both=0, idle=0)
prev = t0
for t, d, c in ev:
    dt = t - prev
    key = "both" if live[0] and live[1] else "nerf_only" if live[0] else "matcher_only" if live[1] else "idle"
    acc[key] += dt
    prev = t
    live[c] += d
acc["idle"] += t1 - prev
out = dict(queries=nq, wall_ms_per_query=(t1 - t0) / nq * 1e-6, kernel_ms_per_query=sum(e - s for s, e, _ in win) / nq * 1e-6,
           kernels_per_query=len(win) / nq, **{f"{k}_ms_per_query": v / nq * 1e-6 for k, v in acc.items()})
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)

"""Cut a rocprofv3 kernel trace of scripts/perf_latency_q1.py (run with NM_LAT_GAP_MS=20) into steps at the idle gaps and print, for the
timed steps: kernels per step, summed kernel duration, first-start-to-last-end span, and the per-kernel totals of a median step.

    python scripts/latency_trace_summarize.py <kernel_trace.csv> <n timed steps> [gap_ms=10] [out.json]"""
import csv
import json
import statistics
import sys

path, n = sys.argv[1], int(sys.argv[2])
gap = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 10e6
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
steps, cur = [], []
for s, e, k in rows:
    if cur and s - cur[-1][1] > gap:
        steps.append(cur)
        cur = []
    cur.append((s, e, k))
if cur:
    steps.append(cur)
# the timed steps are the n consecutive gap-separated groups before the (back-to-back, un-gapped) proxy pass at the end
timed = steps[-(n + 1):-1] if len(steps) > n else steps
ksum = [sum(e - s for s, e, _ in st) * 1e-6 for st in timed]
span = [(st[-1][1] - st[0][0]) * 1e-6 for st in timed]
cnt = [len(st) for st in timed]
mid = sorted(range(len(timed)), key=lambda i: ksum[i])[len(timed) // 2]
per = {}
for s, e, k in timed[mid]:
    name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    c = per.setdefault(name, [0, 0.0])
    c[0] += 1
    c[1] += (e - s) * 1e-6
out = dict(steps=len(timed), kernels_per_step=statistics.median(cnt), kernel_ms=statistics.median(ksum), span_ms=statistics.median(span),
           per_kernel={k: dict(launches=v[0], ms=round(v[1], 4)) for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])})
print(json.dumps({k: v for k, v in out.items() if k != "per_kernel"}))
for k, v in out["per_kernel"].items():
    print(f"  {k:62s} x{v['launches']:3d} {v['ms']:8.4f} ms")
if len(sys.argv) > 4:
    json.dump(out, open(sys.argv[4], "w"), indent=1)
if len(sys.argv) > 5:  # timeline of the median step: start offset, duration, idle gap in front (us)
    with open(sys.argv[5], "w") as f:
        t0, prev = timed[mid][0][0], None
        for s, e, k in timed[mid]:
            name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
            f.write(f"{(s - t0) * 1e-3:9.1f} us  dur {(e - s) * 1e-3:8.1f}  gap {((s - prev) * 1e-3 if prev else 0):7.1f}  {name}\n")
            prev = e

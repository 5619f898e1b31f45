"""GPU idle fraction from a rocprofv3 --kernel-trace CSV: 1 - (union of kernel intervals) / (last end - first start), over the
last `frac` of the trace (the timed regions of bench.py come last).  Usage: python scripts/gpu_idle.py <kernel_trace.csv> [frac]"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = rows[int(len(rows) * (1 - frac)):]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for s, e, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = rows[-1][1] - rows[0][0]
print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  idle {100 * (1 - busy / span):.1f} %")

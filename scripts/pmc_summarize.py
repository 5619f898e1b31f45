"""Summarise the rocprofv3 --pmc passes of scripts/pmc_collect.sh into profiles/<name>.json.

  python scripts/pmc_summarize.py gpurun_out/pmc_bf16x3 nerf_fwd_bf16x3_kernel profiles/r1_pmc_nerf_fwd_bf16x3.json

Per-launch means over the launches of the kernel seen in each pass (the first launch of a process is dropped as warm-up).
HBM-side bytes follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1 KB per
count as reported by rocprofv3 (x1024) and FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (doubled here);
WRITE_SIZE is taken as reported (uncalibrated)."""
import csv, glob, json, sys
from collections import defaultdict

src, kernel, dst = sys.argv[1], sys.argv[2], sys.argv[3]
R, S = 4800, 64
counters, durations = {}, {}
for f in sorted(glob.glob(f"{src}/g*/**/*counter_collection.csv", recursive=True)):
    per_disp = defaultdict(dict)
    for row in csv.DictReader(open(f)):
        if kernel not in row["Kernel_Name"]:
            continue
        per_disp[int(row["Dispatch_Id"])][row["Counter_Name"]] = per_disp[int(row["Dispatch_Id"])].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    disp = sorted(per_disp)[1:]
    names = sorted({n for d in disp for n in per_disp[d]})
    for n in names:
        vals = [per_disp[d][n] for d in disp if n in per_disp[d]]
        counters[n] = sum(vals) / len(vals)
    kt = glob.glob(f.replace("counter_collection", "kernel_trace"))
    if kt:
        ds = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(kt[0])) if kernel in r["Kernel_Name"]][1:]
        durations["_".join(names)[:40]] = sum(ds) / len(ds)
d = {}
if "FETCH_SIZE" in counters:
    d["hbm_fetch_bytes"] = counters["FETCH_SIZE"] * 1024 * 2
if "WRITE_SIZE" in counters:
    d["hbm_write_bytes"] = counters["WRITE_SIZE"] * 1024
if "hbm_fetch_bytes" in d and "hbm_write_bytes" in d:
    d["traffic_bytes"] = d["hbm_fetch_bytes"] + d["hbm_write_bytes"]
# algorithmic bytes per launch: rays 48 B + t (S+1)*4 B in; weights S*4, feat 1024, pts/rgb/depth/acc 32 B out; blob once
d["algorithmic_bytes"] = int(sys.argv[4]) if len(sys.argv) > 4 else R * (48 + (S + 1) * 4 + S * 4 + 1024 + 32) + 2621440
c = counters
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
    # GRBM_GUI_ACTIVE is reported per XCD and summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs
    active = c["GRBM_GUI_ACTIVE"] / 8
    d["mfma_busy_fraction"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (active * 1024)  # 256 CUs x 4 SIMDs
    dur = [v for k, v in durations.items() if "GRBM" in k or "SQ_BUSY" in k]
    if dur:
        d["effective_clock_ghz"] = active / (dur[0] * 1e6)
if "TCC_HIT_sum" in c:
    d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
    d["wait_any_fraction_of_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    d["wait_inst_any_fraction_of_wave_cycles"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    d["valu_active_fraction_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"]
out = {
    "kernel": kernel,
    "workload": (sys.argv[5] if len(sys.argv) > 5 else f"R={R} rays x S={S} samples per launch (scripts/pmc_render.py, NM_PRECISION selects the kernel)"),
    "command": "scripts/pmc_collect.sh <tag>  (rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 scripts/pmc_render.py, one pass per group)",
    "counters_per_launch_mean": counters,
    "duration_ms_under_profiler": durations,
    "derived": d,
}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(d, indent=1))

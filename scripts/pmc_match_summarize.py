"""Per-kernel means of the rocprofv3 --pmc passes of scripts/pmc_match.sh -> profiles/r1_pmc_matcher.json.
Only the launches at the bench shapes are averaged (largest grid of each kernel)."""
import csv, glob, json, sys
from collections import defaultdict

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_match"
kernels = ["attn32_v2_kernel", "gemm_bf16x3_kernel", "row_select_kernel", "col_confmax_kernel", "row_stats_kernel", "col_stats_partial_kernel", "kv_presplit_kernel"]
out = {}
for f in sorted(glob.glob(f"{src}/g*/**/*counter_collection.csv", recursive=True)):
    per = defaultdict(lambda: defaultdict(dict))
    grid = {}
    for row in csv.DictReader(open(f)):
        for k in kernels:
            if k in row["Kernel_Name"]:
                d = int(row["Dispatch_Id"])
                per[k][d][row["Counter_Name"]] = per[k][d].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                grid[(k, d)] = int(row["Grid_Size"])
    for k, disp in per.items():
        gmax = max(grid[(k, d)] for d in disp)
        sel = [d for d in disp if grid[(k, d)] == gmax]
        o = out.setdefault(k, {"grid_size": gmax, "launches_averaged": len(sel), "counters_per_launch_mean": {}})
        for n in {n for d in sel for n in disp[d]}:
            v = [disp[d][n] for d in sel if n in disp[d]]
            o["counters_per_launch_mean"][n] = sum(v) / len(v)
for k, o in out.items():
    c = o["counters_per_launch_mean"]
    d = o["derived"] = {}
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        d["mfma_busy_fraction"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
        d["wait_any_fraction_of_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        d["wait_inst_any_fraction_of_wave_cycles"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
        if "SQ_ACTIVE_INST_VALU" in c:
            d["valu_active_fraction_of_wave_cycles"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in c:
        d["fabric_fetch_bytes"] = c["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in c:
        d["fabric_write_bytes"] = c["WRITE_SIZE"] * 1024
json.dump({"command": "scripts/pmc_match.sh (rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 scripts/perf_match.py)",
           "kernels": out}, open("profiles/r1_pmc_matcher.json", "w"), indent=1)
for k, o in out.items():
    print(k, o["grid_size"], {a: round(b, 3) if b < 10 else int(b) for a, b in o["derived"].items()})

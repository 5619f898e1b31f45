#!/bin/bash
# L2 <-> fabric bytes of EVERY kernel of a bench run (run ON the GPU box from the repo root): two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel trace only),
# aggregated per kernel name -> gpurun_out/r6_traffic_by_kernel.txt.  A screen for kernels that move bytes their arithmetic does not explain (scratch, re-fetches).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; R=$PWD; O=gpurun_out/pmc_all; rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/$c -o all -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/$c.log 2>&1 )
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for c, col in (("FETCH_SIZE", 1), ("WRITE_SIZE", 2)):
    fs = glob.glob("$O/%s/**/*counter_collection.csv" % c, recursive=True)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != c:
                continue
            k = r["Kernel_Name"][:90]
            a = agg[k]
            if col == 1:
                a[0] += 1
            a[col] += float(r["Counter_Value"])
rows = sorted(agg.items(), key=lambda kv: -(kv[1][1] * 64 + kv[1][2] * 64))
with open("gpurun_out/r6_traffic_by_kernel.txt", "w") as out:
    out.write("# kernel, launches, mean MB per launch: fetched (FETCH_SIZE x 1024 x 2, the gfx950 correction of scripts/pmc_summarize.py) and written (WRITE_SIZE x 1024)\n")
    for k, (n, f, w) in rows[:70]:
        n = max(n, 1)
        out.write(f"{k:90s} x{n:5d}  fetched {f / n * 2048 / 1e6:10.2f} MB  written {w / n * 1024 / 1e6:10.2f} MB\n")
print(open("gpurun_out/r6_traffic_by_kernel.txt").read()[:6000])
PY
rm -rf $O

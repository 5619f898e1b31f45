#!/bin/bash
# Round 5: single-query latency of a localisation step (run ON the GPU box from the repo root): plain runs, then kernel traces cut into steps.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$PWD
TAG=${1:-base}
mkdir -p gpurun_out/lat_$TAG
for kind in c2f coarse; do
  python3 scripts/perf_latency_q1.py $kind 30 1 > gpurun_out/lat_$TAG/${kind}_q1.log 2>&1
  ( cd /tmp && NM_LAT_GAP_MS=20 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/lat_$TAG/trace_$kind -o q1 -- python3 $R/scripts/perf_latency_q1.py $kind 30 1 > $R/gpurun_out/lat_$TAG/${kind}_q1_traced.log 2>&1 )
  f=$(find gpurun_out/lat_$TAG/trace_$kind -name "*kernel_trace.csv" | head -1)
  python3 scripts/latency_trace_summarize.py $f 30 10 gpurun_out/lat_$TAG/${kind}_q1_trace.json gpurun_out/lat_$TAG/${kind}_q1_timeline.txt > gpurun_out/lat_$TAG/${kind}_q1_trace.txt 2>&1
  rm -rf gpurun_out/lat_$TAG/trace_$kind
done
python3 scripts/perf_latency_q1.py c2f 10 16 > gpurun_out/lat_$TAG/c2f_q16.log 2>&1
tail -n 40 gpurun_out/lat_$TAG/c2f_q1.log gpurun_out/lat_$TAG/c2f_q1_trace.txt gpurun_out/lat_$TAG/coarse_q1.log gpurun_out/lat_$TAG/coarse_q1_trace.txt
head -3 gpurun_out/lat_$TAG/c2f_q16.log

#!/bin/bash
# Round-6 profile collection (run ON the GPU box from the repo root): kernel-trace stats of the bench (region A alone, then every leg), the
# dominant kernel's HBM-side counters AT THE BENCH'S LAUNCH SIZE (16 queries per launch), kernel traces of the one-query loop on one stream
# and as shipped (render on five XCDs beside the matcher on three): kernel time and idle time per query (the tracer serialises the two queues, so the
# two-stream trace shows the kernels' own durations on their partitions, not their overlap -- the untraced timings show that).
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/prof_r6
mkdir -p $O
# the loop first, untraced, each setting in a process of its own (a PMC pass leaves the box's clocks / queues in a state that showed up as 3.7 ms per query once)
for mode in 1 0; do
  tag=$([ $mode = 1 ] && echo one_stream || echo two_streams)
  NM_LOOP_ONE_STREAM=$mode python3 scripts/perf_loop_q1.py 64 > $O/loop_q1_$tag.log 2>&1
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/regionA -o regionA -- python3 $R/bench.py --steps 10 --warmup 2 --no-extra-legs --no-match --no-cpu-baseline > $R/$O/regionA.log 2>&1 )
cp $(find $O/regionA -name "*kernel_stats.csv" | head -1) $O/r6_regionA_kernel_stats.csv
PMC_SCRIPT=pmc_render_q16.py bash scripts/pmc_collect.sh fp16x3_q16_r6 > $O/pmc_fp16x3_q16.log 2>&1
python scripts/pmc_summarize.py gpurun_out/pmc_fp16x3_q16_r6 nerf_fwd_fp16x3_kernel $O/r6_pmc_nerf_fwd_fp16x3_q16.json $((16 * (4800 * (48 + 65 * 4 + 64 * 4 + 1024 + 32)) + 2621440)) "16 queries x 4800 rays x 64 samples per launch, coarse and fine launches averaged (scripts/pmc_render_q16.py = one region-A step of bench.py)" > /dev/null
for mode in 0 1; do
  tag=$([ $mode = 1 ] && echo one_stream || echo two_streams)
  ( cd /tmp && NM_LOOP_ONE_STREAM=$mode timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_$tag -o loop -- python3 $R/scripts/perf_loop_q1.py 40 > $R/$O/loop_q1_${tag}_traced.log 2>&1 )
  f=$(find $O/trace_$tag -name "*kernel_trace.csv" | head -1)
  python3 scripts/loop_trace_summarize.py $f $O/r6_loop_q1_$tag.json > $O/loop_q1_${tag}_trace.txt 2>&1
  rm -rf $O/trace_$tag
done
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/bench -o bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/$O/bench.log 2>&1 )
cp $(find $O/bench -name "*kernel_stats.csv" | head -1) $O/r6_bench_kernel_stats.csv
cat $O/loop_q1_*.log $O/loop_q1_*_trace.txt
python -c "import json; print(json.load(open('$O/r6_pmc_nerf_fwd_fp16x3_q16.json'))['derived'])"
head -5 $O/r6_regionA_kernel_stats.csv | cut -c1-220
find gpurun_out/pmc_fp16x3_q16_r6 $O/regionA $O/bench -name "*.csv" -size +2M -delete

#!/bin/bash
# Round-5 profile collection (run ON the GPU box from the repo root).  Kernel-trace stats of the bench (region A alone, then every leg),
# PMC passes (own runs, --pmc never combined with other trace domains) of the three kernels whose fractions the bench line quotes -- on
# the FINAL kernels of the round --, the NM_ACC_TAKE A/B that round 4 left without a log, and the one-query latency traces.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/prof_r5
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r5/regionA -o regionA -- python3 $R/bench.py --steps 10 --warmup 2 --no-extra-legs --no-match --no-cpu-baseline > $R/gpurun_out/prof_r5/regionA.log 2>&1 )
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r5/bench -o bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_r5/bench.log 2>&1 )
cp $(find gpurun_out/prof_r5/regionA -name "*kernel_stats.csv" | head -1) gpurun_out/prof_r5/r5_regionA_kernel_stats.csv
cp $(find gpurun_out/prof_r5/bench -name "*kernel_stats.csv" | head -1) gpurun_out/prof_r5/r5_bench_kernel_stats.csv
NM_PRECISION=fp16x3 bash scripts/pmc_collect.sh fp16x3_r5 > gpurun_out/prof_r5/pmc_fp16x3.log 2>&1
PMC_SCRIPT=pmc_attention.py bash scripts/pmc_collect.sh attn_v3_r5 > gpurun_out/prof_r5/pmc_attn.log 2>&1
PMC_SCRIPT=pmc_mini.py bash scripts/pmc_collect.sh mini_r5 > gpurun_out/prof_r5/pmc_mini.log 2>&1
python scripts/pmc_summarize.py gpurun_out/pmc_fp16x3_r5 nerf_fwd_fp16x3_kernel gpurun_out/prof_r5/r5_pmc_nerf_fwd_fp16x3.json > /dev/null
python scripts/pmc_summarize.py gpurun_out/pmc_attn_v3_r5 attn32_v3_kernel gpurun_out/prof_r5/r5_pmc_attn32_v3.json 1 "32 sequences x 8 heads x 4800 x 4800 per launch (scripts/pmc_attention.py)" > /dev/null
python scripts/pmc_summarize.py gpurun_out/pmc_mini_r5 "match_tile_kernel<1" gpurun_out/prof_r5/r5_pmc_match_tile1.json 1 "16 pairs of 4800 x 4800 tokens per launch (scripts/pmc_mini.py)" > /dev/null
python scripts/pmc_summarize.py gpurun_out/pmc_mini_r5 "match_tile_kernel<2" gpurun_out/prof_r5/r5_pmc_match_tile2.json 1 "16 pairs of 4800 x 4800 tokens per launch (scripts/pmc_mini.py)" > /dev/null
{ echo "# NM_ACC_TAKE A/B (round 4's last kernel step, logged in round 5): scripts/variants/nerf_study_switches_r4.patch applied, -DNM_ACC_TAKE=0 vs 1"; echo "# NM_PRECISION=fp16x3 python scripts/ab_nerf.py acctake0 acctake1   (ms per 4 x 4800 x 64 launch: all heads / no feature output / density only; two interleaved rounds)"; NM_PRECISION=fp16x3 python scripts/ab_nerf.py acctake0 acctake1; } > gpurun_out/prof_r5/r5_ab_acc_take.log 2>&1
bash scripts/r5_latency.sh final > /dev/null 2>&1
cp gpurun_out/lat_final/c2f_q1_trace.json gpurun_out/prof_r5/r5_latency_q1_c2f.json
cp gpurun_out/lat_final/coarse_q1_trace.json gpurun_out/prof_r5/r5_latency_q1_coarse.json
cp gpurun_out/lat_final/c2f_q1_timeline.txt gpurun_out/prof_r5/r5_latency_q1_timeline_after.txt
for f in gpurun_out/prof_r5/r5_pmc_*.json; do echo $f; python -c "import json,sys; print(json.load(open('$f'))['derived'])"; done
cat gpurun_out/prof_r5/r5_ab_acc_take.log
head -6 gpurun_out/prof_r5/r5_regionA_kernel_stats.csv | cut -c1-200
# keep only the small summaries in the merge-back (raw counter CSVs are large)
find gpurun_out/pmc_fp16x3_r5 gpurun_out/pmc_attn_v3_r5 gpurun_out/pmc_mini_r5 gpurun_out/prof_r5/regionA gpurun_out/prof_r5/bench -name "*.csv" -size +2M -delete

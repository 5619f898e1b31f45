#!/bin/bash
# Round-4 profile collection (run ON the GPU box from the repo root): kernel-trace stats of the bench command, then PMC passes (own
# runs, --pmc never combined with other traces) for the three kernels whose fractions the bench line quotes.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r4 -o bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_r4_bench.log 2>&1 )
NM_PRECISION=fp16x3 bash scripts/pmc_collect.sh fp16x3_r4 > gpurun_out/pmc_fp16x3_r4.log 2>&1
PMC_SCRIPT=pmc_attention.py bash scripts/pmc_collect.sh attn_v3_r4 > gpurun_out/pmc_attn_v3_r4.log 2>&1
PMC_SCRIPT=pmc_mini.py bash scripts/pmc_collect.sh mini_r4 > gpurun_out/pmc_mini_r4.log 2>&1
python scripts/pmc_summarize.py gpurun_out/pmc_fp16x3_r4 nerf_fwd_fp16x3_kernel gpurun_out/r4_pmc_nerf_fwd_fp16x3.json > /dev/null
python scripts/pmc_summarize.py gpurun_out/pmc_attn_v3_r4 attn32_v3_kernel gpurun_out/r4_pmc_attn32_v3.json 1 "32 sequences x 8 heads x 4800 x 4800 per launch (scripts/pmc_attention.py)" > /dev/null
python scripts/pmc_summarize.py gpurun_out/pmc_mini_r4 "match_tile_kernel<1" gpurun_out/r4_pmc_match_tile1.json 1 "16 pairs of 4800 x 4800 tokens per launch (scripts/pmc_mini.py)" > /dev/null
python scripts/pmc_summarize.py gpurun_out/pmc_mini_r4 "match_tile_kernel<2" gpurun_out/r4_pmc_match_tile2.json 1 "16 pairs of 4800 x 4800 tokens per launch (scripts/pmc_mini.py)" > /dev/null
ls gpurun_out/prof_r4* | head; find gpurun_out/prof_r4 -name "*kernel_stats.csv" | head -2
for f in gpurun_out/r4_pmc_*.json; do echo $f; python -c "import json,sys; print(json.load(open('$f'))['derived'])"; done
# keep only the small summaries in the merge-back (raw counter CSVs are large)
find gpurun_out/pmc_fp16x3_r4 gpurun_out/pmc_attn_v3_r4 gpurun_out/pmc_mini_r4 -name "*.csv" -size +2M -delete

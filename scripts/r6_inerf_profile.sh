cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; R=$PWD; O=gpurun_out/prof_inerf; mkdir -p $O
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/a -o inerf -- python3 $R/scripts/perf_inerf_fused.py > $R/$O/inerf.log 2>&1 )
cp $(find $O/a -name "*kernel_stats.csv" | head -1) $O/r6_inerf_step_kernel_stats.csv
head -16 $O/r6_inerf_step_kernel_stats.csv | cut -c1-160
cat $O/inerf.log | tail -2
find $O -name "*.csv" -size +2M -delete

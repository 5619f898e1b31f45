cd $GRAFT_REPO_ROOT
V=nerfmatch_amd/lib/variants
{
echo "### shipped library"; python scripts/surface_seed_stats.py fp32 fp16x3 fp16x3:neutral bf16x3
echo "### nowscale (-DNM_NO_WSCALE: no operand scaled at all, RNE hi parts), neutral = the round-3 operands"; NERFMATCH_AMD_LIB=$V/lib_nowscale.so python scripts/surface_seed_stats.py fp16x3:neutral
[ -f $V/lib_rtz.so ] && { echo "### rtz (-DNM_HI_RNE=0: needs scripts/variants/nerf_study_switches_r4.patch applied), calibrated"; NERFMATCH_AMD_LIB=$V/lib_rtz.so python scripts/surface_seed_stats.py fp16x3; }
[ -f $V/lib_ipex.so ] && { echo "### ipex (-DNM_IPE_EXACT=1), calibrated"; NERFMATCH_AMD_LIB=$V/lib_ipex.so python scripts/surface_seed_stats.py fp16x3; }
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_surface_seed_stats.log

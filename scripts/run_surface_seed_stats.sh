cd $GRAFT_REPO_ROOT
V=nerfmatch_amd/lib/variants
{
echo "### base library"; python scripts/surface_seed_stats.py fp32 fp16x3 fp16x3:neutral bf16x3
echo "### r3like (-DNM_NO_WSCALE -DNM_HI_RNE=0), neutral = round 3 arithmetic"; NERFMATCH_AMD_LIB=$V/lib_r3like.so python scripts/surface_seed_stats.py fp16x3:neutral
echo "### rtz (-DNM_HI_RNE=0), calibrated"; NERFMATCH_AMD_LIB=$V/lib_rtz.so python scripts/surface_seed_stats.py fp16x3
echo "### ipex (-DNM_IPE_EXACT=1), calibrated"; NERFMATCH_AMD_LIB=$V/lib_ipex.so python scripts/surface_seed_stats.py fp16x3
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_surface_seed_stats.log

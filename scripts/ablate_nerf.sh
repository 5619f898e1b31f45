#!/bin/bash
# Timing-only ablations of the split NeRF kernel (NM_ABL bits, nerf_fwd_bf16.hip): builds the variants HERE (hipcc cross-compiles),
# runs them on the GPU box with scripts/ab_nerf.py:   NM_PRECISION=fp16x3 python scripts/ab_nerf.py base abl1 abl2 ...
set -e
cd "$(dirname "$0")/.."
NM_SRC=nerf_fwd_bf16 scripts/build_variants.sh "base:" "abl1:-DNM_ABL=1" "abl2:-DNM_ABL=2" "abl3:-DNM_ABL=3" "abl4:-DNM_ABL=4" "abl8:-DNM_ABL=8" "abl16:-DNM_ABL=16" "abl32:-DNM_ABL=32" "abl15:-DNM_ABL=15" "abl31:-DNM_ABL=31"

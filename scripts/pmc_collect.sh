#!/bin/bash
# rocprofv3 counter passes for the fused NeRF kernels at the bench shape (run ON the GPU box, from the repo root):
#   NM_PRECISION=bf16x3 scripts/pmc_collect.sh bf16x3
# One rocprofv3 invocation per counter group (--pmc never combined with sys/runtime traces), the program itself after
# `--`; CSVs land in gpurun_out/pmc_<tag>/<group>/ and scripts/pmc_summarize.py turns them into profiles/*.json.
set -u
TAG=${1:-fp32}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
GROUPS_=(
  "FETCH_SIZE"
  "WRITE_SIZE"
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"
  "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE"
  "TCC_HIT_sum TCC_MISS_sum"
)
i=0
for g in "${GROUPS_[@]}"; do
  d=$OUT/g$i
  rm -rf "$d"
  timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$d" -- python3 "$ROOT/scripts/${PMC_SCRIPT:-pmc_render.py}" > "$d.log" 2>&1
  echo "group $i ($g): rc=$?"
  i=$((i + 1))
done

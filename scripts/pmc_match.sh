#!/bin/bash
# rocprofv3 counter passes for the matcher kernels (attention v2, bf16x3 GEMM, match sweeps); run ON the GPU box from the
# repo root.  One invocation per counter group, program directly after `--`.  Output: gpurun_out/pmc_match/g*/
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_match
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
GROUPS_=(
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"
  "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE"
  "FETCH_SIZE"
  "WRITE_SIZE"
)
i=0
for g in "${GROUPS_[@]}"; do
  d=$OUT/g$i
  rm -rf "$d"
  timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$d" -- python3 "$ROOT/scripts/perf_match.py" > "$d.log" 2>&1
  echo "group $i ($g): rc=$?"
  i=$((i + 1))
done

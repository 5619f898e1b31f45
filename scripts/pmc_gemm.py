"""Workload for scripts/pmc_collect.sh (PMC_SCRIPT=pmc_gemm.py): one nm_linear_ex_bf16x3 shape, 8 launches.
GEMM_SHAPE="M,K,N,res" (default 153600,256,256,1)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from nerfmatch_amd import ops

M, K, N, res = (int(v) for v in os.environ.get("GEMM_SHAPE", "153600,256,256,1").split(","))
dev = torch.device("cuda:0")
ops.LINEAR_PRECISION = "bf16x3"
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
r = torch.randn(M, N, device=dev) if res else None
for _ in range(8):
    ops.linear(x, w, b, residual=r)
torch.cuda.synchronize()

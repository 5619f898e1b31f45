"""Workload for scripts/pmc_collect.sh (PMC_SCRIPT=pmc_attention_fp8.py): the e4m3 attention kernel at 32 x 8 heads x 4800 x 4800."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from nerfmatch_amd import ops

dev = torch.device("cuda:0")
ops.ATTENTION_PRECISION = "fp8"
B, L = 32, 4800
qkv = torch.randn(B * L, 768, device=dev)
for _ in range(6):
    ops.attention_fused(qkv, (0, 256), (256, 512), (512, 768), B, L, L, 8, 32 ** -0.5)
torch.cuda.synchronize()

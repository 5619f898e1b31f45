"""Per-shape time of nm_linear_ex_bf16x3 at the matcher's sizes against its two rooflines (HBM bytes, issued bf16 MFMA work)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from nerfmatch_amd import ops

dev = torch.device("cuda:0")
ops.LINEAR_PRECISION = "bf16x3"
for M in (76800, 153600):
    for K, N, res, act in ((256, 768, False, 0), (256, 256, True, 0), (256, 256, False, 2), (256, 512, False, 0), (352, 256, False, 0)):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev) if res else None
        for _ in range(3):
            ops.linear(x, w, b, residual=r, act=act)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.linear(x, w, b, residual=r, act=act)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        byts = 4 * (M * K + M * N * (2 if res else 1))
        fl = 2.0 * M * N * K * 3
        print(f"M={M} K={K} N={N} res={int(res)} act={act}: {us:7.1f} us  {byts / us / 1e6:5.2f} TB/s  {fl / us / 1e6:6.0f} TFLOP/s issued")

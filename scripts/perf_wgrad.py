"""dW = dy^T x at the shapes of a training step: the fp32-MFMA kernel against the split-bf16 one (ops.LINEAR_PRECISION).

    python scripts/perf_wgrad.py
"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from nerfmatch_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for M, N, K in [(7200, 256, 256), (7200, 768, 256), (7200, 256, 768), (14400, 256, 256), (100000, 128, 128), (100000, 384, 128), (4800, 4800, 256), (7200, 256, 352)]:
    dy, x = torch.randn(M, N, generator=g).to(dev), torch.randn(M, K, generator=g).to(dev)
    row = f"M={M:6d} N={N:4d} K={K:4d}:"
    for prec in ("fp32", "bf16x3"):
        ops.LINEAR_PRECISION = prec
        for _ in range(3):
            ops.linear_wgrad(dy, x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30
        e0.record()
        for _ in range(n):
            ops.linear_wgrad(dy, x)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        row += f"  {prec} {us:7.1f} us ({2.0 * M * N * K / us * 1e-6:6.1f} TF/s)"
    print(row, flush=True)

"""One-query shapes of the split-bf16 GEMM: back-to-back launches (the host stays ahead), time per launch by HIP events.
NM_GEMM_SMALL=0 python scripts/perf_gemm_small.py  -> the ring kernel on the same shapes."""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from nerfmatch_amd import ops

dev = torch.device("cuda:0")
ops.LINEAR_PRECISION = "bf16x3"
print("NM_GEMM_SMALL =", os.environ.get("NM_GEMM_SMALL", "1"))
for M in (4800, 9600, 3750, 150, 19200):
    for K, N in ((256, 256), (256, 128), (128, 128), (128, 384), (256, 768), (352, 256)):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        for _ in range(5):
            ops.linear(x, w, b)
        n = 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops.linear(x, w, b)
        e1.record()
        torch.cuda.synchronize()
        print(f"M={M:6d} K={K} N={N}: {e0.elapsed_time(e1) / n * 1e3:7.2f} us per launch")

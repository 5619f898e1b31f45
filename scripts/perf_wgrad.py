import sys, time, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nerfmatch_amd import ops
dev = torch.device("cuda:0")
for M, N, K in [(7200, 256, 256), (7200, 768, 256), (7200, 256, 352), (3600, 3600, 256), (28800, 256, 256)]:
    dy, x = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
    for _ in range(3): ops.linear_wgrad(dy, x)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.linear_wgrad(dy, x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"wgrad {M}x{N}x{K}: {us:.1f} us  {2*M*N*K/us/1e6:.1f} TFLOP/s")

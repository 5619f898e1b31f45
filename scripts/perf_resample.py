"""Time of nm_resample at the bench size (16 x 4800 rays x 64 + 1 fence posts) and at one query."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from nerfmatch_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for R in (76800, 4800):
    S = 64
    t = torch.sort(torch.rand(R, S + 1, generator=g), -1).values.to(dev).contiguous()
    w = torch.rand(R, S, generator=g).to(dev)
    jit = (torch.rand(R, S + 1, generator=g) * (1.0 / (S + 1) - 1.2e-7)).to(dev)
    for _ in range(3):
        out = ops.resample(t, w, jit, 0.01, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = ops.resample(t, w, jit, 0.01, True)
    e1.record(); torch.cuda.synchronize()
    print(f"R={R}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call; checksum {float(out.double().sum()):.10f}")

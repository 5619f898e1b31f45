"""The fused matcher alone, 16 pairs of 4800 x 4800 tokens per call (for rocprofv3 --kernel-trace --stats / --pmc)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from nerfmatch_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops.MATCH_PRECISION = "bf16x3"
T, P = int(sys.argv[1]) if len(sys.argv) > 1 else 4800, 16
g = torch.Generator().manual_seed(1)
im = torch.randn(P, T, 256, generator=g).to(dev); pt = torch.randn(P, T, 256, generator=g).to(dev)
for _ in range(6):
    r = ops.dual_softmax_match_batch(im, pt, 15.0, threshold=0.2, mutual=True, want_conf=False)
torch.cuda.synchronize()
print("matches per pair:", r["count"].float().mean().item())

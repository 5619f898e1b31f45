"""Times one c2f training step (forward + backward + AdamW) of the matcher head at full size on the GPU and prints the
per-kernel breakdown from torch's profiler-independent HIP events.  Usage: python scripts/perf_train.py [H W B]"""
import sys
import time

import numpy as np
import torch

import os

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nerfmatch_amd import ops, synth  # noqa: E402
from nerfmatch_amd.matcher import NeRFMatcherMS  # noqa: E402
from nerfmatch_amd.modules import PrecomputedBackbone  # noqa: E402

H, W, B = (int(a) for a in (sys.argv[1:4] + ["480", "480", "2"][len(sys.argv) - 1:]))
prec = sys.argv[4] if len(sys.argv) > 4 else "fp32"
dev = torch.device("cuda:0")
h, w = H // 8, W // 8
M = N = h * w
g = torch.Generator().manual_seed(0)
cfeat = torch.randn(B, 256, h, w, generator=g).to(dev)
ffeat = torch.randn(B, 128, H // 2, W // 2, generator=g).to(dev)
pt_feat = torch.relu(torch.randn(B, N, 256, generator=g)).to(dev)
pt3d = (torch.randn(B, N, 3, generator=g) * 2).to(dev)
conf_gt = torch.zeros(B, M, N, dtype=torch.bool)
for b in range(B):
    perm = torch.randperm(N, generator=g)
    conf_gt[b, torch.arange(M // 2), perm[: M // 2]] = True
conf_gt = conf_gt.to(dev)
ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
pt2d = (torch.stack([xs, ys], -1).reshape(1, -1, 2).float() * 8 + 4).repeat(B, 1, 1).to(dev)
pt2d_proj = (torch.rand(B, N, 2, generator=g) * torch.tensor([W, H])).to(dev)
model = NeRFMatcherMS(synth.matcher_config("c2f"))
model.load_state_dict(synth.matcher_state_dict("c2f", seed=0), strict=False)
model = model.to(dev)
model.backbone = PrecomputedBackbone((cfeat, ffeat), [256, 128])
ops.LINEAR_PRECISION = ops.ATTENTION_PRECISION = ops.MATCH_PRECISION = prec
opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
torch.set_grad_enabled(True)


def step():
    data = dict(image=torch.zeros(B, 3, 8, 8, device=dev), im_mask=torch.ones(B, M, dtype=torch.bool, device=dev),
                pt_mask=torch.ones(B, N, dtype=torch.bool, device=dev), pt3d=pt3d, pt2d=pt2d, conf_gt=conf_gt, pt2d_proj=pt2d_proj,
                pt_feat=pt_feat)
    m = model.forward_with_metrics(data, training=True)
    opt.zero_grad()
    m["loss"].backward()
    opt.step()
    return m


np.random.seed(0)
for _ in range(2):
    m = step()
torch.cuda.synchronize()
from nerfmatch_amd._lib import steady_gc  # noqa: E402

n = 5
with steady_gc():  # (what trainer.py and bench.py do: without it one step in a few pays a full pass of Python's cyclic collector, 20+ ms)
    t0 = time.perf_counter()
    for _ in range(n):
        m = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
print(f"train step {H}x{W} B={B} tokens={M} precision={prec}: {dt * 1e3:.1f} ms/step  ({B / dt:.1f} pairs/s)  loss {m['loss'].item():.4f}")

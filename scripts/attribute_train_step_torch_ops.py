"""Which autograd node / optimizer call the torch elementwise kernels of a training step belong to (same step as scripts/profile_train_step.py)."""

import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

import nerfmatch_amd
from nerfmatch_amd import _lib, latency, synth
from nerfmatch_amd.matcher import NeRFMatcherMS
from nerfmatch_amd.modules import PrecomputedBackbone

dev = torch.device("cuda:0")
Ht, Wt, Bt = 480, 480, 2
ht, wt = Ht // 8, Wt // 8
Mt = ht * wt
g = torch.Generator().manual_seed(0)
cft, fft = torch.randn(Bt, 256, ht, wt, generator=g).to(dev), torch.randn(Bt, 128, Ht // 2, Wt // 2, generator=g).to(dev)
ptft, p3t = torch.relu(torch.randn(Bt, Mt, 256, generator=g)).to(dev), (torch.randn(Bt, Mt, 3, generator=g) * 2).to(dev)
cgt = torch.zeros(Bt, Mt, Mt, dtype=torch.bool)
for b_ in range(Bt):
    cgt[b_, torch.arange(Mt // 2), torch.randperm(Mt, generator=g)[: Mt // 2]] = True
cgt = cgt.to(dev)
ys, xs = torch.meshgrid(torch.arange(ht), torch.arange(wt), indexing="ij")
p2t = (torch.stack([xs, ys], -1).reshape(1, -1, 2).float() * 8 + 4).repeat(Bt, 1, 1).to(dev)
p2p = (torch.rand(Bt, Mt, 2, generator=g) * torch.tensor([Wt, Ht])).to(dev)
mt = NeRFMatcherMS(synth.matcher_config("c2f"))
mt.load_state_dict(synth.matcher_state_dict("c2f", seed=0), strict=False)
mt = mt.to(dev)
mt.backbone = PrecomputedBackbone((cft, fft), [256, 128])
nerfmatch_amd.set_precision("bf16x3")
np.random.seed(0)
opt = torch.optim.AdamW(mt.parameters(), lr=1e-4)


def train_step():
    d_ = dict(image=torch.zeros(Bt, 3, 8, 8, device=dev), im_mask=torch.ones(Bt, Mt, dtype=torch.bool, device=dev),
              pt_mask=torch.ones(Bt, Mt, dtype=torch.bool, device=dev), pt3d=p3t, pt2d=p2t, conf_gt=cgt, pt2d_proj=p2p, pt_feat=ptft)
    m_ = mt.forward_with_metrics(d_, training=True)
    opt.zero_grad()
    m_["loss"].backward()
    opt.step()



from torch.profiler import ProfilerActivity, profile
with torch.enable_grad(), _lib.steady_gc():
    for _ in range(3):
        train_step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        train_step()
        torch.cuda.synchronize()
agg = {}
for ev in prof.events():
    if not ev.name.startswith("aten::"):
        continue
    dt = getattr(ev, "self_device_time_total", None)
    if dt is None:
        dt = getattr(ev, "self_cuda_time_total", 0)
    if dt <= 0:
        continue
    p, top = ev.cpu_parent, None
    chain = []
    while p is not None:
        chain.append(p.name)
        p = p.cpu_parent
    owner = next((c for c in chain if not c.startswith("aten::")), "(top level)")
    key = (owner[:60], ev.name)
    c = agg.setdefault(key, [0, 0.0])
    c[0] += 1
    c[1] += dt
tot = sum(v[1] for v in agg.values())
print(f"torch's own kernels in one step: {tot / 1e3:.3f} ms")
for (owner, name), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t:8.1f} us  x{n:3d}  {name:22s} under {owner}")

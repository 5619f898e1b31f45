"""Where the matcher's training step goes: native-call spans (HIP events around every C-ABI call, nerfmatch_amd.latency.TimedLib) of one
step of bench.py's `train_step_ms` leg -- forward_with_metrics(training) + backward + AdamW, B = 2 pairs of 3600 + 3600 tokens, bf16x3."""
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


with torch.enable_grad(), _lib.steady_gc():
    for _ in range(3):
        train_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        train_step()
    torch.cuda.synchronize()
    print(f"train step: {(time.perf_counter() - t0) / 6 * 1e3:.2f} ms")
    with latency.timed_lib() as tl:
        tl.spans = []
        t0 = time.perf_counter()
        for _ in range(4):
            train_step()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 4 * 1e3
        per = {}
        for name, e0, e1 in tl.spans:
            c = per.setdefault(name, [0, 0.0])
            c[0] += 1
            c[1] += e0.elapsed_time(e1)
    tot = sum(v[1] for v in per.values()) / 4
    print(f"with the timing proxy: {wall:.2f} ms per step, native-call spans {tot:.2f} ms, {sum(v[0] for v in per.values()) // 4} native calls")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:36s} x{v[0] // 4:3d} {v[1] / 4:8.3f} ms")
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(2):
            train_step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=28, max_name_column_width=70))

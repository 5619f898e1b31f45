"""One synchronised localisation step (eval_batch, one query): the matcher's image side beside the render on compute-unit partitions
(NeRFMatchEvaluator.split_step) against the one-stream step.  One setting per process:  python scripts/ab_split_step.py <off | R,I>
with R / I = xA-B (XCDs A..B-1) for the render / the image side."""
import statistics
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

import nerfmatch_amd
from nerfmatch_amd import synth
from nerfmatch_amd.bench_match import build_evaluator
from nerfmatch_amd.nerf.renderer import NerfRenderer

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
H, W = 480, 640
ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=64), training=False, stop_layer=3)
ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0))
ren.to(dev).eval()
nerfmatch_amd.set_precision("bf16x3")
unnorm = synth.unnorm_scene()
poses = [unnorm @ synth.camera_pose(seed=s) for s in range(64)]
spec = sys.argv[1] if len(sys.argv) > 1 else "off"
ev, make_batch = build_evaluator(dev, H, W, queries=1)
if spec != "off":
    part = lambda t: ("xcd", int(t[1:].split("-")[0]), int(t[1:].split("-")[1]) - int(t[1:].split("-")[0]))
    ev.split_step = tuple(part(t) for t in spec.split(","))
kw = dict(renderer=ren, solver="none", query2query=True, mutual=True)
keys = ("mpt2d_f", "mpt3d", "mconf", "pt_feat")


def step(i, keep=None):
    torch.manual_seed(100 + i)
    b = make_batch(torch.stack([poses[i % 64]]), unnorm)
    out = ev.eval_batch(b, **kw)
    if keep is not None:
        keep.append({k: b[k].clone() for k in keys})
    return out


for i in range(8):
    step(i)
walls = []
for i in range(40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(8 + i)
    torch.cuda.synchronize()
    walls.append((time.perf_counter() - t0) * 1e3)
keep = []
for i in range(4):
    step(200 + i, keep)
torch.cuda.synchronize()
sig = float(sum(float(k["mpt2d_f"].double().sum()) + float(k["mconf"].double().sum()) + float(k["pt_feat"].double().sum()) for k in keep))
walls.sort()
print(f"  {spec:14s} wall {statistics.median(walls):.3f} ms (p10 {walls[4]:.3f}, p90 {walls[36]:.3f})   checksum of four steps' outputs {sig!r}", flush=True)

"""Host-side profile of ONE-query localisation steps (eval_batch): where the Python time in front of the first kernel and behind the
read-back goes.  python scripts/profile_host_q1.py"""
import cProfile
import pstats
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
ev, make_batch = build_evaluator(dev, H, W, queries=1)
unnorm = synth.unnorm_scene()
poses = [unnorm @ synth.camera_pose(seed=s) for s in range(64)]
kw = dict(renderer=ren, solver="none", query2query=True, mutual=True)
step = lambda i: ev.eval_batch(make_batch(torch.stack([poses[i % 64]]), unnorm), **kw)
for i in range(10):
    step(i)
torch.cuda.synchronize()
# host time to ISSUE a step (no synchronisation inside except the matcher's own read-back)
t0 = time.perf_counter()
for i in range(50):
    step(i)
torch.cuda.synchronize()
print(f"wall per step {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for i in range(50):
    step(i)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
pstats.Stats(pr).sort_stats("cumtime").print_stats(40)

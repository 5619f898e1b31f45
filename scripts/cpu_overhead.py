"""Is the localisation loop CPU-bound?  Wall time per batch of NeRFMatchEvaluator.eval_data_loader (the loop bench.py's region B
times) and a cProfile of the host side.  Usage: python scripts/cpu_overhead.py [Q]"""
import cProfile
import pstats
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

import nerfmatch_amd
from bench import Batches
from nerfmatch_amd import synth
from nerfmatch_amd.bench_match import build_evaluator
from nerfmatch_amd.nerf.renderer import NerfRenderer

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
H, W, Q = 480, 640, int(sys.argv[1]) if len(sys.argv) > 1 else 16
ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=64), training=False, stop_layer=3)
ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0))
ren.to(dev).eval()
nerfmatch_amd.set_precision("bf16x3")
ev, make_batch = build_evaluator(dev, H, W, queries=Q)
unnorm = synth.unnorm_scene()
poses = [unnorm @ synth.camera_pose(seed=s) for s in range(64)]
kw = dict(renderer=ren, solver="none", query2query=True, mutual=True)
ev.eval_data_loader(data_loader=Batches(3, 0, Q, poses, unnorm, make_batch), **kw)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
ev.eval_data_loader(data_loader=Batches(n, 3, Q, poses, unnorm, make_batch), **kw)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n
print(f"wall {wall * 1e3:.2f} ms per batch of {Q} = {wall / Q * 1e3:.3f} ms/query = {Q / wall:.1f} queries/s")
pr = cProfile.Profile()
pr.enable()
ev.eval_data_loader(data_loader=Batches(10, 3, Q, poses, unnorm, make_batch), **kw)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(35)

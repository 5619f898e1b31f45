"""The evaluator's loop at ONE query per batch as shipped (round 6: render on five XCDs beside the previous query's matcher on three), or
with NM_LOOP_ONE_STREAM=1 on one stream -- for kernel traces (scripts/loop_trace_summarize.py) and plain timing."""
import os
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
H, W = 480, 640
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=64), training=False, stop_layer=3)
ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0))
ren.to(dev).eval()
nerfmatch_amd.set_precision("bf16x3")
unnorm = synth.unnorm_scene()
poses = [unnorm @ synth.camera_pose(seed=s) for s in range(64)]
ev, mk = build_evaluator(dev, H, W, queries=1)
ev.overlap_render = os.environ.get("NM_LOOP_ONE_STREAM", "0") != "1"
kw = dict(renderer=ren, solver="none", query2query=True, mutual=True)
ev.eval_data_loader(data_loader=Batches(6, 0, 1, poses, unnorm, mk), **kw)
torch.cuda.synchronize()
time.sleep(0.05)  # an idle gap in front of the timed loop (the trace is cut there)
t0 = time.perf_counter()
ev.eval_data_loader(data_loader=Batches(n, 6, 1, poses, unnorm, mk), **kw)
torch.cuda.synchronize()
print(f"{'two streams' if ev.overlap_render else 'one stream'}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per query over {n} queries")

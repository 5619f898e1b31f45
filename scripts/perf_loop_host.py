"""Host time of the evaluator's loop per one-query batch: the same loop on a 96 x 128 image with 8 samples per ray -- 192 rays / tokens, every kernel a few
microseconds --, so that wall per query ~ what the host needs to issue a query's launches (the bound of any deeper pipelining of the loop)."""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

import nerfmatch_amd
from bench import Batches
from nerfmatch_amd import synth
from nerfmatch_amd._lib import steady_gc
from nerfmatch_amd.bench_match import build_evaluator
from nerfmatch_amd.nerf.renderer import NerfRenderer

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
for (H, W, S) in ((96, 128, 32), (480, 640, 64)):
    ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=S), training=False, stop_layer=3)
    ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0))
    ren.to(dev).eval()
    nerfmatch_amd.set_precision("bf16x3")
    unnorm = synth.unnorm_scene()
    poses = [unnorm @ synth.camera_pose(seed=s) for s in range(64)]
    ev, mk = build_evaluator(dev, H, W, queries=1)
    kw = dict(renderer=ren, solver="none", query2query=True, mutual=True)
    for overlap in (False, True):
        ev.overlap_render = overlap
        ev.eval_data_loader(data_loader=Batches(6, 0, 1, poses, unnorm, mk), **kw)
        torch.cuda.synchronize()
        n = 100
        t0 = time.perf_counter()
        ev.eval_data_loader(data_loader=Batches(n, 6, 1, poses, unnorm, mk), **kw)
        torch.cuda.synchronize()
        print(f"{H}x{W}, {S} samples, {'two streams' if overlap else 'one stream'}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per query", flush=True)

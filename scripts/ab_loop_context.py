"""Why is the partitioned loop slow inside bench.py (4.5 ms per query) and fast in scripts/ab_render_stream.py (2.4)?  The loop at one query
per batch, one-stream and partitioned, with a loader that keeps its batch dicts alive (ab_render_stream.Keep) and one that drops them
(bench.Batches), before and after a 16-query region-B run in the same process."""
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
ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=64), training=False, stop_layer=3)
ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0))
ren.to(dev).eval()
nerfmatch_amd.set_precision("bf16x3")
unnorm = synth.unnorm_scene()
poses = [unnorm @ synth.camera_pose(seed=s) for s in range(64)]
kw = dict(renderer=ren, solver="none", query2query=True, mutual=True)


class Keep(Batches):
    def __init__(self, *a):
        super().__init__(*a)
        self.out = []

    def __getitem__(self, b):
        d = super().__getitem__(b)
        self.out.append(d)
        return d


def loop(ev, mk, cls, on, n=40):
    ev.overlap_render = on
    ev.eval_data_loader(data_loader=cls(5, 0, 1, poses, unnorm, mk), **kw)
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev.eval_data_loader(data_loader=cls(n, 5, 1, poses, unnorm, mk), **kw)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / n * 1e3
        best = el if best is None else min(best, el)
    return best


ev1, mk1 = build_evaluator(dev, H, W, queries=1)
for tag in ("fresh process", "after a 16-query region B"):
    for cls in (Keep, Batches):
        print(f"{tag:28s} {cls.__name__:8s} one stream {loop(ev1, mk1, cls, False):.3f}   partitioned {loop(ev1, mk1, cls, True):.3f} ms/query", flush=True)
    if tag == "fresh process":
        ev16, mk16 = build_evaluator(dev, H, W, queries=16)
        ev16.eval_data_loader(data_loader=Batches(6, 0, 16, poses, unnorm, mk16), **kw)
        torch.cuda.synchronize()
print("memory reserved (GB):", torch.cuda.memory_reserved() / 2**30)
import os
os.environ["PYTORCH_NO_CUDA_MEMORY_CACHING"] = "0"

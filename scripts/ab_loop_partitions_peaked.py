"""The one-query loop in the regime a trained matcher produces (~4.0 k matches per query: the fine stage runs on the matcher's partition): per-query time
for one partition split, in a process of its own (streams map to hardware queues by creation order).

    python scripts/ab_loop_partitions_peaked.py <render xcds> <style: peaked|flat>     e.g. 5 peaked  (render on XCDs 0-4, matcher on 5-7), 0 = one stream
"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

import nerfmatch_amd
from bench import Batches
from nerfmatch_amd import synth
from nerfmatch_amd._lib import steady_gc
from nerfmatch_amd.bench_match import CodedRenderer, build_evaluator
from nerfmatch_amd.nerf.renderer import NerfRenderer

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
H, W = 480, 640
nr = int(sys.argv[1])
style = sys.argv[2] if len(sys.argv) > 2 else "peaked"
ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=64), training=False, stop_layer=3)
ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0))
ren.to(dev).eval()
nerfmatch_amd.set_precision("bf16x3")
unnorm = synth.unnorm_scene()
poses = [unnorm @ synth.camera_pose(seed=s) for s in range(64)]
ev, mk = build_evaluator(dev, H, W, queries=1, style="peaked" if style == "peaked" else None)
rr = CodedRenderer(ren, ev.peaked_code) if style == "peaked" else ren
if nr == 0:
    ev.overlap_render = False
else:
    ev.render_part, ev.match_part = ("xcd", 0, nr), ("xcd", nr, 8 - nr)
kw = dict(renderer=rr, solver="none", query2query=True, mutual=True)
ev.eval_data_loader(data_loader=Batches(8, 0, 1, poses, unnorm, mk), **kw)
torch.cuda.synchronize()
walls = []
with steady_gc():
    for rep in range(4):
        n = 48
        t0 = time.perf_counter()
        m = ev.eval_data_loader(data_loader=Batches(n, 8, 1, poses, unnorm, mk), **kw)
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) / n * 1e3)
print(f"{style:6s} render on {nr} XCDs / matcher on {8 - nr}: {min(walls):.3f} ms per query (best of 4: {' '.join(f'{w:.3f}' for w in walls)}), {float(m['num_matches'].mean()):.0f} matches per query", flush=True)

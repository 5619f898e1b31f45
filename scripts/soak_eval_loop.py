"""Soak run of the localisation loop: N batches of Q queries through NeRFMatchEvaluator.eval_data_loader in chunks, throughput and allocator
state per chunk, and the records of one fixed probe batch compared bit for bit between the first and the last chunk (caches, pools and the
speculative fine stage must not drift).    python scripts/soak_eval_loop.py [Q] [batches per chunk] [chunks]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import nerfmatch_amd
from bench import Batches
from nerfmatch_amd import synth
from nerfmatch_amd.bench_match import build_evaluator
from nerfmatch_amd.nerf.renderer import NerfRenderer
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
H, W = 480, 640
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 16
per = int(sys.argv[2]) if len(sys.argv) > 2 else 50
chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=64), training=False, stop_layer=3)
ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0)); ren.to(dev).eval()
nerfmatch_amd.set_precision("bf16x3")
ev, make_batch = build_evaluator(dev, H, W, queries=Q)
unnorm = synth.unnorm_scene()
poses = [unnorm @ synth.camera_pose(seed=s) for s in range(64)]
kw = dict(renderer=ren, solver="none", query2query=True, mutual=True)
torch.manual_seed(0)
def probe():
    torch.manual_seed(1234)  # (the samplers draw from the global generator: same draws for the probe every time)
    out = ev.eval_data_loader(data_loader=Batches(2, 0, Q, poses, unnorm, make_batch), **kw)
    return {k: np.array(v, copy=True) for k, v in out.items()}
first = probe()
tot = 0
for c in range(chunks):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ev.eval_data_loader(data_loader=Batches(per, 3 + c, Q, poses, unnorm, make_batch), **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    tot += per * Q
    s = torch.cuda.memory_stats()
    print(f"chunk {c:3d}: {per * Q / dt:7.1f} queries/s   allocated {s['allocated_bytes.all.current'] / 2**20:7.0f} MB  reserved {s['reserved_bytes.all.current'] / 2**20:7.0f} MB  "
          f"device mallocs {s['num_device_alloc']}", flush=True)
last = probe()
same = all(np.array_equal(first[k], last[k], equal_nan=True) for k in first)
print(f"{tot} queries; probe batch records identical before and after: {same}")
sys.exit(0 if same else 1)

"""Is the localisation step CPU-bound?  Host time spent issuing one step (no synchronisation except the match-count
read-back the algorithm needs) vs the step's wall time."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from nerfmatch_amd import ops, synth
from nerfmatch_amd.nerf.renderer import NerfRenderer
from nerfmatch_amd.bench_match import build_matcher

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
H, W, Q = 480, 640, 4
ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=64), training=False, stop_layer=3)
ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0)); ren.to(dev).eval(); ren.precision = "bf16x3"
ops.ATTENTION_PRECISION = ops.LINEAR_PRECISION = ops.MATCH_PRECISION = "bf16x3"
matcher = build_matcher(dev, H, W, queries=Q)
K, unnorm = synth.intrinsics(H, W), synth.unnorm_scene()
poses = [unnorm @ synth.camera_pose(seed=s) for s in range(16)]
def step(i, stamps):
    t0 = time.perf_counter()
    c2ws = torch.stack([poses[(i * Q + j) % 16] for j in range(Q)])
    out = ren.render_novel_views((H, W), K, c2ws, unnorm, dev, lean=True, want_im_pred=False)
    t1 = time.perf_counter()
    matcher(out)
    t2 = time.perf_counter()
    stamps.append((t1 - t0, t2 - t1))
for i in range(3): step(i, [])
torch.cuda.synchronize()
st = []
t0 = time.perf_counter()
for i in range(20): step(i, st)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 20
print(f"wall {wall*1e3:.2f} ms/step; host time in render call {sum(a for a,_ in st)/20*1e3:.2f} ms, in matcher call (includes waiting at the count read-back) {sum(b for _,b in st)/20*1e3:.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(10): step(i, [])
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(30)

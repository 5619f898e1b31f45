"""iNeRF step WITH the matching term (use_match_loss, nerfmatch_evaluator.py:429-448) at the bench query size: wall per step and the share of
the matcher's own forward + backward (NeRFMatcherMS.match_loss + autograd.grad), which no NeRF-side kernel can shorten."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import nerfmatch_amd
from nerfmatch_amd import inerf, ops, synth
from nerfmatch_amd.bench_match import build_evaluator
from nerfmatch_amd.nerf.renderer import NerfRenderer
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
H, W = 480, 640
ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=128), training=False, stop_layer=3)
ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0)); ren.to(dev).eval()
K = synth.intrinsics(H, W)
img = torch.rand(H, W, 3, device=dev)
pose0 = torch.as_tensor(synth.camera_pose(1), dtype=torch.float32).to(dev)
nerfmatch_amd.set_precision("bf16x3")
ev, make_batch = build_evaluator(dev, H, W, queries=1)
R = (H // 8) * (W // 8)
match = dict(model=ev.model, image=torch.zeros(1, 3, H, W, device=dev), im_mask=torch.ones(1, R, dtype=torch.bool, device=dev),
             pt_mask=torch.ones(1, R, dtype=torch.bool, device=dev), unnorm=synth.unnorm_scene().to(dev))
raw = inerf._match_term
spent = []
def timed(*a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = raw(*a, **k); e1.record(); spent.append((e0, e1)); return out
inerf._match_term = timed
inerf.refine(ren, K, H, W, img, pose0, num_optim=3, match=match)
inerf.refine(ren, K, H, W, img, pose0, num_optim=2)  # (packs the fused kernels' blobs: not inside a timed loop)
torch.cuda.synchronize()
print("precisions:", ops.LINEAR_PRECISION, ops.ATTENTION_PRECISION, ops.MATCH_PRECISION, ren.precision)
spent.clear()
t0 = time.perf_counter()
inerf.refine(ren, K, H, W, img, pose0, num_optim=10, match=match)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 10 * 1e3
mt = sum(a.elapsed_time(b) for a, b in spent) / len(spent)
print(f"iNeRF step with the matching term: {wall:.2f} ms/step; matcher forward + backward (match_loss + grad): {mt:.2f} ms of it; NeRF side {wall - mt:.2f} ms")
t0 = time.perf_counter()
inerf.refine(ren, K, H, W, img, pose0, num_optim=10)
torch.cuda.synchronize()
print(f"without the term: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms/step")
# per-step wall (a step ends with its own read-back)
ts = []
t0 = time.perf_counter()
for _ in inerf.refine_iter(ren, K, H, W, img, pose0, num_optim=12, match=match):
    t1 = time.perf_counter(); ts.append((t1 - t0) * 1e3); t0 = t1
print("per-step wall with the term:", " ".join(f"{t:.1f}" for t in ts))

"""Per-step wall time of iNeRF steps with the matching term, step by step, with and without the loop's gc.freeze()
(nerfmatch_amd._lib.steady_gc): a full cyclic collection of the process's ~2e5 resident objects costs 80-90 ms and lands on one step
in ~17.    python scripts/probe_step_spikes.py [nofreeze]"""
import contextlib, gc, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import nerfmatch_amd
from nerfmatch_amd import _lib, inerf, synth
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
ev, _ = build_evaluator(dev, H, W, queries=1)
R = (H // 8) * (W // 8)
match = dict(model=ev.model, image=torch.zeros(1, 3, H, W, device=dev), im_mask=torch.ones(1, R, dtype=torch.bool, device=dev),
             pt_mask=torch.ones(1, R, dtype=torch.bool, device=dev), unnorm=synth.unnorm_scene().to(dev))
inerf.refine(ren, K, H, W, img, pose0, num_optim=2, match=match)
torch.cuda.synchronize()
guard = contextlib.nullcontext() if "nofreeze" in sys.argv else _lib.steady_gc()
print("tracked objects:", len(gc.get_objects()), "| gc counts", gc.get_count())
ts, t0 = [], time.perf_counter()
with guard:
    for _ in inerf.refine_iter(ren, K, H, W, img, pose0, num_optim=40, match=match):
        t1 = time.perf_counter(); ts.append((t1 - t0) * 1e3); t0 = t1
print(("without" if "nofreeze" in sys.argv else "with") + " gc.freeze(): per-step wall ms:", " ".join(f"{t:.1f}" for t in ts))

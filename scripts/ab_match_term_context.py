import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import nerfmatch_amd
from nerfmatch_amd import inerf, ops, synth
from nerfmatch_amd._lib import steady_gc
from nerfmatch_amd.bench_match import build_evaluator
from nerfmatch_amd.nerf.renderer import NerfRenderer
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
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
def run(tag, n=8):
    spent.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    inerf.refine(ren, K, H, W, img, pose0, num_optim=n, match=match)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    mt = sum(a.elapsed_time(b) for a, b in spent) / len(spent)
    print(f"{tag:40s} {wall:.2f} ms/step, matcher {mt:.2f}", flush=True)
run("plain"); run("plain")
with steady_gc():
    run("steady_gc")
kp = bench.KernelProbe(); ops.KERNEL_PROBE = kp; kp.on = True
run("kernel probe on")
kp.on = False
run("probe off")
match2 = dict(match); match2.pop("_im_tokens", None)
inerf.refine(ren, K, H, W, img, pose0, num_optim=8, match=match2)
spent.clear()
t0=time.perf_counter(); inerf.refine(ren, K, H, W, img, pose0, num_optim=8, match=dict(match2, **{})); torch.cuda.synchronize()
print("fresh match dict each refine (image side recomputed once per refine):", (time.perf_counter()-t0)/8*1e3, sum(a.elapsed_time(b) for a, b in spent) / len(spent))

"""Step time of the iNeRF refinement at the bench query size (640x480/ds8 -> 4800 rays, 128+128 samples)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from nerfmatch_amd import inerf, ops, synth
from nerfmatch_amd.nerf.renderer import NerfRenderer

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
H, W = 480, 640
ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=128), training=False, stop_layer=3)
ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0))
ren.to(dev).eval()
K = synth.intrinsics(H, W)
img = torch.rand(H, W, 3, device=dev)
pose0 = torch.as_tensor(synth.camera_pose(1), dtype=torch.float32).to(dev)
for nerf_prec, lin_prec in (("fp32", "fp32"), ("bf16x3", "bf16x3")):
    ren.precision, ops.LINEAR_PRECISION = nerf_prec, lin_prec
    inerf.refine(ren, K, H, W, img, pose0, num_optim=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    poses, losses, _ = inerf.refine(ren, K, H, W, img, pose0, num_optim=10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"iNeRF step ({nerf_prec} coarse kernel, {lin_prec} GEMMs): {dt*1e3:.2f} ms/step  loss {losses[0]:.5f} -> {losses[-1]:.5f}")
ops.LINEAR_PRECISION = "fp32"

# with the matching term (use_match_loss): + the matcher's training-mode forward and its backward to pt_feat / pt3d
from nerfmatch_amd.bench_match import build_evaluator

ev, _ = build_evaluator(dev, H, W, 1)
R = (H // 8) * (W // 8)
match = dict(model=ev.model, image=torch.rand(1, 3, H, W, device=dev), unnorm=synth.unnorm_scene().to(dev),
             im_mask=torch.ones(1, R, dtype=torch.bool, device=dev), pt_mask=torch.ones(1, R, dtype=torch.bool, device=dev))
for nerf_prec, lin_prec in (("fp32", "fp32"), ("bf16x3", "bf16x3")):
    ren.precision, ops.LINEAR_PRECISION = nerf_prec, lin_prec
    ops.ATTENTION_PRECISION = ops.MATCH_PRECISION = lin_prec
    inerf.refine(ren, K, H, W, img, pose0, num_optim=2, match=match)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    poses, losses, _ = inerf.refine(ren, K, H, W, img, pose0, num_optim=10, match=match)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"iNeRF step with the matching term ({lin_prec}): {dt*1e3:.2f} ms/step  loss {losses[0]:.5f} -> {losses[-1]:.5f}  "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
ops.LINEAR_PRECISION = ops.ATTENTION_PRECISION = ops.MATCH_PRECISION = "fp32"

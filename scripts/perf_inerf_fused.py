"""iNeRF step at the bench query size on the default arithmetic (fused pointwise kernels): wall per step + (under rocprofv3) the kernel list."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import nerfmatch_amd
from nerfmatch_amd import inerf, ops, synth
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
inerf.refine(ren, K, H, W, img, pose0, num_optim=3)
torch.cuda.synchronize()
t0 = time.perf_counter()
inerf.refine(ren, K, H, W, img, pose0, num_optim=10)
torch.cuda.synchronize()
print(f"iNeRF step, renderer precision {ren.precision}, fused fine pass: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms/step")

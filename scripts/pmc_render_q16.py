"""A few launches of the bench's region-A step -- 16 queries x 4800 rays x (64 + 64) samples, every output, every sample evaluated -- for
rocprofv3 --pmc passes (round 6: `roofline.traffic` measured at the bench's own launch size instead of a one-query figure times 16)."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from nerfmatch_amd import synth
from nerfmatch_amd.nerf.renderer import NerfRenderer

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=64), training=False, stop_layer=3)
ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0))
ren.to(dev).eval()
ren.precision = os.environ.get("NM_PRECISION", "fp16x3")
ren.skip_zero_tail = False
unnorm = synth.unnorm_scene()
for step in range(4):
    c2ws = torch.stack([unnorm @ synth.camera_pose(16 * step + q) for q in range(16)])
    ren.render_novel_views((480, 640), synth.intrinsics(), c2ws, unnorm, dev, lean=False)
torch.cuda.synchronize()

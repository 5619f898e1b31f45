"""One warm-up + a few nm_nerf_fwd launches at the bench shape, for rocprofv3 --pmc passes (no events, no timing)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from nerfmatch_amd import synth, ops
from nerfmatch_amd.nerf.renderer import NerfRenderer
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=S), training=False, stop_layer=3)
ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0))
ren.to(dev).eval()
ren.ret_pfeat = True
import os
ren.precision = os.environ.get("NM_PRECISION", "fp32")
ren.skip_zero_tail = os.environ.get("NM_PMC_SKIP", "0") == "1"  # default: every sample evaluated (bench.py's region A since round 3)
lean = os.environ.get("NM_PMC_LEAN", "0") == "1"  # lean render: the coarse pass runs on the fp16x1 kernel (with NM_PRECISION=bf16x3)
for q in range(4):
    ren.render_novel_view((480, 640), synth.intrinsics(), synth.unnorm_scene() @ synth.camera_pose(q), synth.unnorm_scene(), dev, lean=lean)
torch.cuda.synchronize()

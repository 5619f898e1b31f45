"""The coarse-only model on a batch of 16 pairs of 4800 x 4800 tokens (bench.py's `mini` leg), a few forwards, for rocprofv3 --pmc
passes of the fused matching kernels (csrc/match_fused.hip)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import nerfmatch_amd
from nerfmatch_amd import synth
from nerfmatch_amd.matcher import NeRFMatcherCoarse
from nerfmatch_amd.modules import PrecomputedBackbone
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
Q, R = 16, 4800
mm = NeRFMatcherCoarse(synth.matcher_config("coarse"))
mm.load_state_dict(synth.matcher_state_dict("coarse"), strict=False)
im, pt = synth.separated_features(R, R, 256, seed=2)
mm.backbone = PrecomputedBackbone(im.T.reshape(1, 256, 60, 80).expand(Q, -1, -1, -1).contiguous().to(dev), 256)
mm.to(dev).eval()
mm.keep_conf = False
nerfmatch_amd.set_precision("bf16x3")
d = dict(image=torch.zeros(Q, 3, 8, 8, device=dev), im_mask=torch.ones(Q, R, dtype=torch.bool, device=dev), pt3d=torch.zeros(Q, R, 3, device=dev),
         pt_feat=pt[None].expand(Q, -1, -1).contiguous().to(dev), pt_mask=torch.ones(Q, R, dtype=torch.bool, device=dev), pt2d=None)
for _ in range(4):
    mm.forward(d, mutual=True)
torch.cuda.synchronize()

"""Matcher leg of bench.py: NeRFMatcherMS (shipped c2f configuration, PCG64 weights) on the rendered points.
The image backbone (timm ConvFormer, out of scope) is excluded: its two output maps are drawn once with the stub
backbone from a synthetic image and re-used by every step."""
import torch

from . import synth
from .matcher import NeRFMatcherMS
from .modules import PrecomputedBackbone, StubBackbone


def build_matcher(dev, H, W, mutual=True, queries=1):
    model = NeRFMatcherMS(synth.matcher_config("c2f"))
    model.load_state_dict(synth.matcher_state_dict("c2f", seed=0), strict=False)
    g = torch.Generator().manual_seed(3)
    img = torch.randn(queries, 3, H, W, generator=g).to(dev)
    cfeat, ffeat = StubBackbone().to(dev)(img)
    model.backbone = PrecomputedBackbone((cfeat.contiguous(), ffeat.contiguous()), [256, 128])
    model.to(dev).eval()
    M = (H // 8) * (W // 8)
    ys, xs = torch.meshgrid(torch.arange(H // 8), torch.arange(W // 8), indexing="ij")
    pt2d = (torch.stack([xs, ys], -1) * 8 + 4).float().reshape(1, M, 2).expand(queries, M, 2).contiguous().to(dev)
    im_mask = torch.ones(queries, M, dtype=torch.bool, device=dev)

    def begin(render_out):
        """Enqueue the matcher up to its single synchronisation point (the match-count read-back)."""
        pt3d, pt_feat = render_out["pt3d"], render_out["pt_feat"]
        if pt3d.dim() == 2:
            pt3d, pt_feat = pt3d.unsqueeze(0), pt_feat.unsqueeze(0)
        data = dict(image=img, im_mask=im_mask, pt3d=pt3d, pt_feat=pt_feat,
                    pt_mask=torch.ones_like(pt3d[..., 0]), pt2d=pt2d)
        return model.forward_begin(data, mutual=mutual)

    def finish(state):
        """Counts read-back, fine stage, match assembly; returns the number of matches."""
        model.forward_finish(state)
        return float(state["data"]["mpt3d"].shape[0])

    def run(render_out):
        return finish(begin(render_out))

    run.begin, run.finish = begin, finish
    return run

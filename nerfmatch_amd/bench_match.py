"""Matcher side of bench.py: a NeRFMatchEvaluator around NeRFMatcherMS (shipped c2f configuration, PCG64 weights) and the
query batches it localises.  The image backbone (timm ConvFormer, out of scope) is excluded: its two output maps are drawn once
with the stub backbone from a synthetic image and re-used by every step."""
from argparse import Namespace

import torch

from . import synth
from .modules import PrecomputedBackbone, StubBackbone
from .nerfmatch_evaluator import NeRFMatchEvaluator


class CodedRenderer:
    """Measurement aid of the `peaked` bench leg: a NerfRenderer whose rendered point features get a seeded per-ray code added (one
    elementwise launch per batch, inside the timed region).  The features a random-weight NeRF renders are not discriminative (DESIGN.md
    section 4), a trained matcher sits on features that are; the code is the synthetic discriminative part, the same one the image
    tokens of build_evaluator(style="peaked") carry -- ray i <-> image token i, as in iNeRF's matching term."""

    def __init__(self, renderer, code):
        self._r, self._code = renderer, code

    def __getattr__(self, name):
        return getattr(self._r, name)

    def render_novel_views(self, *a, **kw):
        out = self._r.render_novel_views(*a, **kw)
        out["pt_feat"] = out["pt_feat"] + self._code
        return out

    def render_novel_view(self, *a, **kw):
        out = self._r.render_novel_view(*a, **kw)
        out["pt_feat"] = out["pt_feat"] + self._code
        return out


def build_evaluator(dev, H, W, queries=1, kind="c2f", style=None):
    """style="peaked": see below (wrap the renderer in CodedRenderer(ren, ev.peaked_code)).
    -> (evaluator, make_batch); kind "c2f" (NeRFMatcherMS) or "coarse" (NeRFMatcherCoarse, the Mini model).  make_batch(c2ws (Q,4,4) world poses, unnorm) builds one batch dict in the reference's
    schema (nerfmatch_dataset.py:311-325) whose large tensors are shared, device-resident buffers (inputs are in HBM when the
    timed region starts) and whose small per-query tensors (K, poses, scene normalisation) live on the host, as a DataLoader
    would deliver them."""
    ev = NeRFMatchEvaluator(Namespace(model=synth.matcher_config(kind), exp=Namespace(seed=0), data=Namespace()))
    peaked = style == "peaked"
    ev.model.load_state_dict(synth.matcher_state_dict(kind, seed=0, **(dict(temperature=30.0, style="aligned") if peaked else {})), strict=False)
    g = torch.Generator().manual_seed(3)
    img = torch.randn(queries, 3, H, W, generator=g).to(dev)
    cfeat, ffeat = StubBackbone().to(dev)(img)
    if peaked:
        # the regime a TRAINED matcher produces (round 6; extract_matches.py:21-36 returns ~1e3 matches on real data):
        # `style="aligned"` weights at temperature 30 and planted correspondences -- image token i = code_i + noise 0.25, point token i =
        # rendered feature + code_i (CodedRenderer), 70 % of the image tokens planted -- give ~3.4 k mutual matches per query with row maxima near 1
        M_ = (H // 8) * (W // 8)
        code = torch.randn(M_, 256, generator=torch.Generator().manual_seed(41))
        planted = (torch.arange(M_) % 10 < 7)[:, None]  # 70 % of the image tokens have a counterpart among the points, the rest are clutter
        noisy = torch.stack([torch.where(planted, code, torch.randn(M_, 256, generator=torch.Generator().manual_seed(200 + q))) +
                             0.25 * torch.randn(M_, 256, generator=torch.Generator().manual_seed(100 + q)) for q in range(queries)])
        cfeat = noisy.transpose(1, 2).reshape(queries, 256, H // 8, W // 8).contiguous().to(dev)
        ev.peaked_code = code.to(dev)
    if kind == "c2f":
        ev.model.backbone = PrecomputedBackbone((cfeat.contiguous(), ffeat.contiguous()), [256, 128])
    else:  # NeRFMatch-Mini: one 1/8-resolution map (coarse_trainer.py:94-107)
        ev.model.backbone = PrecomputedBackbone(cfeat.contiguous(), 256)
    ev.model.to(dev).eval()
    M = (H // 8) * (W // 8)
    ys, xs = torch.meshgrid(torch.arange(H // 8), torch.arange(W // 8), indexing="ij")
    pt2d = (torch.stack([xs, ys], -1) * 8 + 4).float().reshape(1, M, 2).expand(queries, M, 2).contiguous().to(dev)
    im_mask = torch.ones(queries, M, dtype=torch.bool, device=dev)
    K = synth.intrinsics(H, W)[None].expand(queries, 3, 3).contiguous()

    def make_batch(c2ws, unnorm):
        return dict(image=img, im_mask=im_mask, pt2d=pt2d, K=K, c2w=c2ws, unnorm_scene=unnorm[None].expand(queries, 4, 4).contiguous())

    return ev, make_batch

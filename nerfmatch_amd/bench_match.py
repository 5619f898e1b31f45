"""Matcher side of bench.py: a NeRFMatchEvaluator around NeRFMatcherMS (shipped c2f configuration, PCG64 weights) and the
query batches it localises.  The image backbone (timm ConvFormer, out of scope) is excluded: its two output maps are drawn once
with the stub backbone from a synthetic image and re-used by every step."""
from argparse import Namespace

import torch

from . import synth
from .modules import PrecomputedBackbone, StubBackbone
from .nerfmatch_evaluator import NeRFMatchEvaluator


def build_evaluator(dev, H, W, queries=1, kind="c2f"):
    """-> (evaluator, make_batch); kind "c2f" (NeRFMatcherMS) or "coarse" (NeRFMatcherCoarse, the Mini model).  make_batch(c2ws (Q,4,4) world poses, unnorm) builds one batch dict in the reference's
    schema (nerfmatch_dataset.py:311-325) whose large tensors are shared, device-resident buffers (inputs are in HBM when the
    timed region starts) and whose small per-query tensors (K, poses, scene normalisation) live on the host, as a DataLoader
    would deliver them."""
    ev = NeRFMatchEvaluator(Namespace(model=synth.matcher_config(kind), exp=Namespace(seed=0), data=Namespace()))
    ev.model.load_state_dict(synth.matcher_state_dict(kind, seed=0), strict=False)
    g = torch.Generator().manual_seed(3)
    img = torch.randn(queries, 3, H, W, generator=g).to(dev)
    cfeat, ffeat = StubBackbone().to(dev)(img)
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

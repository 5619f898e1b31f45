"""Image backbones (contract only).

The reference uses timm's ConvFormer/CAFormer-B36 patched to emit 1/8 and 1/2 resolution maps
(nerfmatch/modules/__init__.py:14-113).  timm is third-party code that is absent from the reference tree and
is out of the hot-path scope; only the OUTPUT CONTRACT matters here:
    init_backbone_8_2(...)  -> module with .feat_dim == [256, 128], forward(img) -> (cfeat (B,256,H/8,W/8),
                                                                                    ffeat (B,128,H/2,W/2))
    init_backbone(...)      -> module with .feat_dim == 256,        forward(img) -> cfeat (B,256,H/8,W/8)
`backbone="stub"` selects a small deterministic torch stand-in with that contract (synthetic benchmarks, tests);
any other name needs timm and is delegated to it when importable."""
import torch
from torch import nn
import torch.nn.functional as F


class StubBackbone(nn.Module):
    """Deterministic stand-in: average pooling + fixed random 1x1 projections (PCG64 seeded)."""

    def __init__(self, coarse_dim=256, fine_dim=128, two_scales=True, seed=7):
        super().__init__()
        import numpy as np

        rng = np.random.default_rng(seed)
        self.two_scales = two_scales
        self.register_buffer("wc", torch.from_numpy(rng.standard_normal((coarse_dim, 3 * 16, 1, 1)).astype("float32")) / 3.0, persistent=False)
        self.register_buffer("wf", torch.from_numpy(rng.standard_normal((fine_dim, 3 * 4, 1, 1)).astype("float32")) / 2.0, persistent=False)
        self.feat_dim = [coarse_dim, fine_dim] if two_scales else coarse_dim

    def forward(self, img):
        # 1x1 projections as plain contractions (a conv2d would send MIOpen into its solver search on every fresh process)
        c = torch.einsum("oc,bchw->bohw", self.wc[:, :, 0, 0], F.pixel_unshuffle(F.avg_pool2d(img, 2), 4)).contiguous()
        if not self.two_scales:
            return c
        f = torch.einsum("oc,bchw->bohw", self.wf[:, :, 0, 0], F.pixel_unshuffle(img, 2)).contiguous()
        return c, f


class PrecomputedBackbone(nn.Module):
    """Returns feature maps computed elsewhere (parity tests start from the backbone's outputs)."""

    def __init__(self, outs, feat_dim):
        super().__init__()
        self.outs, self.feat_dim = outs, feat_dim

    def forward(self, img):
        return self.outs


def _need_timm(name):
    try:
        import timm  # noqa: F401
    except ImportError as e:
        raise NotImplementedError(
            f"backbone '{name}' is timm's ConvFormer/CAFormer (third-party, not part of the reference tree, out of scope); "
            "install timm and plug a module with the documented contract, or use backbone='stub'") from e
    raise NotImplementedError("timm is importable but the ConvFormer stride patching is not part of this build; "
                              "pass a module via model.backbone = ...")


def init_backbone_8_2(backbone="stub", pretrained=False):
    if backbone == "stub":
        return StubBackbone(two_scales=True)
    _need_timm(backbone)


def init_backbone(backbone="stub", pretrained=False, downsample=8):
    if backbone == "stub":
        return StubBackbone(two_scales=False)
    _need_timm(backbone)

"""Thin torch-tensor wrappers over the C ABI (one function per entry point of include/nerfmatch_amd.h).

Everything here enqueues hand-written gfx950 kernels on torch's current stream; there is no eager
fallback.  Tensors must live on the GPU, be contiguous and fp32 (int64 / uint8 where stated)."""
import ctypes as C

import torch

from . import _lib
from ._lib import check, dptr, hptr, lib, stream

NEAR_PLANE = 0.01  # reference: nerfmatch/nerf/render_utils.py:72


def _f32(t):
    return t.to(torch.float32).contiguous()


# ----------------------------------------------------------------------------- NeRF half
def raygen(K, c2w_norm, H, W, device, ds=8, near=NEAR_PLANE):
    """rays (R,12) on `device` for the sub-sampled pixel grid; also returns the far-fallback flag tensor."""
    L = lib()
    kinv = torch.linalg.inv(K.detach().to("cpu", torch.float32)).contiguous()
    pose = c2w_norm.detach().to("cpu", torch.float32).contiguous()
    R = L.nm_raygen_count(int(H), int(W), int(ds))
    rays = torch.empty(R, 12, device=device, dtype=torch.float32)
    flag = torch.empty(1, device=device, dtype=torch.int32)
    check(L.nm_raygen(hptr(kinv), hptr(pose), int(H), int(W), int(ds), float(near), dptr(rays), dptr(flag, torch.int32), stream()), "nm_raygen")
    return rays, flag


def sample_coarse(rays, t_rand, S):
    R = rays.shape[0]
    assert t_rand.shape == (R, S + 1)
    t = torch.empty(R, S + 1, device=rays.device, dtype=torch.float32)
    check(lib().nm_sample_coarse(dptr(rays), dptr(t_rand), R, int(S), dptr(t), stream()), "nm_sample_coarse")
    return t


def resample(t, weights, jitter, padding=0.01, randomized=True):
    R, n = t.shape
    S = n - 1
    assert weights.shape == (R, S)
    out = torch.empty_like(t)
    check(lib().nm_resample(dptr(t), dptr(weights), dptr(jitter), R, S, float(padding), int(bool(randomized)), dptr(out), stream()), "nm_resample")
    return out


def nerf_fwd(blob, rays, t, app_row=None, tap_layer=-1, white_bg=False, var_scale=-1.0, need_rgb=True, need_feat=True,
             feat_max=False, want_raw=False, want_sample_feat=False):
    """One fused pass.  Returns dict(weights, feat, pts, rgb, depth, acc[, raw, sample_feat])."""
    R, n = t.shape
    S = n - 1
    dev = rays.device
    new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
    out = dict(weights=new(R, S), pts=new(R, 3), depth=new(R), acc=new(R))
    out["feat"] = new(R, 256) if need_feat else None
    out["rgb"] = new(R, 3) if need_rgb else None
    out["raw"] = new(R, S, 4) if want_raw else None
    out["sample_feat"] = new(R, S, 256) if want_sample_feat else None
    flags = (0 if need_rgb else _lib.NM_NERF_SKIP_RGB) | (_lib.NM_NERF_FEAT_MAX if feat_max else 0)
    check(lib().nm_nerf_fwd(dptr(blob), dptr(rays), dptr(t), dptr(app_row), R, S, int(tap_layer), int(bool(white_bg)),
                            float(var_scale), flags, dptr(out["weights"]), dptr(out["feat"]), dptr(out["pts"]),
                            dptr(out["rgb"]), dptr(out["depth"]), dptr(out["acc"]), dptr(out["raw"]),
                            dptr(out["sample_feat"]), stream()), "nm_nerf_fwd")
    return out


def unnormalize_points(pts, unnorm):
    m = unnorm.detach().to("cpu", torch.float32).contiguous()
    out = torch.empty_like(pts)
    check(lib().nm_unnormalize_points(dptr(pts), hptr(m), pts.shape[0], dptr(out), stream()), "nm_unnormalize_points")
    return out

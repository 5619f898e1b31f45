"""ORACLE (test infrastructure, NOT product code) -- iNeRF pose refinement.

A CPU restatement, with torch autograd, of `NeRFMatchEvaluator.inerf_refinement`
(nerfmatch/nerfmatch_evaluator.py:288-500, `eval_pose=True`, with and without the `use_match_loss` term) and of the ray
generator it differentiates through (`gen_rays`, :232-286).  Only tests/ may import it.

Parity status: PINNED against golden trajectories (and, for the matching term, per-step pose gradients) produced by the
reference's own `inerf_refinement` in the build container (tests/golden/make_golden.py -> tests/golden/inerf_*.npz;
checked by tests/test_oracle_golden.py).

What the reference does per Adam step (all of it reproduced here, quirks included):
  * rays = gen_rays(cam_pose) with grad; the samplers see `rays.detach()` and run with their default
    `randomized=True` (stratified jitter + the `u + u + jitter` resampler), 128 + 128 samples hard-coded (:354-370);
  * the Gaussians' variances come from the detached sampler (`scale_var=1`: a multiplication by 1), the means are
    re-derived WITH grad as o + t_mean * viewdir, t_mean from mu and the sign-flipped hw = (t0 - t1)/2 (only hw^2 is
    used) (:372-383);
  * coarse network under no_grad, fine network with grad; appearance row of ray_id 1 if the model has one (:391-399);
  * compositing with white_bg=True and rays_d = rays[:, 3:6] (:410-418); loss = MSE(rgb_map, image[ds//2::ds, ds//2::ds]);
  * torch.optim.Adam on the full 4x4 pose matrix (not re-orthonormalised), optional cosine lr decay (:336-346);
  * `use_match_loss` (:420-441): pt_feat = sum_s w_s feats_s (fine weights and the fine network's layer-`stop_layer`
    activations, both WITH grad), pt3d = unnormalise(sum_s w_s mean_s) with the sampler's DETACHED Gaussian means, the
    matcher's forward_match(mutual=True) on them, and the focal loss of its conf_matrix against the identity (image token i
    <-> ray i; the sub-sampled ray grid IS the coarse token grid at ds = 8) added to the photometric loss.
The random tensors the samplers draw are explicit inputs (`t_rands[j]`, `jitters[j]`, one pair per step).
"""
import math

import numpy as np
import torch

from . import matcher_oracle as mo
from . import nerf_oracle as no
from . import train_oracle as to


def gen_rays(pose, W, H, K, ds=8, z_near=0.01):
    """(4,4) pose -> rays (R,12) = [o, viewdir, near, far, viewdir, radius], differentiable w.r.t. `pose`.
    nerfmatch/nerfmatch_evaluator.py:232-286 (the far plane falls back to 1 when the sphere test fails)."""
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    xys = torch.stack([xs, ys, torch.ones_like(xs)], dim=-1).float()
    dirs = xys @ torch.linalg.inv(K).T
    centers = pose[None, None, :3, 3].expand(H, W, -1)
    raydir = torch.matmul(pose[None, None, :3, :3], dirs.unsqueeze(-1))[..., 0]
    view = raydir / raydir.norm(dim=-1, keepdim=True)
    near = torch.full((H, W, 1), z_near)
    far, ok = no.sphere_far(centers.reshape(-1, 3), view.reshape(-1, 3))
    far = far.reshape(H, W, 1) if ok else torch.ones(H, W, 1)
    dx = torch.sqrt(torch.sum((view[:-1] - view[1:]) ** 2, -1))
    dx = torch.cat([dx, dx[-2:-1]], 0)
    radii = dx[..., None] * 2 / np.sqrt(12)
    rays = torch.cat((centers, view, near, far, view, radii), dim=-1)
    return rays[ds // 2 :: ds, ds // 2 :: ds].reshape(-1, 12)


def step_loss(params, pose, K, H, W, img_ds, t_rand, jitter, app_row=None, ds=8, num_pts=128, match=None, stop_layer=3):
    """Loss of one refinement step (and the rendered colours), differentiable w.r.t. `pose`.  match: None (photometric loss
    only) or dict(p=matcher parameters, cfg, cfeat, ffeat, unnorm, im_mask, pt_mask) for the `use_match_loss` term."""
    rays = gen_rays(pose, W, H, K, ds)
    rd = rays.detach()
    view = rays[:, 8:11]
    out = None
    t = w = None
    for key in ("coarse", "fine"):
        with torch.no_grad():
            if key == "coarse":
                t = no.sample_coarse(rd, num_pts, t_rand)
            else:
                t = no.resample(t, w, jitter, padding=0.01, randomized=True)
            mean, var = no.frustum_gaussians(t, rd[:, :3], rd[:, 3:6], rd[:, 11:12])
            var = 1 * var
        mu = (t[:, :-1] + t[:, 1:]) / 2
        hw = (t[:, :-1] - t[:, 1:]) / 2
        eps = torch.tensor(torch.finfo(torch.float32).eps)
        t_mean = mu + (2 * mu * hw**2) / torch.maximum(eps, 3 * mu**2 + hw**2)
        R, S = t_mean.shape
        pts = rays[:, None, :3].expand(R, S, 3) + t_mean[:, :, None] * view[:, None, :].expand(R, S, 3)
        x_pts = no.ipe(pts.reshape(-1, 3), var.reshape(-1, 3), 15)
        x_dir = no.dir_pe(view[:, None, :].expand(R, S, 3).reshape(-1, 3), 4)
        x_app = None if app_row is None else app_row.view(1, -1).expand(R * S, -1)
        if key == "coarse":
            with torch.no_grad():
                raw, _ = no.nerf_mlp(params, "nerf_coarse", x_pts, x_dir, x_app)
        else:
            raw, feats = no.nerf_mlp(params, "nerf_fine", x_pts, x_dir, x_app, stop_layer=stop_layer)
        out = no.composite(raw.reshape(R, S, 4), t, rays[:, 3:6], white_bg=True)
        w = out[3].detach()
    rgb_map = out[0]
    loss = torch.mean((rgb_map - img_ds) ** 2)
    if match is not None:
        weights = out[3]
        pt_feat = torch.sum(weights[..., None] * feats.reshape(R, S, -1), dim=-2)[None]
        pts = torch.sum(weights[..., None] * mean, dim=-2)
        un = match["unnorm"]
        hom = torch.cat([pts, torch.ones_like(pts[..., 0:1])], dim=-1)[None]  # unnormaliz_pts, utils/geometry.py:76-85
        pt3d = torch.bmm(un[None], hom.transpose(-1, -2)).transpose(-1, -2)[..., :3]
        if match.get("ffeat") is None:  # the coarse-only model class (its forward_match has the same call signature, coarse_trainer.py:236-288)
            preds = mo.coarse_forward_match_cfg(match["p"], match["cfg"], match["cfeat"], pt_feat, pt3d, match.get("im_mask"), match.get("pt_mask"),
                                                mutual=True)
        else:
            preds = mo.c2f_forward_match(match["p"], match["cfg"], match["cfeat"], match["ffeat"], pt_feat, pt3d, match.get("im_mask"),
                                         match.get("pt_mask"), mutual=True)
        loss = loss + to.matching_loss(preds["conf_matrix"], torch.eye(R)[None])
    return loss, rgb_map


def refine(params, K, H, W, image_hw3, pose0, t_rands, jitters, lrate=0.001, lrdecay=False, app_row=None, ds=8, match=None, grads=None):
    """`len(t_rands)` Adam steps from `pose0` (normalised-scene c2w).  Returns (poses after every step, losses); the pose
    gradient of every step is appended to `grads` when given."""
    img_ds = image_hw3[ds // 2 :: ds, ds // 2 :: ds].contiguous().view(-1, 3)
    pose = pose0.clone().requires_grad_(True)
    opt = torch.optim.Adam(params=[pose], lr=lrate)
    n = len(t_rands)
    poses, losses = [], []
    for j in range(n):
        if lrdecay:
            for g in opt.param_groups:
                g["lr"] = lrate * (1 + math.cos(math.pi * j / n)) / 2
        with torch.enable_grad():
            loss, _ = step_loss(params, pose, K, H, W, img_ds, t_rands[j], jitters[j], app_row, ds, match=match)
            loss.backward()
        if grads is not None:
            grads.append(pose.grad.detach().clone())
        opt.step()
        opt.zero_grad()
        poses.append(pose.detach().clone())
        losses.append(float(loss.detach()))
    return poses, losses

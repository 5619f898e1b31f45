#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REFERENCE implementation on CPU.

Runs only in the build container (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

What it does
  * puts /root/reference on sys.path and installs empty stub modules for third-party packages
    that are not installed here and are not on the hot path (cv2, pycolmap, imageio, torchvision,
    timm, pytorch_lightning, transforms3d, kornia) -- see SURVEY.md section 8c;
  * hands the reference (a) a fixed-output stand-in for the timm backbone (out of scope: the
    fixtures start at the backbone's outputs) and (b) a restatement of the two kornia helpers
    used by fine matching (so that part is "parity unpinned" w.r.t. real kornia);
  * builds the reference's NerfRenderer / NeRFMatcherMS / NeRFMatcherCoarse from the shipped
    hyper-parameter values, loads PCG64 weights from nerfmatch_amd.synth, feeds seeded inputs and
    stores inputs + outputs as small .npz files.

The random tensors the reference draws at inference (torch.rand for the stratified jitter,
Tensor.uniform_ for the resampling jitter) are captured by replaying the same seeded global
generator; the replay is validated by the oracle-vs-golden test.
Only data (inputs / expected outputs) is written; no reference source is copied.
"""
import sys
import types
from argparse import Namespace
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[2]
REF = Path("/root/reference")
import os  # noqa: E402

OUT = Path(os.environ.get("NM_GOLDEN_OUT", Path(__file__).resolve().parent))  # (tests/test_golden_regen_cpu.py regenerates into a temp dir)
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REF))

from nerfmatch_amd import synth  # noqa: E402


# ----------------------------------------------------------------------------- stubs
def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _grid(h, w, normalized_coordinates=True, device=None):
    xs = torch.linspace(-1, 1, w, device=device)
    ys = torch.linspace(-1, 1, h, device=device)
    g = torch.stack(torch.meshgrid([xs, ys], indexing="ij"), dim=-1)
    return g.permute(1, 0, 2).unsqueeze(0)


def _spatial_expectation2d(inp, normalized_coordinates=True):
    b, n, h, w = inp.shape
    g = _grid(h, w, normalized_coordinates, inp.device)
    px, py = g[..., 0].reshape(-1), g[..., 1].reshape(-1)
    flat = inp.reshape(b, n, -1)
    return torch.cat([(px * flat).sum(-1, keepdim=True), (py * flat).sum(-1, keepdim=True)], -1)


def _rodrigues(Rm):
    """Stand-in for cv2.Rodrigues(matrix) -> (rotation vector, None); only feeds the pose-error metric (utils/metrics.py:366-368)."""
    Rm = np.asarray(Rm, dtype=np.float64)
    ang = np.arccos(np.clip((np.trace(Rm) - 1.0) / 2.0, -1.0, 1.0))
    ax = np.array([Rm[2, 1] - Rm[1, 2], Rm[0, 2] - Rm[2, 0], Rm[1, 0] - Rm[0, 1]])
    n = np.linalg.norm(ax)
    return ((ax / n * ang) if n > 1e-12 else np.zeros(3)).reshape(3, 1), None


def install_stubs():
    for n in ("pycolmap", "imageio", "timm", "h5py"):
        _stub(n)
    _stub("cv2", COLORMAP_JET=2, Rodrigues=_rodrigues)
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms")
    t3 = _stub("transforms3d")
    t3.quaternions = _stub("transforms3d.quaternions", qinverse=None, qmult=None, rotate_vector=None, quat2mat=None, mat2quat=None)
    pl = _stub("pytorch_lightning", LightningModule=torch.nn.Module, seed_everything=lambda *a, **k: None, Trainer=object)
    pl.callbacks = _stub("pytorch_lightning.callbacks", ModelCheckpoint=object, LearningRateMonitor=object)
    pl.loggers = _stub("pytorch_lightning.loggers", TensorBoardLogger=object)
    pl.plugins = _stub("pytorch_lightning.plugins", DDPPlugin=object)
    k = _stub("kornia")
    k.geometry = _stub("kornia.geometry")
    k.geometry.subpix = _stub("kornia.geometry.subpix")
    k.geometry.subpix.dsnt = _stub("kornia.geometry.subpix.dsnt", spatial_expectation2d=_spatial_expectation2d)
    k.utils = _stub("kornia.utils")
    k.utils.grid = _stub("kornia.utils.grid", create_meshgrid=_grid)


class FixedBackbone(torch.nn.Module):
    """Stands in for the timm backbone: returns pre-drawn feature maps."""

    def __init__(self, outs, feat_dim):
        super().__init__()
        self.outs, self.feat_dim = outs, feat_dim

    def forward(self, img):
        return self.outs


def to_np(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.detach().cpu().numpy()
        else:
            out[k] = np.asarray(v)
    return out


# ----------------------------------------------------------------------------- NeRF half
def nerf_fixture(tag, scene_type, H, W, S, stop_layer, seed, sub_rays=4, style=None, focal=60.0, pose_seed=None, density_shift=0.0):
    from nerfmatch.nerf.renderer import NerfRenderer
    from nerfmatch.nerf import render_utils as ru

    torch.set_grad_enabled(False)
    cfg = synth.nerf_config(scene_type, num_pts=S, img_wh=(W, H))
    app = scene_type == "cambridge"
    # style "surface": the trained-like regime (activations O(10), densities in the thousands, alpha saturating within
    # 2-4 coarse samples; synth.SURFACE_STYLE); default: smooth random field with density_bias 3
    sd = synth.nerf_state_dict(seed=seed, app_vocab=5 if app else 0, density_bias=3.0 if style is None else density_shift, style=style)
    ren = NerfRenderer(cfg, num_frames=5 if app else None, training=False, stop_layer=stop_layer)
    missing = ren.load_state_dict(sd, strict=True)
    ren.eval()

    K = torch.tensor([[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1]])
    unnorm = synth.unnorm_scene()
    c2w_n = synth.camera_pose(seed=seed + 10 if pose_seed is None else pose_seed)
    c2w = unnorm @ c2w_n  # world pose whose normalised version is c2w_n (up to rounding)
    fx = dict(H=H, W=W, S=S, stop_layer=stop_layer, K=K, c2w=c2w, unnorm=unnorm, app=int(app), white_bg=int(ren.white_bg))

    # R1-R3
    c2w_norm = unnorm.inverse() @ c2w
    rays = ru.sample_nerf_rays(H, W, K, c2w_norm, ds=8, embed_type="mip")
    fx["c2w_norm"] = c2w_norm
    fx["rays"] = rays
    R = rays.shape[0]
    # the random draws of one predict(): torch.rand(R,S+1) then empty(R,S+1).uniform_(to=1/(S+1)-eps)
    rng_seed = 1000 + seed
    torch.manual_seed(rng_seed)
    t_rand = torch.rand(R, S + 1)
    jitter = torch.empty(R, S + 1).uniform_(to=(1 / (S + 1) - torch.finfo(torch.float32).eps))
    fx["t_rand"], fx["jitter"] = t_rand, jitter

    # R4a/R4b coarse sampling on its own
    torch.manual_seed(rng_seed)
    (mean_c, var_c), t_c = ru.sample_smth_along_rays(rays, num_pts=S, embed_type="mip", model_type="coarse")
    NS = sub_rays * S  # per-sample arrays are stored for the first `sub_rays` rays only (fixture size)
    fx.update(t_coarse=t_c, mean_coarse=mean_c[:sub_rays], var_coarse=var_c[:sub_rays], sub_rays=sub_rays)

    # N0
    x_pts = ren.xyz_encoder(mean_c.reshape(-1, 3), y=var_c.reshape(-1, 3))[0]
    x_dir = ren.dirs_encoder(rays[:, 8:11])
    fx.update(ipe_coarse=x_pts[:NS], dir_pe=x_dir)

    # N1 on the coarse samples with the FINE network (exercises stop_layer) and the coarse one
    view = rays[:, 8:11][:, None, :].expand(R, S, 3).reshape(-1, 3)
    inp = torch.cat([x_pts, ren.dirs_encoder(view)], -1)
    app_row = None
    if app:
        app_row = ren.embedding_a(torch.ones(1).long())[0]
        inp = torch.cat([inp, app_row[None].expand(R * S, -1)], -1)
        fx["app_row"] = app_row
    raw_f, feat_f = ren.nerf_fine(inp, ret_pfeat=1, val=True)
    raw_c, feat_c = ren.nerf_coarse(inp, ret_pfeat=1, val=True)
    fx.update(mlp_raw_fine=raw_f[:NS], mlp_feat_fine=feat_f[:NS], mlp_raw_coarse=raw_c[:NS], mlp_feat_coarse=feat_c[:NS])

    # R6
    rgb, disp, acc, w, depth, _ = ru.volume_render_radiance_field(
        raw_c.reshape(R, S, 4), t_c, rays[:, 3:6], noise_std=0.0, white_bg=ren.white_bg, embed_type="mip", input_dim=4
    )
    fx.update(comp_rgb=rgb, comp_acc=acc, comp_weights=w, comp_depth=depth)

    # R5 on those weights
    torch.manual_seed(rng_seed)
    torch.rand(R, S + 1)
    (mean_f, var_f), t_f = ru.sample_smth_along_rays(
        rays, num_pts=S, z_vals=t_c, weights=w, embed_type="mip", model_type="fine"
    )
    fx.update(t_fine=t_f, mean_fine=mean_f[:sub_rays], var_fine=var_f[:sub_rays])
    w_f = None
    if style is not None:
        # the fine pass's own compositing weights and accumulated opacity (predict() does not return them in validation mode)
        xf = ren.xyz_encoder(mean_f.reshape(-1, 3), y=var_f.reshape(-1, 3))[0]
        inp_f = torch.cat([xf, ren.dirs_encoder(view)], -1)
        if app:
            inp_f = torch.cat([inp_f, app_row[None].expand(R * S, -1)], -1)
        raw_ff, _ = ren.nerf_fine(inp_f, ret_pfeat=1, val=True)
        _, _, acc_f, w_f, _, _ = ru.volume_render_radiance_field(
            raw_ff.reshape(R, S, 4), t_f, rays[:, 3:6], noise_std=0.0, white_bg=ren.white_bg, embed_type="mip", input_dim=4
        )
        fx.update(fine_weights=w_f, fine_acc=acc_f, fine_sigma=raw_ff.reshape(R, S, 4)[:, :, 3])

    # R7: full predict (ret_pfeat on) and R8: render_novel_view, same draws
    ren.ret_pfeat = True
    torch.manual_seed(rng_seed)
    preds = ren.predict(rays, W // 8, H // 8, out_raw=True)
    for k, v in preds.items():
        fx[f"pred_{k}"] = v
    torch.manual_seed(rng_seed)
    nv = ren.render_novel_view((H, W), K, c2w, unnorm, torch.device("cpu"), downsample=8)
    fx.update(nv_im_pred=nv["im_pred"], nv_pt3d=nv["pt3d"], nv_pt_feat=nv["pt_feat"])
    fx["weights_seed"] = seed
    fx["style"] = "" if style is None else style
    fx["density_shift"] = density_shift
    np.savez_compressed(OUT / f"nerf_{tag}.npz", **to_np(fx))
    w_c, w_f = w, (w if w_f is None else w_f)
    print(f"nerf_{tag}: R={R} S={S} app={app} keys={len(fx)} missing={missing}  |tap|max={float(feat_f.abs().max()):.2f} "
          f"|last|max={float(feat_c.abs().max()):.2f} sigma {float(raw_c[:, 3].min()):.0f}..{float(raw_c[:, 3].max()):.0f}  "
          f"median max-weight coarse {float(w_c.max(-1)[0].median()):.3f} fine {float(w_f.max(-1)[0].median()):.3f}")


def surface_seed_fixture(seed, pose_seed, H=64, W=96, S=64, focal=90.0, density_shift=None):
    """Round 4: one small trained-like ("surface" style) fixture per (weight seed, pose) -- the reference's fp32 render and, next
    to it, the SAME reference modules evaluated in float64 on the same fp32 inputs (IPE values and fence posts of the fp32 run):
    the "truth" that says how far the reference's own fp32 arithmetic is from the exact result in this ill-conditioned regime
    (activations ~20, density head x2600).  Stored as fp32 differences truth - fp32 (small numbers: no precision lost).
    Reference path: nerf/renderer.py:182-295 (predict), models/nerf.py:94-144, render_utils.py:176-230."""
    import copy
    from nerfmatch.nerf.renderer import NerfRenderer
    from nerfmatch.nerf import render_utils as ru

    torch.set_grad_enabled(False)
    cfg = synth.nerf_config("7scenes", num_pts=S, img_wh=(W, H))
    K = torch.tensor([[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1]])
    unnorm = synth.unnorm_scene()
    c2w = unnorm @ synth.camera_pose(seed=pose_seed)
    c2w_norm = unnorm.inverse() @ c2w
    rays = ru.sample_nerf_rays(H, W, K, c2w_norm, ds=8, embed_type="mip")
    R = rays.shape[0]
    if density_shift is None:
        # SURFACE_STYLE's density bias was tuned on weight seed 0; other seeds get their own shift so that ~25 % of the sampled
        # space is occupied (the 75 % quantile of the raw density along these rays becomes 0), rounded to an integer
        ren0 = NerfRenderer(cfg, num_frames=None, training=False, stop_layer=3)
        ren0.load_state_dict(synth.nerf_state_dict(seed=seed, style="surface"), strict=True)
        torch.manual_seed(1)
        (m0, v0), _ = ru.sample_smth_along_rays(rays, num_pts=S, embed_type="mip", model_type="coarse")
        x0 = ren0.xyz_encoder(m0.reshape(-1, 3), y=v0.reshape(-1, 3))[0]
        raw0 = ren0.nerf_coarse(torch.cat([x0, ren0.dirs_encoder(rays[:, 8:11][:, None, :].expand(R, S, 3).reshape(-1, 3))], -1), val=True)
        density_shift = -float(torch.round(torch.quantile(raw0[:, 3], 0.75)))
    sd = synth.nerf_state_dict(seed=seed, density_bias=density_shift, style="surface")
    ren = NerfRenderer(cfg, num_frames=None, training=False, stop_layer=3)
    ren.load_state_dict(sd, strict=True)
    ren.eval()
    rng_seed = 2000 + 17 * seed + pose_seed
    torch.manual_seed(rng_seed)
    t_rand = torch.rand(R, S + 1)
    jitter = torch.empty(R, S + 1).uniform_(to=(1 / (S + 1) - torch.finfo(torch.float32).eps))
    ren.ret_pfeat = True
    torch.manual_seed(rng_seed)
    preds = ren.predict(rays, W // 8, H // 8, out_raw=True)
    fx = dict(H=H, W=W, S=S, stop_layer=3, K=K, c2w=c2w, unnorm=unnorm, c2w_norm=c2w_norm, rays=rays, t_rand=t_rand, jitter=jitter,
              app=0, white_bg=0, weights_seed=seed, pose_seed=pose_seed, style="surface", density_shift=density_shift)
    # the two passes again, stage by stage, fp32 (for the fence posts / weights, which predict() does not return) and fp64
    torch.manual_seed(rng_seed)
    (mean_c, var_c), t_c = ru.sample_smth_along_rays(rays, num_pts=S, embed_type="mip", model_type="coarse")
    view = rays[:, 8:11][:, None, :].expand(R, S, 3).reshape(-1, 3)
    xdir = ren.dirs_encoder(view)
    nets64 = {k: copy.deepcopy(getattr(ren, f"nerf_{k}")).double() for k in ("coarse", "fine")}

    def one_pass(key, mean, var, t):
        x = ren.xyz_encoder(mean.reshape(-1, 3), y=var.reshape(-1, 3))[0]
        inp = torch.cat([x, xdir], -1)
        raw, feat = getattr(ren, f"nerf_{key}")(inp, ret_pfeat=1, val=True)
        out32 = ru.volume_render_radiance_field(raw.reshape(R, S, 4), t, rays[:, 3:6], noise_std=0.0, white_bg=False, embed_type="mip", input_dim=4)
        raw64, feat64 = nets64[key](inp.double(), ret_pfeat=1, val=True)
        out64 = ru.volume_render_radiance_field(raw64.reshape(R, S, 4), t.double(), rays[:, 3:6].double(), noise_std=0.0, white_bg=False,
                                                embed_type="mip", input_dim=4)
        w32, w64 = out32[3], out64[3]
        f32 = (w32[..., None] * feat.reshape(R, S, -1)).sum(-2)
        f64 = (w64[..., None] * feat64.reshape(R, S, -1)).sum(-2)
        return dict(w=w32, feat=f32, sigma=raw.reshape(R, S, 4)[..., 3], w_d=(w64 - w32.double()).float(), feat_d=(f64 - f32.double()).float(),
                    sigma_d=(raw64.reshape(R, S, 4)[..., 3] - raw.reshape(R, S, 4)[..., 3].double()).float(), act_max=float(feat64.abs().max()))

    pc = one_pass("coarse", mean_c, var_c, t_c)
    torch.manual_seed(rng_seed)
    torch.rand(R, S + 1)
    (mean_f, var_f), t_f = ru.sample_smth_along_rays(rays, num_pts=S, z_vals=t_c, weights=pc["w"], embed_type="mip", model_type="fine")
    pf = one_pass("fine", mean_f, var_f, t_f)
    assert torch.equal(pc["feat"], preds["feat_coarse"]) and torch.equal(pf["feat"], preds["feat_fine"]), "stage-by-stage replay differs from predict()"
    fx.update(t_coarse=t_c, t_fine=t_f, comp_weights=pc["w"], fine_weights=pf["w"], sigma_coarse=pc["sigma"], sigma_fine=pf["sigma"])
    for k in ("feat_coarse", "feat_fine", "pts_coarse", "pts_fine", "rgb_coarse", "rgb_fine", "depth_coarse", "depth_fine"):
        fx[f"pred_{k}"] = preds[k]
    fx.update(truth_d_weights_coarse=pc["w_d"], truth_d_weights_fine=pf["w_d"], truth_d_feat_coarse=pc["feat_d"], truth_d_feat_fine=pf["feat_d"],
              truth_d_sigma_coarse=pc["sigma_d"], truth_d_sigma_fine=pf["sigma_d"])
    tag = f"surf_w{seed}_p{pose_seed}"
    np.savez_compressed(OUT / f"nerf_{tag}.npz", **to_np(fx))
    print(f"nerf_{tag}: R={R} S={S}  |h7|max {pc['act_max']:.1f} |h3|max {pf['act_max']:.1f}  sigma {float(pc['sigma'].min()):.0f}..{float(pc['sigma'].max()):.0f}  "
          f"median max-weight coarse {float(pc['w'].max(-1)[0].median()):.2f} fine {float(pf['w'].max(-1)[0].median()):.2f}  |feat| coarse {float(pc['feat'].abs().max()):.1f} fine {float(pf['feat'].abs().max()):.1f}  "
          f"REFERENCE fp32 vs its own fp64: weights {float(pc['w_d'].abs().max()):.1e}/{float(pf['w_d'].abs().max()):.1e}  feat {float(pc['feat_d'].abs().max()):.1e}/{float(pf['feat_d'].abs().max()):.1e}")


def fine_count_fixture():
    """coarse_nerf.num_pts != fine_nerf.num_pts in a mip configuration: the reference's fine pass still has the coarse count
    (render_utils.py:299-309, :594-597).  Stores the reference's predict() outputs for (32, 64) -- and asserts here, at generation time,
    that they EQUAL its outputs for (32, 32) on the same random draws."""
    from nerfmatch.nerf.renderer import NerfRenderer
    from nerfmatch.nerf import render_utils as ru

    torch.set_grad_enabled(False)
    H, W, seed = 32, 64, 0
    sd = synth.nerf_state_dict(seed=seed, density_bias=3.0)
    K = torch.tensor([[60.0, 0, W / 2], [0, 60.0, H / 2], [0, 0, 1]])
    unnorm = synth.unnorm_scene()
    c2w_norm = unnorm.inverse() @ (unnorm @ synth.camera_pose(seed=10))
    rays = ru.sample_nerf_rays(H, W, K, c2w_norm, ds=8, embed_type="mip")
    outs = {}
    for sf in (32, 64):
        ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=32, img_wh=(W, H), num_pts_fine=sf), training=False, stop_layer=3)
        ren.load_state_dict(sd, strict=True)
        ren.eval()
        ren.ret_pfeat = True
        torch.manual_seed(1234)
        outs[sf] = ren.predict(rays, W // 8, H // 8, out_raw=True)
    for k in outs[32]:
        assert torch.equal(outs[32][k], outs[64][k]), k
    R = rays.shape[0]
    torch.manual_seed(1234)
    t_rand = torch.rand(R, 33)
    jitter = torch.empty(R, 33).uniform_(to=(1 / 33 - torch.finfo(torch.float32).eps))
    fx = dict(H=H, W=W, S_coarse=32, S_fine=64, rays=rays, t_rand=t_rand, jitter=jitter, weights_seed=seed, **{f"pred_{k}": v for k, v in outs[64].items()})
    np.savez_compressed(OUT / "nerf_fine_count_c32_f64.npz", **to_np(fx))
    print("nerf_fine_count_c32_f64: the reference's outputs for fine_nerf.num_pts = 64 equal those for 32 (fine pass has", outs[64]["feat_fine"].shape, ")")


SURFACE_SEEDS = [(1, 21), (1, 22), (2, 23), (2, 24), (3, 25), (3, 26), (4, 27), (4, 28), (5, 29), (5, 30)]  # (weight seed, pose seed)


def far_fallback_fixture():
    """Camera outside the unit sphere looking away: the discriminant goes negative somewhere, the
    reference's assert fires and EVERY ray gets far = 1 (render_utils.py:62-68)."""
    from nerfmatch.nerf import render_utils as ru

    H, W = 32, 32
    K = torch.tensor([[60.0, 0, W / 2], [0, 60.0, H / 2], [0, 0, 1]])
    c2w = torch.eye(4)
    c2w[:3, :3] = torch.tensor([[0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [-1.0, 0.0, 0.0]])  # camera looks along +x
    c2w[:3, 3] = torch.tensor([0.0, 0.0, 1.5])
    rays = ru.sample_nerf_rays(H, W, K, c2w, ds=8, embed_type="mip")
    np.savez_compressed(OUT / "nerf_far_fallback.npz", H=H, W=W, K=K.numpy(), c2w=c2w.numpy(), rays=rays.numpy())
    print("nerf_far_fallback: far ->", rays[:, 7].unique())


# ----------------------------------------------------------------------------- matcher half
def matcher_fixtures(seed=0):
    import nerfmatch.nerfmatch_c2f_trainer as c2f
    import nerfmatch.nerfmatch_coarse_trainer as crs
    from nerfmatch.modules.attention import GenericEncoderLayer
    from nerfmatch.utils.geometry import get_pixel_coords_grid

    torch.set_grad_enabled(False)
    g = torch.Generator().manual_seed(77 + seed)
    h, w, N = 6, 8, 64  # M = 48 image tokens, N = 64 points (M != N catches transposes)
    Himg, Wimg = h * 8, w * 8
    cfeat = torch.randn(1, 256, h, w, generator=g)
    ffeat = torch.randn(1, 128, h * 4, w * 4, generator=g)
    pt_feat = torch.relu(torch.randn(1, N, 256, generator=g))
    pt3d = torch.randn(1, N, 3, generator=g) * 2.0
    # plant real correspondences so that mutual matches exist: image token i ~ point perm[i]
    perm = torch.randperm(N, generator=g)[: h * w]
    planted = cfeat.flatten(-2).permute(0, 2, 1).clone()
    pt_feat[0, perm[:30]] = torch.relu(planted[0, :30]) + 0.05 * torch.randn(30, 256, generator=g)
    img = torch.zeros(1, 3, Himg, Wimg)
    pt2d = get_pixel_coords_grid(Wimg, Himg, ds=8).reshape(1, -1, 2)
    im_mask = torch.ones(1, h * w, dtype=torch.bool)
    pt_mask = torch.ones(1, N, dtype=torch.bool)
    im_mask_p = im_mask.clone()
    im_mask_p[0, -7:] = False
    pt_mask_p = pt_mask.clone()
    pt_mask_p[0, 5:11] = False

    # ---- c2f
    THR = 0.004
    cfg = synth.matcher_config("c2f")
    sd = synth.matcher_state_dict("c2f", seed=seed)
    c2f.init_backbone_8_2 = lambda *a, **k: FixedBackbone((cfeat, ffeat), [256, 128])
    model = c2f.NeRFMatcherMS(cfg)
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res
    assert all(k.startswith("im_sa.") for k in res.missing_keys), res.missing_keys  # aliases of pt_sa
    model.eval()
    fx = dict(cfeat=cfeat, ffeat=ffeat, pt_feat=pt_feat, pt3d=pt3d, pt2d=pt2d, weights_seed=seed, thr=THR,
              im_mask_partial=im_mask_p, pt_mask_partial=pt_mask_p)
    for tag, mutual, thr, imm, ptm in (("mut", True, 0.0, im_mask, pt_mask), ("nomut", False, 0.0, im_mask, pt_mask),
                                       ("mask", True, 0.0, im_mask_p, pt_mask_p), ("thr", True, THR, im_mask, pt_mask), ("empty", True, 0.5, im_mask, pt_mask)):
        data = dict(image=img, im_mask=imm, pt3d=pt3d.clone(), pt_feat=pt_feat.clone(), pt_mask=ptm, pt2d=pt2d)
        model.forward(data, ret_feats=True, mutual=mutual, match_thres=thr)
        b, i, j = data["match_ids"]
        fx.update({f"{tag}_b_ids": b, f"{tag}_i_ids": i, f"{tag}_j_ids": j, f"{tag}_mconf": data["mconf"],
                   f"{tag}_expec_f": data["expec_f"], f"{tag}_mpt2d_f": data["mpt2d_f"], f"{tag}_mpt2d_c": data["mpt2d_c"],
                   f"{tag}_mpt3d": data["mpt3d"], f"{tag}_m_bids": data["m_bids"]})
        if tag in ("mut", "mask"):
            fx[f"{tag}_conf"] = data["conf_matrix"]
            fx[f"{tag}_im_cfeat"] = data["im_cfeat"]
            fx[f"{tag}_pt_cfeat"] = data["pt_cfeat"]
        print(f"c2f {tag}: matches={len(b)} mconf range {float(data['mconf'].min()) if len(b) else 0:.4f}..{float(data['mconf'].max()) if len(b) else 0:.4f}")
    # per-row intermediates
    im_tok, _ = model.extract_im_feat(img)
    pt_tok = model.extract_pt_feat(pt_feat, pt3d)
    fx.update(im_tokens_sa=im_tok, pt_tokens_sa=pt_tok)
    fx["pe_table"] = model.im_pe.pe[0, :, :h, :w]
    fx["fourier_pt3d"] = model.pt_pe(pt3d)
    x0 = cfeat.flatten(-2).permute(0, 2, 1)
    fx["enc_self_in"] = x0
    fx["enc_self_out"] = model.pt_sa.layers[0](x0)
    fx["enc_cross_out"] = model.coarse_former(x0, pt_feat)
    np.savez_compressed(OUT / "matcher_c2f.npz", **to_np(fx))

    # ---- LSA encoder layer (A2) on fine-sized tokens
    lsa = GenericEncoderLayer(model_dim=128, head_dim=16, att_type="lsa", att_mode="self")
    rng = np.random.default_rng(seed + 5)
    lsa_sd = {}
    synth._encoder_layer(lsa_sd, rng, "L", 128)
    lsa_sd = {k[2:]: v for k, v in lsa_sd.items()}
    lsa_sd["attention.attend.scale"] = torch.log(torch.tensor(16**-0.5)) + 0.1
    lsa.load_state_dict(lsa_sd, strict=True)
    xin = torch.randn(6, 25, 128, generator=g)
    np.savez_compressed(OUT / "matcher_lsa.npz", x=xin.numpy(), y=lsa(xin).numpy(), weights_seed=seed + 5,
                        scale=lsa_sd["attention.attend.scale"].numpy())

    # ---- coarse-only (Mini)
    cfgc = synth.matcher_config("coarse")
    crs.init_backbone = lambda *a, **k: FixedBackbone(cfeat, 256)
    mini = crs.NeRFMatcherCoarse(cfgc)
    mini.load_state_dict(synth.matcher_state_dict("coarse", seed=seed), strict=False)
    mini.eval()
    # Mini consumes raw features: use normalised relu features like BASELINE config C2
    fxc = dict(cfeat=cfeat, pt_feat=pt_feat)
    for tag, mutual in (("mut", True), ("nomut", False)):
        data = dict(image=img, im_mask=im_mask, pt3d=pt3d.clone(), pt_feat=pt_feat.clone(), pt_mask=pt_mask, pt2d=pt2d)
        mini.forward(data, mutual=mutual)
        b, i, j = data["match_ids"]
        fxc.update({f"{tag}_b_ids": b, f"{tag}_i_ids": i, f"{tag}_j_ids": j, f"{tag}_mconf": data["mconf"]})
        if mutual:
            fxc["conf"] = data["conf_matrix"]
        print(f"coarse {tag}: matches={len(b)}")
    np.savez_compressed(OUT / "matcher_coarse.npz", **to_np(fxc))


def postnorm_fixture(seed=8):
    """Round 5: post-norm encoder layers (attention.py:209-221; no shipped yaml selects them), self and cross attention, 256-d."""
    from nerfmatch.modules.attention import GenericEncoderLayer
    from nerfmatch_amd import synth

    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    fx = dict(weights_seed=seed)
    for mode in ("self", "cross"):
        layer = GenericEncoderLayer(model_dim=256, context_dim=256, head_dim=32, norm_type="post", att_type="full", att_mode=mode)
        sd = {}
        synth._encoder_layer(sd, rng, "L", 256, cross=False)  # (post-norm: norm1 holds ONE LayerNorm also in cross mode, attention.py:195-198)
        layer.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
        x, c = torch.randn(2, 64, 256, generator=g), torch.randn(2, 40, 256, generator=g)
        fx[f"{mode}_x"] = x
        if mode == "cross":
            fx["cross_c"] = c
        fx[f"{mode}_y"] = layer(x) if mode == "self" else layer(x, c)
    np.savez_compressed(OUT / "matcher_postnorm.npz", **to_np(fx))
    print("matcher_postnorm:", {k: tuple(v.shape) for k, v in fx.items() if hasattr(v, "shape")})


def envelope_fixture(seed=0):
    """Round 6: the option values the reference's constructors accept beyond the shipped yamls -- pt_ftype pe3d / pt3d
    with the pt_proj layer, pt_pe_type "id", a pre-attention PE, pt_feat_norm (synth.matcher_variant) -- run through the reference's own
    model classes: point tokens, confidence matrix, mutual match lists, scores and (c2f) the fine-stage outputs."""
    import nerfmatch.nerfmatch_c2f_trainer as c2f
    import nerfmatch.nerfmatch_coarse_trainer as crs
    from nerfmatch.utils.geometry import get_pixel_coords_grid

    torch.set_grad_enabled(False)
    g = torch.Generator().manual_seed(177 + seed)
    h, w, N = 6, 8, 64
    Himg, Wimg = h * 8, w * 8
    cfeat = torch.randn(1, 256, h, w, generator=g)
    ffeat = torch.randn(1, 128, h * 4, w * 4, generator=g)
    pt3d = torch.randn(1, N, 3, generator=g) * 2.0
    feat256 = torch.relu(torch.randn(1, N, 256, generator=g))
    feat128 = torch.relu(torch.randn(1, N, 128, generator=g))
    img = torch.zeros(1, 3, Himg, Wimg)
    pt2d = get_pixel_coords_grid(Wimg, Himg, ds=8).reshape(1, -1, 2)
    im_mask, pt_mask = torch.ones(1, h * w, dtype=torch.bool), torch.ones(1, N, dtype=torch.bool)
    fx = dict(cfeat=cfeat, ffeat=ffeat, pt3d=pt3d, feat256=feat256, feat128=feat128, pt2d=pt2d, weights_seed=seed)
    for name in synth.MATCHER_VARIANTS:
        cfg, sd = synth.matcher_variant(name, seed)
        pf = feat128 if name == "nerf128_id" else feat256
        if name == "coarse_norm":
            crs.init_backbone = lambda *a, **k: FixedBackbone(cfeat, 256)
            model = crs.NeRFMatcherCoarse(cfg)
        else:
            c2f.init_backbone_8_2 = lambda *a, **k: FixedBackbone((cfeat, ffeat), [256, 128])
            model = c2f.NeRFMatcherMS(cfg)
        res = model.load_state_dict(sd, strict=False)
        assert not res.unexpected_keys and all(k.startswith("im_sa.") for k in res.missing_keys), (name, res)
        model.eval()
        fx[f"{name}_pt_tokens"] = model.extract_pt_feat(pf.clone(), pt3d.clone())
        data = dict(image=img, im_mask=im_mask, pt3d=pt3d.clone(), pt_feat=pf.clone(), pt_mask=pt_mask, pt2d=pt2d)
        if name == "coarse_norm":
            model.forward(data, mutual=True)
            fx[f"{name}_pt3d_after"], fx[f"{name}_pt_feat_after"] = data["pt3d"], data["pt_feat"]  # centred in place by feature_normalization
        else:
            model.forward(data, ret_feats=False, mutual=True, match_thres=0.0)
            fx.update({f"{name}_expec_f": data["expec_f"], f"{name}_mpt2d_f": data["mpt2d_f"], f"{name}_mpt3d": data["mpt3d"]})
        b, i, j = data["match_ids"]
        fx.update({f"{name}_conf": data["conf_matrix"], f"{name}_i_ids": i, f"{name}_j_ids": j, f"{name}_mconf": data["mconf"]})
        print(f"envelope {name}: {len(i)} mutual matches, conf max {float(data['conf_matrix'].max()):.4f}")
    np.savez_compressed(OUT / "matcher_envelope.npz", **to_np(fx))


def layer_grads_fixture(seed=12):
    """Round 6: gradients of ONE encoder layer through the reference's own modules and torch autograd for the two training options outside
    the shipped yamls -- att_type "lsa" (the learnable log-scale gets a gradient; fine-stage shape: 128-d, 8 heads of 16, 25-token windows, and
    a 256-d / 96-token case on the large-sequence kernels, a subset of its parameter gradients kept) and act_fn "relu": loss = sum(y * g)."""
    from nerfmatch.modules.attention import GenericEncoderLayer
    from nerfmatch_amd import synth

    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    fx = dict(weights_seed=seed)
    torch.set_grad_enabled(True)
    for tag, dim, hd, att, act, B, L in (("lsa128", 128, 16, "lsa", "gelu", 6, 25), ("lsa256", 256, 32, "lsa", "gelu", 2, 96), ("relu128", 128, 16, "full", "relu", 6, 25)):
        layer = GenericEncoderLayer(model_dim=dim, head_dim=hd, att_type=att, att_mode="self", act_fn=act)
        sd = {}
        synth._encoder_layer(sd, rng, "L", dim)
        sd = {k[2:]: v for k, v in sd.items()}
        if att == "lsa":
            sd["attention.attend.scale"] = torch.log(torch.tensor(hd**-0.5)) + 0.2
        layer.load_state_dict(sd, strict=True)
        x = torch.randn(B, L, dim, generator=g).requires_grad_(True)
        gy = torch.randn(B, L, dim, generator=g)
        y = layer(x)
        (y * gy).sum().backward()
        fx.update({f"{tag}_x": x.detach(), f"{tag}_gy": gy, f"{tag}_y": y.detach(), f"{tag}_dx": x.grad})
        for n, p_ in layer.named_parameters():
            if dim == 128 or n in ("attention.attend.scale", "attention.proj_q.weight", "norm2.weight", "feedforward.layers.0.bias"):  # (file size)
                fx[f"{tag}_d.{n}"] = p_.grad
    torch.set_grad_enabled(False)
    np.savez_compressed(OUT / "matcher_layer_grads.npz", **to_np(fx))
    print("matcher_layer_grads:", sorted(k for k in fx if k.endswith("_dx")))


def peaked_matcher_fixture(seed=0):
    """Round 3: the c2f and the coarse-only model in a PEAKED-confidence regime, M = 320 image tokens x N = 352 points
    (11 key tiles of 32, 3 GEMM row tiles of 128), weights `style="aligned"` (synth.matcher_state_dict), 280 of the 320
    image tokens planted on points with noise 0.25, temperature 15: most rows are mutual matches with a row maximum near 1,
    the unplanted rows stay diffuse.  Run through the reference's own NeRFMatcherMS.forward / NeRFMatcherCoarse.forward."""
    import nerfmatch.nerfmatch_c2f_trainer as c2f
    import nerfmatch.nerfmatch_coarse_trainer as crs
    from nerfmatch.utils.geometry import get_pixel_coords_grid

    torch.set_grad_enabled(False)
    g = torch.Generator().manual_seed(377 + seed)
    h, w, N, n_plant, noise, temp = 16, 20, 352, 280, 0.25, 15.0
    M = h * w
    Himg, Wimg = h * 8, w * 8
    cfeat = torch.randn(1, 256, h, w, generator=g)
    ffeat = torch.randn(1, 128, h * 4, w * 4, generator=g)
    pt_feat = torch.randn(1, N, 256, generator=g)
    pt3d = torch.randn(1, N, 3, generator=g) * 2.0
    perm = torch.randperm(N, generator=g)[:M]
    tok = cfeat.flatten(-2).permute(0, 2, 1)
    pt_feat[0, perm[:n_plant]] = tok[0, :n_plant] + noise * torch.randn(n_plant, 256, generator=g)
    img = torch.zeros(1, 3, Himg, Wimg)
    pt2d = get_pixel_coords_grid(Wimg, Himg, ds=8).reshape(1, -1, 2)
    im_mask = torch.ones(1, M, dtype=torch.bool)
    pt_mask = torch.ones(1, N, dtype=torch.bool)
    im_mask_p = im_mask.clone()
    im_mask_p[0, 100:131] = False
    pt_mask_p = pt_mask.clone()
    pt_mask_p[0, 7:40] = False

    cfg = synth.matcher_config("c2f")
    sd = synth.matcher_state_dict("c2f", seed=seed, temperature=temp, style="aligned")
    c2f.init_backbone_8_2 = lambda *a, **k: FixedBackbone((cfeat, ffeat), [256, 128])
    model = c2f.NeRFMatcherMS(cfg)
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and all(k.startswith("im_sa.") for k in res.missing_keys), res
    model.eval()
    fx = dict(cfeat=cfeat, ffeat=ffeat, pt_feat=pt_feat, pt3d=pt3d, pt2d=pt2d, weights_seed=seed, temperature=temp, perm=perm,
              n_plant=n_plant, im_mask_partial=im_mask_p, pt_mask_partial=pt_mask_p, thr=0.2)
    for tag, mutual, thr, imm, ptm in (("mut", True, 0.0, im_mask, pt_mask), ("nomut", False, 0.0, im_mask, pt_mask),
                                       ("mask", True, 0.0, im_mask_p, pt_mask_p), ("thr", True, 0.2, im_mask, pt_mask)):
        data = dict(image=img, im_mask=imm, pt3d=pt3d.clone(), pt_feat=pt_feat.clone(), pt_mask=ptm, pt2d=pt2d)
        model.forward(data, ret_feats=True, mutual=mutual, match_thres=thr)
        b, i, j = data["match_ids"]
        fx.update({f"{tag}_b_ids": b, f"{tag}_i_ids": i, f"{tag}_j_ids": j, f"{tag}_mconf": data["mconf"],
                   f"{tag}_expec_f": data["expec_f"], f"{tag}_mpt2d_f": data["mpt2d_f"], f"{tag}_mpt2d_c": data["mpt2d_c"],
                   f"{tag}_mpt3d": data["mpt3d"], f"{tag}_m_bids": data["m_bids"]})
        if tag in ("mut", "mask"):
            fx[f"{tag}_conf"] = data["conf_matrix"]
        if tag == "mut":
            fx["mut_im_cfeat"], fx["mut_pt_cfeat"] = data["im_cfeat"], data["pt_cfeat"]
        rm = data["conf_matrix"][0].max(1)[0]
        print(f"peaked c2f {tag}: matches={len(b)}/{M} planted-correct={(perm[i] == j).sum().item()} row-max median {float(rm.median()):.3f} "
              f"rows with max >= 0.2: {float((rm >= 0.2).float().mean()):.2f}  mconf {float(data['mconf'].min()):.4f}..{float(data['mconf'].max()):.4f}")
    np.savez_compressed(OUT / "matcher_peaked.npz", **to_np(fx))

    # coarse-only model on the same (aligned) tokens, L2-normalised by the model itself
    crs.init_backbone = lambda *a, **k: FixedBackbone(cfeat, 256)
    mini = crs.NeRFMatcherCoarse(synth.matcher_config("coarse"))
    mini.load_state_dict(synth.matcher_state_dict("coarse", seed=seed, temperature=temp), strict=False)
    mini.eval()
    fxc = dict(temperature=temp)
    for tag, mutual in (("mut", True), ("nomut", False)):
        data = dict(image=img, im_mask=im_mask, pt3d=pt3d.clone(), pt_feat=pt_feat.clone(), pt_mask=pt_mask, pt2d=pt2d)
        mini.forward(data, mutual=mutual)
        b, i, j = data["match_ids"]
        fxc.update({f"{tag}_b_ids": b, f"{tag}_i_ids": i, f"{tag}_j_ids": j, f"{tag}_mconf": data["mconf"]})
        if mutual:
            fxc["conf"] = data["conf_matrix"]
        print(f"peaked coarse {tag}: matches={len(b)} row-max median {float(data['conf_matrix'][0].max(1)[0].median()):.3f}")
    np.savez_compressed(OUT / "matcher_peaked_coarse.npz", **to_np(fxc))


def inerf_fixture(tag, scene_type, H, W, seed, num_optim=3, lrate=0.002, lrdecay=False, use_match_loss=False, matcher="c2f"):
    """The reference's own `NeRFMatchEvaluator.inerf_refinement` (nerfmatch_evaluator.py:288-500) run for a few Adam
    steps on a synthetic scene, with `eval_pose=True` (pose error from the refined pose, no matcher in the loop).
    The method is called unbound on a minimal stand-in for `self` (it uses self.device, self.gen_rays, self.timer)."""
    from collections import defaultdict
    from functools import partial

    from nerfmatch.nerf.renderer import NerfRenderer
    from nerfmatch import nerfmatch_evaluator as ne

    app = scene_type == "cambridge"
    S = 128  # hard-coded in the reference (:354, :360)
    cfg = synth.nerf_config(scene_type, num_pts=S, img_wh=(W, H))
    sd = synth.nerf_state_dict(seed=seed, app_vocab=5 if app else 0, density_bias=3.0)
    torch.set_grad_enabled(False)  # the evaluator disables grad globally; inerf re-enables it locally
    ren = NerfRenderer(cfg, num_frames=5 if app else None, training=False, stop_layer=3)
    ren.load_state_dict(sd, strict=True)
    ren.eval()
    K = torch.tensor([[60.0, 0, W / 2], [0, 60.0, H / 2], [0, 0, 1]])
    unnorm = synth.unnorm_scene()
    c2w_gt = unnorm @ synth.camera_pose(seed=seed + 20)
    c2w_est = unnorm @ synth.camera_pose(seed=seed + 21)  # a different, nearby pose to start from
    g = torch.Generator().manual_seed(seed + 5)
    image = torch.rand(1, 3, H, W, generator=g)
    R = (H // 8) * (W // 8)
    rng_seed = 3000 + seed
    # the draws of step j: torch.rand(R,129) (coarse jitter) then empty(R,129).uniform_(to=1/129-eps) (resampler)
    torch.manual_seed(rng_seed)
    t_rands, jitters = [], []
    for _ in range(num_optim):
        t_rands.append(torch.rand(R, S + 1))
        jitters.append(torch.empty(R, S + 1).uniform_(to=(1 / (S + 1) - torch.finfo(torch.float32).eps)))
    fake = types.SimpleNamespace(device=torch.device("cpu"), timer=defaultdict(list))
    fake.gen_rays = partial(ne.NeRFMatchEvaluator.gen_rays, fake)
    conf = Namespace(lrate=lrate, lrdecay=lrdecay, num_optim=num_optim, eval_pose=True, use_match_loss=use_match_loss, ds=8)
    batch = dict(c2w=c2w_gt[None], K=K[None], image=image)
    extra = {}
    if use_match_loss:
        # the matching term (:420-441): the c2f matcher (stand-in backbone returning fixed maps) on the rendered features
        import nerfmatch.nerfmatch_c2f_trainer as c2f

        cfeat = torch.randn(1, 256, H // 8, W // 8, generator=g)
        ffeat = torch.randn(1, 128, H // 2, W // 2, generator=g)
        if matcher == "c2f":
            c2f.init_backbone_8_2 = lambda *a, **k: FixedBackbone((cfeat, ffeat), [256, 128])
            model = c2f.NeRFMatcherMS(synth.matcher_config("c2f"))
            model.load_state_dict(synth.matcher_state_dict("c2f", seed=seed), strict=False)
        else:  # round 6: the coarse-only model class in the matching term (synth.matcher_variant("coarse_full"): PE, one self / cross layer)
            import nerfmatch.nerfmatch_coarse_trainer as crs

            crs.init_backbone = lambda *a, **k: FixedBackbone(cfeat, 256)
            mcfg, msd = synth.matcher_variant(matcher, seed)
            model = crs.NeRFMatcherCoarse(mcfg)
            res = model.load_state_dict(msd, strict=False)
            assert not res.unexpected_keys, res
        fake.model = model.eval()
        batch.update(im_mask=torch.ones(1, R, dtype=torch.bool), pt_mask=torch.ones(1, R, dtype=torch.bool))
        extra = dict(cfeat=cfeat, ffeat=ffeat)
    # d loss / d pose of every step, recorded as the optimiser sees it
    grads = []
    adam = torch.optim.Adam

    class RecordingAdam(adam):
        def step(self, *a, **k):
            grads.append(self.param_groups[0]["params"][0].grad.detach().clone()[0])
            return super().step(*a, **k)

    poses = []
    # the refined pose is only returned at the end: run the reference once per prefix length to get the trajectory
    for n in range(1, num_optim + 1):
        conf.num_optim = n
        torch.manual_seed(rng_seed)
        grads.clear()
        torch.optim.Adam = RecordingAdam
        try:
            est, R_err, t_err = ne.NeRFMatchEvaluator.inerf_refinement(fake, batch, ren, unnorm, c2w_est, conf)
        finally:
            torch.optim.Adam = adam
        poses.append(est)
    if lrdecay:
        # with the cosine schedule the prefix runs differ from the full run (the rate depends on num_optim): keep the last
        poses = poses[-1:]
    fx = dict(H=H, W=W, K=K, unnorm=unnorm, c2w_gt=c2w_gt, c2w_est0=c2w_est, image=image, app=int(app), weights_seed=seed,
              lrate=lrate, lrdecay=int(lrdecay), num_optim=num_optim, t_rands=torch.stack(t_rands), jitters=torch.stack(jitters),
              poses=torch.stack(poses), R_err=R_err, t_err=t_err, pose_grads=torch.stack(grads), use_match_loss=int(use_match_loss), **extra)
    np.savez_compressed(OUT / f"inerf_{tag}.npz", **to_np(fx))
    print(f"inerf_{tag}: R={R} steps={num_optim} final R_err={R_err:.4f} t_err={t_err:.4f}")


# ----------------------------------------------------------------------------- matcher training step
def train_fixture(seed=5):
    """Loss and gradients of one c2f training step, computed by the reference: NeRFMatcherMS.forward(training=True) with the
    GT-padded match sampling (extract_matches.py:38-56, numpy global RNG seeded here), then the statements of
    forward_with_metrics (c2f_trainer.py:502-551) without the pose metrics (PnP is third-party), then loss.backward()."""
    import nerfmatch.nerfmatch_c2f_trainer as c2f
    from nerfmatch.utils.geometry import get_pixel_coords_grid
    from nerfmatch.utils.metrics import compute_fine_match_loss_l2_std, compute_matching_loss

    torch.set_grad_enabled(True)
    g = torch.Generator().manual_seed(177 + seed)
    B, h, w, N = 2, 6, 8, 64
    M = h * w
    Himg, Wimg = h * 8, w * 8
    cfeat = torch.randn(B, 256, h, w, generator=g)
    ffeat = torch.randn(B, 128, h * 4, w * 4, generator=g)
    pt_feat = torch.relu(torch.randn(B, N, 256, generator=g))
    pt3d = torch.randn(B, N, 3, generator=g) * 2.0
    pt2d = get_pixel_coords_grid(Wimg, Himg, ds=8).reshape(1, -1, 2).repeat(B, 1, 1)
    conf_gt = torch.zeros(B, M, N, dtype=torch.bool)
    pt2d_proj = torch.rand(B, N, 2, generator=g) * torch.tensor([Wimg, Himg])
    for b in range(B):
        perm = torch.randperm(N, generator=g)[:M]
        n_pl = 26 + 4 * b
        planted = cfeat[b].flatten(-2).T
        pt_feat[b, perm[:n_pl]] = torch.relu(planted[:n_pl]) + 0.05 * torch.randn(n_pl, 256, generator=g)
        conf_gt[b, torch.arange(n_pl), perm[:n_pl]] = True
        pt2d_proj[b, perm[:n_pl]] = pt2d[b, :n_pl] + (torch.rand(n_pl, 2, generator=g) - 0.5) * 6.0
    im_mask = torch.ones(B, M, dtype=torch.bool)
    pt_mask = torch.ones(B, N, dtype=torch.bool)
    im_mask[1, -5:] = False
    pt_mask[0, 3:9] = False
    img = torch.zeros(B, 3, Himg, Wimg)

    cfg = synth.matcher_config("c2f")
    sd = synth.matcher_state_dict("c2f", seed=seed)
    cfeat_p, ffeat_p, pt_feat_p = cfeat.clone().requires_grad_(), ffeat.clone().requires_grad_(), pt_feat.clone().requires_grad_()
    c2f.init_backbone_8_2 = lambda *a, **k: FixedBackbone((cfeat_p, ffeat_p), [256, 128])
    model = c2f.NeRFMatcherMS(cfg)
    model.load_state_dict(sd, strict=False)
    model.train()
    NP_SEED = 1234
    np.random.seed(NP_SEED)
    data = dict(image=img, im_mask=im_mask, pt3d=pt3d, pt_feat=pt_feat_p, pt_mask=pt_mask, pt2d=pt2d, conf_gt=conf_gt, pt2d_proj=pt2d_proj)
    model.forward(data, training=True, ret_feats=True)
    coarse_loss = compute_matching_loss(data["conf_matrix"], conf_gt)
    mpt2d_f_gt, mpt2d_f, mpt2d_c, expec_f = data["mpt2d_f_gt_train"], data["mpt2d_f_train"], data["mpt2d_c_train"], data["expec_f"]
    coarse_dist = (mpt2d_f_gt - mpt2d_c).norm(dim=-1)
    coarse_pos = coarse_dist < model.coarse_dthres
    fine_loss = compute_fine_match_loss_l2_std(mpt2d_f, mpt2d_f_gt, expec_f[:, 2], mask=coarse_pos)
    loss = coarse_loss + fine_loss
    loss.backward()
    b_ids, i_ids, j_ids = data["match_ids"]
    fx = dict(cfeat=cfeat, ffeat=ffeat, pt_feat=pt_feat, pt3d=pt3d, pt2d=pt2d, conf_gt=conf_gt, pt2d_proj=pt2d_proj, im_mask=im_mask,
              pt_mask=pt_mask, weights_seed=seed, np_seed=NP_SEED, coarse_loss=coarse_loss, fine_loss=fine_loss, loss=loss,
              b_ids=b_ids, i_ids=i_ids, j_ids=j_ids, mconf=data["mconf"], pred_num=data["pred_num"], expec_f=expec_f,
              coarse_pos=coarse_pos, conf_matrix=data["conf_matrix"],
              g_cfeat=cfeat_p.grad, g_pt_feat=pt_feat_p.grad, g_ffeat_norm=ffeat_p.grad.norm(),
              g_ffeat_sub=ffeat_p.grad.flatten()[::211])
    # coarse-only step (coarse_only_epochs): gradients of the coarse loss alone
    for p_ in model.parameters():
        p_.grad_coarse = None
    names = dict(model.named_parameters())
    full = [k for k, v in names.items() if v.grad is not None and v.numel() <= 512]
    for k, v in names.items():
        if v.grad is None:
            continue
        key = k.replace(".", "__")
        gflat = v.grad.flatten()
        fx[f"gn__{key}"] = gflat.norm()
        fx[f"gs__{key}"] = gflat if v.numel() <= 512 else gflat[::97]
    no_grad = sorted(k for k, v in names.items() if v.grad is None)
    print("train fixture: coarse %.6f fine %.6f  matches %d (pred %d)  params with grad %d, without %d: %s" % (
        float(coarse_loss), float(fine_loss), len(b_ids), data["pred_num"], sum(v.grad is not None for v in names.values()), len(no_grad), no_grad[:8]))
    # second backward: coarse loss only
    model.zero_grad()
    cfeat_p.grad = None
    pt_feat_p.grad = None
    np.random.seed(NP_SEED)
    data2 = dict(image=img, im_mask=im_mask, pt3d=pt3d, pt_feat=pt_feat_p, pt_mask=pt_mask, pt2d=pt2d, conf_gt=conf_gt, pt2d_proj=pt2d_proj)
    model.forward(data2, training=True, ret_feats=True)
    cl = compute_matching_loss(data2["conf_matrix"], conf_gt)
    cl.backward()
    fx.update(c_g_cfeat=cfeat_p.grad, c_g_pt_feat=pt_feat_p.grad)
    for k, v in names.items():
        if v.grad is None:
            continue
        key = k.replace(".", "__")
        gflat = v.grad.flatten()
        fx[f"c_gn__{key}"] = gflat.norm()
        fx[f"c_gs__{key}"] = gflat if v.numel() <= 512 else gflat[::97]
    torch.set_grad_enabled(False)
    np.savez_compressed(OUT / "matcher_train.npz", **to_np(fx))


# ----------------------------------------------------------------------------- multi-pair and scene cache (rank 2)
def multi_pair_fixture(seed=0, k=3):
    """Top-k reference frames: the reference's own forward_multi_pair (nerfmatch_c2f_trainer.py:371-427,
    nerfmatch_coarse_trainer.py:290-336), B = 2 queries x k = 3 point sets, one of them partially masked."""
    import nerfmatch.nerfmatch_c2f_trainer as c2f
    import nerfmatch.nerfmatch_coarse_trainer as crs
    from nerfmatch.utils.geometry import get_pixel_coords_grid

    torch.set_grad_enabled(False)
    g = torch.Generator().manual_seed(277 + seed)
    B, h, w, N = 2, 6, 8, 64
    M = h * w
    Himg, Wimg = h * 8, w * 8
    cfeat = torch.randn(B, 256, h, w, generator=g)
    ffeat = torch.randn(B, 128, h * 4, w * 4, generator=g)
    pt_feat = torch.relu(torch.randn(B, k, N, 256, generator=g))
    pt3d = torch.randn(B, k, N, 3, generator=g) * 2.0
    for b in range(B):
        tok = cfeat[b].flatten(-2).T
        for f in range(k):  # every frame sees a different subset of the image tokens
            perm = torch.randperm(N, generator=g)[:M]
            lo, n_pl = 6 * f, 18 + 3 * b
            pt_feat[b, f, perm[lo:lo + n_pl]] = torch.relu(tok[lo:lo + n_pl]) + 0.05 * torch.randn(n_pl, 256, generator=g)
    img = torch.zeros(B, 3, Himg, Wimg)
    pt2d = get_pixel_coords_grid(Wimg, Himg, ds=8).reshape(1, -1, 2).repeat(B, 1, 1)
    im_mask = torch.ones(B, M, dtype=torch.bool)
    pt_mask = torch.ones(B, k, N, dtype=torch.bool)
    pt_mask[1, 2, 10:20] = False
    fx = dict(cfeat=cfeat, ffeat=ffeat, pt_feat=pt_feat, pt3d=pt3d, pt2d=pt2d, im_mask=im_mask, pt_mask=pt_mask, weights_seed=seed, k=k)

    c2f.init_backbone_8_2 = lambda *a, **kw: FixedBackbone((cfeat, ffeat), [256, 128])
    model = c2f.NeRFMatcherMS(synth.matcher_config("c2f"))
    model.load_state_dict(synth.matcher_state_dict("c2f", seed=seed), strict=False)
    model.eval()
    for tag, mutual in (("mut", True), ("nomut", False)):
        data = dict(image=img, im_mask=im_mask, pt3d=pt3d.clone(), pt_feat=pt_feat.clone(), pt_mask=pt_mask, pt2d=pt2d)
        out = model.forward(data, mutual=mutual)
        assert out is None
        fx.update({f"c2f_{tag}_{key}": data[key] for key in ("mpt2d_f", "mpt2d_c", "mpt3d", "m_bids", "mconf")})
        print(f"multi-pair c2f {tag}: {len(data['m_bids'])} matches over {k} frames x {B} queries")

    crs.init_backbone = lambda *a, **kw: FixedBackbone(cfeat, 256)
    mini = crs.NeRFMatcherCoarse(synth.matcher_config("coarse"))
    mini.load_state_dict(synth.matcher_state_dict("coarse", seed=seed), strict=False)
    mini.eval()
    for tag, mutual in (("mut", True), ("nomut", False)):
        data = dict(image=img, im_mask=im_mask, pt3d=pt3d.clone(), pt_feat=pt_feat.clone(), pt_mask=pt_mask, pt2d=pt2d)
        out = mini.forward(data, mutual=mutual)
        assert out is data
        b_, i_, j_ = data["match_ids"]
        fx.update({f"coarse_{tag}_b_ids": b_, f"coarse_{tag}_i_ids": i_, f"coarse_{tag}_j_ids": j_, f"coarse_{tag}_mconf": data["mconf"]})
        print(f"multi-pair coarse {tag}: {len(b_)} matches")
    np.savez_compressed(OUT / "matcher_multipair.npz", **to_np(fx))


def scene_cache_fixture(seed=6):
    """One frame dict of the scene-feature cache with the arithmetic of NerfEvaluator.cache_scene_pts
    (nerf_evaluator.py:340-372): the reference's predict() on the frame's ray bundle (ret_pfeat on), points un-normalised on
    the host by NerfEvaluator.unnorm (:234-238, called unbound), colours clamped to [0,1].  The evaluator's constructor needs
    the out-of-scope dataset classes, so the three statements are replayed here around the reference's own methods."""
    from nerfmatch.nerf.renderer import NerfRenderer
    from nerfmatch.nerf import render_utils as ru
    from nerfmatch import nerf_evaluator as nev

    torch.set_grad_enabled(False)
    H, W, S = 48, 64, 32
    cfg = synth.nerf_config("cambridge", num_pts=S, img_wh=(W, H))
    sd = synth.nerf_state_dict(seed=seed, app_vocab=5, density_bias=3.0)
    ren = NerfRenderer(cfg, num_frames=5, training=False, stop_layer=3)
    ren.load_state_dict(sd, strict=True)
    ren.eval()
    K = torch.tensor([[60.0, 0, W / 2], [0, 60.0, H / 2], [0, 0, 1]])
    unnorm = synth.unnorm_scene()
    c2w_norm = synth.camera_pose(seed=seed + 30)
    rays = ru.sample_nerf_rays(H, W, K, c2w_norm, ds=8, embed_type="mip")
    R = rays.shape[0]
    ts = torch.full((R,), 3, dtype=torch.long)  # the frame's appearance id (batch["ts"])
    rng_seed = 5000 + seed
    torch.manual_seed(rng_seed)
    t_rand = torch.rand(R, S + 1)
    jitter = torch.empty(R, S + 1).uniform_(to=(1 / (S + 1) - torch.finfo(torch.float32).eps))
    ren.ret_pfeat, ren.feat_comb = True, "lin"
    torch.manual_seed(rng_seed)
    preds = ren.predict(rays, W // 8, H // 8, ray_id=ts)
    pt3d = nev.NerfEvaluator.unnorm(None, unnorm, preds["pts_fine"].cpu())
    frame = dict(pt3d=pt3d.numpy(), unnorm_scene=unnorm.numpy(), pt_feat=preds["feat_fine"].cpu().numpy(),
                 pt_color=preds["rgb_fine"].reshape(-1, 3).clamp(0, 1).cpu().numpy())
    fx = dict(H=H, W=W, S=S, K=K, unnorm=unnorm, c2w_norm=c2w_norm, rays=rays, ts=ts, t_rand=t_rand, jitter=jitter, weights_seed=seed,
              rgb_fine=preds["rgb_fine"], depth_fine=preds["depth_fine"], **{f"frame_{k_}": v for k_, v in frame.items()})
    np.savez_compressed(OUT / "scene_cache_frame.npz", **to_np(fx))
    print(f"scene_cache_frame: R={R} keys={sorted(frame)}")



if __name__ == "__main__":
    assert REF.exists(), "the reference is only present in the build container"
    install_stubs()
    torch.set_num_threads(8)
    if sys.argv[1:] == ["check"]:  # round 6: the three fixtures tests/test_golden_regen_cpu.py regenerates and compares bit for bit
        nerf_fixture("r32_s32", "7scenes", H=32, W=64, S=32, stop_layer=3, seed=0)
        matcher_fixtures(seed=0)
        postnorm_fixture()
        sys.exit(0)
    if sys.argv[1:] == ["layer_grads"]:  # round 6: gradients of an lsa / relu encoder layer
        layer_grads_fixture()
        sys.exit(0)
    if sys.argv[1:] == ["envelope"]:  # round 6: option values beyond the shipped yamls
        envelope_fixture()
        sys.exit(0)
    if sys.argv[1:] == ["smooth"]:  # only the three smooth NeRF fixtures
        nerf_fixture("r32_s32", "7scenes", H=32, W=64, S=32, stop_layer=3, seed=0)
        nerf_fixture("r128_s64_app", "cambridge", H=64, W=128, S=64, stop_layer=3, seed=1, sub_rays=2)
        nerf_fixture("r32_s32_last", "7scenes", H=32, W=64, S=32, stop_layer=-1, seed=2)
        sys.exit(0)
    if sys.argv[1:] == ["surface"]:  # only the trained-like NeRF fixture (round 3)
        nerf_fixture("surface_r512_s128", "7scenes", H=128, W=256, S=128, stop_layer=3, seed=0, sub_rays=2, style="surface", focal=240.0, pose_seed=11)
        sys.exit(0)
    if sys.argv[1:] == ["surface_app"]:  # only the trained-like fixture of the Cambridge variant (appearance embedding, white background)
        nerf_fixture("surface_r256_s64_app", "cambridge", H=64, W=256, S=64, stop_layer=3, seed=0, sub_rays=2, style="surface", focal=240.0, pose_seed=12, density_shift=7500.0)
        sys.exit(0)
    if sys.argv[1:] == ["fine_count"]:  # round 4: coarse_nerf.num_pts = 32, fine_nerf.num_pts = 64 -- the mip resampler ignores the latter
        fine_count_fixture()
        sys.exit(0)
    if sys.argv[1:] == ["surface_seeds"]:  # round 4: five trained-like weight seeds x two poses, with the reference's fp64 evaluation
        for ws, ps in SURFACE_SEEDS:
            surface_seed_fixture(ws, ps)
        sys.exit(0)
    if sys.argv[1:] == ["postnorm"]:  # round 5: post-norm encoder layers
        postnorm_fixture()
        sys.exit(0)
    if sys.argv[1:] == ["peaked"]:  # only the peaked-confidence matcher fixture (round 3)
        peaked_matcher_fixture(seed=0)
        sys.exit(0)
    if sys.argv[1:] == ["train"]:  # only the training-step fixture
        train_fixture(seed=5)
        sys.exit(0)
    if sys.argv[1:] == ["next"]:  # only the multi-pair and scene-cache fixtures
        multi_pair_fixture(seed=0)
        scene_cache_fixture(seed=6)
        sys.exit(0)
    if sys.argv[1:] == ["inerf_match"]:  # only the iNeRF fixture with the matching term
        inerf_fixture("match", "7scenes", H=48, W=64, seed=7, num_optim=3, use_match_loss=True)
        sys.exit(0)
    if sys.argv[1:] == ["inerf_match_coarse"]:  # round 6: the matching term through the coarse-only model class
        inerf_fixture("match_coarse", "7scenes", H=48, W=64, seed=8, num_optim=3, use_match_loss=True, matcher="coarse_full")
        sys.exit(0)
    nerf_fixture("r32_s32", "7scenes", H=32, W=64, S=32, stop_layer=3, seed=0)
    nerf_fixture("r128_s64_app", "cambridge", H=64, W=128, S=64, stop_layer=3, seed=1, sub_rays=2)
    nerf_fixture("r32_s32_last", "7scenes", H=32, W=64, S=32, stop_layer=-1, seed=2)
    nerf_fixture("surface_r512_s128", "7scenes", H=128, W=256, S=128, stop_layer=3, seed=0, sub_rays=2, style="surface", focal=240.0, pose_seed=11)
    nerf_fixture("surface_r256_s64_app", "cambridge", H=64, W=256, S=64, stop_layer=3, seed=0, sub_rays=2, style="surface", focal=240.0, pose_seed=12, density_shift=7500.0)
    for ws, ps in SURFACE_SEEDS:
        surface_seed_fixture(ws, ps)
    fine_count_fixture()
    far_fallback_fixture()
    matcher_fixtures(seed=0)
    peaked_matcher_fixture(seed=0)
    inerf_fixture("7s", "7scenes", H=32, W=64, seed=3, num_optim=3)
    inerf_fixture("cam_decay", "cambridge", H=32, W=32, seed=4, num_optim=2, lrdecay=True)
    train_fixture(seed=5)
    postnorm_fixture()
    envelope_fixture()
    layer_grads_fixture()
    multi_pair_fixture(seed=0)
    scene_cache_fixture(seed=6)
    inerf_fixture("match", "7scenes", H=48, W=64, seed=7, num_optim=3, use_match_loss=True)
    inerf_fixture("match_coarse", "7scenes", H=48, W=64, seed=8, num_optim=3, use_match_loss=True, matcher="coarse_full")
    print("done")

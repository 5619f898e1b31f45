"""Deterministic synthetic hyper-parameters, weights and inputs.

There is no network here (no datasets, no checkpoints), so every test, fixture and benchmark uses
  * hyper-parameters equal to the VALUES of the reference's shipped yaml files
    (configs/nerf/nerf_7scenes_mip_sfm.yaml, configs/nerf/nerf_cambridge_mip_app.yaml,
     configs/nerfmatch/nerfmatch_7scenes_sfm_{c2f,coarse}.yaml), and
  * weights drawn from numpy's PCG64 (`default_rng(seed)`) so that the reference (in the survey
    container), the oracle and the HIP path can all regenerate bit-identical parameters
    without relying on torch's RNG streams.

State-dict key names are exactly the reference's so the same dict loads into
the reference modules, the oracle and `nerfmatch_amd`'s classes.
"""
from argparse import Namespace
from collections import OrderedDict
import math

import numpy as np
import torch

from .utils.config import dict2namespace


# --------------------------------------------------------------------------------------
# hyper-parameters
# --------------------------------------------------------------------------------------
def nerf_config(scene_type="7scenes", num_pts=128, img_wh=(480, 480), num_pts_fine=None):
    """Namespace with the fields NerfRenderer reads (reference: nerf/renderer.py:27-114).  num_pts_fine: the fine network's `num_pts`
    (default: the same; in the mip configuration the reference's resampler ignores it, render_utils.py:299-309)."""
    app = scene_type == "cambridge"
    cfg = dict(
        data=dict(img_wh=list(img_wh), white_bg=app),
        coarse_nerf=dict(method="NeRF", layer_num=8, hid_dim=256, output_dim=4, skips=[4], num_pts=num_pts),
        fine_nerf=dict(method="NeRF", layer_num=8, hid_dim=256, output_dim=4, skips=[4], num_pts=num_pts if num_pts_fine is None else num_pts_fine),
        embedding=dict(xyz_num_freqs=15, dirs_num_freqs=4, type="mip"),
        render=dict(chunksize=16384, use_viewdirs=True, use_disp=False, perturb=True, white_bg=app, noise_std=1.0),
        loss=dict(use_sem_mask=app, ray_reg_weight=0.01),
    )
    if app:
        cfg["embedding"]["appearance_embed"] = True
    return dict2namespace(cfg)


def matcher_config(kind="c2f", backbone="stub"):
    """Namespace equal to the `model:` block of the shipped matcher yamls."""
    if kind == "c2f":
        m = dict(
            backbone=backbone, pretrained=False, im_pe=True, im_sa_type="share", im_sa=3, temp_type="mul",
            pt_sa=3, pt_dim=256, pt_sa_type="full", pt_pe=True, pt_pe_type="fourier", post_pt_pe=True,
            cfeat_dim=256, ffeat_dim=128, cformer_type="crs", coarse_layers=1, pt_ftype="nerf",
            fine_sa=1, fsa_type="full", win_sz=5, cat_c_feat=True,
            fine_loss="match", coarse_percent=0.3, coarse_dthres=10,
        )
    elif kind == "coarse":
        m = dict(
            backbone=backbone, pretrained=False, im_pe=False, im_sa_type="share", im_sa=0, temp_type="mul",
            pt_dim=256, pt_sa=0, pt_sa_type="full", pt_pe=False, pt_pe_type="fourier", post_pt_pe=False,
            cfeat_dim=256, cformer_type="crs", coarse_layers=0, pt_ftype="nerf",
        )
    else:
        raise ValueError(kind)
    return Namespace(**m)


# --------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------
def _uniform(rng, shape, bound):
    return torch.from_numpy(rng.uniform(-bound, bound, size=shape).astype(np.float32))


def _linear(sd, rng, name, fan_out, fan_in, bias=True):
    b = 1.0 / math.sqrt(fan_in)
    sd[f"{name}.weight"] = _uniform(rng, (fan_out, fan_in), b)
    if bias:
        sd[f"{name}.bias"] = _uniform(rng, (fan_out,), b)


def _layernorm(sd, rng, name, dim):
    sd[f"{name}.weight"] = 1.0 + _uniform(rng, (dim,), 0.1)
    sd[f"{name}.bias"] = _uniform(rng, (dim,), 0.1)


# "surface" style (round 3): a random-weight stand-in for a TRAINED NeRF.  U(+-1/sqrt(fan_in)) weights give a smooth field
# with activations ~0.1 and compositing weights spread over the whole ray; a trained scene has hidden activations of O(10),
# densities in the thousands (normalised units) and a surface that saturates alpha within 2-4 coarse samples.  Constants found
# with scripts/trained_like_study.py on the oracle: layer gain 3.2 (activations up to ~18), IPE input columns of frequency
# 2^i damped by 2^(4-i) for i > 4 (a trained net does not weight the 2^14 band like the 2^0 band: without this the density is
# white noise along the ray), density head scaled / shifted so that ~25 % of space is occupied, and the fine net sharing the
# coarse net's density trunk (both describe the same scene; the other heads stay independent).
SURFACE_STYLE = dict(layer_gain=3.2, freq_cut=4, density_gain=2600.0, density_bias=-8600.0)


def nerf_state_dict(seed=0, app_vocab=0, hid=256, xyz_freqs=15, dirs_freqs=4, density_bias=0.0, style=None):
    """Keys as stored under `model.` in a reference NeRF checkpoint (prefix already stripped).

    `density_bias` shifts alpha_linear.bias so random-init densities are not almost all <= 0 (which
    would make every compositing weight ~0 and the parity tests vacuous).  `style="surface"`: see SURFACE_STYLE."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    xyz_dim = 2 * 3 * xyz_freqs
    dirs_dim = 2 * 3 * dirs_freqs + 3
    app_dim = 16 if app_vocab > 0 else 0
    for net in ("nerf_coarse", "nerf_fine"):
        _linear(sd, rng, f"{net}.pts_linears.0", hid, xyz_dim)
        for i in range(1, 8):
            _linear(sd, rng, f"{net}.pts_linears.{i}", hid, hid + (xyz_dim if i == 5 else 0))
        _linear(sd, rng, f"{net}.views_linears.0", hid // 2, hid + dirs_dim + app_dim)
        _linear(sd, rng, f"{net}.feature_linear", hid, hid)
        _linear(sd, rng, f"{net}.alpha_linear", 1, hid)
        _linear(sd, rng, f"{net}.rgb_linear", 3, hid // 2)
        sd[f"{net}.alpha_linear.bias"] = sd[f"{net}.alpha_linear.bias"] + density_bias
    sd["xyz_encoder.scales"] = torch.tensor([2**i for i in range(xyz_freqs)], dtype=torch.int64)
    sd["dirs_encoder.scales"] = torch.tensor([2**i for i in range(dirs_freqs)], dtype=torch.int64)
    if app_vocab > 0:
        sd["embedding_a.weight"] = torch.from_numpy(rng.standard_normal((app_vocab, 16)).astype(np.float32))
    if style == "surface":
        st = SURFACE_STYLE
        damp = torch.tensor([min(1.0, 2.0 ** (st["freq_cut"] - i)) for i in range(xyz_freqs)]).repeat_interleave(3).repeat(2)
        for net in ("nerf_coarse", "nerf_fine"):
            for i in range(8):
                sd[f"{net}.pts_linears.{i}.weight"] = sd[f"{net}.pts_linears.{i}.weight"] * st["layer_gain"]
            sd[f"{net}.pts_linears.0.weight"] = sd[f"{net}.pts_linears.0.weight"] * damp[None]
            sd[f"{net}.pts_linears.5.weight"][:, :xyz_dim] *= damp[None]
            sd[f"{net}.alpha_linear.weight"] = sd[f"{net}.alpha_linear.weight"] * st["density_gain"]
            sd[f"{net}.alpha_linear.bias"] = torch.full((1,), st["density_bias"] + density_bias)  # (density_bias: per-fixture shift)
        for k in list(sd):
            if k.startswith("nerf_coarse.pts_linears") or k.startswith("nerf_coarse.alpha_linear"):
                sd[k.replace("nerf_coarse", "nerf_fine")] = sd[k].clone()
    elif style is not None:
        raise ValueError(style)
    return sd


def _encoder_layer(sd, rng, name, dim, cross=False):
    for p in ("q", "k", "v"):
        _linear(sd, rng, f"{name}.attention.proj_{p}", dim, dim, bias=False)
    _linear(sd, rng, f"{name}.attention.proj_out.0", dim, dim, bias=False)
    _layernorm(sd, rng, f"{name}.norm1.0", dim)
    if cross:
        _layernorm(sd, rng, f"{name}.norm1.1", dim)
    _linear(sd, rng, f"{name}.feedforward.layers.0", dim, dim)
    _linear(sd, rng, f"{name}.feedforward.layers.2", dim, dim)
    _layernorm(sd, rng, f"{name}.norm2", dim)


def matcher_state_dict(kind="c2f", seed=0, temperature=10.0, style=None):
    """Keys of NeRFMatcherMS / NeRFMatcherCoarse without the backbone.

    `style="aligned"` (round 3): a stand-in for a TRAINED matcher -- `pt_pe_proj` keeps the feature block (identity + a small
    random mixture of the Fourier block) so that image tokens and the points planted on them stay aligned through the shared
    self-attention stack and the confidence matrix comes out peaked (row maxima near 1) instead of nearly uniform."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    sd["temperature"] = torch.tensor(float(temperature))
    if kind == "coarse":
        return sd
    C, Cf = 256, 128
    _linear(sd, rng, "pt_pe_proj", C, C + 93)
    _linear(sd, rng, "pt_ffeat_proj.0", Cf, C)
    _linear(sd, rng, "pt_ffeat_proj.1", Cf, Cf)
    for i in range(3):
        _encoder_layer(sd, rng, f"pt_sa.layers.{i}", C)
    _encoder_layer(sd, rng, "coarse_former", C, cross=True)
    _linear(sd, rng, "fine_preprocess.down_proj", Cf, C)
    _linear(sd, rng, "fine_preprocess.merge_feat", Cf, 2 * Cf)
    _encoder_layer(sd, rng, "fine_sa.layers.0", Cf)
    if style == "aligned":
        rng2 = np.random.default_rng(seed + 99)
        wp = torch.zeros(C, C + 93)
        wp[:, :C] = torch.eye(C)
        sd["pt_pe_proj.weight"] = wp + 0.3 / math.sqrt(C + 93) * torch.from_numpy(rng2.standard_normal((C, C + 93)).astype(np.float32))
    elif style is not None:
        raise ValueError(style)
    return sd


MATCHER_VARIANTS = ("pe3d", "pe3d_pre", "pt3d_id", "nerf128_id", "coarse_norm")


def matcher_variant(name, seed=0):
    """(config, state dict) of the option values outside the shipped yamls that the reference's constructors accept (round 6, VERDICT r5
    item 5; nerfmatch_c2f_trainer.py:121-147, nerfmatch_coarse_trainer.py:91-124):
      pe3d        c2f, points described by the Fourier embedding of their coordinates (pt_proj 93 -> 256), Fourier PE behind the self-attention
      pe3d_pre    the same with the PE in FRONT of the self-attention (post_pt_pe False)
      pt3d_id     c2f, points described by their coordinates (pt_proj 3 -> 256), pt_pe_type "id": the coordinates again as the "encoding"
      nerf128_id  c2f, 128-d rendered features (pt_proj 128 -> 256), pt_pe_type "id"
      coarse_norm coarse-only model with image PE, one shared self-attention layer, Fourier PE in front, one cross layer and pt_feat_norm"""
    rng = np.random.default_rng(1000 + seed)
    C = 256
    if name in ("coarse_norm", "coarse_full"):  # coarse_full: the same model without pt_feat_norm (the iNeRF matching term's coarse fixture)
        cfg = matcher_config("coarse")
        cfg.im_pe, cfg.im_sa, cfg.pt_sa, cfg.pt_pe, cfg.coarse_layers, cfg.pt_feat_norm = True, 1, 1, True, 1, name == "coarse_norm"
        sd = matcher_state_dict("coarse", seed=seed)
        _linear(sd, rng, "pt_pe_proj", C, C + 93)
        _encoder_layer(sd, rng, "pt_sa.layers.0", C)
        _encoder_layer(sd, rng, "coarse_former", C, cross=True)
        return cfg, sd
    cfg = matcher_config("c2f")
    sd = matcher_state_dict("c2f", seed=seed)
    if name in ("pe3d", "pe3d_pre"):
        cfg.pt_ftype, cfg.post_pt_pe = "pe3d", name == "pe3d"
        _linear(sd, rng, "pt_proj", C, 93)
    elif name == "pt3d_id":
        cfg.pt_ftype, cfg.pt_pe_type, cfg.pt_dim = "pt3d", "id", 3
        _linear(sd, rng, "pt_proj", C, 3)
        _linear(sd, rng, "pt_pe_proj", C, C + 3)
    elif name == "nerf128_id":
        cfg.pt_dim, cfg.pt_pe_type = 128, "id"
        _linear(sd, rng, "pt_proj", C, 128)
        _linear(sd, rng, "pt_pe_proj", C, C + 128)
    else:
        raise ValueError(name)
    return cfg, sd


# --------------------------------------------------------------------------------------
# inputs
# --------------------------------------------------------------------------------------
K_7SCENES = [[525.0, 0.0, 320.0], [0.0, 525.0, 240.0], [0.0, 0.0, 1.0]]


def intrinsics(H=480, W=640, f=525.0):
    return torch.tensor([[f, 0.0, 0.5 * W], [0.0, f, 0.5 * H], [0.0, 0.0, 1.0]], dtype=torch.float32)


def camera_pose(seed=0, max_t=0.5, max_angle=0.3):
    """A normalised c2w (4x4 f32) inside the unit sphere: small-angle rotation, |t| <= max_t."""
    rng = np.random.default_rng(seed)
    axis = rng.standard_normal(3)
    axis /= np.linalg.norm(axis)
    ang = rng.uniform(-max_angle, max_angle)
    Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + math.sin(ang) * Kx + (1 - math.cos(ang)) * (Kx @ Kx)
    t = rng.standard_normal(3)
    t = t / np.linalg.norm(t) * rng.uniform(0, max_t)
    c2w = np.eye(4)
    c2w[:3, :3] = R
    c2w[:3, 3] = t
    return torch.from_numpy(c2w.astype(np.float32))


def unnorm_scene(scale=3.0, shift=(0.4, -0.2, 1.1)):
    """normalised-scene -> world transform (uniform scale + shift), like compute_world2nscene()^-1."""
    T = torch.eye(4)
    T[:3, :3] *= scale
    T[:3, 3] = torch.tensor(shift)
    return T


def uniform01(shape, seed):
    """U[0,1) float32 from a seeded CPU torch generator (the explicit random inputs of R4a/R5)."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g, dtype=torch.float32)


def resample_jitter(shape, seed):
    """The `uniform_(to=1/n - eps)` tensor of the reference's resampler (render_utils.py:483-485)."""
    n = shape[-1]
    hi = 1.0 / n - float(torch.finfo(torch.float32).eps)
    g = torch.Generator().manual_seed(seed)
    return torch.empty(*shape, dtype=torch.float32).uniform_(0.0, hi, generator=g)


def separated_features(m, n, dim=256, seed=2):
    """relu(N(0,1)) features, L2-normalised (BASELINE config C2 inputs)."""
    g = torch.Generator().manual_seed(seed)
    a = torch.relu(torch.randn(m, dim, generator=g))
    b = torch.relu(torch.randn(n, dim, generator=g))
    a = a / a.norm(dim=-1, keepdim=True).clamp_min(1e-6)
    b = b / b.norm(dim=-1, keepdim=True).clamp_min(1e-6)
    return a, b

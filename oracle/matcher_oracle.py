"""ORACLE (test infrastructure, NOT product code) -- coarse-to-fine 2D-3D matcher half.

CPU restatement (torch-CPU fp32) of SURVEY.md section 8(a) rows M1-M5, A1/A2, F1-F4, C0.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Parity status: PINNED against golden vectors produced by importing the reference
(tests/golden/make_golden.py -> tests/golden/matcher_*.npz), with two caveats that are stated in
DESIGN.md: (1) the image backbone (timm ConvFormer-B36) is third-party code absent from the
reference tree -- fixtures start from backbone OUTPUTS; (2) the two kornia helpers used by fine
matching (kornia.geometry.subpix.dsnt.spatial_expectation2d, kornia.utils.grid.create_meshgrid;
requirements.txt:12, unpinned, not vendored) are restated here from their published definition
and the same restatement was handed to the reference when the fixtures were made, so that part
is "parity unpinned" w.r.t. real kornia.

`params` uses the reference's state-dict key names without the "model." prefix.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- encodings
def sine_pe_table(d_model, h, w):
    """2-D sinusoidal table (d_model,h,w): channels 0::4 sin x, 1::4 cos x, 2::4 sin y, 3::4 cos y,
    1-based positions.  third_party/loftr/position_encoding.py:24-43 (temp_bug_fix=True)."""
    ypos = torch.ones(h, w).cumsum(0).float().unsqueeze(0)
    xpos = torch.ones(h, w).cumsum(1).float().unsqueeze(0)
    div = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / (d_model // 2)))[:, None, None]
    pe = torch.zeros(d_model, h, w)
    pe[0::4] = torch.sin(xpos * div)
    pe[1::4] = torch.cos(xpos * div)
    pe[2::4] = torch.sin(ypos * div)
    pe[3::4] = torch.cos(ypos * div)
    return pe


def fourier_embed(x, num_freqs=15):
    """[x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...]  nerfmatch/nerf/embedding.py:35-46."""
    out = [x]
    for f in 2 ** torch.linspace(0, num_freqs - 1, num_freqs):
        out += [torch.sin(f * x * 1.0), torch.cos(f * x * 1.0)]
    return torch.cat(out, -1)


# ----------------------------------------------------------------------------- A1 / A2
def multi_head_attention(p, name, q_in, kv_in, heads, att_type="full"):
    """Bias-free q/k/v/out projections around softmax attention.
    nerfmatch/modules/attention.py:119-133, :53-57 (full), :71-81 (lsa; the diagonal mask the
    reference builds is never applied -- softmax is taken of the unmasked scores)."""
    q = F.linear(q_in, p[f"{name}.proj_q.weight"])
    k = F.linear(kv_in, p[f"{name}.proj_k.weight"])
    v = F.linear(kv_in, p[f"{name}.proj_v.weight"])
    b, l, c = q.shape
    d = c // heads
    q, k, v = (t.reshape(t.shape[0], t.shape[1], heads, d) for t in (q, k, v))
    if att_type == "lsa":
        score = torch.einsum("blhd,bshd->blsh", q, k) * p[f"{name}.attend.scale"].exp()
    else:
        score = torch.einsum("blhd,bshd->blsh", q / d**0.5, k)
    att = torch.softmax(score, dim=2)
    o = torch.einsum("blsh,bshd->blhd", att, v).reshape(b, l, c)
    return F.linear(o, p[f"{name}.proj_out.0.weight"])


def encoder_layer(p, name, x, ctx=None, heads=8, att_type="full", act="gelu"):
    """Pre-norm encoder layer.  nerfmatch/modules/attention.py:223-241.
    y = xh + FFN(LN2(xh + MHA(xh, ch)))  with xh = LN1[0](x), ch = LN1[1 or 0](ctx):
    BOTH residuals add onto the normalised input (the reference rebinds x = norm_x(x))."""
    dim = x.shape[-1]
    cross = ctx is not None
    xh = F.layer_norm(x, (dim,), p[f"{name}.norm1.0.weight"], p[f"{name}.norm1.0.bias"])
    if cross:
        ch = F.layer_norm(ctx, (dim,), p[f"{name}.norm1.1.weight"], p[f"{name}.norm1.1.bias"])
    else:
        ch = xh
    a = xh + multi_head_attention(p, f"{name}.attention", xh, ch, heads, att_type)
    a = F.layer_norm(a, (dim,), p[f"{name}.norm2.weight"], p[f"{name}.norm2.bias"])
    f = F.linear(a, p[f"{name}.feedforward.layers.0.weight"], p[f"{name}.feedforward.layers.0.bias"])
    f = F.linear(F.gelu(f) if act == "gelu" else F.relu(f), p[f"{name}.feedforward.layers.2.weight"], p[f"{name}.feedforward.layers.2.bias"])  # act_fn, attention.py:136-154
    return xh + f


def encoder_layer_post_norm(p, name, x, ctx=None, heads=8, att_type="full"):
    """Post-norm encoder layer.  nerfmatch/modules/attention.py:209-221 (selected by norm_type != "pre"; no shipped yaml does):
    a = LN1(x + MHA(x, ctx)); y = LN2(x + FFN(a)) -- the second residual is again the RAW input x, and norm1 holds one LayerNorm even in
    cross mode (:195-198)."""
    dim = x.shape[-1]
    c = x if ctx is None else ctx
    a = x + multi_head_attention(p, f"{name}.attention", x, c, heads, att_type)
    a = F.layer_norm(a, (dim,), p[f"{name}.norm1.0.weight"], p[f"{name}.norm1.0.bias"])
    f = F.linear(a, p[f"{name}.feedforward.layers.0.weight"], p[f"{name}.feedforward.layers.0.bias"])
    f = F.linear(F.gelu(f), p[f"{name}.feedforward.layers.2.weight"], p[f"{name}.feedforward.layers.2.bias"])
    return F.layer_norm(x + f, (dim,), p[f"{name}.norm2.weight"], p[f"{name}.norm2.bias"])


def self_attention_block(p, name, x, num_layers, heads=8, att_type="full"):
    """nerfmatch/modules/attention.py:255-285."""
    for i in range(num_layers):
        x = encoder_layer(p, f"{name}.layers.{i}", x, None, heads, att_type)
    return x


# ----------------------------------------------------------------------------- M4 / M5
def coarse_matching(im_feat, pt_feat, temperature, im_mask=None, pt_mask=None, temp_type="mul"):
    """Dual-softmax confidence (B,M,N).  nerfmatch/nerfmatch_c2f_trainer.py:289-300."""
    im = im_feat / (im_feat.norm(dim=-1, keepdim=True) + 1e-6)
    pt = pt_feat / (pt_feat.norm(dim=-1, keepdim=True) + 1e-6)
    sim = torch.einsum("bmd,bnd->bmn", im, pt)
    sim = sim * temperature if temp_type == "mul" else sim / temperature
    im_m = torch.ones_like(im[..., 0]) if im_mask is None else im_mask
    pt_m = torch.ones_like(pt[..., 0]) if pt_mask is None else pt_mask
    sim = sim.masked_fill(~(im_m[..., None] * pt_m[:, None]).bool(), -1e9)
    return F.softmax(sim, 1) * F.softmax(sim, 2), im, pt


def mutual_matches(conf, mutual=True, threshold=0.0):
    """Inference branch of nerfmatch/modules/extract_matches.py:21-36.
    Returns (b_ids, i_ids, j_ids) int64 sorted by (b,i) and mconf."""
    mask = conf > threshold
    row_best = conf == conf.max(dim=2, keepdim=True)[0]
    mask = mask * row_best
    if mutual:
        mask = mask * (conf == conf.max(dim=1, keepdim=True)[0])
    any_j, first_j = mask.max(dim=2)
    b_ids, i_ids = torch.where(any_j)
    j_ids = first_j[b_ids, i_ids]
    return (b_ids, i_ids, j_ids), conf[b_ids, i_ids, j_ids]


# ----------------------------------------------------------------------------- F2 / F3
def fine_windows(ffeat, b_ids, i_ids, win=5, stride=4):
    """5x5 windows (stride 4, pad 2) of the fine map gathered at the matched coarse cells:
    (K, win*win, C).  third_party/loftr/fine_matching.py:46-55."""
    c = ffeat.shape[1]
    unf = F.unfold(ffeat, kernel_size=(win, win), stride=stride, padding=win // 2)
    unf = unf.reshape(ffeat.shape[0], c, win * win, -1).permute(0, 3, 2, 1)
    return unf[b_ids, i_ids]


def spatial_expectation_5x5(heat):
    """kornia dsnt.spatial_expectation2d(heat[None], normalized_coordinates=True)[0] restated:
    grid = linspace(-1,1,W) (x) / linspace(-1,1,H) (y); returns (M,2) = (E[x], E[y])."""
    m, h, w = heat.shape
    xs = torch.linspace(-1, 1, w)
    ys = torch.linspace(-1, 1, h)
    gx = xs[None, :].expand(h, w).reshape(-1)
    gy = ys[:, None].expand(h, w).reshape(-1)
    flat = heat.reshape(m, -1)
    return torch.stack([(gx * flat).sum(-1), (gy * flat).sum(-1)], -1), torch.stack([gx, gy], -1)


def fine_matching(pt_ffeat, win_feat):
    """(K,C),(K,WW,C) -> expec_f (K,3) = [E[x], E[y], std].  third_party/loftr/fine_matching.py:88-121."""
    k, ww, c = win_feat.shape
    if k == 0:
        return torch.empty(0, 3)
    w = int(math.sqrt(ww))
    sim = torch.einsum("mc,mrc->mr", pt_ffeat, win_feat)
    heat = torch.softmax(sim * (1.0 / c**0.5), dim=1).view(-1, w, w)
    coords, grid = spatial_expectation_5x5(heat)
    var = (grid[None] ** 2 * heat.view(-1, ww, 1)).sum(1) - coords**2
    std = torch.sqrt(torch.clamp(var, min=1e-10)).sum(-1)
    return torch.cat([coords, std[:, None]], -1)


# ----------------------------------------------------------------------------- models
def pad_matches_with_gt(ids, mconf, conf_gt, coarse_percent=0.3, train_percent=0.3):
    """Training branch of extract_mutual_matches (nerfmatch/modules/extract_matches.py:38-56): a fixed number of training
    matches, at most `coarse_percent` of them predictions (re-drawn WITH replacement), the rest ground-truth pairs with
    mconf = 0; the draws are numpy's global RNG, in the reference's order.  Returns (ids, mconf, pred_num)."""
    import numpy as np

    b_ids, i_ids, j_ids = ids
    b, d2, d3 = conf_gt.shape
    pred_num = len(b_ids)
    total_pts = b * min(d2, d3)
    b_gt, i_gt, j_gt = torch.where(conf_gt)
    train_num = int(total_pts * train_percent)
    pred_num = min(int(train_num * coarse_percent), pred_num)
    gt_num = train_num - pred_num
    pred_idx = np.random.choice(len(b_ids), pred_num)
    gt_idx = np.random.choice(len(b_gt), gt_num)
    ids = (torch.cat([b_ids[pred_idx], b_gt[gt_idx]]), torch.cat([i_ids[pred_idx], i_gt[gt_idx]]),
           torch.cat([j_ids[pred_idx], j_gt[gt_idx]]))
    return ids, torch.cat([mconf[pred_idx], torch.zeros(gt_num).to(mconf)]), pred_num


def feature_normalization(x):
    """nerfmatch/nerfmatch_coarse_trainer.py:42-47 -- note the IN-PLACE centring of the argument (`x -= centroid`)."""
    centroid = x.mean(dim=1)
    x -= centroid[:, None, :]
    max_norm = x.norm(dim=-1).max(dim=-1)[0]
    return x / max_norm[:, None, None]


def extract_pt_feat(p, cfg, pt_feat, pt3d):
    """Point tokens for every option value the reference's constructors accept: nerfmatch/nerfmatch_c2f_trainer.py:121-147 (options),
    :258-287 (cat_pe, extract_pt_feat); nerfmatch/nerfmatch_coarse_trainer.py:91-124, :186-224 (the same plus `pt_feat_norm`, :198-200).
    pt_ftype "rand" draws noise and has no restatement."""
    ftype = getattr(cfg, "pt_ftype", "nerf")
    if getattr(cfg, "pt_feat_norm", False):
        pt_feat, pt3d = feature_normalization(pt_feat), feature_normalization(pt3d)
    if ftype == "pt3d":
        pt_feat = pt3d
    elif ftype == "pe3d":
        pt_feat = fourier_embed(pt3d)
    elif ftype != "nerf":
        raise ValueError(ftype)
    pt_in = pt_feat
    pt = pt_feat
    if "pt_proj.weight" in p:
        pt = F.linear(pt, p["pt_proj.weight"], p["pt_proj.bias"])
    has_pe, post = getattr(cfg, "pt_pe", True), getattr(cfg, "post_pt_pe", False)

    def cat_pe(t):
        emb = pt_in if getattr(cfg, "pt_pe_type", "fourier") == "id" else fourier_embed(pt3d)
        return F.linear(torch.cat([t, emb], -1), p["pt_pe_proj.weight"], p["pt_pe_proj.bias"])

    if has_pe and not post:
        pt = cat_pe(pt)
    n_sa = getattr(cfg, "pt_sa", 3)
    if getattr(cfg, "pt_sa_type", "full") == "full" and n_sa > 0:
        pt = self_attention_block(p, "pt_sa", pt, n_sa)
    if has_pe and post:
        pt = cat_pe(pt)
    return pt


def c2f_forward_match(p, cfg, cfeat_map, ffeat_map, pt_feat, pt3d, im_mask=None, pt_mask=None,
                      mutual=False, match_thres=0.0, conf_gt=None):
    """NeRFMatcherMS.forward_match from backbone outputs on.
    nerfmatch/nerfmatch_c2f_trainer.py:237-256 (image side), :263-287 (point side),
    :319-328 (sequential cross attention, same weights), :330-351 (matching + fine stage).
    cfeat_map (B,256,h,w), ffeat_map (B,128,4h,4w) are the backbone's two outputs."""
    b, c, h, w = cfeat_map.shape
    im = cfeat_map.flatten(-2).permute(0, 2, 1)
    if getattr(cfg, "im_pe", True):
        im = (cfeat_map + sine_pe_table(c, h, w)[None]).flatten(-2).permute(0, 2, 1)
    n_sa = getattr(cfg, "pt_sa", 3)
    if getattr(cfg, "im_sa_type", None) == "share" and getattr(cfg, "im_sa", 3) > 0:
        im = self_attention_block(p, "pt_sa", im, n_sa)
    pt = extract_pt_feat(p, cfg, pt_feat, pt3d)
    if getattr(cfg, "coarse_layers", 1) > 0:
        im = encoder_layer(p, "coarse_former", im, pt)
        pt = encoder_layer(p, "coarse_former", pt, im)
    conf, im_n, pt_n = coarse_matching(im, pt, p["temperature"], im_mask, pt_mask, getattr(cfg, "temp_type", "mul"))
    ids, mconf = mutual_matches(conf, mutual=mutual, threshold=match_thres)
    pred_num = len(ids[0])
    if conf_gt is not None:  # training: forward_match passes conf_gt on (c2f_trainer.py:333-339)
        ids, mconf, pred_num = pad_matches_with_gt(ids, mconf, conf_gt, getattr(cfg, "coarse_percent", 0.3))
    b_ids, i_ids, j_ids = ids
    # fine stage
    pf = F.linear(pt, p["pt_ffeat_proj.0.weight"], p["pt_ffeat_proj.0.bias"])
    pf = F.linear(pf, p["pt_ffeat_proj.1.weight"], p["pt_ffeat_proj.1.bias"])
    if b_ids.shape[0] == 0:
        expec = torch.empty(0, 3)
    else:
        win = fine_windows(ffeat_map, b_ids, i_ids, win=int(getattr(cfg, "win_sz", 5)))
        win = self_attention_block(p, "fine_sa", win, getattr(cfg, "fine_sa", 1), heads=8,
                                   att_type=getattr(cfg, "fsa_type", "full"))
        expec = fine_matching(pf[b_ids, j_ids], win)
    return dict(conf_matrix=conf, expec_f=expec, match_ids=ids, mconf=mconf, pred_mask=mconf != 0, pred_num=pred_num,
                im_cfeat=im_n, pt_cfeat=pt_n, im_tokens=im, pt_tokens=pt)


def c2f_assemble(preds, pt2d, pt3d, win_sz=5, fine_ds=2):
    """Match assembly of NeRFMatcherMS.forward.  nerfmatch/nerfmatch_c2f_trainer.py:457-483."""
    b_ids, i_ids, j_ids = preds["match_ids"]
    mpt2d_c = pt2d[b_ids, i_ids]
    mpt3d = pt3d[b_ids, j_ids]
    mpt2d_f = mpt2d_c + preds["expec_f"][:, :2] * win_sz / 2 * fine_ds
    keep = preds["pred_mask"]
    return dict(m_bids=b_ids[keep], mpt2d_c=mpt2d_c[keep], mpt2d_f=mpt2d_f[keep], mpt3d=mpt3d[keep])


def coarse_forward_match(p, cfeat_map, pt_feat, im_mask=None, pt_mask=None, mutual=False, match_thres=0.0,
                         temp_type="mul"):
    """NeRFMatch-Mini: backbone -> dual softmax -> mutual NN (im_pe/im_sa/pt_sa/pt_pe off,
    coarse_layers 0).  nerfmatch/nerfmatch_coarse_trainer.py:236-288 with
    configs/nerfmatch/nerfmatch_7scenes_sfm_coarse.yaml:32-48."""
    im = cfeat_map.flatten(-2).permute(0, 2, 1)
    conf, im_n, pt_n = coarse_matching(im, pt_feat, p["temperature"], im_mask, pt_mask, temp_type)
    ids, mconf = mutual_matches(conf, mutual=mutual, threshold=match_thres)
    return dict(conf_matrix=conf, match_ids=ids, mconf=mconf, im_cfeat=im_n, pt_cfeat=pt_n)


def coarse_forward_match_cfg(p, cfg, cfeat_map, pt_feat, pt3d, im_mask=None, pt_mask=None, mutual=False, match_thres=0.0):
    """NeRFMatcherCoarse.forward_match for ANY option values (the shipped Mini configuration is coarse_forward_match above):
    nerfmatch/nerfmatch_coarse_trainer.py:169-185 (image side), :186-224 (point side), :236-288."""
    b, c, h, w = cfeat_map.shape
    im = cfeat_map.flatten(-2).permute(0, 2, 1)
    if getattr(cfg, "im_pe", True):
        im = (cfeat_map + sine_pe_table(c, h, w)[None]).flatten(-2).permute(0, 2, 1)
    if getattr(cfg, "im_sa_type", None) == "share" and getattr(cfg, "im_sa", 3) > 0 and getattr(cfg, "pt_sa", 3) > 0:
        im = self_attention_block(p, "pt_sa", im, getattr(cfg, "pt_sa", 3))
    pt = extract_pt_feat(p, cfg, pt_feat, pt3d)
    if getattr(cfg, "coarse_layers", 1) > 0:
        im = encoder_layer(p, "coarse_former", im, pt)
        pt = encoder_layer(p, "coarse_former", pt, im)
    conf, im_n, pt_n = coarse_matching(im, pt, p["temperature"], im_mask, pt_mask, getattr(cfg, "temp_type", "mul"))
    ids, mconf = mutual_matches(conf, mutual=mutual, threshold=match_thres)
    return dict(conf_matrix=conf, match_ids=ids, mconf=mconf, im_cfeat=im_n, pt_cfeat=pt_n, im_tokens=im, pt_tokens=pt)


def pixel_grid(w, h, ds=8):
    """pt2d = cell * ds + ds/2, (h/ds * w/ds, 2) as (x,y).  nerfmatch/utils/geometry.py:94-104."""
    ys, xs = torch.meshgrid(torch.arange(int(h) // ds), torch.arange(int(w) // ds), indexing="ij")
    return (torch.stack([xs, ys], -1) * ds + ds / 2).float().reshape(-1, 2)

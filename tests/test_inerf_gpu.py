"""GPU parity of the iNeRF refinement: HIP forward/backward kernels against the oracle's
autograd (oracle/inerf_oracle.py) and against the trajectory the reference itself produced (tests/golden/inerf_*.npz)."""
import pytest
import torch

from conftest import load_golden
from nerfmatch_amd import inerf, ops, synth
from nerfmatch_amd.nerf.renderer import NerfRenderer
from oracle import inerf_oracle as io

pytestmark = pytest.mark.gpu


def build(fx, gpu):
    app = bool(fx["app"])
    H, W = int(fx["H"]), int(fx["W"])
    cfg = synth.nerf_config("cambridge" if app else "7scenes", num_pts=128, img_wh=(W, H))
    ren = NerfRenderer(cfg, num_frames=5 if app else None, training=False, stop_layer=3)
    sd = synth.nerf_state_dict(seed=int(fx["weights_seed"]), app_vocab=5 if app else 0, density_bias=3.0)
    ren.load_state_dict(sd, strict=True)
    return ren.to(gpu).eval(), sd, H, W


@pytest.mark.parametrize("tag", ["7s", "cam_decay"])
@pytest.mark.parametrize("skip", [True, False])
def test_step_gradient_vs_oracle_autograd(gpu, built_lib, tag, skip):
    """Loss and d loss / d pose of one step: hand-written backward vs torch autograd through the oracle."""
    fx = load_golden(f"inerf_{tag}")
    ren, sd, H, W = build(fx, gpu)
    un = fx["unnorm"]
    pose0 = un.inverse() @ fx["c2w_est0"]
    img = fx["image"][0].permute(1, 2, 0)
    img_ds = img[4::8, 4::8].contiguous().view(-1, 3)
    app_row = sd["embedding_a.weight"][1] if bool(fx["app"]) else None
    p = pose0.clone().requires_grad_(True)
    with torch.enable_grad():
        loss_ref, rgb_ref = io.step_loss(sd, p, fx["K"], H, W, img_ds, fx["t_rands"][0], fx["jitters"][0], app_row)
        loss_ref.backward()
    loss, g_pose, ctx = inerf.step_gradient(ren, pose0.to(gpu), fx["K"], H, W, img_ds.to(gpu), fx["t_rands"][0], fx["jitters"][0],
                                            skip_zero_tail=skip)
    assert abs(float(loss) - float(loss_ref)) < 1e-5
    assert (ctx["rgb_map"].cpu() - rgb_ref.detach()).abs().max().item() < 1e-4
    g_ref = p.grad
    scale = g_ref.abs().max().item()
    assert (g_pose.cpu() - g_ref).abs().max().item() < 2e-3 * scale + 1e-7, (g_pose.cpu(), g_ref)
    assert float(g_pose[3].abs().max()) == 0.0  # the homogeneous row has no influence


@pytest.mark.parametrize("tag", ["7s", "cam_decay"])
def test_refinement_trajectory_vs_reference(gpu, built_lib, tag):
    """Poses after every Adam step against the reference's own inerf_refinement."""
    fx = load_golden(f"inerf_{tag}")
    ren, sd, H, W = build(fx, gpu)
    un = fx["unnorm"]
    pose0 = (un.inverse() @ fx["c2w_est0"]).to(gpu)
    n = int(fx["num_optim"])
    poses, losses, _ = inerf.refine(ren, fx["K"], H, W, fx["image"][0].permute(1, 2, 0), pose0, num_optim=n, lrate=float(fx["lrate"]),
                                    lrdecay=bool(fx["lrdecay"]), t_rands=list(fx["t_rands"][:n]), jitters=list(fx["jitters"][:n]))
    got = torch.stack([un @ p.cpu() for p in poses])
    want = fx["poses"]
    got = got[-want.shape[0]:]
    # Adam normalises the gradient: the first step moves every entry by +-lr (x3, the scene scale), later steps depend on
    # ratios of successive gradients -- ill-conditioned for the entries whose gradient nearly cancels (one entry of this
    # fixture moves by 0.6 lr in step 2), so: nearly all entries to 3e-4, every entry to a fraction of one step.
    err = (got - want).abs()
    assert (err < 3e-4).float().mean().item() > 0.9 and err.max().item() < 2e-3, err
    assert losses[-1] == losses[-1]


def test_bf16x3_linear_path(gpu, built_lib):
    """The same step with the MLP GEMMs on the split-bf16 path."""
    fx = load_golden("inerf_7s")
    ren, sd, H, W = build(fx, gpu)
    un = fx["unnorm"]
    pose0 = (un.inverse() @ fx["c2w_est0"]).to(gpu)
    img_ds = fx["image"][0].permute(1, 2, 0)[4::8, 4::8].contiguous().view(-1, 3).to(gpu)
    l32, g32, _ = inerf.step_gradient(ren, pose0, fx["K"], H, W, img_ds, fx["t_rands"][0], fx["jitters"][0])
    ops.LINEAR_PRECISION = "bf16x3"
    try:
        l16, g16, _ = inerf.step_gradient(ren, pose0, fx["K"], H, W, img_ds, fx["t_rands"][0], fx["jitters"][0])
    finally:
        ops.LINEAR_PRECISION = "fp32"
    assert abs(float(l32) - float(l16)) < 1e-5
    assert (g32 - g16).abs().max().item() < 5e-3 * g32.abs().max().item()


# ------------------------------------------------------------------------------------------------ matching term (use_match_loss)
def _match_setup(gpu):
    from nerfmatch_amd.matcher import NeRFMatcherMS
    from nerfmatch_amd.modules import PrecomputedBackbone

    fx = load_golden("inerf_match")
    ren, sd, H, W = build(fx, gpu)
    seed = int(fx["weights_seed"])
    model = NeRFMatcherMS(synth.matcher_config("c2f"))
    msd = synth.matcher_state_dict("c2f", seed=seed)
    model.load_state_dict(msd, strict=False)
    model.backbone = PrecomputedBackbone((fx["cfeat"].to(gpu), fx["ffeat"].to(gpu)), [256, 128])
    model.to(gpu).eval()
    un = fx["unnorm"]
    R = (H // 8) * (W // 8)
    match = dict(model=model, image=fx["image"].to(gpu), unnorm=un.to(gpu), im_mask=torch.ones(1, R, dtype=torch.bool, device=gpu),
                 pt_mask=torch.ones(1, R, dtype=torch.bool, device=gpu))
    omatch = dict(p=msd, cfg=synth.matcher_config("c2f"), cfeat=fx["cfeat"], ffeat=fx["ffeat"], unnorm=un)
    return fx, ren, sd, H, W, match, omatch


def test_ray_sums_and_weight_gradient_kernels(gpu, built_lib):
    """nm_inerf_composite_ex / _bwd_ex (weights and their extra gradient) and nm_inerf_ray_sums / _bwd against torch autograd
    over the oracle's compositing and frustum means."""
    from oracle import nerf_oracle as no

    g = torch.Generator().manual_seed(11)
    R, S, Sa, Cf = 37, 16, 9, 256
    o = torch.randn(R, 3, generator=g) * 0.2
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    rays = torch.cat([o, d, torch.full((R, 1), 0.01), torch.ones(R, 1), d, torch.full((R, 1), 0.002)], -1)
    z = torch.sort(torch.rand(R, S + 1, generator=g) * 0.9 + 0.05, dim=-1).values
    logit = torch.randn(R * Sa, 8, generator=g)
    sig = torch.randn(R * Sa, 8, generator=g) * 3 + 1
    feats = torch.relu(torch.randn(R * Sa, Cf, generator=g))
    G = torch.randn(R, 3, generator=g)
    g_pf = torch.randn(R, Cf, generator=g)
    g_pts = torch.randn(R, 3, generator=g)
    # reference: autograd over the oracle
    lg, sg, ft = logit[:, :3].clone().requires_grad_(True), sig[:, :1].clone().requires_grad_(True), feats.clone().requires_grad_(True)
    with torch.enable_grad():
        raw = torch.cat([torch.sigmoid(lg), sg], -1).reshape(R, Sa, 4)
        out = no.composite(raw, z[:, : Sa + 1], d, white_bg=True)
        mean, _ = no.frustum_gaussians(z[:, : Sa + 1], o, d, rays[:, 11:12])
        w_ref = out[3]
        pf_ref = (w_ref[..., None] * ft.reshape(R, Sa, Cf)).sum(-2)
        pts_ref = (w_ref[..., None] * mean).sum(-2)
        ((out[0] * G).sum() + (pf_ref * g_pf).sum() + (pts_ref * g_pts).sum()).backward()
    # kernels
    dv = lambda t: t.to(gpu).contiguous()
    rgb, w = inerf._composite(dv(logit), dv(sig), dv(z), dv(rays), Sa, want_weights=True)
    assert (w.cpu() - w_ref.detach()).abs().max().item() < 1e-6
    pf, pts = inerf._ray_sums(w, dv(feats), dv(rays), dv(z), Sa)
    assert (pf.cpu() - pf_ref.detach()).abs().max().item() < 1e-5
    assert (pts.cpu() - pts_ref.detach()).abs().max().item() < 1e-6
    g_feats, g_w = inerf._ray_sums_bwd(w, dv(feats), dv(rays), dv(z), Sa, dv(g_pf), dv(g_pts))
    assert (g_feats.cpu() - ft.grad).abs().max().item() < 1e-5
    g_logit, g_sig, _ = inerf._composite_bwd(dv(logit), dv(sig), dv(z), dv(rays), Sa, dv(G), g_w)
    sc = sg.grad.abs().max().item()
    assert (g_sig[:, :1].cpu() - sg.grad).abs().max().item() < 2e-5 * sc, ((g_sig[:, :1].cpu() - sg.grad).abs().max(), sc)
    assert (g_logit[:, :3].cpu() - lg.grad).abs().max().item() < 1e-5 * lg.grad.abs().max().item() + 1e-7
    # without the extra term the sigma gradient is a different one (the test would notice a dropped g_w)
    _, g_sig0, _ = inerf._composite_bwd(dv(logit), dv(sig), dv(z), dv(rays), Sa, dv(G))
    assert (g_sig0[:, :1].cpu() - sg.grad).abs().max().item() > 1e-2 * sc


def test_fourier_backward_vs_autograd(gpu, built_lib):
    from oracle import matcher_oracle as mo

    g = torch.Generator().manual_seed(12)
    n, Cc = 50, 256
    x = (torch.randn(n, 3, generator=g) * 2.0).requires_grad_(True)
    dy = torch.randn(n, ((Cc + 93 + 7) // 8) * 8, generator=g)
    with torch.enable_grad():
        (mo.fourier_embed(x) * dy[:, Cc : Cc + 93]).sum().backward()
    got = ops.cat_fourier_bwd(dy.to(gpu), x.detach().to(gpu), Cc, 15).cpu()
    # 2^14 x carries ~1e-3 rad of argument rounding at these magnitudes, times the 2^14 factor of the derivative
    assert (got - x.grad).abs().max().item() < 2e-3 * x.grad.abs().max().item(), (got - x.grad).abs().max()


def test_match_term_gradients_vs_oracle(gpu, built_lib):
    """d match_loss / d pt_feat and d match_loss / d pt3d for the SAME rendered inputs: the matcher's HIP backward against
    torch autograd over the oracle's forward_match + focal loss."""
    from oracle import matcher_oracle as mo
    from oracle import train_oracle as to

    fx, ren, sd, H, W, match, om = _match_setup(gpu)
    g = torch.Generator().manual_seed(13)
    R = (H // 8) * (W // 8)
    pt_feat = torch.relu(torch.randn(R, 256, generator=g))
    pt3d = torch.randn(R, 3, generator=g) * 0.25  # small coordinates keep the 2^14-frequency features well conditioned
    pf, p3 = pt_feat[None].clone().requires_grad_(True), pt3d[None].clone().requires_grad_(True)
    with torch.enable_grad():
        preds = mo.c2f_forward_match(om["p"], om["cfg"], om["cfeat"], om["ffeat"], pf, p3, mutual=True)
        loss_ref = to.matching_loss(preds["conf_matrix"], torch.eye(R)[None])
        loss_ref.backward()
    loss, g_pf, g_p3 = inerf._match_term(match, pt_feat.to(gpu), pt3d.to(gpu))
    assert abs(float(loss) - float(loss_ref)) < 1e-5 * max(1.0, abs(float(loss_ref)))
    assert (g_pf.cpu() - pf.grad[0]).abs().max().item() < 2e-3 * pf.grad.abs().max().item()
    assert (g_p3.cpu() - p3.grad[0]).abs().max().item() < 5e-3 * p3.grad.abs().max().item()
    assert all(p.requires_grad for p in match["model"].parameters())  # un-frozen again


@pytest.mark.parametrize("skip", [True, False])
def test_match_loss_step_vs_oracle_and_reference(gpu, built_lib, skip):
    """Loss and pose gradient of the first step with the matching term, against the oracle's autograd and the gradient the
    reference's own run recorded.  One ulp on the rendered points moves this gradient by ~6e-3 of its size (2^14-frequency
    Fourier features of world coordinates; tests/test_oracle_golden.py), hence the tolerance; the kernel-level tests above are
    the tight ones."""
    fx, ren, sd, H, W, match, om = _match_setup(gpu)
    un = fx["unnorm"]
    pose0 = un.inverse() @ fx["c2w_est0"]
    img_ds = fx["image"][0].permute(1, 2, 0)[4::8, 4::8].contiguous().view(-1, 3)
    p = pose0.clone().requires_grad_(True)
    with torch.enable_grad():
        loss_ref, _ = io.step_loss(sd, p, fx["K"], H, W, img_ds, fx["t_rands"][0], fx["jitters"][0], match=om)
        loss_ref.backward()
    loss, g_pose, _ = inerf.step_gradient(ren, pose0.to(gpu), fx["K"], H, W, img_ds.to(gpu), fx["t_rands"][0], fx["jitters"][0],
                                          skip_zero_tail=skip, match=match)
    assert abs(float(loss) - float(loss_ref)) < 2e-3 * abs(float(loss_ref))
    for want in (p.grad, fx["pose_grads"][0]):
        scale = want.abs().max().item()
        assert (g_pose.cpu() - want).abs().max().item() < 5e-2 * scale, (g_pose.cpu(), want)
    l0, g0, _ = inerf.step_gradient(ren, pose0.to(gpu), fx["K"], H, W, img_ds.to(gpu), fx["t_rands"][0], fx["jitters"][0])
    assert (g_pose - g0).abs().max().item() > 0.2 * g_pose.abs().max().item()  # the matching term dominates this gradient


def test_match_loss_trajectory_vs_reference(gpu, built_lib):
    fx, ren, sd, H, W, match, om = _match_setup(gpu)
    un = fx["unnorm"]
    pose0 = (un.inverse() @ fx["c2w_est0"]).to(gpu)
    n = int(fx["num_optim"])
    poses, losses, _ = inerf.refine(ren, fx["K"], H, W, fx["image"][0].permute(1, 2, 0), pose0, num_optim=n, lrate=float(fx["lrate"]),
                                    t_rands=list(fx["t_rands"][:n]), jitters=list(fx["jitters"][:n]), match=match)
    got = torch.stack([un @ p.cpu() for p in poses])
    err = (got - fx["poses"]).abs()
    # Adam: +-lr per entry in the first step (x3 scene scale), ratios of noisy gradients afterwards
    assert err[0].max().item() < 1e-5 and err.max().item() < 2e-3, err


def test_match_loss_trajectory_coarse_model_vs_reference(gpu, built_lib):
    """Round 6: `use_match_loss` with the coarse-only model class -- the reference's call takes either class
    (nerfmatch_evaluator.py:429-441) -- against the reference's own trajectory (tests/golden/inerf_match_coarse.npz), through
    NeRFMatchEvaluator.inerf_refinement."""
    from argparse import Namespace

    from nerfmatch_amd.modules import PrecomputedBackbone
    from nerfmatch_amd.nerfmatch_evaluator import NeRFMatchEvaluator

    fx = load_golden("inerf_match_coarse")
    ren, sd, H, W = build(fx, gpu)
    seed = int(fx["weights_seed"])
    cfg, msd = synth.matcher_variant("coarse_full", seed)
    ev = NeRFMatchEvaluator(Namespace(model=cfg, exp=Namespace(seed=1), data=Namespace()))
    assert ev.coarse_only
    ev.model.load_state_dict(msd, strict=False)
    ev.model.backbone = PrecomputedBackbone(fx["cfeat"].to(gpu), 256)
    ev.model.to(gpu).eval()
    R = (H // 8) * (W // 8)
    n = int(fx["num_optim"])
    batch = dict(image=fx["image"].to(gpu), K=fx["K"][None], c2w=fx["c2w_gt"][None], im_mask=torch.ones(1, R, dtype=torch.bool, device=gpu),
                 pt_mask=torch.ones(1, R, dtype=torch.bool, device=gpu))
    conf = Namespace(lrate=float(fx["lrate"]), lrdecay=False, num_optim=n, eval_pose=True, ds=8, use_match_loss=True)
    est, R_err, t_err = ev.inerf_refinement(batch, ren, fx["unnorm"], fx["c2w_est0"], conf, t_rands=list(fx["t_rands"]), jitters=list(fx["jitters"]))
    assert (est - fx["poses"][-1]).abs().max().item() < 2e-3 and abs(t_err - float(fx["t_err"])) < 2e-3
    no_match = Namespace(**{**vars(conf), "use_match_loss": False})
    est0, _, _ = ev.inerf_refinement(batch, ren, fx["unnorm"], fx["c2w_est0"], no_match, t_rands=list(fx["t_rands"]), jitters=list(fx["jitters"]))
    assert (est0 - est).abs().max().item() > 1e-3  # a different trajectory without the term


# ------------------------------------------------------------------------------------------------ fused pointwise kernels, tapped layer
def _points_case(gpu, app, R, Sa, seed):
    """rays / fence posts of a small bundle + the fine network both as GEMM chain (fp32) and as fused kernels"""
    S = 128
    cfg = synth.nerf_config("cambridge" if app else "7scenes", num_pts=S)
    ren = NerfRenderer(cfg, num_frames=5 if app else None, training=False, stop_layer=3)
    ren.load_state_dict(synth.nerf_state_dict(seed=seed, app_vocab=5 if app else 0, density_bias=3.0))
    ren.to(gpu).eval()
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(R, 3, generator=g) * 0.2
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    rays = torch.cat([o, d, torch.full((R, 1), 0.01), torch.ones(R, 1), d, torch.full((R, 1), 0.002)], -1).to(gpu).contiguous()
    z = torch.sort(torch.rand(R, S + 1, generator=g) * 0.9 + 0.05, dim=-1).values.to(gpu).contiguous()
    app_row = ren.embedding_a.weight[1].detach().float().contiguous() if app else None
    return ren, rays, z, app_row, g


@pytest.mark.parametrize("app,R,Sa,tap", [(False, 61, 65, 3), (True, 37, 65, 7), (False, 19, 128, 0), (False, 300, 65, 5)])
def test_tapped_points_kernels_vs_gemm_chain(gpu, built_lib, app, R, Sa, tap):
    """nm_nerf_points_fwd_rays_tap_bf16x3 / nm_nerf_points_bwd_tap_bf16x3 (round 5: the matching term on the fused pair) against the fp32
    GEMM chain (FineField, which the oracle / reference tests above pin): tapped activations, outputs, and d loss / d (xi, xd) with a
    gradient entering at the tapped layer as w_n . g_pt_feat[ray] -- including a ragged last tile (R Sa is no multiple of 128)."""
    ren, rays, z, app_row, g = _points_case(gpu, app, R, Sa, seed=21 + tap)
    n = R * Sa
    chain = inerf.FineField(ren.nerf_fine, gpu)
    xi, xd = inerf._encode(rays, z, Sa, app_row)
    logit, sig, saved = chain.forward(xi, xd)
    fused = inerf.FusedField(ren.nerf_fine, gpu)
    out4, gates, feats = fused.forward_rays(rays, z, Sa, app_row, tap)
    h_ref = saved[0][tap]
    assert feats.shape == (n, 256)
    assert (feats - h_ref).abs().max().item() < 1e-5 * max(1.0, h_ref.abs().max().item())
    assert ((feats > 0) == (h_ref > 0)).float().mean().item() > 0.9999  # (a ReLU may flip where the pre-activation is within rounding of zero)
    assert (out4[:, :3] - logit[:, :3]).abs().max().item() < 1e-5 * max(1.0, logit.abs().max().item())
    assert (out4[:, 3] - sig[:, 0]).abs().max().item() < 1e-5 * max(1.0, sig.abs().max().item())
    # the un-tapped entry point computes the same outputs and gates, bit for bit
    out4_b, gates_b = fused.forward_rays(rays, z, Sa, app_row)
    assert torch.equal(out4_b, out4) and torch.equal(gates_b, gates)
    # backward
    g_logit = torch.zeros(n, 8, device=gpu)
    g_logit[:, :3] = torch.randn(n, 3, generator=g).to(gpu) * 1e-4
    g_sig = torch.zeros(n, 8, device=gpu)
    g_sig[:, 0] = torch.randn(n, generator=g).to(gpu) * 1e-5
    w = torch.rand(R, Sa, generator=g).to(gpu) * 0.1
    g_pf = torch.randn(R, 256, generator=g).to(gpu) * 1e-3
    g_feats = (w.reshape(n, 1) * g_pf.repeat_interleave(Sa, 0)).contiguous()
    gxi_ref, gxd_ref = chain.backward(g_logit, g_sig, saved, (tap, g_feats))
    g4 = torch.cat([g_logit[:, :3], g_sig[:, :1]], 1).contiguous()
    (a0, a5), gxd = fused.backward(g4, gates, (tap, w, g_pf))
    gxi = a0 + a5
    # The two passes round differently, so a ReLU whose pre-activation is within rounding of zero can gate differently: a few samples in
    # a thousand then carry a different (equally valid) sub-gradient.  Hence: nearly every row to 2e-4 of the largest entry, all rows
    # together to 1e-2 in the L2 sense (a wrong layer or column order would be an O(1) error in every row).
    for got, want in ((gxi, gxi_ref), (gxd, gxd_ref)):
        assert torch.isfinite(got).all()
        row = (got - want).abs().max(1).values / want.abs().max().item()
        assert (row <= 2e-4).float().mean().item() >= 0.995, (row > 2e-4).sum().item()
        assert ((got - want).norm() / want.norm()).item() < 1e-2
    # given the gates the pass is linear: (photometric + matching) = photometric alone + matching alone
    (m0, m5), m_xd = fused.backward(torch.zeros_like(g4), gates, (tap, w, g_pf))
    (p0, p5), p_xd = fused.backward(g4, gates)
    lin = (m0 + m5) + (p0 + p5)
    assert (lin - gxi).abs().max().item() < 1e-5 * gxi.abs().max().item()
    # the injected gradient matters (a dropped addend would be noticed) ...
    (b0, b5), _ = fused.backward(g4, gates)
    assert ((b0 + b5) - gxi_ref).abs().max().item() > 1e-2 * gxi_ref.abs().max().item()
    # ... and a zero one changes nothing
    (c0, c5), c_xd = fused.backward(g4, gates, (tap, torch.zeros_like(w), g_pf))
    assert torch.equal(c0, b0) and torch.equal(c5, b5)


def test_ray_sums_bwd_without_the_feature_gradient(gpu, built_lib):
    """g_feats NULL: the weights' gradient alone, the same bits"""
    g = torch.Generator().manual_seed(5)
    R, S, Sa = 23, 16, 9
    o = torch.randn(R, 3, generator=g) * 0.2
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    rays = torch.cat([o, d, torch.full((R, 1), 0.01), torch.ones(R, 1), d, torch.full((R, 1), 0.002)], -1).to(gpu).contiguous()
    z = torch.sort(torch.rand(R, S + 1, generator=g) * 0.9 + 0.05, dim=-1).values.to(gpu).contiguous()
    w, feats = torch.rand(R, Sa, generator=g).to(gpu), torch.rand(R * Sa, 256, generator=g).to(gpu)
    g_pf, g_pts = torch.randn(R, 256, generator=g).to(gpu), torch.randn(R, 3, generator=g).to(gpu)
    a_f, a_w = inerf._ray_sums_bwd(w, feats, rays, z, Sa, g_pf, g_pts)
    b_f, b_w = inerf._ray_sums_bwd(w, feats, rays, z, Sa, g_pf, g_pts, want_g_feats=False)
    assert b_f is None and torch.equal(a_w, b_w)


@pytest.mark.parametrize("skip", [True, False])
def test_match_loss_step_on_the_fused_pair(gpu, built_lib, skip):
    """The step with the matching term under the split arithmetic: fused kernel pair (tapped) against the bf16x3 GEMM chain of round 4 and
    against the oracle's autograd / the reference's recorded gradient (the tolerances of test_match_loss_step_vs_oracle_and_reference)."""
    fx, ren, sd, H, W, match, om = _match_setup(gpu)
    un = fx["unnorm"]
    pose0 = un.inverse() @ fx["c2w_est0"]
    img_ds = fx["image"][0].permute(1, 2, 0)[4::8, 4::8].contiguous().view(-1, 3)
    p = pose0.clone().requires_grad_(True)
    with torch.enable_grad():
        loss_ref, _ = io.step_loss(sd, p, fx["K"], H, W, img_ds, fx["t_rands"][0], fx["jitters"][0], match=om)
        loss_ref.backward()
    args = (ren, pose0.to(gpu), fx["K"], H, W, img_ds.to(gpu), fx["t_rands"][0], fx["jitters"][0])
    ops.LINEAR_PRECISION = "bf16x3"
    try:
        assert inerf.FUSED_FINE
        loss, g_pose, _ = inerf.step_gradient(*args, skip_zero_tail=skip, match=match)
        inerf.FUSED_FINE = False
        loss_c, g_chain, _ = inerf.step_gradient(*args, skip_zero_tail=skip, match=match)
    finally:
        inerf.FUSED_FINE = True
        ops.LINEAR_PRECISION = "fp32"
    assert abs(float(loss) - float(loss_ref)) < 2e-3 * abs(float(loss_ref))
    assert abs(float(loss) - float(loss_c)) < 2e-3 * abs(float(loss_c))
    for want in (p.grad, fx["pose_grads"][0], g_chain.cpu()):
        scale = want.abs().max().item()
        assert (g_pose.cpu() - want).abs().max().item() < 5e-2 * scale, (g_pose.cpu(), want)
    assert float(g_pose[3].abs().max()) == 0.0


@pytest.mark.parametrize("R,S,Sa,two", [(37, 40, 33, False), (21, 128, 128, True), (9, 160, 129, True), (5, 64, 64, False)])
def test_encode_backward_vs_fp64_sums(gpu, built_lib, R, S, Sa, two):
    """nm_inerf_encode_bwd / _bwd2 alone (round 6: rewritten for row-order loads): d loss / d origin and d loss / d view direction of every ray
    for random upstream gradients on the IPE columns (one or two contributions) and the view-direction row, against fp64 sums of the same
    expression (nerfmatch_evaluator.py:364-393; the Gaussians' variances see detached rays).  A 1-ulp difference of a sample's mean is a phase
    difference of 2^i ulp at frequency i, so the inputs are chosen such that the fp32 mean is the SAME number however it is formed: fence posts on
    a 2^-7 grid (mu, hw, their squares and 3 mu^2 + hw^2 are exact; t_mean is one correctly rounded division and one sum), direction components
    +-1 / +-0.5 (t_mean * v exact: fused or not, o + t_mean v rounds once), origins on a 2^-6 grid."""
    import numpy as np
    from nerfmatch_amd import _lib
    from oracle import nerf_oracle as no

    g = torch.Generator().manual_seed(R * 1000 + Sa)
    o = torch.randint(-19, 20, (R, 3), generator=g).float() / 64
    v = torch.tensor([-1.0, -0.5, 0.5, 1.0])[torch.randint(0, 4, (R, 3), generator=g)]
    rays = torch.cat([o, v, torch.full((R, 1), 0.05), torch.full((R, 1), 1.5), v, torch.full((R, 1), 2e-3)], 1).contiguous()
    grid = torch.stack([torch.sort(torch.randperm(185, generator=g)[: S + 1]).values for _ in range(R)]).float()
    z = ((grid + 7) / 128).contiguous()
    n = R * Sa
    gxi_a, gxi_b = torch.randn(n, 96, generator=g), torch.randn(n, 96, generator=g)
    gxd = torch.randn(n, 48, generator=g)
    # the kernel's fp32 arguments, re-formed on the host in its order of operations
    f32 = np.float32
    zn = z.numpy()
    t0, t1 = zn[:, :-1], zn[:, 1:]
    mu, hw = (t0 + t1) / f32(2), (t1 - t0) / f32(2)
    hw2 = hw * hw
    denom = np.maximum(f32(1.1920928955078125e-07), f32(3) * (mu * mu) + hw2)
    t_mean32 = (mu + (f32(2) * mu * hw2) / denom).astype(f32)[:, :Sa]
    mean32 = (o.numpy()[:, None, :] + t_mean32[..., None] * v.numpy()[:, None, :]).astype(f32)
    _, var = no.frustum_gaussians(z.double(), o.double(), v.double(), rays[:, 11:12].double())
    var = var.reshape(R, S, 3)[:, :Sa]
    sc = 2.0 ** torch.arange(15, dtype=torch.float64)
    xe = (torch.from_numpy(mean32).double()[:, :, None, :] * sc[:, None]).reshape(R, Sa, 45)
    damp = torch.exp(-0.5 * (var[:, :, None, :] * (sc * sc)[:, None])).reshape(R, Sa, 45)
    gs = ((gxi_a + gxi_b) if two else gxi_a).double().reshape(R, Sa, 96)  # (the kernel adds the two contributions in fp32 first)
    sc3 = sc.repeat_interleave(3)
    gm = ((gs[..., :45] * (damp * torch.cos(xe)) - gs[..., 45:90] * (damp * torch.sin(xe))) * sc3).reshape(R, Sa, 15, 3).sum(2)  # d loss / d mean
    want_o = gm.sum(1)
    gd = gxd.double().reshape(R, Sa, 48).sum(1)
    vv = v.double()
    want_v = (gm * torch.from_numpy(t_mean32).double()[..., None]).sum(1) + gd[:, 24:27]
    for k in range(4):
        # the forward's second half is sin(fl(x + pi/2)) with x = v 2^k <= 8: its derivative at the rounded argument
        arg2 = (v * 2.0**k + 0.5 * torch.pi).double()
        want_v = want_v + (gd[:, 3 * k:3 * k + 3] * torch.cos(vv * 2.0**k) + gd[:, 12 + 3 * k:15 + 3 * k] * torch.cos(arg2)) * 2.0**k
    dev = lambda t: t.to(gpu).contiguous()
    rays_d, z_d, ga_d, gb_d, gd_d = dev(rays), dev(z), dev(gxi_a), dev(gxi_b), dev(gxd)
    g_o, g_v = torch.empty(R, 3, device=gpu), torch.empty(R, 3, device=gpu)
    L, p = _lib.lib(), _lib.dptr
    if two:
        _lib.check(L.nm_inerf_encode_bwd2(p(rays_d), p(z_d), R, S, Sa, p(ga_d), p(gb_d), p(gd_d), p(g_o), p(g_v), ops.stream()), "nm_inerf_encode_bwd2")
    else:
        _lib.check(L.nm_inerf_encode_bwd(p(rays_d), p(z_d), R, S, Sa, p(ga_d), p(gd_d), p(g_o), p(g_v), ops.stream()), "nm_inerf_encode_bwd")
    so, sv = float(want_o.abs().max()), float(want_v.abs().max())
    eo = float((g_o.cpu().double() - want_o).abs().max()) / so
    evv = float((g_v.cpu().double() - want_v).abs().max()) / sv
    print(f"encode backward R={R} S={S} Sa={Sa} two={two}: |g_o - fp64| {eo:.2e}, |g_v - fp64| {evv:.2e} of the largest entry")
    assert eo < 1e-6 and evv < 1e-6  # fp32 accumulation of Sa * 45 terms, fp32 sine / cosine / exponential (measured: 0.9e-7 ... 1.8e-7)


@pytest.mark.parametrize("R,S,Sa,with_gw", [(50, 64, 33, False), (37, 128, 65, True), (9, 256, 129, True), (4801, 128, 128, False)])
def test_wavefront_compositing_vs_per_ray_loops(gpu, built_lib, R, S, Sa, with_gw):
    """nm_inerf_composite4 / _bwd (round 6: one wavefront per ray, prefix product and suffix sum over lanes, the fused field's (n, 4) rows as they
    lie) against the sequential one-thread-per-ray kernels (nm_inerf_composite_ex / _bwd_ex, themselves pinned by the step-gradient tests against
    the oracle's autograd): same expression, different association of the products and sums."""
    g = torch.Generator().manual_seed(R + Sa)
    n = R * Sa
    out4 = torch.randn(n, 4, generator=g)
    out4[:, 3] = out4[:, 3] * 40 + 10  # raw sigma: ~40 % negative (ReLU gate closed), opacity from thin to saturating
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    rays = torch.cat([torch.zeros(R, 3), d, torch.full((R, 1), 0.05), torch.full((R, 1), 1.5), d, torch.full((R, 1), 2e-3)], 1)
    z = 0.05 + 1.45 * torch.sort(torch.rand(R, S + 1, generator=g), dim=1).values
    G = torch.randn(R, 3, generator=g)
    g_w = torch.randn(R, Sa, generator=g) if with_gw else None
    dev = lambda t: None if t is None else t.to(gpu).contiguous()
    out4_d, rays_d, z_d, G_d, gw_d = dev(out4), dev(rays), dev(z), dev(G), dev(g_w)
    rgb4, w4 = inerf._composite4(out4_d, z_d, rays_d, Sa, want_weights=True)
    g4, gd4 = inerf._composite4_bwd(out4_d, z_d, rays_d, Sa, G_d, gw_d)
    sig = torch.zeros_like(out4_d)
    sig[:, 0] = out4_d[:, 3]
    rgb, w = inerf._composite(out4_d, sig, z_d, rays_d, Sa, want_weights=True)
    g_logit, g_sig, gd = inerf._composite_bwd(out4_d, sig, z_d, rays_d, Sa, G_d, gw_d)
    rel = lambda a, b: float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
    errs = dict(rgb=rel(rgb4, rgb), w=rel(w4, w), g_logit=rel(g4[:, :3], g_logit[:, :3]), g_sigma=rel(g4[:, 3], g_sig[:, 0]), g_d=rel(gd4, gd))
    print(f"R={R} S={S} Sa={Sa}: " + "  ".join(f"{k} {v:.1e}" for k, v in errs.items()))
    assert all(v < 5e-6 for v in errs.values()), errs
    assert torch.equal(g4[:, 3] == 0, g_sig[:, 0] == 0) or float(((g4[:, 3] == 0) != (g_sig[:, 0] == 0)).float().mean()) < 1e-4  # same closed gates

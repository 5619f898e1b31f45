"""GPU parity of the training side: every backward kernel against torch-CPU autograd of the
same op, and one full c2f training step (losses, GT-padded match lists, gradients of every parameter and of the backbone
outputs) against the reference's own numbers (tests/golden/matcher_train.npz) and the training oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from nerfmatch_amd import autograd as ag
from nerfmatch_amd import ops, synth
from nerfmatch_amd.matcher import NeRFMatcherMS
from nerfmatch_amd.modules import PrecomputedBackbone
from oracle import matcher_oracle as mo
from oracle import train_oracle as to

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _grad_enabled():
    """The evaluator side of the package switches autograd off globally (like the reference, nerf_evaluator.py:155)."""
    with torch.enable_grad():
        yield


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def rel(a, b, floor=1e-3):
    """max |a - b| relative to the largest reference entry (at least `floor`)."""
    a, b = a.detach().cpu().double(), torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(floor)).item()


_WGRAD_SHAPES = [(4800, 256, 256), (333, 128, 352), (50, 96, 40), (19200, 768, 256), (4800, 4800, 256), (7, 8, 8), (100000, 128, 128), (1000, 70, 132),
                 (5000, 192, 100), (130, 64, 128)]


@pytest.mark.parametrize("M,N,K,precision", [(m, n, k, p) for (m, n, k) in _WGRAD_SHAPES for p in ("fp32", "bf16x3")
                                             if p == "bf16x3" or (n % 4 == 0 and k % 4 == 0)])  # (the fp32 kernel moves 16-byte row pieces)
def test_linear_wgrad_and_col_sum(gpu, built_lib, M, N, K, precision):
    """dW = dy^T x: the fp32-MFMA kernel and the split-bf16 one (round 6: lane = column, eight rows per operand register set straight from
    global memory) against fp64.  The split product carries 16 mantissa bits per operand; over M random-sign terms: < 1e-5 of the largest entry."""
    dy, x = rnd(M, N, seed=1), rnd(M, K, seed=2)
    ref = dy.double().T @ x.double()
    tol = 2e-6 if precision == "fp32" else 1e-5
    ops.LINEAR_PRECISION, keep = precision, ops.LINEAR_PRECISION
    try:
        dw = ops.linear_wgrad(dy.to(gpu), x.to(gpu))
        assert rel(dw, ref) < tol
        acc = torch.ones(N, K, device=gpu)
        ops.linear_wgrad(dy.to(gpu), x.to(gpu), out=acc)
        assert rel(acc, ref + 1.0) < tol
        again = ops.linear_wgrad(dy.to(gpu), x.to(gpu))
        assert torch.equal(again, dw)  # partial tiles summed in a fixed order
        dw2, db2 = ops.linear_wgrad_bias(dy.to(gpu), x.to(gpu))  # (one launch for both on the split path)
        assert torch.equal(dw2, dw) and rel(db2, dy.double().sum(0)) < 2e-6
    finally:
        ops.LINEAR_PRECISION = keep
    assert rel(ops.col_sum(dy.to(gpu)), dy.double().sum(0)) < 2e-6


def test_gelu_and_backward(gpu, built_lib):
    u = (rnd(1000, 256, seed=3) * 2.0).requires_grad_()
    dh = rnd(1000, 256, seed=4)
    h = F.gelu(u)
    h.backward(dh)
    assert rel(ops.gelu(u.detach().to(gpu)), h) < 1e-6
    assert rel(ops.gelu_bwd(u.detach().to(gpu), dh.to(gpu)), u.grad) < 2e-6


@pytest.mark.parametrize("rows,dim", [(4800, 256), (701, 128), (3, 256)])
def test_layernorm_backward(gpu, built_lib, rows, dim):
    x = (rnd(rows, dim, seed=1, scale=3.0) + 0.5).requires_grad_()
    g = (1 + 0.1 * rnd(dim, seed=2)).requires_grad_()
    b = (0.1 * rnd(dim, seed=3)).requires_grad_()
    dy = rnd(rows, dim, seed=4)
    F.layer_norm(x, (dim,), g, b).backward(dy)
    dx, dg, db = ops.layernorm_bwd(x.detach().to(gpu), g.detach().to(gpu), dy.to(gpu))
    assert rel(dx, x.grad) < 1e-5 and rel(dg, g.grad) < 1e-5 and rel(db, b.grad) < 1e-5


def test_l2norm_backward(gpu, built_lib):
    f = rnd(500, 256, seed=1).requires_grad_()
    dy = rnd(500, 256, seed=2)
    (f / (f.norm(dim=-1, keepdim=True) + 1e-6)).backward(dy)
    assert rel(ops.l2norm_bwd(f.detach().to(gpu), dy.to(gpu)), f.grad) < 1e-5


@pytest.fixture(params=["fp32", "bf16x3"])
def attn_precision(request):
    ops.ATTENTION_PRECISION = request.param
    yield request.param
    ops.ATTENTION_PRECISION = "fp32"


@pytest.mark.parametrize("B,L,S,H,D", [(1, 80, 96, 8, 32), (2, 200, 333, 8, 32), (1, 33, 1000, 8, 32), (7, 25, 25, 8, 16), (1, 1, 1, 8, 32),
                                       (3, 129, 64, 8, 32), (1, 385, 127, 5, 32)])
def test_attention_backward(gpu, built_lib, attn_precision, B, L, S, H, D):
    q, k, v = (rnd(B, n, H * D, seed=s).requires_grad_() for n, s in ((L, 1), (S, 2), (S, 3)))
    d_o = rnd(B, L, H * D, seed=4)
    scale = D**-0.5
    qh, kh, vh = (t.view(B, -1, H, D).transpose(1, 2) for t in (q, k, v))
    att = torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1) @ vh
    o = att.transpose(1, 2).reshape(B, L, H * D)
    o.backward(d_o)
    dq, dk, dv = ops.attention_bwd(q.detach().to(gpu), k.detach().to(gpu), v.detach().to(gpu), o.detach().to(gpu), d_o.to(gpu), H, scale)
    # a single key makes dq exactly 0 (P (dP - D) with dP = D): floor of 0.1, i.e. 2e-6 absolute in fp32; the split-bf16
    # products (16 significand bits per operand) leave ~3e-6 of |dO||V| ~ 10 there
    tol = 5e-4 if (S == 1 and attn_precision == "bf16x3") else 2e-5
    assert rel(dq, q.grad, 0.1) < tol and rel(dk, k.grad, 0.1) < tol and rel(dv, v.grad, 0.1) < tol


def test_attention_backward_full_size_property(gpu, built_lib, attn_precision):
    """4800 x 4800 tokens: gradients of sum(O * W) against torch's own attention backward on the GPU (fp32)."""
    B, L, S, H, D = 1, 4800, 4800, 8, 32
    q, k, v = (rnd(B, n, H * D, seed=s).to(gpu).requires_grad_() for n, s in ((L, 1), (S, 2), (S, 3)))
    d_o = rnd(B, L, H * D, seed=4).to(gpu)
    o_ref = F.scaled_dot_product_attention(q.view(B, L, H, D).transpose(1, 2), k.view(B, S, H, D).transpose(1, 2),
                                           v.view(B, S, H, D).transpose(1, 2), scale=D**-0.5).transpose(1, 2).reshape(B, L, H * D)
    o_ref.backward(d_o)
    o = ops.attention(q.detach(), k.detach(), v.detach(), H, D**-0.5)
    dq, dk, dv = ops.attention_bwd(q.detach(), k.detach(), v.detach(), o, d_o, H, D**-0.5)
    assert rel(dq, q.grad.cpu()) < 1e-4 and rel(dk, k.grad.cpu()) < 1e-4 and rel(dv, v.grad.cpu()) < 1e-4


@pytest.mark.parametrize("B,M,N,masked", [(1, 48, 64, False), (2, 100, 72, True), (1, 600, 520, True)])
def test_coarse_match_loss_and_gradients(gpu, built_lib, B, M, N, masked):
    g = torch.Generator().manual_seed(5)
    im = rnd(B, M, 256, seed=1).requires_grad_()
    pt = rnd(B, N, 256, seed=2).requires_grad_()
    temp = torch.tensor(10.0, requires_grad=True)
    conf_gt = torch.zeros(B, M, N, dtype=torch.bool)
    for b in range(B):
        perm = torch.randperm(N, generator=g)[: min(M, N) // 2]
        conf_gt[b, torch.arange(len(perm)), perm] = True
        with torch.no_grad():
            pt[b, perm] = im[b, : len(perm)] + 0.3 * torch.randn(len(perm), 256, generator=g)
    im_mask = pt_mask = None
    if masked:
        im_mask, pt_mask = torch.ones(B, M, dtype=torch.bool), torch.ones(B, N, dtype=torch.bool)
        im_mask[0, -9:] = False
        pt_mask[B - 1, 4:17] = False
    conf, _, _ = mo.coarse_matching(im, pt, temp, im_mask, pt_mask)
    loss = to.matching_loss(conf, conf_gt)
    (3.0 * loss).backward()
    dev = lambda t: None if t is None else t.to(gpu)
    im_g, pt_g = im.detach().to(gpu).requires_grad_(), pt.detach().to(gpu).requires_grad_()
    temp_g = temp.detach().to(gpu).requires_grad_()
    out = ag.coarse_match_loss(im_g, pt_g, temp_g, 10.0, dev(im_mask), dev(pt_mask), dev(conf_gt))
    assert abs(out[0].item() - loss.item()) < 1e-5 * max(1.0, abs(loss.item()))
    assert (out[1].cpu() - conf.detach()).abs().max() < 1e-5
    (3.0 * out[0]).backward()
    assert rel(im_g.grad, im.grad) < 1e-4 and rel(pt_g.grad, pt.grad) < 1e-4
    assert abs(temp_g.grad.item() - temp.grad.item()) < 1e-4 * max(abs(temp.grad.item()), 1e-3)
    # loss_only (what the pose refinement's matching term asks for): no confidence matrix, no selection -- same loss, same gradients, bit for bit
    im_l, pt_l = im.detach().to(gpu).requires_grad_(), pt.detach().to(gpu).requires_grad_()
    temp_l = temp.detach().to(gpu).requires_grad_()
    lo = ag.coarse_match_loss(im_l, pt_l, temp_l, 10.0, dev(im_mask), dev(pt_mask), dev(conf_gt), loss_only=True)
    assert lo[1].numel() == 0 and lo[2].numel() == 0 and torch.equal(lo[0], out[0])
    (3.0 * lo[0]).backward()
    assert torch.equal(im_l.grad, im_g.grad) and torch.equal(pt_l.grad, pt_g.grad) and torch.equal(temp_l.grad, temp_g.grad)
    # and the sums are deterministic (one-pass form: partial sums merged in a fixed order)
    again = ag.coarse_match_loss(im_g.detach(), pt_g.detach(), temp_g.detach(), 10.0, dev(im_mask), dev(pt_mask), dev(conf_gt), loss_only=True)
    assert torch.equal(again[0], out[0])


def test_focal_loss_two_kernel_form_for_ragged_widths(gpu, built_lib):
    """N % 4 != 0 takes the rounds-1-5 form of the loss (row sweep + column sweep); forward only -- the backward's GEMMs need N % 8 == 0."""
    g = torch.Generator().manual_seed(9)
    B, M, N = 1, 90, 70
    im, pt = rnd(B, M, 256, seed=3), rnd(B, N, 256, seed=4)
    conf_gt = torch.zeros(B, M, N, dtype=torch.bool)
    perm = torch.randperm(N, generator=g)[:30]
    conf_gt[0, torch.arange(30), perm] = True
    pt[0, perm] = im[0, :30] + 0.3 * torch.randn(30, 256, generator=g)
    conf, _, _ = mo.coarse_matching(im, pt, torch.tensor(10.0), None, None)
    loss = to.matching_loss(conf, conf_gt)
    ops.MATCH_PRECISION, keep = "fp32", ops.MATCH_PRECISION
    try:
        out = ag.coarse_match_loss(im.to(gpu), pt.to(gpu), torch.tensor(10.0, device=gpu), 10.0, None, None, conf_gt.to(gpu))
    finally:
        ops.MATCH_PRECISION = keep
    assert abs(out[0].item() - loss.item()) < 1e-5 * max(1.0, abs(loss.item()))
    assert (out[1].cpu() - conf).abs().max() < 1e-5


def test_fine_stage_backward(gpu, built_lib):
    K, C, win = 37, 128, 5
    pt_f = rnd(K, C, seed=1).requires_grad_()
    win_f = rnd(K, win * win, C, seed=2).requires_grad_()
    d = rnd(K, 3, seed=3)
    mo.fine_matching(pt_f, win_f).backward(d)
    cnt = torch.tensor([K], device=gpu, dtype=torch.int32)
    d_pt, d_win = ops.fine_expectation_bwd(pt_f.detach().to(gpu), win_f.detach().to(gpu), d.to(gpu), cnt, win)
    assert rel(d_pt, pt_f.grad) < 1e-5 and rel(d_win, win_f.grad) < 1e-5
    # window scatter: repeated and border cells
    ffeat = rnd(2, C, 24, 32, seed=4).requires_grad_()
    b_ids = torch.tensor([0, 1, 1, 0, 1, 0, 0])
    i_ids = torch.tensor([0, 47, 5, 0, 47, 7, 40])
    dw = rnd(len(b_ids), win * win, C, seed=5)
    mo.fine_windows(ffeat, b_ids, i_ids).backward(dw)
    fg = ffeat.detach().to(gpu).requires_grad_()
    w = ag.fine_windows(fg, b_ids.to(gpu), i_ids.to(gpu), win, 4)
    assert (w.cpu() - mo.fine_windows(ffeat, b_ids, i_ids).detach()).abs().max() == 0
    w.backward(dw.to(gpu))
    assert rel(fg.grad, ffeat.grad) < 1e-6


def build_model(fx, gpu):
    cfg = synth.matcher_config("c2f")
    model = NeRFMatcherMS(cfg)
    model.load_state_dict(synth.matcher_state_dict("c2f", seed=int(fx["weights_seed"])), strict=False)
    model = model.to(gpu)
    cfeat = fx["cfeat"].to(gpu).requires_grad_()
    ffeat = fx["ffeat"].to(gpu).requires_grad_()
    model.backbone = PrecomputedBackbone((cfeat, ffeat), [256, 128])
    return model, cfeat, ffeat


def batch(fx, gpu):
    t = lambda k: fx[k].to(gpu)
    B = fx["cfeat"].shape[0]
    d = dict(image=torch.zeros(B, 3, 8, 8, device=gpu), im_mask=t("im_mask"), pt_mask=t("pt_mask"), pt3d=t("pt3d"), pt2d=t("pt2d"),
             conf_gt=t("conf_gt"), pt2d_proj=t("pt2d_proj"))
    d["pt_feat"] = t("pt_feat").requires_grad_()
    return d


@pytest.mark.parametrize("coarse_only", [False, True])
def test_training_step_vs_reference(gpu, built_lib, coarse_only):
    """One c2f training step on the reference's own fixture: losses, sampled matches and gradients."""
    fx = load_golden("matcher_train")
    model, cfeat, ffeat = build_model(fx, gpu)
    data = batch(fx, gpu)
    np.random.seed(int(fx["np_seed"]))
    metrics = model.forward_with_metrics(data, training=True, coarse_only=coarse_only)
    assert abs(metrics["coarse_loss"].item() - float(fx["coarse_loss"])) < 2e-6 * float(fx["coarse_loss"]) + 1e-6
    b, i, j = data["match_ids"]
    assert torch.equal(b.cpu(), fx["b_ids"]) and torch.equal(i.cpu(), fx["i_ids"]) and torch.equal(j.cpu(), fx["j_ids"])
    assert data["pred_num"] == int(fx["pred_num"])
    assert (data["conf_matrix"].cpu() - fx["conf_matrix"]).abs().max() < 1e-6
    assert (data["expec_f"].detach().cpu() - fx["expec_f"]).abs().max() < 1e-4
    pre = "c_" if coarse_only else ""
    if not coarse_only:
        assert abs(metrics["fine_loss"].item() - float(fx["fine_loss"])) < 1e-4 * float(fx["fine_loss"])
        assert abs(metrics["loss"].item() - float(fx["loss"])) < 1e-4 * float(fx["loss"])
    metrics["loss"].backward()
    assert rel(cfeat.grad, fx[pre + "g_cfeat"]) < 1e-3
    assert rel(data["pt_feat"].grad, fx[pre + "g_pt_feat"]) < 1e-3
    if not coarse_only:
        assert rel(ffeat.grad.flatten()[::211], fx["g_ffeat_sub"]) < 1e-3
        assert abs(ffeat.grad.norm().item() - float(fx["g_ffeat_norm"])) < 1e-3 * float(fx["g_ffeat_norm"])
    names = dict(model.named_parameters())
    checked = 0
    for key in fx.keys():
        if not key.startswith(pre + "gn__"):
            continue
        name = key[len(pre) + 4:].replace("__", ".")
        g = names[name].grad
        assert g is not None, name
        gf = g.flatten()
        mine = gf if gf.numel() <= 512 else gf[::97]
        ref = fx[pre + "gs__" + key[len(pre) + 4:]]
        assert rel(mine, ref) < 2e-3, (name, rel(mine, ref))
        # (gradients that vanish analytically, e.g. the last fine_sa bias under the window soft-max, are rounding noise ~1e-8)
        assert abs(gf.norm().item() - float(fx[key])) < 1e-3 * float(fx[key]) + 1e-6, name
        checked += 1
    assert checked >= (50 if coarse_only else 65)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_training_step_mid_size_vs_oracle(gpu, built_lib, precision):
    """600 image tokens x 520 points (several workgroups per kernel, ragged tiles), B = 1: one full step against the oracle's
    autograd on the CPU, in both arithmetic configurations."""
    import nerfmatch_amd

    g = torch.Generator().manual_seed(11)
    h, w, N = 20, 30, 520
    M = h * w
    cfeat = torch.randn(1, 256, h, w, generator=g)
    ffeat = torch.randn(1, 128, 4 * h, 4 * w, generator=g)
    pt_feat = torch.relu(torch.randn(1, N, 256, generator=g))
    pt3d = torch.randn(1, N, 3, generator=g) * 2.0
    perm = torch.randperm(N, generator=g)[:300]
    pt_feat[0, perm] = torch.relu(cfeat[0].flatten(-2).T[:300]) + 0.05 * torch.randn(300, 256, generator=g)
    conf_gt = torch.zeros(1, M, N, dtype=torch.bool)
    conf_gt[0, torch.arange(300), perm] = True
    pt2d = mo.pixel_grid(w * 8, h * 8).reshape(1, -1, 2)
    pt2d_proj = torch.rand(1, N, 2, generator=g) * torch.tensor([w * 8.0, h * 8.0])
    pt2d_proj[0, perm] = pt2d[0, :300] + (torch.rand(300, 2, generator=g) - 0.5) * 6.0
    im_mask, pt_mask = torch.ones(1, M, dtype=torch.bool), torch.ones(1, N, dtype=torch.bool)
    pt_mask[0, 500:] = False
    cfg = synth.matcher_config("c2f")
    sd = synth.matcher_state_dict("c2f", seed=3)
    p = {k: v.clone().requires_grad_() for k, v in sd.items()}
    ptf = pt_feat.clone().requires_grad_()
    np.random.seed(21)
    ref = to.c2f_train_step(p, cfg, cfeat, ffeat, ptf, pt3d, pt2d, pt2d_proj, conf_gt, im_mask, pt_mask)
    ref["loss"].backward()
    model = NeRFMatcherMS(cfg)
    model.load_state_dict(sd, strict=False)
    model = model.to(gpu)
    cg, fg = cfeat.to(gpu).requires_grad_(), ffeat.to(gpu).requires_grad_()
    model.backbone = PrecomputedBackbone((cg, fg), [256, 128])
    data = dict(image=torch.zeros(1, 3, 8, 8, device=gpu), im_mask=im_mask.to(gpu), pt_mask=pt_mask.to(gpu), pt3d=pt3d.to(gpu),
                pt2d=pt2d.to(gpu), conf_gt=conf_gt.to(gpu), pt2d_proj=pt2d_proj.to(gpu), pt_feat=pt_feat.to(gpu).requires_grad_())
    nerfmatch_amd.set_precision(precision)
    try:
        np.random.seed(21)
        m = model.forward_with_metrics(data, training=True)
        m["loss"].backward()
    finally:
        nerfmatch_amd.set_precision("fp32")
    assert abs(m["coarse_loss"].item() - ref["coarse_loss"].item()) < 1e-5 * ref["coarse_loss"].item()
    assert abs(m["fine_loss"].item() - ref["fine_loss"].item()) < 2e-4 * ref["fine_loss"].item()
    ids = ref["preds"]["match_ids"]
    assert torch.equal(data["match_ids"][1].cpu(), ids[1]) and torch.equal(data["match_ids"][2].cpu(), ids[2])
    tol = 2e-3 if precision == "fp32" else 5e-3
    assert rel(data["pt_feat"].grad, ptf.grad) < tol
    names = dict(model.named_parameters())
    worst = 0.0
    for k, v in p.items():
        if v.grad is None or k not in names or names[k].grad is None:
            continue
        worst = max(worst, rel(names[k].grad, v.grad))
    assert worst < tol, worst


def test_fine_loss_exp_and_feat_l2_vs_oracle(gpu, built_lib):
    """fine_loss = "exp" (LoFTR's window-level loss) and the feat_l2 diagnostic against the training oracle."""
    fx = load_golden("matcher_train")
    cfg = synth.matcher_config("c2f")
    cfg.fine_loss = "exp"
    p = {k: v.clone().requires_grad_() for k, v in synth.matcher_state_dict("c2f", seed=int(fx["weights_seed"])).items()}
    ptf = fx["pt_feat"].clone().requires_grad_()
    np.random.seed(7)
    ref = to.c2f_train_step(p, cfg, fx["cfeat"], fx["ffeat"], ptf, fx["pt3d"], fx["pt2d"], fx["pt2d_proj"], fx["conf_gt"], fx["im_mask"],
                            fx["pt_mask"], fine_loss="exp")
    ref["loss"].backward()
    model, cfeat, ffeat = build_model(fx, gpu)
    model.fine_loss = "exp"
    data = batch(fx, gpu)
    np.random.seed(7)
    m = model.forward_with_metrics(data, training=True)
    assert abs(m["fine_loss"].item() - ref["fine_loss"].item()) < 1e-4 * abs(ref["fine_loss"].item())
    assert abs(m["feat_l2"].item() - ref["feat_l2"].item()) < 1e-5
    m["loss"].backward()
    assert rel(data["pt_feat"].grad, ptf.grad) < 1e-3
    g = dict(model.named_parameters())["fine_sa.layers.0.attention.proj_q.weight"].grad
    assert rel(g, p["fine_sa.layers.0.attention.proj_q.weight"].grad) < 2e-3


def test_trainer_steps_reduce_loss(gpu, built_lib):
    """NeRFMatchMSTrainer (injected optimizer / scheduler factories): a few steps on the fixture batch lower the loss, the
    first epoch is coarse-only, validation runs without a graph."""
    from argparse import Namespace

    from nerfmatch_amd.trainer import NeRFMatchMSTrainer

    fx = load_golden("matcher_train")
    optim = Namespace(lr=0.0004, coarse_only_epochs=1)
    tr = NeRFMatchMSTrainer(Namespace(model=synth.matcher_config("c2f"), optim=optim, gpu_num=1), device=gpu,
                            optimizer_factory=lambda params: torch.optim.Adam(params, lr=optim.lr),
                            scheduler_factory=lambda opt: torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=4, eta_min=1e-8))
    tr.model.load_state_dict(synth.matcher_state_dict("c2f", seed=int(fx["weights_seed"])), strict=False)
    tr.model.backbone = PrecomputedBackbone((fx["cfeat"].to(gpu), fx["ffeat"].to(gpu)), [256, 128])
    losses = []
    for step in range(6):
        np.random.seed(100)
        m = tr.training_step(batch(fx, gpu), step)
        if step == 0:
            assert "fine_loss" not in m  # coarse-only epoch
            tr.on_epoch_end()
        else:
            losses.append(m["loss"].item())
    assert losses[-1] < losses[0]
    assert tr.optimizer.param_groups[0]["lr"] < optim.lr  # cosine schedule stepped
    v = tr.validation_step(batch(fx, gpu))
    assert not v["loss"].requires_grad and torch.isfinite(v["loss"])


def test_coarse_model_loss_and_gradients(gpu, built_lib):
    """NeRFMatcherCoarse.forward_with_metrics (focal loss without clamp) against the oracle: loss, d/d backbone features, d/d T."""
    from nerfmatch_amd.matcher import NeRFMatcherCoarse

    fx = load_golden("matcher_train")
    cfeat = fx["cfeat"].clone().requires_grad_()
    pt_feat = fx["pt_feat"].clone().requires_grad_()
    temp = torch.tensor(10.0, requires_grad=True)
    # no masks: without the clamp a masked ground-truth pair has conf = 0 and the loss is inf (in the reference too)
    out = mo.coarse_forward_match({"temperature": temp}, cfeat, pt_feat, None, None)
    conf = out["conf_matrix"]
    pos, neg = fx["conf_gt"] == 1, fx["conf_gt"] == 0
    loss = (-0.25 * (1 - conf[pos]) ** 2 * conf[pos].log()).mean() + (-0.25 * conf[neg] ** 2 * (1 - conf[neg]).log()).mean()
    loss.backward()
    model = NeRFMatcherCoarse(synth.matcher_config("coarse")).to(gpu)
    cg = fx["cfeat"].to(gpu).requires_grad_()
    model.backbone = PrecomputedBackbone(cg, 256)
    data = batch(fx, gpu)
    data["im_mask"], data["pt_mask"] = torch.ones_like(data["im_mask"]), torch.ones_like(data["pt_mask"])
    m = model.forward_with_metrics(data)
    assert torch.isfinite(loss) and abs(m["loss"].item() - loss.item()) < 1e-5 * abs(loss.item())
    m["loss"].backward()
    assert rel(cg.grad, cfeat.grad) < 1e-4 and rel(data["pt_feat"].grad, pt_feat.grad) < 1e-4
    assert abs(model.temperature.grad.item() - temp.grad.item()) < 1e-4 * abs(temp.grad.item())
    assert torch.equal(data["match_ids"][1].cpu(), out["match_ids"][1]) and torch.equal(data["match_ids"][2].cpu(), out["match_ids"][2])


def test_inference_builds_no_graph(gpu, built_lib):
    """Outside autograd.training() the modules run the fused inference kernels even when parameters require grad."""
    fx = load_golden("matcher_train")
    model, cfeat, ffeat = build_model(fx, gpu)
    data = batch(fx, gpu)
    model.forward(data, mutual=True)
    assert not data["expec_f"].requires_grad and not data["conf_matrix"].requires_grad


def test_forward_match_with_conf_gt(gpu, built_lib):
    """NeRFMatcherMS.forward_match(conf_gt=...) -- the reference's training-time call signature (c2f_trainer.py:302-369; its
    iNeRF match loss calls it directly, nerfmatch_evaluator.py:436-444): GT-padded matches identical to the reference's step;
    the focal loss evaluated by the kernels that hold the similarity matrix (`coarse_loss`) and `expec_f` carry the graph."""
    fx = load_golden("matcher_train")
    model, cfeat, ffeat = build_model(fx, gpu)
    d = batch(fx, gpu)
    np.random.seed(int(fx["np_seed"]))
    with torch.enable_grad():
        preds = model.forward_match(d["image"], d["pt_feat"], d["pt3d"], im_mask=d["im_mask"], pt_mask=d["pt_mask"], conf_gt=d["conf_gt"],
                                    ret_feats=True)
    b, i, j = preds["match_ids"]
    assert torch.equal(i.cpu(), fx["i_ids"]) and torch.equal(j.cpu(), fx["j_ids"]) and torch.equal(b.cpu(), fx["b_ids"])
    assert preds["pred_num"] == int(fx["pred_num"]) and preds["coarse_loss"].requires_grad and preds["expec_f"].requires_grad
    assert abs(float(preds["coarse_loss"]) - float(fx["coarse_loss"])) < 1e-5 * abs(float(fx["coarse_loss"]))
    assert (preds["conf_matrix"].detach().cpu() - fx["conf_matrix"]).abs().max() < 1e-5


def test_transposed_weight_cache_follows_the_tensor(gpu, built_lib):
    """ops.transposed: one copy per (tensor, version) -- an in-place update (an optimiser step) is seen, a frozen weight is transposed once"""
    w = torch.randn(24, 40, device=gpu)
    t0 = ops.transposed(w)
    assert torch.equal(t0, w.t()) and t0.is_contiguous()
    assert ops.transposed(w) is t0
    w.mul_(2.0)  # (bumps the version counter)
    t1 = ops.transposed(w)
    assert t1 is not t0 and torch.equal(t1, w.t())
    ops.invalidate_caches()
    assert torch.equal(ops.transposed(w), w.t())


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("tag,dim,hd,att,act", [("lsa128", 128, 16, "lsa", "gelu"), ("lsa256", 256, 32, "lsa", "gelu"), ("relu128", 128, 16, "full", "relu")])
def test_lsa_and_relu_layers_train_vs_reference(gpu, built_lib, tag, dim, hd, att, act, precision):
    """Round 6 (VERDICT r5 'missing' 4): training through an encoder layer with att_type "lsa" (the learnable log-scale receives its
    gradient: the scores are q.k exp(p), attention.py:60-81) and with act_fn "relu" (FeedForwardNetwork, :136-154) -- output, input gradient
    and parameter gradients of loss = sum(y * g) against the reference's own modules under torch autograd (tests/golden/matcher_layer_grads.npz)."""
    import nerfmatch_amd
    from nerfmatch_amd import autograd as ag
    from nerfmatch_amd.modules.attention import GenericEncoderLayer

    fx = load_golden("matcher_layer_grads")
    rng = np.random.default_rng(int(fx["weights_seed"]))
    sds = {}
    for t_, d_, h_, a_ in (("lsa128", 128, 16, "lsa"), ("lsa256", 256, 32, "lsa"), ("relu128", 128, 16, "full")):  # (the generator's draw order)
        sd = {}
        synth._encoder_layer(sd, rng, "L", d_)
        sd = {k[2:]: v for k, v in sd.items()}
        if a_ == "lsa":
            sd["attention.attend.scale"] = torch.log(torch.tensor(h_**-0.5)) + 0.2
        sds[t_] = sd
    layer = GenericEncoderLayer(model_dim=dim, head_dim=hd, att_type=att, att_mode="self", act_fn=act)
    layer.load_state_dict(sds[tag], strict=True)
    layer.to(gpu)
    x = fx[f"{tag}_x"].to(gpu).requires_grad_(True)
    nerfmatch_amd.set_precision(precision)
    try:
        with torch.enable_grad(), ag.training():
            y = layer(x)
            (y * fx[f"{tag}_gy"].to(gpu)).sum().backward()
    finally:
        nerfmatch_amd.set_precision("fp32")
    rel = lambda a, b: float((a.detach().cpu() - torch.as_tensor(b)).abs().max() / max(1e-6, float(torch.as_tensor(b).abs().max())))
    assert rel(y, fx[f"{tag}_y"]) < 1e-4
    assert rel(x.grad, fx[f"{tag}_dx"]) < 2e-4
    n_checked = 0
    for n, p_ in layer.named_parameters():
        key = f"{tag}_d.{n}"
        if key in fx:
            assert p_.grad is not None, n
            assert rel(p_.grad, fx[key]) < 5e-4, (n, rel(p_.grad, fx[key]))
            n_checked += 1
    assert n_checked >= 4 and (att != "lsa" or f"{tag}_d.attention.attend.scale" in fx)


@pytest.mark.parametrize("rows", [4800, 333, 7200])
def test_fused_tail_backward_vs_separate_kernels_and_fp64(gpu, built_lib, rows):
    """Round 6: the encoder tail's backward for frozen parameters as ONE kernel (nm_encoder_tail_bwd_bf16x3) against
    (a) the separate backward launches it replaces (three GEMMs, gelu_bwd, layernorm_bwd, the adds) on the same inputs and (b) torch autograd
    in float64 -- incl. a ragged last workgroup (333 rows) and more than one round of workgroups (7200)."""
    import nerfmatch_amd
    from nerfmatch_amd.modules.attention import GenericEncoderLayer

    rng = np.random.default_rng(17)
    sd = {}
    synth._encoder_layer(sd, rng, "L", 256)
    layer = GenericEncoderLayer(model_dim=256, head_dim=32, att_type="full", att_mode="self")
    layer.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
    layer.to(gpu)
    for p_ in layer.parameters():
        p_.requires_grad_(False)
    g = torch.Generator().manual_seed(rows)
    att0 = torch.randn(rows, 256, generator=g)
    xh0 = torch.randn(rows, 256, generator=g)
    gy = torch.randn(rows, 256, generator=g).to(gpu)
    at, n2, ff = layer.attention, layer.norm2, layer.feedforward

    def run(fused):
        ops.ENCODER_TAIL_BWD_FUSED = fused
        att, xh = att0.to(gpu).requires_grad_(True), xh0.to(gpu).requires_grad_(True)
        with torch.enable_grad(), ag.training():
            if fused:
                y = ag.encoder_tail_frozen(att, xh, at.proj_out[0].weight, n2, ff.layers[0], ff.layers[2])
            else:
                a = ag.linear(att, at.proj_out[0].weight, residual=xh)
                y = ff(ag.layernorm(a, n2.weight, n2.bias, n2.eps), residual=xh)
            (y * gy).sum().backward()
        return y.detach(), att.grad, xh.grad

    nerfmatch_amd.set_precision("bf16x3")
    try:
        y1, da1, dx1 = run(True)
        y0, da0, dx0 = run(False)
    finally:
        nerfmatch_amd.set_precision("fp32")
        ops.ENCODER_TAIL_BWD_FUSED = True
    # the forward that keeps its intermediates = the inference path's fused tail, bit for bit; against the separate launches to rounding
    ops.LINEAR_PRECISION = "bf16x3"
    try:
        y_inf = ops.encoder_tail(att0.to(gpu), xh0.to(gpu), at.proj_out[0].weight, n2, ff.layers[0], ff.layers[2])
    finally:
        ops.LINEAR_PRECISION = "fp32"
    assert torch.equal(y1, y_inf)
    assert float((y1 - y0).abs().max()) < 2e-5 * float(y0.abs().max())
    # fp64 truth
    att, xh = att0.double().requires_grad_(True), xh0.double().requires_grad_(True)
    W = lambda t: t.detach().cpu().double()
    with torch.enable_grad():
        a = xh + att @ W(at.proj_out[0].weight).T
        an = F.layer_norm(a, (256,), W(n2.weight), W(n2.bias), n2.eps)
        y = xh + F.gelu(an @ W(ff.layers[0].weight).T + W(ff.layers[0].bias)) @ W(ff.layers[2].weight).T + W(ff.layers[2].bias)
        (y * gy.cpu().double()).sum().backward()
    for name, got, sep, ref in (("d_att", da1, da0, att.grad), ("d_xh", dx1, dx0, xh.grad)):
        scale = float(ref.abs().max())
        e_sep, e_ref, e_sep_ref = float((got - sep).abs().max()), float((got.cpu().double() - ref).abs().max()), float((sep.cpu().double() - ref).abs().max())
        print(f"fused tail backward {rows} rows {name}: |fused - separate| {e_sep:.2e}, |fused - fp64| {e_ref:.2e}, |separate - fp64| {e_sep_ref:.2e} (scale {scale:.1f})")
        assert torch.isfinite(got).all()
        assert e_sep < 2e-5 * scale and e_ref < 2e-5 * scale, name

"""iNeRF pose refinement (reference: NeRFMatchEvaluator.inerf_refinement, nerfmatch/nerfmatch_evaluator.py:288-500).

Per Adam step the reference renders the query view from the current pose with autograd through the FINE network and
minimises the MSE to the (sub-sampled) query image.  Here every arithmetic stage is a HIP kernel with a hand-written
backward:
  rays              nm_raygen (values) -- only o and viewdir carry gradient; d(o, viewdir)/d(pose) is a 16-parameter torch
                    autograd expression over the sub-sampled pixel grid (plumbing-sized)
  sampling, coarse  nm_sample_coarse, fused nm_nerf_fwd (weights only, no grad), nm_resample -- as in render_rays
  fine pass         nm_inerf_encode -> nm_linear(_bf16x3) x 12 -> nm_inerf_composite, and the mirrored backward
                    nm_inerf_composite_bwd -> nm_linear(_bf16x3) with transposed weights -> nm_inerf_encode_bwd
  optimiser         torch.optim.Adam on the 4x4 pose (as the reference)
  matching term     (`use_match_loss`, :420-441) nm_inerf_composite_ex (weights) -> nm_inerf_ray_sums (pt_feat, points) ->
                    matcher.match_loss (the training kernels of the matcher under torch.autograd.Function, parameters frozen:
                    only d loss / d pt_feat and d loss / d pt3d are computed) -> nm_inerf_ray_sums_bwd -> the gradients of
                    the weights and of the tapped layer's activations join the photometric backward
                    (nm_inerf_composite_bwd_ex, FineField.backward(g_h=...)).  Under the split arithmetic the fine pass is the
                    fused kernel pair here too (round 5): the forward kernel writes the tapped activations, the backward kernel
                    forms w_n . d loss / d pt_feat[ray] itself and adds it at the tapped layer (nm_nerf_points_*_tap_bf16x3)
Only the first S/2 + 1 fine samples of a ray are evaluated: the randomized resampler leaves the later intervals with zero
width, i.e. zero weight and zero gradient (see NM_NERF_ZERO_TAIL in include/nerfmatch_amd.h).
"""
import ctypes as C
import math

import torch

from . import ops
from ._lib import check, dptr, lib, steady_gc, stream

NUM_PTS = 128  # hard-coded by the reference (:354, :360)
XI, XD = 96, 48
F32_EPS = float(torch.finfo(torch.float32).eps)


def _new(*shape, dev):
    return torch.empty(*shape, device=dev, dtype=torch.float32)


class FineField:
    """nerf_fine as padded GEMM operands: forward weights and their transposes (for the backward GEMMs), cached per
    renderer.  Layers with a concatenated input are split into two GEMMs joined through the `pre` addend of nm_linear_ex:
    layer 5 = relu(xi . W5x^T + h4 . W5h^T + b), views = relu(feature . Wvf^T + xd . Wvd^T + b); xi has 96 columns
    (90 IPE + padding), xd 48 (27 dir PE + 16 appearance + padding); the density / rgb heads are padded to 8 outputs.
    The ReLU derivatives of the backward pass are the `gate` of the same epilogue (no elementwise passes)."""

    def __init__(self, nerf_fine, dev):
        sd = {k: v.detach().to(dev, torch.float32) for k, v in nerf_fine.state_dict().items()}
        z = lambda *s: torch.zeros(*s, device=dev)

        def padded(w, cols):
            out = z(w.shape[0], cols)
            out[:, : w.shape[1]] = w
            return out.contiguous()

        W = [sd[f"pts_linears.{l}.weight"].contiguous() for l in range(8)]
        self.b = [sd[f"pts_linears.{l}.bias"].contiguous() for l in range(8)]
        self.W5x, W[5] = padded(W[5][:, :90], XI), W[5][:, 90:].contiguous()
        W[0] = padded(W[0], XI)
        self.W = W
        self.Wa, self.ba = z(8, 256), z(8)
        self.Wa[:1] = sd["alpha_linear.weight"]
        self.ba[:1] = sd["alpha_linear.bias"]
        self.Wf, self.bf = sd["feature_linear.weight"].contiguous(), sd["feature_linear.bias"].contiguous()
        wv = sd["views_linears.0.weight"]  # (128, 256 + 27 [+ 16])
        self.Wvf, self.Wvd, self.bv = wv[:, :256].contiguous(), padded(wv[:, 256:], XD), sd["views_linears.0.bias"].contiguous()
        self.Wr, self.br = z(8, 128), z(8)
        self.Wr[:3] = sd["rgb_linear.weight"]
        self.br[:3] = sd["rgb_linear.bias"]
        t = lambda w: w.t().contiguous()
        self.WT = [t(w) for w in W]
        self.W5xT, self.WaT, self.WfT, self.WvfT, self.WvdT, self.WrT = t(self.W5x), t(self.Wa), t(self.Wf), t(self.Wvf), t(self.Wvd), t(self.Wr)

    def forward(self, xi, xd):
        """xi (n,96), xd (n,48) -> rgb logits (n,8), raw sigma (n,8) (columns 0..2 / 0), saved activations."""
        lin = ops.linear
        h = [lin(xi, self.W[0], self.b[0], act=1)]
        for l in range(1, 8):
            h.append(lin(h[-1], self.W[l], self.b[l], act=1, pre=lin(xi, self.W5x) if l == 5 else None))
        sig = lin(h[7], self.Wa, self.ba)
        feat = lin(h[7], self.Wf, self.bf)
        hv = lin(feat, self.Wvf, self.bv, act=1, pre=lin(xd, self.Wvd))
        logit = lin(hv, self.Wr, self.br)
        return logit, sig, (h, hv)

    def backward(self, g_logit, g_sig, saved, g_h=None):
        """d loss / d logits, d loss / d sigma -> d loss / d xi (n,96), d loss / d xd (n,48).  g_h = (layer, d loss / d h[layer])
        adds a gradient arriving at one layer's (post-ReLU) activations from outside the network: the tapped features."""
        lin = ops.linear
        h, hv = saved
        tap, g_tap = g_h if g_h is not None else (-1, None)
        g_hv = lin(g_logit, self.WrT, gate=hv)
        g_xd = lin(g_hv, self.WvdT)
        g_sig_h = lin(g_sig, self.WaT, residual=g_tap if tap == 7 else None)
        g = lin(lin(g_hv, self.WvfT), self.WfT, residual=g_sig_h, gate=h[7])
        g_xi_skip = None
        for l in range(7, 0, -1):
            if l == 5:
                g_xi_skip = lin(g, self.W5xT)
            g = lin(g, self.WT[l], residual=g_tap if tap == l - 1 else None, gate=h[l - 1])
        g_xi = lin(g, self.WT[0], residual=g_xi_skip)
        return g_xi, g_xd


class FusedField:
    """nerf_fine's pointwise forward / backward as TWO fused kernels (round 4; csrc/nerf_fwd_bf16.hip, nm_nerf_points_fwd_bf16x3 /
    nm_nerf_points_bwd_bf16x3) instead of 12 + 14 GEMM launches: the K-loop machinery of the render kernel, activations in
    registers, and between the passes only one BIT per ReLU activation (9 x 16 bytes per sample lane).  dX only -- the pose is the
    only parameter of the refinement.  With the matching term (round 5) the forward kernel also writes the tapped layer's activations
    and the backward kernel takes the term's gradient in at that layer (nm_nerf_points_*_tap_bf16x3)."""

    def __init__(self, nerf_fine, dev):
        self.blob = nerf_fine.packed(dev, "bf16x3")
        self.blob_bwd = nerf_fine.packed(dev, "bwd_bf16x3")

    def forward(self, xi, xd):
        n, dev = xi.shape[0], xi.device
        out4 = _new(n, 4, dev=dev)
        gates = torch.empty(lib().nm_nerf_points_gate_bytes(n), dtype=torch.uint8, device=dev)
        check(lib().nm_nerf_points_fwd_bf16x3(dptr(self.blob, torch.uint8), dptr(xi), dptr(xd), n, dptr(out4), dptr(gates, torch.uint8), stream()),
              "nm_nerf_points_fwd_bf16x3")
        return out4, gates

    def forward_rays(self, rays, z, S_act, app_row, tap=-1):
        """The same forward pass with the encoding done inside the kernel (no xi / xd arrays, no nm_inerf_encode launch).
        tap >= 0: also returns the post-ReLU activations (n, 256) of pts layer `tap` (the rendered features of the matching term)."""
        R, S, dev = z.shape[0], z.shape[1] - 1, rays.device
        n = R * S_act
        out4 = _new(n, 4, dev=dev)
        gates = torch.empty(lib().nm_nerf_points_gate_bytes(n), dtype=torch.uint8, device=dev)
        feats = _new(n, 256, dev=dev) if tap >= 0 else None
        check(lib().nm_nerf_points_fwd_rays_tap_bf16x3(dptr(self.blob, torch.uint8), dptr(rays), dptr(z), R, S, int(S_act), dptr(app_row), int(tap),
                                                       dptr(out4), dptr(gates, torch.uint8), dptr(feats), stream()), "nm_nerf_points_fwd_rays_tap_bf16x3")
        return (out4, gates, feats) if tap >= 0 else (out4, gates)

    def backward(self, g4, gates, tap=None):
        """tap = (layer, compositing weights (R, S_act), d loss / d pt_feat (R, 256)): the matching term's gradient enters at that layer's
        activations inside the kernel (w_n . g_pt_feat[ray]: no (n, 256) gradient array)."""
        n, dev = g4.shape[0], g4.device
        g_xi0, g_xi5, g_xd = _new(n, XI, dev=dev), _new(n, XI, dev=dev), _new(n, XD, dev=dev)
        if tap is None:
            check(lib().nm_nerf_points_bwd_bf16x3(dptr(self.blob_bwd, torch.uint8), dptr(g4), dptr(gates, torch.uint8), n, dptr(g_xi0), dptr(g_xi5),
                                                  dptr(g_xd), stream()), "nm_nerf_points_bwd_bf16x3")
        else:
            layer, w, g_pf = tap
            w, g_pf = w.contiguous(), g_pf.contiguous()
            R, S_act = w.shape
            assert R * S_act == n and g_pf.shape == (R, 256)
            check(lib().nm_nerf_points_bwd_tap_bf16x3(dptr(self.blob_bwd, torch.uint8), dptr(g4), dptr(gates, torch.uint8), R, S_act, int(layer),
                                                      dptr(w), dptr(g_pf), dptr(g_xi0), dptr(g_xi5), dptr(g_xd), stream()),
                  "nm_nerf_points_bwd_tap_bf16x3")
        return (g_xi0, g_xi5), g_xd  # the two contributions to d loss / d xi; nm_inerf_encode_bwd2 adds them while reading


def _encode(rays, z, S_act, app_row):
    R, S = z.shape[0], z.shape[1] - 1
    xi, xd = _new(R * S_act, XI, dev=rays.device), _new(R * S_act, XD, dev=rays.device)
    check(lib().nm_inerf_encode(dptr(rays), dptr(z), R, S, S_act, dptr(app_row), dptr(xi), dptr(xd), stream()), "nm_inerf_encode")
    return xi, xd


def _encode_bwd(rays, z, S_act, g_xi, g_xd):
    R, S = z.shape[0], z.shape[1] - 1
    g_o, g_v = _new(R, 3, dev=rays.device), _new(R, 3, dev=rays.device)
    g_xd = g_xd.contiguous()  # (named: a temporary made inside the argument list is freed before the launch and may be re-used by the next one)
    if isinstance(g_xi, tuple):
        check(lib().nm_inerf_encode_bwd2(dptr(rays), dptr(z), R, S, S_act, dptr(g_xi[0]), dptr(g_xi[1]), dptr(g_xd), dptr(g_o), dptr(g_v),
                                         stream()), "nm_inerf_encode_bwd2")
    else:
        g_xi = g_xi.contiguous()
        check(lib().nm_inerf_encode_bwd(dptr(rays), dptr(z), R, S, S_act, dptr(g_xi), dptr(g_xd), dptr(g_o), dptr(g_v),
                                        stream()), "nm_inerf_encode_bwd")
    return g_o, g_v


def _pose_grad(K, p_host, H, W, ds, g_o, g_v, g_d):
    """d loss / d pose (4,4) on the device from the per-ray gradients (nm_inerf_pose_grad: one launch instead of an autograd graph of
    a dozen tiny kernels).  o = pose[:3,3]; view = normalise(pose[:3,:3] . K^-1 (x, y, 1)) on the sub-sampled pixel grid."""
    kinv = torch.linalg.inv(torch.as_tensor(K, dtype=torch.float32).reshape(3, 3).cpu()).contiguous()
    g_pose = _new(4, 4, dev=g_o.device)
    check(lib().nm_inerf_pose_grad(ops.hptr(kinv), ops.hptr(p_host.contiguous()), int(H), int(W), int(ds), dptr(g_o), dptr(g_v), dptr(g_d), g_o.shape[0],
                                   dptr(g_pose), stream()), "nm_inerf_pose_grad")
    return g_pose


def _composite(logit, sig, z, rays, S_act, want_weights=False):
    R, S = z.shape[0], z.shape[1] - 1
    rgb = _new(R, 3, dev=rays.device)
    w = _new(R, S_act, dev=rays.device) if want_weights else None
    check(lib().nm_inerf_composite_ex(dptr(logit), dptr(sig), logit.shape[1], dptr(z), dptr(rays), R, S, S_act, dptr(rgb), dptr(w), stream()),
          "nm_inerf_composite_ex")
    return (rgb, w) if want_weights else rgb


def _composite_bwd(logit, sig, z, rays, S_act, G, g_w=None):
    R, S = z.shape[0], z.shape[1] - 1
    g_logit, g_sig, g_d = torch.empty_like(logit), torch.empty_like(sig), _new(R, 3, dev=rays.device)
    G = G.contiguous()
    check(lib().nm_inerf_composite_bwd_ex(dptr(logit), dptr(sig), logit.shape[1], dptr(z), dptr(rays), dptr(G), dptr(g_w), R, S,
                                          S_act, dptr(g_logit), dptr(g_sig), dptr(g_d), stream()), "nm_inerf_composite_bwd_ex")
    return g_logit, g_sig, g_d


def _composite4(out4, z, rays, S_act, want_weights=False):
    """_composite on the fused field's own output rows (n, 4) = rgb logits | raw sigma: one wavefront per ray (nm_inerf_composite4)."""
    R, S = z.shape[0], z.shape[1] - 1
    rgb = _new(R, 3, dev=rays.device)
    w = _new(R, S_act, dev=rays.device) if want_weights else None
    check(lib().nm_inerf_composite4(dptr(out4), dptr(z), dptr(rays), R, S, S_act, dptr(rgb), dptr(w), stream()), "nm_inerf_composite4")
    return (rgb, w) if want_weights else rgb


def _composite4_bwd(out4, z, rays, S_act, G, g_w=None):
    """-> g4 (n, 4) = d loss / d (logits, sigma), g_d (R, 3)."""
    R, S = z.shape[0], z.shape[1] - 1
    g4, g_d = torch.empty_like(out4), _new(R, 3, dev=rays.device)
    G = G.contiguous()
    check(lib().nm_inerf_composite4_bwd(dptr(out4), dptr(z), dptr(rays), dptr(G), dptr(g_w), R, S, S_act, dptr(g4), dptr(g_d), stream()),
          "nm_inerf_composite4_bwd")
    return g4, g_d


def _ray_sums(w, feats, rays, z, S_act):
    """pt_feat (R,C) = sum_s w_s feats_s, pts (R,3) = sum_s w_s mean_s (normalised scene)."""
    R, S, Cf = z.shape[0], z.shape[1] - 1, feats.shape[1]
    pt_feat, pts = _new(R, Cf, dev=rays.device), _new(R, 3, dev=rays.device)
    check(lib().nm_inerf_ray_sums(dptr(w), dptr(feats), Cf, dptr(rays), dptr(z), R, S, S_act, dptr(pt_feat), dptr(pts), stream()),
          "nm_inerf_ray_sums")
    return pt_feat, pts


def _ray_sums_bwd(w, feats, rays, z, S_act, g_pt_feat, g_pts, want_g_feats=True):
    R, S, Cf = z.shape[0], z.shape[1] - 1, feats.shape[1]
    g_feats, g_w = torch.empty_like(feats) if want_g_feats else None, torch.empty_like(w)
    g_pt_feat, g_pts = g_pt_feat.contiguous(), g_pts.contiguous()
    check(lib().nm_inerf_ray_sums_bwd(dptr(w), dptr(feats), Cf, dptr(rays), dptr(z), dptr(g_pt_feat), dptr(g_pts), R, S,
                                      S_act, dptr(g_feats), dptr(g_w), stream()), "nm_inerf_ray_sums_bwd")
    return g_feats, g_w


def _match_term(match, pt_feat, pt3d):
    """Matching loss of the rendered view against the query image and its gradients w.r.t. the rendered features (R,C) and
    (world) points (R,3) -- reference :429-441: forward_match(mutual=True), focal loss of conf_matrix against the identity
    (image token i <-> ray i).  The matcher's parameters are frozen for the call, so its backward computes input gradients
    only."""
    from . import autograd as ag

    model = match["model"]
    R = pt_feat.shape[0]
    pf = pt_feat.detach()[None].requires_grad_(True)
    p3 = pt3d.detach()[None].requires_grad_(True)
    conf_gt = match.get("_conf_gt")  # (the identity, as uint8 -- what the loss kernels read: built once per refinement, not 23 MB filled and converted per step)
    if conf_gt is None or conf_gt.shape[1] != R or conf_gt.device != pt_feat.device:
        conf_gt = match["_conf_gt"] = torch.eye(R, dtype=torch.uint8, device=pt_feat.device)[None]
    frozen = [p for p in model.parameters() if p.requires_grad]
    for p in frozen:
        p.requires_grad_(False)
    try:
        if "_im_tokens" not in match:  # the image side does not depend on the pose: once per refinement, on the fused inference kernels
            match["_im_tokens"] = model.image_tokens(match["image"]) if hasattr(model, "image_tokens") else None
        with torch.enable_grad(), ag.training():
            loss = model.match_loss(match["image"], pf, p3, match.get("im_mask"), match.get("pt_mask"), conf_gt, im_tokens=match["_im_tokens"])
            g_pf, g_p3 = torch.autograd.grad(loss, [pf, p3])
    finally:
        for p in frozen:
            p.requires_grad_(True)
    return loss.detach(), g_pf[0], g_p3[0]


FUSED_FINE = True  # the fine pass on the two fused pointwise kernels when the split arithmetic is selected (ops.LINEAR_PRECISION == "bf16x3");
                   # False: always the GEMM chain (A/B runs, and the arithmetic the fp32 setting uses)


def fused_field(renderer, dev):
    ff = renderer.__dict__.get("_inerf_fused")
    key = tuple((p.data_ptr(), p._version) for p in renderer.nerf_fine.parameters())
    if ff is None or ff[0] != key or ff[1].blob.device != dev:
        ff = renderer.__dict__["_inerf_fused"] = (key, FusedField(renderer.nerf_fine, dev))
    return ff[1]


def fine_field(renderer, dev):
    ff = renderer.__dict__.get("_inerf_field")
    key = tuple((p.data_ptr(), p._version) for p in renderer.nerf_fine.parameters())
    if ff is None or ff[0] != key or ff[1].Wf.device != dev:
        ff = renderer.__dict__["_inerf_field"] = (key, FineField(renderer.nerf_fine, dev))
    return ff[1]


def step_gradient(renderer, pose, K, H, W, img_ds, t_rand, jitter, ds=8, skip_zero_tail=True, match=None):
    """One refinement step's forward + backward: returns (loss, d loss / d pose (4,4), context for the caller).
    `pose`: normalised-scene c2w (4,4), on the device or on the host; img_ds (R,3) on the device.  t_rand / jitter: the samplers' random tensors (R,129).
    match: None, or dict(model=NeRFMatcherMS, image (1,3,H,W), unnorm (4,4) on the device, im_mask, pt_mask) to add the
    matching loss of the rendered view (`use_match_loss`)."""
    dev = img_ds.device  # (the pose may live on the host: refine_iter keeps it and its Adam state there)
    S = NUM_PTS
    app_row = None
    if renderer.embedding_a is not None:
        app_row = renderer.embedding_a.weight[1].detach().to(dev, torch.float32).contiguous()  # ray_id 1 (:391-393)
    p_host = pose.detach().to("cpu", torch.float32)  # (free when the caller keeps the pose on the host, as refine_iter does)
    rays, _ = ops.raygen(K, p_host, H, W, dev, ds=ds)
    R = rays.shape[0]
    # sampling and coarse weights: no gradient (the reference hands the samplers rays.detach(), coarse net under no_grad)
    t_c = ops.sample_coarse(rays, t_rand.to(dev, torch.float32).contiguous(), S)
    # (only the weights of that pass are read: single-product fp16 kernel when the renderer allows it)
    cprec = "fp16x1" if renderer.precision in ("bf16x3", "fp16x3") and getattr(renderer, "coarse_precision", "same") == "fp16x1" else renderer.precision
    w_c = renderer.nerf_coarse.fused(cprec, rays, t_c, app_row, tap_layer=-1, white_bg=True,
                       need_rgb=False, need_feat=False)["weights"]
    t_f = ops.resample(t_c, w_c, jitter.to(dev, torch.float32).contiguous(), 0.01, True)
    S_act = S // 2 + 1 if skip_zero_tail else S
    # fine pass, forward
    fused = FUSED_FINE and ops.LINEAR_PRECISION == "bf16x3"
    tap = -1
    if match is not None:
        tap = renderer.nerf_fine.stop_layer if renderer.nerf_fine.stop_layer >= 0 else 7
    if fused:
        # two fused kernels (forward here, backward below) instead of 12 + 14 GEMM launches; between them: one bit per ReLU; the
        # forward kernel encodes its samples itself.  With the matching term (round 5) the forward kernel also writes the tapped layer's
        # activations and the backward kernel takes the term's gradient in at that layer
        field = fused_field(renderer, dev)
        if match is not None:
            out4, gates, feats = field.forward_rays(rays, t_f, S_act, app_row, tap)
        else:
            out4, gates = field.forward_rays(rays, t_f, S_act, app_row)
        rgb_map, weights = _composite4(out4, t_f, rays, S_act, want_weights=True)  # (n, 4) = rgb logits | raw sigma, composited as they lie
    else:
        field = fine_field(renderer, dev)
        xi, xd = _encode(rays, t_f, S_act, app_row)
        logit, sig, saved = field.forward(xi, xd)
        rgb_map, weights = _composite(logit, sig, t_f, rays, S_act, want_weights=True)
    diff = rgb_map - img_ds
    loss = torch.mean(diff * diff)
    g_w = g_h = None
    if match is not None:
        if not fused:
            feats = saved[0][tap]
        pt_feat, pts = _ray_sums(weights, feats, rays, t_f, S_act)
        un = match["unnorm"]
        loss_m, g_pf, g_p3 = _match_term(match, pt_feat, pts @ un[:3, :3].T + un[:3, 3])
        g_feats, g_w = _ray_sums_bwd(weights, feats, rays, t_f, S_act, g_pf, g_p3 @ un[:3, :3], want_g_feats=not fused)
        g_h = (tap, weights, g_pf) if fused else (tap, g_feats)
        loss = loss + loss_m
    # backward
    G = diff * (2.0 / diff.numel())
    if fused:
        g4, g_d = _composite4_bwd(out4, t_f, rays, S_act, G, g_w)  # (n, 4) = d loss / d (logits, sigma)
        g_xi, g_xd = field.backward(g4, gates, g_h)
    else:
        g_logit, g_sig, g_d = _composite_bwd(logit, sig, t_f, rays, S_act, G, g_w)
        g_xi, g_xd = field.backward(g_logit, g_sig, saved, g_h)
    g_o, g_v = _encode_bwd(rays, t_f, S_act, g_xi, g_xd)
    # rays -> pose: o = pose[:3,3]; viewdir = normalise(pose[:3,:3] . K^-1 [x, y, 1]) on the sub-sampled pixel grid
    # (rays[:, 3:6] and rays[:, 8:11] are the same tensor in gen_rays, :281-283: the direction takes g_v and g_d)
    g_pose = _pose_grad(K, p_host, H, W, ds, g_o, g_v, g_d)
    return loss, g_pose, dict(rays=rays, t_fine=t_f, rgb_map=rgb_map, app_row=app_row)


def refine_iter(renderer, K, H, W, image_hw3, pose0, num_optim=5, lrate=0.001, lrdecay=False, ds=8, t_rands=None, jitters=None,
                skip_zero_tail=True, match=None):
    """Generator over the Adam steps: yields (j, pose after step j, loss of step j, ctx of step j).  ctx holds the rays and
    fine fence posts the step rendered with, i.e. BEFORE its update ("1 iteration less than the pose", :469)."""
    dev = pose0.device
    img = torch.as_tensor(image_hw3, dtype=torch.float32).to(dev)
    img_ds = img[ds // 2 :: ds, ds // 2 :: ds].contiguous().view(-1, 3)
    R = img_ds.shape[0]
    # The 16 pose parameters and their Adam state live on the HOST (round 4): the ray generator takes the pose from the host anyway, and a
    # step then has exactly one synchronisation -- the read-back of (loss, d loss / d pose) -- instead of a pose download at its start plus
    # a tail of a dozen tiny optimiser kernels.  Same arithmetic as torch.optim.Adam on the device (fp32).
    pose = pose0.detach().to("cpu", torch.float32).clone().requires_grad_(True)
    opt = torch.optim.Adam(params=[pose], lr=lrate)
    for j in range(num_optim):
        if lrdecay:
            for grp in opt.param_groups:
                grp["lr"] = lrate * (1 + math.cos(math.pi * j / num_optim)) / 2
        t_rand = t_rands[j] if t_rands is not None else torch.rand(R, NUM_PTS + 1, device=dev)
        jit = jitters[j] if jitters is not None else torch.rand(R, NUM_PTS + 1, device=dev) * (1.0 / (NUM_PTS + 1) - F32_EPS)
        loss, g_pose, ctx = step_gradient(renderer, pose.detach(), K, H, W, img_ds, t_rand, jit, ds, skip_zero_tail, match)
        both = torch.cat([g_pose.reshape(-1), loss.reshape(1).to(g_pose.dtype)]).cpu()  # the step's one synchronisation
        pose.grad = both[:16].reshape(4, 4)
        opt.step()
        opt.zero_grad()
        yield j, pose.detach().clone().to(dev), float(both[16]), ctx


def refine(renderer, K, H, W, image_hw3, pose0, num_optim=5, lrate=0.001, lrdecay=False, ds=8, t_rands=None, jitters=None,
           skip_zero_tail=True, match=None):
    """`num_optim` Adam steps on the normalised-scene pose.  Returns (poses after every step, losses, ctx of the last step).
    t_rands / jitters: optional explicit random tensors (one (R,129) pair per step)."""
    poses, losses, ctx = [], [], None
    with steady_gc():
        for _, p, l, ctx in refine_iter(renderer, K, H, W, image_hw3, pose0, num_optim, lrate, lrdecay, ds, t_rands, jitters, skip_zero_tail,
                                        match):
            poses.append(p)
            losses.append(l)
    return poses, losses, ctx


def rendered_points(renderer, ctx):
    """pt3d (normalised) and pt_feat of the view a step rendered (fused forward kernel on the step's own fence posts):
    what the reference re-matches with when `eval_pose` is off (:470-479)."""
    dev = ctx["rays"].device
    o = renderer.nerf_fine.fused(renderer.precision, ctx["rays"], ctx["t_fine"], ctx["app_row"],
                     tap_layer=renderer.nerf_fine.stop_layer, white_bg=True, need_rgb=False, zero_tail=True)
    return o["pts"], o["feat"]

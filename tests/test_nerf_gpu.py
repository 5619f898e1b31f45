"""GPU parity of the NeRF render half: HIP kernels (through the C ABI) vs golden vectors / the oracle.

Tolerances: sampling rows are elementwise fp32 -> 1e-6; anything through the 8-layer MLP -> 1e-4
(BASELINE.json north_star: "rendered features and match scores within 1e-4 fp32"), taken relative to the tensor's scale
max(1, max|reference|): absolute 1e-4 for the unit-scale fixtures, and the same 1e-4 of full scale for the trained-like
"surface" fixture (round 3: activations up to 18, densities +-1e4 whose fp32 ulp alone is 1e-3)."""
import numpy as np
import pytest
import torch

from conftest import abs_bound, load_golden
from nerfmatch_amd import synth, ops, _lib
from nerfmatch_amd.nerf.renderer import NerfRenderer
from oracle import nerf_oracle as no

pytestmark = pytest.mark.gpu
CASES = ["r32_s32", "r128_s64_app", "r32_s32_last", "surface_r512_s128", "surface_r256_s64_app"]
TOL = 1e-4


def maxdiff(a, b):
    return (a.detach().cpu().float() - torch.as_tensor(b).float()).abs().max().item()


def relerr(a, b, what=None):
    """max |a - b| in units of the reference tensor's scale max(1, max|b|).  With `what`: the ABSOLUTE maximum is also held to its
    recorded bound (conftest.abs_bound: 1.5 x the value measured when tests/golden/abs_bounds.json was made) -- an of-scale bar alone
    could hide a drift on the trained-like fixtures, whose scales are 20 (activations) to 1e4 (densities)."""
    b = torch.as_tensor(b).float()
    d = maxdiff(a, b)
    if what is not None:
        abs_bound(what, d)
    return d / max(1.0, b.abs().max().item() if b.numel() else 1.0)


SPLIT = ["fp16x3", "bf16x3"]  # the two operand splits of the 16-bit matrix-core kernel (fp16x3 = the default parity arithmetic)


def tol_for(precision, case):
    """1e-4 everywhere, with ONE stated exception (round 3 finding): the bf16 split (16 mantissa bits) on the trained-like
    "surface" fixture -- densities of +-1e4 come out ~0.25 off, compositing weights up to 7e-4, rendered features 1.4e-4 of
    scale.  It is no longer the default; its bound there is 2e-3 and the measured values are printed."""
    return 2e-3 if (precision == "bf16x3" and case.startswith("surface")) else TOL


def make_renderer(fx, gpu, S=None):
    app = bool(fx["app"])
    cfg = synth.nerf_config("cambridge" if app else "7scenes", num_pts=S or fx["S"], img_wh=(fx["W"], fx["H"]))
    ren = NerfRenderer(cfg, num_frames=5 if app else None, training=False, stop_layer=fx["stop_layer"])
    style = str(fx["style"]) if "style" in fx and str(fx["style"]) else None
    sd = synth.nerf_state_dict(seed=int(fx["weights_seed"]), app_vocab=5 if app else 0, density_bias=float(fx["density_shift"]) if (style and "density_shift" in fx) else (0.0 if style else 3.0), style=style)
    ren.load_state_dict(sd, strict=True)
    ren.precision = "fp32"  # the tests of the split kernels switch it explicitly (the class default is "fp16x3")
    return ren.to(gpu).eval(), sd


def test_library_loaded(gpu, built_lib):
    assert _lib.lib().nm_abi_version() == 1


def test_mfma_probe_runs(gpu, built_lib):
    """bench.py's measurement aid: every thread adds up 24 * rounds identical products -- the result is known in closed form."""
    import ctypes as C
    wgs, rounds = 8, 5
    sink = torch.empty(wgs * 256, device=gpu)
    _lib.check(_lib.lib().nm_probe_mfma_f16(C.c_void_p(sink.data_ptr()), wgs, rounds, ops.stream()), "nm_probe_mfma_f16")
    torch.cuda.synchronize()
    assert torch.isfinite(sink).all() and (sink > 0).any()
    assert torch.equal(sink[:256], sink[256:512])  # every workgroup computes the same numbers


@pytest.mark.parametrize("case", CASES)
def test_raygen(gpu, built_lib, case):
    fx = load_golden(f"nerf_{case}")
    rays, flag = ops.raygen(fx["K"], fx["c2w_norm"], fx["H"], fx["W"], gpu)
    assert int(flag.item()) == 0
    assert maxdiff(rays, fx["rays"]) < 2e-6


def test_raygen_far_fallback(gpu, built_lib):
    fx = load_golden("nerf_far_fallback")
    rays, flag = ops.raygen(fx["K"], fx["c2w"], fx["H"], fx["W"], gpu)
    assert int(flag.item()) == 1
    assert torch.all(rays[:, 7] == 1.0)
    assert maxdiff(rays, fx["rays"]) < 2e-6


@pytest.mark.parametrize("case", CASES)
def test_sampling(gpu, built_lib, case):
    fx = load_golden(f"nerf_{case}")
    rays = fx["rays"].to(gpu)
    t = ops.sample_coarse(rays, fx["t_rand"].to(gpu), fx["S"])
    assert maxdiff(t, fx["t_coarse"]) < 1e-6
    t2 = ops.resample(fx["t_coarse"].to(gpu), fx["comp_weights"].to(gpu), fx["jitter"].to(gpu))
    # 2e-6 everywhere since round 4 (rounds 1-3 needed 5e-6 on the trained-like fixtures): the pdf normaliser is now summed in fp64
    # and rounded once, which puts the HIP fence posts 2x CLOSER to the fp64 evaluation than the reference's own fp32 run
    # (tests/test_resample_truth_gpu.py: 7e-7 against 1.8e-6 at worst); what is left against the golden vectors is the reference's
    # own distance from the exact value (inverse cdf over padding-only bins: cdf steps of ~1e-4 amplify one ulp of the normaliser)
    assert maxdiff(t2, fx["t_fine"]) < 2e-6
    assert torch.all(t2[:, 1:] >= t2[:, :-1])
    # deterministic branch against the oracle
    t3 = ops.resample(fx["t_coarse"].to(gpu), fx["comp_weights"].to(gpu), None, randomized=False)
    ref = no.resample(fx["t_coarse"], fx["comp_weights"], None, randomized=False)
    assert maxdiff(t3, ref) < 2e-6


@pytest.mark.parametrize("case", CASES)
def test_fused_pass_vs_golden(gpu, built_lib, case):
    """One fused pass on the golden coarse fence posts: per-sample MLP outputs, compositing weights and sums."""
    fx = load_golden(f"nerf_{case}")
    ren, sd = make_renderer(fx, gpu)
    rays, t = fx["rays"].to(gpu), fx["t_coarse"].to(gpu)
    app = fx["app_row"].to(gpu) if fx["app"] else None
    n = fx["sub_rays"] * fx["S"]
    for net, tap, kraw, kfeat in ((ren.nerf_coarse, -1, "mlp_raw_coarse", "mlp_feat_coarse"),
                                  (ren.nerf_fine, fx["stop_layer"], "mlp_raw_fine", "mlp_feat_fine")):
        o = ops.nerf_fwd(net.packed(gpu), rays, t, app, tap_layer=tap, white_bg=bool(fx["white_bg"]), want_raw=True, want_sample_feat=True)
        assert relerr(o["raw"].reshape(-1, 4)[:n, :3], fx[kraw][:, :3], f"{kraw}.rgb") < TOL  # colours
        assert relerr(o["raw"].reshape(-1, 4)[:n, 3], fx[kraw][:, 3], f"{kraw}.density") < TOL  # raw density (scale: its own maximum)
        assert relerr(o["sample_feat"].reshape(-1, 256)[:n], fx[kfeat], kfeat) < TOL
    # the coarse network's compositing is pinned by the reference's volume_render_radiance_field outputs
    o = ops.nerf_fwd(ren.nerf_coarse.packed(gpu), rays, t, app, tap_layer=-1, white_bg=bool(fx["white_bg"]))
    assert maxdiff(o["weights"], fx["comp_weights"]) < TOL
    assert maxdiff(o["rgb"], fx["comp_rgb"]) < TOL
    assert maxdiff(o["depth"], fx["comp_depth"]) < TOL
    assert maxdiff(o["acc"], fx["comp_acc"]) < TOL


@pytest.mark.parametrize("case", CASES)
def test_render_rays_and_novel_view(gpu, built_lib, case):
    fx = load_golden(f"nerf_{case}")
    ren, sd = make_renderer(fx, gpu)
    ren.ret_pfeat = True
    preds = ren.predict(fx["rays"].to(gpu), fx["W"] // 8, fx["H"] // 8, out_raw=True, t_rand=fx["t_rand"], jitter=fx["jitter"])
    for k in ("feat_coarse", "pts_coarse", "rgb_coarse", "depth_coarse", "feat_fine", "pts_fine", "rgb_fine", "depth_fine"):
        assert relerr(preds[k], fx[f"pred_{k}"], k) < TOL, k
    for lean in (True, False):
        nv = ren.render_novel_view((fx["H"], fx["W"]), fx["K"], fx["c2w"], fx["unnorm"], gpu, t_rand=fx["t_rand"], jitter=fx["jitter"], lean=lean)
        assert nv["im_pred"].shape == fx["nv_im_pred"].shape
        assert maxdiff(nv["im_pred"], fx["nv_im_pred"]) < TOL
        assert relerr(nv["pt_feat"], fx["nv_pt_feat"], "nv_pt_feat") < TOL
        assert maxdiff(nv["pt3d"], fx["nv_pt3d"]) < 3 * TOL  # world units (scene scale 3)


@pytest.mark.parametrize("S,R", [(32, 203), (64, 131), (128, 77), (256, 40), (96, 131), (192, 77), (160, 50), (48, 90), (20, 64)])
def test_render_vs_oracle_sizes(gpu, built_lib, S, R):
    """Ragged ray counts (tail workgroups) and samples-per-ray, against the oracle on the same inputs: the natively tiled row lengths
    (32, 64, k * 128) and -- round 5 -- arbitrary `num_pts` like the reference (96, 192, 160, 48, 20: run on the next native length with
    zero-width padding intervals, which carry weight exactly 0; ops.nerf_fwd)."""
    fx = load_golden("nerf_r32_s32")
    ren, sd = make_renderer(fx, gpu, S=S)
    ren.ret_pfeat = True
    g = torch.Generator().manual_seed(5)
    H, W = 8 * 16, 8 * 16
    K = torch.tensor([[100.0, 0, W / 2], [0, 100.0, H / 2], [0, 0, 1]])
    rays = no.make_rays(H, W, K, synth.camera_pose(3), ds=8)[:R].contiguous()
    t_rand, jit = synth.uniform01((R, S + 1), 11), synth.resample_jitter((R, S + 1), 12)
    ref = no.render_rays(sd, rays, t_rand, jit, S, S, stop_layer=fx["stop_layer"])
    preds = ren.predict(rays.to(gpu), 1, 1, out_raw=True, t_rand=t_rand, jitter=jit, debug=True)
    assert maxdiff(preds["t_fine"], ref["t_fine"]) < 1e-4
    for k in ("weights_coarse", "weights_fine", "feat_coarse", "feat_fine", "pts_fine", "rgb_fine", "depth_fine", "acc_fine"):
        assert maxdiff(preds[k], ref[k]) < TOL, k
    assert ref["weights_fine"].sum(-1).max() > 0.3  # not vacuous


def test_feat_comb_max(gpu, built_lib):
    fx = load_golden("nerf_r32_s32")
    ren, sd = make_renderer(fx, gpu)
    ren.ret_pfeat, ren.feat_comb = True, "max"
    ref = no.render_rays(sd, fx["rays"], fx["t_rand"], fx["jitter"], fx["S"], fx["S"], stop_layer=fx["stop_layer"], feat_comb="max")
    preds = ren.predict(fx["rays"].to(gpu), 1, 1, out_raw=True, t_rand=fx["t_rand"], jitter=fx["jitter"])
    for k in ("feat_fine", "pts_fine", "feat_coarse", "pts_coarse", "rgb_fine"):
        assert maxdiff(preds[k], ref[k]) < TOL, k


def test_unsupported_shapes_fail_loudly(gpu, built_lib):
    fx = load_golden("nerf_r32_s32")
    ren, sd = make_renderer(fx, gpu)
    rays = fx["rays"].to(gpu)
    t = torch.zeros(rays.shape[0], 49, device=gpu)  # S = 48 is not a tile shape of the C entry point (ops.nerf_fwd pads it: test_render_vs_oracle_sizes)
    blob = ren.nerf_fine.packed(gpu)
    w = torch.empty(rays.shape[0], 48, device=gpu)
    v = lambda x: _lib.dptr(x)
    rc = _lib.lib().nm_nerf_fwd(v(blob), v(rays), v(t), None, rays.shape[0], 48, 3, 0, -1.0, 0, v(w), None, v(torch.empty(rays.shape[0], 3, device=gpu)), None,
                                v(torch.empty(rays.shape[0], device=gpu)), v(torch.empty(rays.shape[0], device=gpu)), None, None, ops.stream())
    assert rc == _lib.NM_ERR_UNSUPPORTED
    with pytest.raises(_lib.NerfmatchAmdError):
        ops.nerf_fwd(blob, rays, torch.zeros(rays.shape[0], 1, device=gpu))  # no interval at all


def test_full_size_properties(gpu, built_lib):
    """BASELINE shapes (4800 rays x 64 / 128 samples): size-independent properties of the rendering."""
    fx = load_golden("nerf_r32_s32")
    for S in (64, 128):
        ren, sd = make_renderer(fx, gpu, S=S)
        nv = ren.render_novel_view((480, 640), synth.intrinsics(), synth.unnorm_scene() @ synth.camera_pose(1), synth.unnorm_scene(), gpu, lean=False)
        assert nv["pt_feat"].shape == (4800, 256) and nv["pt3d"].shape == (4800, 3) and nv["im_pred"].shape == (60, 80, 3)
        assert torch.isfinite(nv["pt_feat"]).all() and torch.isfinite(nv["pt3d"]).all()
        assert (nv["im_pred"] >= 0).all() and (nv["im_pred"] <= 1.0 + 1e-5).all()
        rays, flag = ops.raygen(synth.intrinsics(), synth.camera_pose(1), 480, 640, gpu)
        t = ops.sample_coarse(rays, torch.rand(4800, S + 1, device=gpu), S)
        assert torch.all(t[:, 1:] >= t[:, :-1]) and torch.all(t[:, 0] >= 0.01 - 1e-6) and torch.all(t[:, -1] <= rays[:, 7] + 1e-5)
        o = ops.nerf_fwd(ren.nerf_coarse.packed(gpu), rays, t)
        w = o["weights"]
        assert (w >= 0).all() and (w.sum(-1) <= 1.0 + 1e-4).all()
        assert torch.allclose(w.sum(-1), o["acc"], atol=1e-5)
        # linearity of the weighted feature sum in the weights: identical rays give identical outputs
        o2 = ops.nerf_fwd(ren.nerf_coarse.packed(gpu), rays, t)
        assert torch.equal(o["feat"], o2["feat"]) and torch.equal(o["weights"], o2["weights"])  # deterministic


def test_batched_novel_views_equal_single(gpu, built_lib):
    """render_novel_views (Q poses, one launch per kernel) == Q x render_novel_view on the same random inputs."""
    fx = load_golden("nerf_r32_s32")
    ren, sd = make_renderer(fx, gpu)
    H, W, S = fx["H"], fx["W"], fx["S"]
    R = (H // 8) * (W // 8)
    unnorm = fx["unnorm"]
    c2ws = torch.stack([unnorm @ synth.camera_pose(s) for s in (3, 4, 5)])
    t_rand, jit = synth.uniform01((3 * R, S + 1), 21), synth.resample_jitter((3 * R, S + 1), 22)
    nb = ren.render_novel_views((H, W), fx["K"], c2ws, unnorm, gpu, t_rand=t_rand, jitter=jit)
    assert nb["pt_feat"].shape == (3, R, 256) and nb["pt3d"].shape == (3, R, 3) and nb["im_pred"].shape == (3, H // 8, W // 8, 3)
    for q in range(3):
        one = ren.render_novel_view((H, W), fx["K"], c2ws[q], unnorm, gpu, t_rand=t_rand[q * R:(q + 1) * R], jitter=jit[q * R:(q + 1) * R])
        assert torch.equal(one["pt_feat"], nb["pt_feat"][q]) and torch.equal(one["pt3d"], nb["pt3d"][q]) and torch.equal(one["im_pred"], nb["im_pred"][q])


# ----------------------------------------------------------------------------- split kernels (fp16x3 / bf16x3)
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("precision", SPLIT)
def test_split_fused_pass_vs_golden(gpu, built_lib, case, precision):
    """The split kernels hold the SAME 1e-4 tolerance against the reference's golden vectors (fp16x3 on every fixture; bf16x3
    on the smooth ones -- tol_for)."""
    TOL = tol_for(precision, case)
    fx = load_golden(f"nerf_{case}")
    ren, sd = make_renderer(fx, gpu)
    rays, t = fx["rays"].to(gpu), fx["t_coarse"].to(gpu)
    app = fx["app_row"].to(gpu) if fx["app"] else None
    n = fx["sub_rays"] * fx["S"]
    for net, tap, kraw, kfeat in ((ren.nerf_coarse, -1, "mlp_raw_coarse", "mlp_feat_coarse"),
                                  (ren.nerf_fine, fx["stop_layer"], "mlp_raw_fine", "mlp_feat_fine")):
        o = ops.nerf_fwd(net.packed(gpu, precision), rays, t, app, tap_layer=tap, white_bg=bool(fx["white_bg"]), want_raw=True, want_sample_feat=True)
        e_rgb, e_sig = relerr(o["raw"].reshape(-1, 4)[:n, :3], fx[kraw][:, :3], f"{kraw}.rgb"), relerr(o["raw"].reshape(-1, 4)[:n, 3], fx[kraw][:, 3], f"{kraw}.density")
        e_feat = relerr(o["sample_feat"].reshape(-1, 256)[:n], fx[kfeat], kfeat)
        print(f"{precision} {case} {kraw}: rgb err {e_rgb:.2e} density err {e_sig:.2e} (of scale {float(fx[kraw][:, 3].abs().max()):.0f}) "
              f"feat err {e_feat:.2e} (of scale {float(fx[kfeat].abs().max()):.1f}; absolute {maxdiff(o['sample_feat'].reshape(-1, 256)[:n], fx[kfeat]):.2e})")
        assert e_rgb < TOL and e_sig < TOL and e_feat < TOL
    o = ops.nerf_fwd(ren.nerf_coarse.packed(gpu, precision), rays, t, app, tap_layer=-1, white_bg=bool(fx["white_bg"]))
    print(f"{precision} {case}: coarse compositing weights err {maxdiff(o['weights'], fx['comp_weights']):.2e}")
    assert maxdiff(o["weights"], fx["comp_weights"]) < TOL
    assert maxdiff(o["rgb"], fx["comp_rgb"]) < TOL
    assert maxdiff(o["depth"], fx["comp_depth"]) < TOL
    assert maxdiff(o["acc"], fx["comp_acc"]) < TOL


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("precision", SPLIT)
def test_split_render_vs_golden(gpu, built_lib, case, precision):
    TOL = tol_for(precision, case)
    fx = load_golden(f"nerf_{case}")
    ren, sd = make_renderer(fx, gpu)
    ren.precision = precision
    ren.ret_pfeat = True
    preds = ren.predict(fx["rays"].to(gpu), fx["W"] // 8, fx["H"] // 8, out_raw=True, t_rand=fx["t_rand"], jitter=fx["jitter"])
    for k in ("feat_coarse", "pts_coarse", "rgb_coarse", "depth_coarse", "feat_fine", "pts_fine", "rgb_fine", "depth_fine"):
        print(f"{precision} {case} {k}: abs err {maxdiff(preds[k], fx[f'pred_{k}']):.2e} of scale {float(fx[f'pred_{k}'].abs().max()):.2f}")
        assert relerr(preds[k], fx[f"pred_{k}"], k) < TOL, k
    nv = ren.render_novel_view((fx["H"], fx["W"]), fx["K"], fx["c2w"], fx["unnorm"], gpu, t_rand=fx["t_rand"], jitter=fx["jitter"])
    assert relerr(nv["pt_feat"], fx["nv_pt_feat"], "nv_pt_feat") < TOL and maxdiff(nv["pt3d"], fx["nv_pt3d"]) < 3 * TOL


@pytest.mark.parametrize("S,R", [(32, 203), (64, 131), (128, 77), (256, 40)])
@pytest.mark.parametrize("precision", SPLIT)
def test_split_sizes_vs_oracle(gpu, built_lib, S, R, precision):
    fx = load_golden("nerf_r32_s32")
    ren, sd = make_renderer(fx, gpu, S=S)
    ren.precision, ren.ret_pfeat = precision, True
    H, W = 8 * 16, 8 * 16
    K = torch.tensor([[100.0, 0, W / 2], [0, 100.0, H / 2], [0, 0, 1]])
    rays = no.make_rays(H, W, K, synth.camera_pose(3), ds=8)[:R].contiguous()
    t_rand, jit = synth.uniform01((R, S + 1), 11), synth.resample_jitter((R, S + 1), 12)
    ref = no.render_rays(sd, rays, t_rand, jit, S, S, stop_layer=fx["stop_layer"])
    preds = ren.predict(rays.to(gpu), 1, 1, out_raw=True, t_rand=t_rand, jitter=jit, debug=True)
    for k in ("weights_coarse", "weights_fine", "feat_coarse", "feat_fine", "pts_fine", "rgb_fine", "depth_fine", "acc_fine"):
        assert maxdiff(preds[k], ref[k]) < TOL, k
    for mode in ("max",):
        ren.feat_comb = mode
        refm = no.render_rays(sd, rays, t_rand, jit, S, S, stop_layer=fx["stop_layer"], feat_comb="max")
        pm = ren.predict(rays.to(gpu), 1, 1, out_raw=True, t_rand=t_rand, jitter=jit)
        assert maxdiff(pm["feat_fine"], refm["feat_fine"]) < TOL and maxdiff(pm["pts_fine"], refm["pts_fine"]) < TOL


@pytest.mark.parametrize("precision", ["fp32", "fp16x3", "bf16x3"])
def test_mip_var_scale_and_white_bg(gpu, built_lib, precision):
    """`mip_var_scale` (render_utils.py:311-312) and the white-background add (:224-225), both kernels."""
    fx = load_golden("nerf_r32_s32")
    ren, sd = make_renderer(fx, gpu)
    ren.precision, ren.ret_pfeat, ren.mip_var_scale, ren.white_bg = precision, True, 2.5, True
    ref = no.render_rays(sd, fx["rays"], fx["t_rand"], fx["jitter"], fx["S"], fx["S"], stop_layer=fx["stop_layer"], var_scale=2.5, white_bg=True)
    preds = ren.predict(fx["rays"].to(gpu), 1, 1, out_raw=True, t_rand=fx["t_rand"], jitter=fx["jitter"])
    for k in ("feat_fine", "pts_fine", "rgb_fine", "rgb_coarse", "depth_fine"):
        assert maxdiff(preds[k], ref[k]) < TOL, k
    base = no.render_rays(sd, fx["rays"], fx["t_rand"], fx["jitter"], fx["S"], fx["S"], stop_layer=fx["stop_layer"])
    assert (base["feat_fine"] - ref["feat_fine"]).abs().max() > 10 * TOL  # the option really changes the result


def test_single_ray_and_tiny_bundles(gpu, built_lib):
    """R = 1, 2, 3 rays (every workgroup mostly padding) for all supported S."""
    fx = load_golden("nerf_r32_s32")
    for S in (32, 64, 128):
        ren, sd = make_renderer(fx, gpu, S=S)
        ren.ret_pfeat = True
        for R in (1, 3):
            rays = fx["rays"][:R].contiguous()
            t_rand, jit = synth.uniform01((R, S + 1), 31), synth.resample_jitter((R, S + 1), 32)
            ref = no.render_rays(sd, rays, t_rand, jit, S, S, stop_layer=fx["stop_layer"])
            for prec in ("fp32", "fp16x3", "bf16x3"):
                ren.precision = prec
                preds = ren.predict(rays.to(gpu), 1, 1, out_raw=True, t_rand=t_rand, jitter=jit)
                assert maxdiff(preds["feat_fine"], ref["feat_fine"]) < TOL and maxdiff(preds["pts_fine"], ref["pts_fine"]) < TOL


@pytest.mark.parametrize("S,R,white", [(64, 131, False), (64, 4800, True), (128, 77, False), (64, 1, False), (64, 515, True), (256, 67, False),
                                       (512, 9, True)])
@pytest.mark.parametrize("precision", SPLIT)
def test_split_zero_tail_skip(gpu, built_lib, S, R, white, precision):
    """NM_NERF_ZERO_TAIL: the fine pass evaluates samples 0..S/2 only (the randomized resampler leaves the intervals
    s > S/2 with zero width) and must reproduce the full evaluation: weights of the tail exactly 0, everything else
    within rounding of the full pass, and the oracle within the usual tolerance."""
    fx = load_golden("nerf_r32_s32")
    ren, sd = make_renderer(fx, gpu, S=S)
    ren.precision, ren.ret_pfeat, ren.white_bg = precision, True, white
    K = torch.as_tensor(synth.intrinsics(), dtype=torch.float32).reshape(3, 3)
    rays = no.make_rays(480, 640, K, torch.as_tensor(synth.camera_pose(2), dtype=torch.float32), ds=8).reshape(-1, 12)[:R].contiguous()
    t_rand, jit = synth.uniform01((R, S + 1), 41), synth.resample_jitter((R, S + 1), 42)
    out = {}
    for skip in (False, True):
        ren.skip_zero_tail = skip
        out[skip] = ren.predict(rays.to(gpu), 1, 1, out_raw=True, t_rand=t_rand, jitter=jit)
    for k in ("feat_fine", "pts_fine", "rgb_fine", "depth_fine", "feat_coarse", "rgb_coarse"):
        assert maxdiff(out[True][k], out[False][k].cpu()) < 5e-6, k
    if R <= 600:
        ref = no.render_rays(sd, rays, t_rand, jit, S, S, stop_layer=fx["stop_layer"], white_bg=white)
        for k in ("feat_fine", "pts_fine", "rgb_fine", "depth_fine"):
            assert maxdiff(out[True][k], ref[k]) < TOL, k
    # the kernel itself: tail weights exactly zero, head weights equal to the full pass
    dev = gpu
    rg = rays.to(dev)
    t_c = ops.sample_coarse(rg, t_rand.to(dev), S)
    blob_c, blob_f = ren.nerf_coarse.packed(dev, precision), ren.nerf_fine.packed(dev, precision)
    wc = ops.nerf_fwd(blob_c, rg, t_c, need_rgb=False, need_feat=False)["weights"]
    t_f = ops.resample(t_c, wc, jit.to(dev), 0.01, True)
    assert bool((t_f[:, S // 2 + 1:] == t_f[:, S // 2 + 1: S // 2 + 2]).all())  # the premise: fence posts > S/2 coincide
    full = ops.nerf_fwd(blob_f, rg, t_f, tap_layer=3, white_bg=white)
    fast = ops.nerf_fwd(blob_f, rg, t_f, tap_layer=3, white_bg=white, zero_tail=True)
    assert float(fast["weights"][:, S // 2 + 1:].abs().max()) == 0.0 and float(full["weights"][:, S // 2 + 1:].abs().max()) == 0.0
    for k in ("weights", "feat", "pts", "rgb", "depth", "acc"):
        assert maxdiff(fast[k], full[k].cpu()) < 5e-6, k


@pytest.mark.parametrize("precision", ["fp32", "fp16x3", "bf16x3"])
def test_novel_view_without_im_pred(gpu, built_lib, precision):
    """want_im_pred=False (the localisation loop's render): the fine pass skips its colour heads, pt3d / pt_feat unchanged."""
    fx = load_golden("nerf_r128_s64_app")
    ren, sd = make_renderer(fx, gpu)
    ren.precision = precision
    kw = dict(t_rand=fx["t_rand"], jitter=fx["jitter"])
    full = ren.render_novel_view((fx["H"], fx["W"]), fx["K"], fx["c2w"], fx["unnorm"], gpu, **kw)
    fast = ren.render_novel_view((fx["H"], fx["W"]), fx["K"], fx["c2w"], fx["unnorm"], gpu, want_im_pred=False, **kw)
    assert fast["im_pred"] is None and full["im_pred"] is not None
    assert maxdiff(fast["pt_feat"], full["pt_feat"].cpu()) < 1e-6 and maxdiff(fast["pt3d"], full["pt3d"].cpu()) < 1e-6
    assert maxdiff(fast["pt_feat"], fx["nv_pt_feat"]) < TOL
    batched = ren.render_novel_views((fx["H"], fx["W"]), fx["K"], torch.stack([fx["c2w"], fx["c2w"]]), fx["unnorm"], gpu, want_im_pred=False,
                                     t_rand=torch.cat([fx["t_rand"]] * 2), jitter=torch.cat([fx["jitter"]] * 2))
    assert batched["im_pred"] is None and maxdiff(batched["pt_feat"][1], full["pt_feat"].cpu()) < 1e-6


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_nerf_module_forward_vs_golden(gpu, built_lib, case, precision):
    """NeRF.forward(x, ret_pfeat=1, val=True) -- the per-sample call the reference's iNeRF loop makes on the sub-modules
    (nerfmatch_evaluator.py:402-406) -- against the reference's own per-sample outputs (GEMM-chain path, both arithmetic
    settings of nm_linear)."""
    fx = load_golden(f"nerf_{case}")
    ren, sd = make_renderer(fx, gpu)
    n, S = fx["sub_rays"] * fx["S"], fx["S"]
    dirs = fx["dir_pe"][torch.arange(n) // S]
    cols = [fx["ipe_coarse"], dirs] + ([fx["app_row"][None].expand(n, -1)] if fx["app"] else [])
    x = torch.cat(cols, -1).to(gpu)
    ops.LINEAR_PRECISION = precision
    try:
        for net, kraw, kfeat in ((ren.nerf_coarse, "mlp_raw_coarse", "mlp_feat_coarse"), (ren.nerf_fine, "mlp_raw_fine", "mlp_feat_fine")):
            raw, feat = net(x.reshape(fx["sub_rays"], S, -1), ret_pfeat=1, val=True)
            assert raw.shape == (fx["sub_rays"], S, 4) and feat.shape == (fx["sub_rays"], S, 256)
            assert relerr(raw.reshape(-1, 4)[:, :3], fx[kraw][:, :3]) < TOL and relerr(raw.reshape(-1, 4)[:, 3], fx[kraw][:, 3]) < TOL
            assert relerr(feat.reshape(-1, 256), fx[kfeat]) < TOL
            assert relerr(net(x)[:, :3], fx[kraw][:, :3]) < TOL and relerr(net(x)[:, 3], fx[kraw][:, 3]) < TOL  # ret_pfeat = 0: outputs only
    finally:
        ops.LINEAR_PRECISION = "fp32"


def test_fp16x1_coarse_pass(gpu, built_lib):
    """The single-product fp16 kernel (coarse pass of the lean render): its own outputs against the oracle (fp16-class error),
    and -- what it is for -- the FINE outputs of a render whose coarse pass ran on it against the fp32 oracle: unchanged at the
    1e-6 level, because the coarse weights only place the fine samples."""
    fx = load_golden("nerf_r32_s32")
    for S, R in ((64, 600), (128, 200)):
        ren, sd = make_renderer(fx, gpu, S=S)
        ren.precision, ren.ret_pfeat = "fp16x3", True
        H, W = 8 * 32, 8 * 32
        K = torch.tensor([[120.0, 0, W / 2], [0, 120.0, H / 2], [0, 0, 1]])
        rays = no.make_rays(H, W, K, synth.camera_pose(3), ds=8)[:R].contiguous()
        t_rand, jit = synth.uniform01((R, S + 1), 11), synth.resample_jitter((R, S + 1), 12)
        ref = no.render_rays(sd, rays, t_rand, jit, S, S, stop_layer=fx["stop_layer"])
        # the kernel itself: weights of the coarse pass
        t_c = ops.sample_coarse(rays.to(gpu), t_rand.to(gpu), S)
        w16 = ops.nerf_fwd(ren.nerf_coarse.packed(gpu, "fp16x1"), rays.to(gpu), t_c, tap_layer=-1, need_rgb=False, need_feat=False)["weights"]
        w48 = ops.nerf_fwd(ren.nerf_coarse.packed(gpu, "fp16x3"), rays.to(gpu), t_c, tap_layer=-1, need_rgb=False, need_feat=False)["weights"]
        e16, e48 = maxdiff(w16, ref["weights_coarse"]), maxdiff(w48, ref["weights_coarse"])
        print(f"S={S}: coarse weights vs oracle: fp16x1 {e16:.2e}, fp16x3 {e48:.2e}")
        assert e16 < 2e-3 and e48 < 1e-5
        # the render: lean (fp16x1 coarse) against the oracle, and against the same render with the bf16x3 coarse pass
        errs = {}
        for cp in ("fp16x1", "same"):
            ren.coarse_precision = cp
            p = ren.render_rays(rays.to(gpu), validation=True, t_rand=t_rand, jitter=jit, lean=True)
            errs[cp] = {k: maxdiff(p[k], ref[k]) for k in ("feat_fine", "pts_fine", "rgb_fine", "depth_fine")}
            assert ("pts_coarse" in p) == (cp == "same")
        print(f"S={S}: fine outputs vs oracle: coarse fp16x1 {errs['fp16x1']}, coarse fp16x3 {errs['same']}")
        for k, v in errs["fp16x1"].items():
            assert v < 1e-5, (k, v)


def test_fp16x1_saturates_instead_of_overflowing(gpu, built_lib):
    """Activations beyond the fp16 range (weights scaled up by 300 in two layers) saturate at 65504 in the single-product fp16 pass:
    the weights it returns stay finite."""
    fx = load_golden("nerf_r32_s32")
    ren, sd = make_renderer(fx, gpu, S=64)
    with torch.no_grad():
        ren.nerf_coarse.pts_linears[1].weight.mul_(300.0)
        ren.nerf_coarse.pts_linears[2].weight.mul_(300.0)
    rays = fx["rays"].to(gpu)
    t_c = ops.sample_coarse(rays, torch.rand(rays.shape[0], 65, device=gpu), 64)
    w = ops.nerf_fwd(ren.nerf_coarse.packed(gpu, "fp16x1"), rays, t_c, tap_layer=-1, need_rgb=False, need_feat=False)["weights"]
    assert torch.isfinite(w).all() and float(w.min()) >= 0.0 and float(w.sum(1).max()) <= 1.0 + 1e-5


# ----------------------------------------------------------------------------- trained-like regime (round 3)
@pytest.mark.parametrize("precision", ["fp32", "fp16x3", "bf16x3"])
def test_surface_fine_weights_and_zero_tail(gpu, built_lib, precision):
    """Trained-like NeRF (tests/golden/nerf_surface_r512_s128.npz, generated by the reference): the fine pass's own
    compositing weights / opacity -- 20 samples across a surface whose density runs into the thousands -- and the
    zero-width-tail skip in that regime."""
    fx = load_golden("nerf_surface_r512_s128")
    ren, sd = make_renderer(fx, gpu)
    rays, t_f = fx["rays"].to(gpu), fx["t_fine"].to(gpu)
    blob = ren.nerf_fine.packed(gpu, precision)
    o = ops.nerf_fwd(blob, rays, t_f, tap_layer=3, want_raw=True)
    e_w, e_a = maxdiff(o["weights"], fx["fine_weights"]), maxdiff(o["acc"], fx["fine_acc"])
    e_s = relerr(o["raw"][..., 3], fx["fine_sigma"], "fine_sigma")
    print(f"surface fine pass [{precision}]: weights {e_w:.2e} acc {e_a:.2e} density {e_s:.2e} of scale {float(fx['fine_sigma'].abs().max()):.0f}")
    tol = tol_for(precision, "surface")
    assert e_w < tol and e_a < tol and e_s < tol
    assert relerr(o["feat"], fx["pred_feat_fine"], "feat_fine") < tol and maxdiff(o["pts"], fx["pred_pts_fine"]) < tol
    if precision != "fp32":
        fast = ops.nerf_fwd(blob, rays, t_f, tap_layer=3, zero_tail=True)
        S = int(fx["S"])
        assert float(fast["weights"][:, S // 2 + 1:].abs().max()) == 0.0
        for k in ("weights", "feat", "pts", "rgb", "depth", "acc"):
            assert relerr(fast[k], o[k].cpu()) < 1e-5, k


def test_surface_fp16x1_coarse_pass_measured(gpu, built_lib):
    """the fp16x1 coarse pass MEASURED on the trained-like fixture.  It is opt-in since round 3
    (`NerfRenderer.coarse_precision = "same"` by default); this test records what it costs in this regime: the coarse
    weights themselves and the FINE outputs of a lean render whose coarse pass ran on one fp16 product, against the
    reference's golden outputs, next to the same render with the coarse pass on fp16x3 (the parity arithmetic).
    Measured (round 3): coarse weights 6.6e-2 off with fp16x1 (fp16x3: < 1e-4), rendered features 1.8e-3 of scale, points 4.6e-4 --
    18x / 13x the parity render's error and outside the 1e-4 class: on a scene with surfaces the single-product coarse pass DOES
    show in the fine outputs, so it stays opt-in."""
    fx = load_golden("nerf_surface_r512_s128")
    ren, sd = make_renderer(fx, gpu)
    ren.precision, ren.ret_pfeat = "fp16x3", True
    rays, t_c = fx["rays"].to(gpu), fx["t_coarse"].to(gpu)
    w16 = ops.nerf_fwd(ren.nerf_coarse.packed(gpu, "fp16x1"), rays, t_c, tap_layer=-1, need_rgb=False, need_feat=False)["weights"]
    w48 = ops.nerf_fwd(ren.nerf_coarse.packed(gpu, "fp16x3"), rays, t_c, tap_layer=-1, need_rgb=False, need_feat=False)["weights"]
    e16, e48 = maxdiff(w16, fx["comp_weights"]), maxdiff(w48, fx["comp_weights"])
    errs = {}
    for cp in ("fp16x1", "same"):
        ren.coarse_precision = cp
        nv = ren.render_novel_view((fx["H"], fx["W"]), fx["K"], fx["c2w"], fx["unnorm"], gpu, t_rand=fx["t_rand"], jitter=fx["jitter"], lean=True)
        errs[cp] = (relerr(nv["pt_feat"], fx["nv_pt_feat"], f"nv_pt_feat.coarse_{cp}"), maxdiff(nv["pt3d"], fx["nv_pt3d"]), maxdiff(nv["im_pred"], fx["nv_im_pred"]))
    print(f"surface: coarse weights vs reference: fp16x1 {e16:.2e}, fp16x3 {e48:.2e}; lean render (pt_feat rel, pt3d abs, rgb abs): "
          f"coarse fp16x1 {errs['fp16x1']}, coarse fp16x3 {errs['same']}")
    assert e48 < TOL
    assert errs["same"][0] < TOL and errs["same"][1] < 3 * TOL and errs["same"][2] < TOL
    # the single-product pass is a throughput option: finite, normalised weights; its error is reported above, and bounded
    # here only loosely (fp16 has 11 significant bits; a density of +-1e4 moves by several units)
    assert torch.isfinite(w16).all() and float(w16.sum(1).max()) <= 1.0 + 1e-4
    assert errs["fp16x1"][0] < 5e-2 and errs["fp16x1"][1] < 5e-2


@pytest.mark.parametrize("case", ["r128_s64_app", "surface_r512_s128"])
def test_single_product_render_error_stated(gpu, built_lib, case):
    """BASELINE config 3's "bf16" throughput configuration (16-bit operands, ONE product per block, fp32
    accumulate, looser tolerance, reported separately): the WHOLE render -- coarse and fine pass, every head -- on the
    single-product fp16 kernel (`NerfRenderer.precision = "fp16x1"`).  Not a parity arithmetic; stated bounds against the
    reference's golden outputs: smooth field (r128_s64_app) features 2e-3 of scale, colours 2e-3, points 2e-3; trained-like
    field (surface) features 5e-2 of scale, points 2e-2 (the density of +-1e4 moves by ~10 units: sample weights shift)."""
    fx = load_golden(f"nerf_{case}")
    ren, sd = make_renderer(fx, gpu)
    ren.precision, ren.ret_pfeat = "fp16x1", True
    preds = ren.predict(fx["rays"].to(gpu), fx["W"] // 8, fx["H"] // 8, out_raw=True, t_rand=fx["t_rand"], jitter=fx["jitter"])
    errs = {k: relerr(preds[k], fx[f"pred_{k}"]) for k in ("feat_coarse", "feat_fine", "pts_fine", "rgb_fine", "depth_fine")}
    print(f"single-product (fp16x1) whole render {case}: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert all(torch.isfinite(preds[k]).all() for k in errs)
    lim = dict(feat=5e-2, other=2e-2) if case.startswith("surface") else dict(feat=2e-3, other=2e-3)
    assert errs["feat_fine"] < lim["feat"] and errs["feat_coarse"] < lim["feat"]
    assert errs["pts_fine"] < lim["other"] and errs["rgb_fine"] < lim["other"] and errs["depth_fine"] < lim["other"]
    assert errs["feat_fine"] > 1e-6  # (it is the single-product path: the parity kernels are at 1e-7 here)


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_fine_sample_count_differs_from_coarse(gpu, built_lib, precision):
    """Round 4 (VERDICT r3 missing 2): a config with coarse_nerf.num_pts != fine_nerf.num_pts is accepted and behaves like the reference --
    in the mip configuration its resampler draws as many fence posts as it is given, so the fine pass has the COARSE count
    (tests/golden/nerf_fine_count_c32_f64.npz: the reference's own outputs for (32, 64), equal to its (32, 32) run)."""
    fx = load_golden("nerf_fine_count_c32_f64")
    cfg = synth.nerf_config("7scenes", num_pts=int(fx["S_coarse"]), img_wh=(int(fx["W"]), int(fx["H"])), num_pts_fine=int(fx["S_fine"]))
    ren = NerfRenderer(cfg, training=False, stop_layer=3)
    ren.load_state_dict(synth.nerf_state_dict(seed=int(fx["weights_seed"]), density_bias=3.0), strict=True)
    ren.to(gpu).eval()
    ren.precision, ren.ret_pfeat = precision, True
    assert ren.num_pts_fine == 64 and ren.num_pts_coarse == 32
    preds = ren.predict(fx["rays"].to(gpu), 1, 1, out_raw=True, t_rand=fx["t_rand"], jitter=fx["jitter"])
    for k in ("feat_coarse", "pts_coarse", "rgb_coarse", "depth_coarse", "feat_fine", "pts_fine", "rgb_fine", "depth_fine"):
        assert maxdiff(preds[k], fx[f"pred_{k}"]) < TOL, k


@pytest.mark.parametrize("R,S,randomized", [(4801, 64, True), (13, 64, False), (100, 32, True), (7, 20, True), (1, 64, True), (15, 63, True)])
def test_packed_resampler_equals_the_one_ray_per_workgroup_form(gpu, built_lib, R, S, randomized):
    """nm_resample for rows of at most 64 intervals packs seven rays into a workgroup (round 5); NM_RESAMPLE_PACK=0 selects the original
    kernel: the fence posts and the zero-tail flag are the same bits, for ragged last workgroups and short rows too."""
    import os

    g = torch.Generator().manual_seed(R + S)
    t = torch.sort(torch.rand(R, S + 1, generator=g) * 3 + 0.1, -1).values.to(gpu).contiguous()
    w = (torch.rand(R, S, generator=g) ** 4).to(gpu)
    jit = (torch.rand(R, S + 1, generator=g) * (1.0 / (S + 1) - 1.2e-7)).to(gpu)
    if R > 3:
        jit[3, S] = 0.5  # (breaks the zero-tail premise of that ray: the flag must be raised by both forms)
    a, fa = ops.resample(t, w, jit, 0.01, randomized, want_tail_flag=True)
    os.environ["NM_RESAMPLE_PACK"] = "0"
    try:
        b, fb = ops.resample(t, w, jit, 0.01, randomized, want_tail_flag=True)
    finally:
        del os.environ["NM_RESAMPLE_PACK"]
    assert torch.equal(a, b) and int(fa) == int(fb)
    assert torch.isfinite(a).all() and bool((a[:, 1:] >= a[:, :-1]).all())


@pytest.mark.parametrize("H,W,ds", [(480, 640, 8), (60, 81, 8), (37, 53, 4), (16, 24, 1), (33, 40, 5)])
def test_per_pixel_raygen_equals_the_per_ray_form(gpu, built_lib, H, W, ds):
    """nm_raygen_batch gives every full-resolution pixel's far-plane discriminant its own thread (round 5); NM_RAYGEN_PIXELS=0 selects the
    original one-thread-per-ray kernel: identical rays and flags, image sizes that are no multiples of ds, a pose whose full-resolution grid
    has a negative discriminant (the far fall-back) beside ordinary ones."""
    import os

    K = synth.intrinsics(H, W)
    inside = torch.stack([synth.camera_pose(seed=s) for s in range(3)])
    outside = synth.camera_pose(seed=7).clone()
    outside[:3, 3] = torch.tensor([3.0, 0.5, -2.0])  # outside the unit sphere: rays that miss it have a negative discriminant
    poses = torch.cat([inside, outside[None]])
    a, fa = ops.raygen_batch(K, poses, H, W, gpu, ds=ds)
    os.environ["NM_RAYGEN_PIXELS"] = "0"
    try:
        b, fb = ops.raygen_batch(K, poses, H, W, gpu, ds=ds)
    finally:
        del os.environ["NM_RAYGEN_PIXELS"]
    assert torch.equal(fa, fb) and torch.equal(a, b)
    assert fa.tolist()[:3] == [0, 0, 0] and int(fa[3]) == 1


@pytest.mark.parametrize("S", [64, 128, 20])
def test_resampler_scales_the_jitter_like_an_elementwise_product(gpu, built_lib, S):
    """nm_resample_scaled (round 5): jitter * scale inside the kernel = the tensor scaled by torch first, bit for bit (the renderer's own
    draw goes in unscaled: one launch less in front of the coarse pass)."""
    g = torch.Generator().manual_seed(S)
    R = 333
    t = torch.sort(torch.rand(R, S + 1, generator=g) * 3 + 0.1, -1).values.to(gpu).contiguous()
    w = torch.rand(R, S, generator=g).to(gpu)
    raw = torch.rand(R, S + 1, generator=g).to(gpu)
    scale = 1.0 / (S + 1) - float(torch.finfo(torch.float32).eps)
    a, fa = ops.resample(t, w, raw, 0.01, True, want_tail_flag=True, jitter_scale=scale)
    b, fb = ops.resample(t, w, raw * scale, 0.01, True, want_tail_flag=True)
    assert torch.equal(a, b) and int(fa) == int(fb) == 0

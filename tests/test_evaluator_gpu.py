"""GPU: the evaluator's render -> match loop (eval_batch / eval_data_loader) and the checkpoint loaders."""
import pytest
import torch
from argparse import Namespace

from nerfmatch_amd import synth
from nerfmatch_amd.nerf_evaluator import load_nerf_render_from_ckpt, save_nerf_ckpt
from nerfmatch_amd.nerfmatch_evaluator import NeRFMatchEvaluator, load_nerfmatch_from_ckpt
from nerfmatch_amd.modules import StubBackbone

pytestmark = pytest.mark.gpu


def make_batch(H, W, q):
    unnorm = synth.unnorm_scene()
    M = (H // 8) * (W // 8)
    ys, xs = torch.meshgrid(torch.arange(H // 8), torch.arange(W // 8), indexing="ij")
    g = torch.Generator().manual_seed(q)
    return dict(image=torch.randn(1, 3, H, W, generator=g), im_mask=torch.ones(1, M, dtype=torch.bool), K=synth.intrinsics(H, W, 120.0)[None],
                c2w=(unnorm @ synth.camera_pose(q))[None], rc2w=(unnorm @ synth.camera_pose(q + 100))[None],
                pt2d=(torch.stack([xs, ys], -1) * 8 + 4).float().reshape(1, M, 2), unnorm_scene=unnorm[None])


def test_ckpt_roundtrip_and_eval_loop(gpu, built_lib, tmp_path):
    H, W, S = 96, 128, 32
    cfg = synth.nerf_config("7scenes", num_pts=S, img_wh=(W, H))
    sd = synth.nerf_state_dict(seed=0, density_bias=3.0)
    save_nerf_ckpt(tmp_path / "nerf.ckpt", cfg, sd, unnorm_scene=synth.unnorm_scene())
    ren = load_nerf_render_from_ckpt(str(tmp_path / "nerf.ckpt"), gpu, stop_layer=3)
    assert ren.nerf_fine.stop_layer == 3 and ren.unnorm_scene is not None
    assert torch.equal(ren.nerf_fine.pts_linears[5].weight.cpu(), sd["nerf_fine.pts_linears.5.weight"])

    mcfg = Namespace(model=synth.matcher_config("c2f"), exp=Namespace(seed=1), data=Namespace())
    msd = {f"model.{k}": v for k, v in synth.matcher_state_dict("c2f").items()}
    torch.save(dict(state_dict=msd, hyper_parameters=vars(mcfg), epoch=1, global_step=2), tmp_path / "m.ckpt")
    ev = load_nerfmatch_from_ckpt(str(tmp_path / "m.ckpt"))
    assert isinstance(ev, NeRFMatchEvaluator) and not ev.coarse_only
    assert torch.equal(ev.model.pt_pe_proj.weight.cpu(), synth.matcher_state_dict("c2f")["pt_pe_proj.weight"])

    batches = [make_batch(H, W, q) for q in range(3)]
    # query2query: render at the query pose, match; no PnP solver in this image -> solver "none" returns matches only
    out = ev.eval_data_loader(renderer=ren, data_loader=batches, solver="none", query2query=True, mutual=True)
    assert out["query_idx"].tolist() == [0, 1, 2]
    assert (out["num_matches"] >= 0).all() and out["c2w_est"].shape == (3, 4, 4)
    b = batches[0]
    assert b["pt3d"].shape == (1, (H // 8) * (W // 8), 3) and b["pt_feat"].shape[-1] == 256  # filled in by the render
    assert "mpt2d_f" in b and b["mpt2d_f"].shape[0] == b["mpt3d"].shape[0]
    with pytest.raises(ImportError):
        ev.eval_batch(make_batch(H, W, 5), renderer=ren, solver="colmap", query2query=True)


def test_scene_cache_roundtrip(gpu, built_lib, tmp_path):
    """cache_scene_pts writes the reference's per-frame dict format; cached points then drive the matcher (cached_pt path)."""
    import numpy as np
    from nerfmatch_amd.nerf.renderer import NerfRenderer
    from nerfmatch_amd.scene_cache import cache_scene_pts, load_frame_3d

    H, W, S = 64, 96, 32
    ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=S, img_wh=(W, H)), training=False, stop_layer=3)
    ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0))
    ren.to(gpu).eval()
    unnorm = synth.unnorm_scene()
    frames = [(f"seq-01_frame-{i:06d}", unnorm @ synth.camera_pose(i)) for i in range(5)]
    files = cache_scene_pts(ren, frames, synth.intrinsics(H, W, 80.0), (H, W), unnorm, tmp_path, gpu, batch=2)
    assert len(files) == 5 and files[0].parent.name == "ds8lin"
    R = (H // 8) * (W // 8)
    d = np.load(files[3], allow_pickle=True).item()
    assert set(d) == {"pt3d", "unnorm_scene", "pt_feat", "pt_color"}
    assert d["pt3d"].shape == (R, 3) and d["pt_feat"].shape == (R, 256) and d["pt_color"].shape == (R, 3)
    assert d["pt_color"].min() >= 0 and d["pt_color"].max() <= 1 and np.isfinite(d["pt_feat"]).all()
    pt3d, pt_feat, mask, un = load_frame_3d(files[3])
    assert mask.all() and np.allclose(un, unnorm.numpy())


def test_inerf_refinement_through_the_evaluator(gpu, built_lib):
    """NeRFMatchEvaluator.inerf_refinement: `eval_pose` branch against the reference's final pose (golden), the re-match
    branch end to end (solver "none": no PnP package in this image), and the unsupported options fail loudly."""
    from conftest import load_golden
    from nerfmatch_amd.nerf.renderer import NerfRenderer

    fx = load_golden("inerf_7s")
    H, W = int(fx["H"]), int(fx["W"])
    ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=128, img_wh=(W, H)), training=False, stop_layer=3)
    ren.load_state_dict(synth.nerf_state_dict(seed=int(fx["weights_seed"]), density_bias=3.0), strict=True)
    ren.to(gpu).eval()
    ev = NeRFMatchEvaluator(Namespace(model=synth.matcher_config("c2f"), exp=Namespace(seed=1), data=Namespace()))
    ev.model.load_state_dict(synth.matcher_state_dict("c2f"), strict=False)
    ev.model.backbone = StubBackbone().to(gpu)
    n = int(fx["num_optim"])
    M = (H // 8) * (W // 8)
    ys, xs = torch.meshgrid(torch.arange(H // 8), torch.arange(W // 8), indexing="ij")
    batch = dict(image=fx["image"].to(gpu), K=fx["K"][None], c2w=fx["c2w_gt"][None], im_mask=torch.ones(1, M, dtype=torch.bool, device=gpu),
                 pt2d=(torch.stack([xs, ys], -1) * 8 + 4).float().reshape(1, M, 2).to(gpu))
    conf = Namespace(lrate=float(fx["lrate"]), lrdecay=False, num_optim=n, eval_pose=True, ds=8)
    est, R_err, t_err = ev.inerf_refinement(batch, ren, fx["unnorm"], fx["c2w_est0"], conf, t_rands=list(fx["t_rands"]), jitters=list(fx["jitters"]))
    err = (est - fx["poses"][-1]).abs()  # (see test_inerf_gpu.py: Adam makes a few entries ill-conditioned)
    assert (err < 3e-4).float().mean().item() > 0.9 and err.max().item() < 2e-3
    assert abs(t_err - float(fx["t_err"])) < 2e-3
    assert len(ev.timer["inerf_step_time"]) == n
    conf.eval_pose = False
    est2, R2, t2 = ev.inerf_refinement(batch, ren, fx["unnorm"], fx["c2w_est0"], conf, solver="none")
    assert est2 is None and batch["pt3d"].shape == (1, M, 3) and batch["pt_feat"].shape == (1, M, 256) and "mpt3d" in batch
    with pytest.raises(NotImplementedError):
        ev.inerf_refinement(batch, ren, fx["unnorm"], fx["c2w_est0"], Namespace(use_match_loss=True))

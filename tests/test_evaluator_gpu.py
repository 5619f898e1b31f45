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
    branch end to end (solver "none": no PnP package in this image), and the `use_match_loss` option against the reference's
    own run with the matching term (tests/golden/inerf_match.npz)."""
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
    # the matching term: the reference's trajectory of the same call (first Adam step is +-lr per entry: sign-exact)
    from nerfmatch_amd.modules import PrecomputedBackbone

    fx = load_golden("inerf_match")
    H, W, seed = int(fx["H"]), int(fx["W"]), int(fx["weights_seed"])
    ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=128, img_wh=(W, H)), training=False, stop_layer=3)
    ren.load_state_dict(synth.nerf_state_dict(seed=seed, density_bias=3.0), strict=True)
    ren.to(gpu).eval()
    ev.model.load_state_dict(synth.matcher_state_dict("c2f", seed=seed), strict=False)
    ev.model.backbone = PrecomputedBackbone((fx["cfeat"].to(gpu), fx["ffeat"].to(gpu)), [256, 128])
    M = (H // 8) * (W // 8)
    batch = dict(image=fx["image"].to(gpu), K=fx["K"][None], c2w=fx["c2w_gt"][None], im_mask=torch.ones(1, M, dtype=torch.bool, device=gpu),
                 pt_mask=torch.ones(1, M, dtype=torch.bool, device=gpu))
    conf = Namespace(lrate=float(fx["lrate"]), lrdecay=False, num_optim=int(fx["num_optim"]), eval_pose=True, ds=8, use_match_loss=True)
    est, R_err, t_err = ev.inerf_refinement(batch, ren, fx["unnorm"], fx["c2w_est0"], conf, t_rands=list(fx["t_rands"]), jitters=list(fx["jitters"]))
    assert (est - fx["poses"][-1]).abs().max().item() < 2e-3 and abs(t_err - float(fx["t_err"])) < 2e-3
    no_match = Namespace(**{**vars(conf), "use_match_loss": False})
    est0, _, _ = ev.inerf_refinement(batch, ren, fx["unnorm"], fx["c2w_est0"], no_match, t_rands=list(fx["t_rands"]), jitters=list(fx["jitters"]))
    assert (est0 - est).abs().max().item() > 2e-3  # a different trajectory without the term


# ----------------------------------------------------------------------------------------------- batched / pipelined localisation
def _c2f_evaluator(gpu, H, W):
    ev = NeRFMatchEvaluator(Namespace(model=synth.matcher_config("c2f"), exp=Namespace(seed=1), data=Namespace()))
    ev.model.load_state_dict(synth.matcher_state_dict("c2f"), strict=False)
    ev.model.backbone = StubBackbone().to(gpu)
    return ev


def _renderer(gpu, H, W, S=32):
    from nerfmatch_amd.nerf.renderer import NerfRenderer

    ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=S, img_wh=(W, H)), training=False, stop_layer=3)
    ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0))
    return ren.to(gpu).eval()


def _stack(batches):
    return {k: torch.cat([b[k] for b in batches]) for k in batches[0]}


def test_eval_data_loader_batches_of_queries(gpu, built_lib):
    """A loader whose batches hold Q > 1 queries (the reference's `batch_size`, nerfmatch_evaluator.py:726-731,864-869): the Q
    queries are rendered and matched as one launch sequence and consecutive batches are pipelined across the matcher's
    synchronisation point.  With the device generator re-seeded, the batched result equals what render_novel_views + the
    matcher give for the same batch directly; records come back per query, in order."""
    H, W = 96, 128
    ev, ren = _c2f_evaluator(gpu, H, W), _renderer(gpu, H, W)
    singles = [make_batch(H, W, q) for q in range(5)]
    loader = [_stack(singles[0:2]), _stack(singles[2:4]), _stack(singles[4:5])]  # batch_size 2, ragged tail
    torch.manual_seed(7)
    out = ev.eval_data_loader(renderer=ren, data_loader=loader, solver="none", query2query=True, mutual=True)
    assert out["query_idx"].tolist() == [0, 1, 2, 3, 4] and out["c2w_est"].shape == (5, 4, 4)
    assert len(ev.timer["localize_time"]) == 3 and len(ev.timer["match_time"]) == 3
    # solver "none": the pose the points were rendered from is kept (query2query: the query pose)
    for q in range(5):
        assert torch.allclose(torch.from_numpy(out["c2w_est"][q]), singles[q]["c2w"][0], atol=1e-6)
    # the same three batches by hand, same generator state
    torch.manual_seed(7)
    nums = []
    for b in [_stack(singles[0:2]), _stack(singles[2:4]), _stack(singles[4:5])]:
        o = ren.render_novel_views((H, W), b["K"][0], b["c2w"], b["unnorm_scene"][0], gpu, want_im_pred=False)
        d = dict(image=b["image"].to(gpu), im_mask=b["im_mask"].to(gpu), pt2d=b["pt2d"].to(gpu), pt3d=o["pt3d"], pt_feat=o["pt_feat"],
                 pt_mask=torch.ones_like(o["pt3d"][..., 0]))
        ev.model.forward(d, mutual=True)
        nums += torch.bincount(d["m_bids"].cpu(), minlength=b["image"].shape[0]).tolist()
    assert out["num_matches"].tolist() == nums and sum(nums) > 0
    # per-query lists from eval_batch; the batch dict carries the per-query match counts
    m = ev.eval_batch(_stack(singles[0:2]), renderer=ren, solver="none", query2query=True)
    assert len(m["R_err"]) == len(m["num_matches"]) == len(m["c2w_ests"]) == 2


class _SceneDataset(torch.utils.data.Dataset):
    """Stand-in for the reference's per-scene dataset objects (NeRFMatchPair, out of scope): `.scene`, `.scene_dir` and
    per-sample dicts WITHOUT the batch dimension (the DataLoader's default collate adds it)."""

    def __init__(self, scene, n, H, W):
        self.scene, self.scene_dir, self.samples = scene, f"cache/{scene}/inter_layer3/ds8lin", [make_batch(H, W, q) for q in range(n)]

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        return {k: v[0] for k, v in self.samples[i].items()}


def test_eval_multi_scenes_with_the_benchmark_call(gpu, built_lib, tmp_path):
    """The keyword call of model_eval/benchmark_nerfmatch.py:126-151 (eval_ckpt) against nerfmatch_amd, dataset classes and
    NeRF checkpoints replaced by the two factories; result-cache files carry the reference's names and are re-used."""
    H, W = 96, 128
    ev = _c2f_evaluator(gpu, H, W)
    made = []

    def renderer_factory(scene, scene_dir, stop_layer):
        made.append((scene, stop_layer))
        r = _renderer(gpu, H, W)
        r.unnorm_scene = synth.unnorm_scene()
        return r

    ev.dataset_factory = lambda conf, split: [_SceneDataset("chess", 3, H, W), _SceneDataset("fire", 2, H, W)]
    ev.renderer_factory = renderer_factory
    args = Namespace(rthres=1, center_subpixel=False, solver="none", split="test", mutual=True, match_thres=0.0, iters=1, nerf_path=None,
                     test_pair_txt=None, scene_dir=None, query2query=True, ow_cache=False, debug=False, no_cache_pt=False,
                     cache_dir=str(tmp_path / "res"), cache_iters=False, retrieval_only=False, match_oracle=False, visualize=False, seed=3)
    call = lambda: ev.eval_multi_scenes(  # the statement at benchmark_nerfmatch.py:126-151, verbatim keywords
        rthres=args.rthres, center_subpixel=args.center_subpixel, solver=args.solver, split=args.split, mutual=args.mutual,
        match_thres=args.match_thres, iters=args.iters, nerf_path=args.nerf_path, test_pair_txt=args.test_pair_txt, scene_dir=args.scene_dir,
        data_conf=Namespace(), query2query=args.query2query, ow_cache=args.ow_cache, inerf_conf=None, debug=args.debug,
        cached_pt=not args.no_cache_pt, cache_dir=args.cache_dir, cache_iters=args.cache_iters, retrieval_only=args.retrieval_only,
        match_oracle=args.match_oracle, visualize=args.visualize, seed=args.seed)
    summ = call()
    assert [s_ for s_, _ in made] == ["chess", "fire"] and made[0][1] == 3  # stop layer parsed from scene_dir ("inter_layer3")
    assert len(summ) == 2 and set(summ[0]) >= {"t_med", "r_med", "recall", "match_time"}
    import numpy as np
    f = tmp_path / "res" / "chess_rth1test_none_itr1.query2query.npy"  # the reference's naming scheme (:782-850)
    assert f.exists() and (tmp_path / "res" / "fire_rth1test_none_itr1.query2query.npy").exists()
    cached = np.load(f, allow_pickle=True).item()
    assert cached["query_idx"].tolist() == [0, 1, 2] and len(cached["match_time"]) == 3 and (cached["num_matches"] >= 0).all()
    made.clear()
    summ2 = call()  # second call: metrics come from the cache files, nothing is rendered
    assert made == [] and len(summ2) == 2
    # batch_size > 1 through the same entry point
    args.ow_cache = True
    s3 = ev.eval_multi_scenes(batch_size=2, solver="none", query2query=True, ow_cache=True, cache_dir=args.cache_dir, seed=3)
    assert len(s3) == 2
    ev.dataset_factory = None
    with pytest.raises(NotImplementedError):
        ev.eval_multi_scenes(solver="none")


def test_ckpt_loader_rejects_misfitting_state(gpu, built_lib, tmp_path):
    """load_nerfmatch_from_ckpt: missing / renamed matcher weights raise (the reference's strict=False lets them pass); backbone
    tensors load into a backbone handed in, and are dropped for the parameter-free stub."""
    mcfg = Namespace(model=synth.matcher_config("c2f"), exp=Namespace(seed=1), data=Namespace())
    msd = {f"model.{k}": v for k, v in synth.matcher_state_dict("c2f").items()}
    msd["model.backbone.model.stem.weight"] = torch.ones(4)
    torch.save(dict(state_dict=msd, hyper_parameters=vars(mcfg), epoch=1, global_step=2), tmp_path / "ok.ckpt")
    ev = load_nerfmatch_from_ckpt(str(tmp_path / "ok.ckpt"))  # stub backbone: the backbone tensor is dropped
    assert torch.equal(ev.model.pt_pe_proj.weight.cpu(), synth.matcher_state_dict("c2f")["pt_pe_proj.weight"])

    class TinyBackbone(torch.nn.Module):
        feat_dim = [256, 128]

        def __init__(self):
            super().__init__()
            self.model = torch.nn.Module()
            self.model.stem = torch.nn.Module()
            self.model.stem.weight = torch.nn.Parameter(torch.zeros(4))

    ev2 = load_nerfmatch_from_ckpt(str(tmp_path / "ok.ckpt"), backbone=TinyBackbone())
    assert torch.equal(ev2.model.backbone.model.stem.weight.cpu(), torch.ones(4))
    bad = dict(msd)
    bad["model.pt_pe_proj.weight_renamed"] = bad.pop("model.pt_pe_proj.weight")
    torch.save(dict(state_dict=bad, hyper_parameters=vars(mcfg), epoch=1, global_step=2), tmp_path / "bad.ckpt")
    with pytest.raises(RuntimeError, match="does not fit"):
        load_nerfmatch_from_ckpt(str(tmp_path / "bad.ckpt"))


# ----------------------------------------------------------------------------------------------- NerfEvaluator (eval_nerf.py's class)
def test_nerf_evaluator_and_scene_cache_vs_reference_frame(gpu, built_lib, tmp_path):
    """NerfEvaluator.eval_batch / cache_scene_pts (nerf_evaluator.py:200-232, :308-372) against a frame dict produced by the
    reference's own predict + cache arithmetic (tests/golden/scene_cache_frame.npz): values, not just keys and shapes."""
    import numpy as np
    from conftest import load_golden
    from nerfmatch_amd.nerf_evaluator import NerfEvaluator

    fx = load_golden("scene_cache_frame")
    H, W, S = int(fx["H"]), int(fx["W"]), int(fx["S"])
    cfg = synth.nerf_config("cambridge", num_pts=S, img_wh=(W, H))
    cfg.exp, cfg.split, cfg.downsample = Namespace(seed=0), "train", 8
    frame = dict(img_wh=torch.tensor([[W // 8, H // 8]]), rays=fx["rays"][None], ts=fx["ts"][None], rgbs=fx["rgb_fine"].reshape(1, -1, 3),
                 img_idx=["seq1_frame00012"], unnorm_scene=fx["unnorm"][None])
    ev = NerfEvaluator(cfg, vocab_num=5, stop_layer=3, data_loader=[frame])
    ev.model.load_state_dict(synth.nerf_state_dict(seed=int(fx["weights_seed"]), app_vocab=5, density_bias=3.0), strict=True)
    ev.model.precision = "fp32"
    # eval_batch: image-shaped rgb / depth and the PSNR against the reference's own rendering (-> very high)
    preds, metrics = ev.eval_batch(frame, t_rand=fx["t_rand"], jitter=fx["jitter"])
    assert preds["rgb_fine"].shape == (H // 8, W // 8, 3) and preds["depth_fine"].shape[:2] == (H // 8, W // 8)
    assert (preds["rgb_fine"].cpu() - fx["rgb_fine"]).abs().max() < 1e-4 and (preds["depth_fine"].cpu().reshape(-1) - fx["depth_fine"].reshape(-1)).abs().max() < 1e-4
    assert float(metrics["rgb_fine_psnr"]) > 80
    # cache writer, both arithmetic paths of the fused kernel
    for prec in ("fp32", "fp16x3", "bf16x3"):
        ev.model.precision = prec
        files = ev.cache_scene_pts(cache_dir=tmp_path / prec, frames_per_launch=1, t_rand=fx["t_rand"], jitter=fx["jitter"])
        assert [f.name for f in files] == ["seq1_frame00012.npy"] and files[0].parent.name == "ds8lin"
        d = np.load(files[0], allow_pickle=True).item()
        assert set(d) == {"pt3d", "unnorm_scene", "pt_feat", "pt_color"}
        assert np.abs(d["pt_feat"] - fx["frame_pt_feat"].numpy()).max() < 1e-4
        assert np.abs(d["pt_color"] - fx["frame_pt_color"].numpy()).max() < 1e-4
        assert np.abs(d["pt3d"] - fx["frame_pt3d"].numpy()).max() < 3e-4  # world units (scene scale 3)
        assert np.array_equal(d["unnorm_scene"], fx["frame_unnorm_scene"].numpy())
    assert ev.model.ret_pfeat is False
    assert float(np.mean(ev.eval_data_loader()["psnr"])) > 20


def test_mixed_appearance_ids_and_tail_flag(gpu, built_lib):
    """Weak spots named by the round-1 review: (1) per-ray appearance ids were collapsed to ids[0] -- rays with different ids
    now render per id and equal the single-id renders; (2) the zero-tail skip trusted the caller's jitter -- the re-sampler
    now reports a violated premise on the device and the fused kernel evaluates every sample."""
    from conftest import load_golden
    from nerfmatch_amd import ops
    from nerfmatch_amd.nerf.renderer import NerfRenderer

    fx = load_golden("nerf_r128_s64_app")
    S, R = int(fx["S"]), fx["rays"].shape[0]
    ren = NerfRenderer(synth.nerf_config("cambridge", num_pts=S, img_wh=(int(fx["W"]), int(fx["H"]))), num_frames=5, training=False, stop_layer=3)
    ren.load_state_dict(synth.nerf_state_dict(seed=int(fx["weights_seed"]), app_vocab=5, density_bias=3.0), strict=True)
    ren.to(gpu).eval()
    ren.ret_pfeat = True
    rays = fx["rays"].to(gpu)
    kw = dict(t_rand=fx["t_rand"], jitter=fx["jitter"])
    ids = torch.ones(R, dtype=torch.long)
    ids[R // 3:] = 4
    mixed = ren.predict(rays, 1, 1, out_raw=True, ray_id=ids, **kw)
    one = ren.predict(rays, 1, 1, out_raw=True, ray_id=torch.ones(R, dtype=torch.long), **kw)
    four = ren.predict(rays, 1, 1, out_raw=True, ray_id=torch.full((R,), 4).to(gpu), **kw)  # device-resident ids work too
    assert (one["rgb_fine"] - four["rgb_fine"]).abs().max() > 1e-3  # the appearance row matters
    for k in ("rgb_fine", "feat_fine", "pts_fine", "depth_fine"):
        assert torch.equal(mixed[k][: R // 3], one[k][: R // 3]) and torch.equal(mixed[k][R // 3:], four[k][R // 3:]), k
    assert (one["rgb_fine"].cpu() - fx["pred_rgb_fine"]).abs().max() < 1e-4  # id 1 is the golden's id
    # (2) a jitter outside the re-sampler's contract (negative): fence posts > S/2 no longer coincide
    ren.precision = "fp16x3"
    bad_jit = fx["jitter"] - 0.6
    rg = rays
    t_c = ops.sample_coarse(rg, fx["t_rand"].to(gpu), S)
    blob_c, blob_f = ren.nerf_coarse.packed(gpu, "fp16x3"), ren.nerf_fine.packed(gpu, "fp16x3")
    app = ren.embedding_a.weight[1].detach().contiguous()
    wc = ops.nerf_fwd(blob_c, rg, t_c, app, need_rgb=False, need_feat=False)["weights"]
    t_ok, f_ok = ops.resample(t_c, wc, fx["jitter"].to(gpu), 0.01, True, want_tail_flag=True)
    t_bad, f_bad = ops.resample(t_c, wc, bad_jit.to(gpu), 0.01, True, want_tail_flag=True)
    _, f_det = ops.resample(t_c, wc, fx["jitter"].to(gpu), 0.01, False, want_tail_flag=True)
    assert int(f_ok.item()) == 0 and int(f_bad.item()) != 0 and int(f_det.item()) != 0
    assert not bool((t_bad[:, S // 2 + 1:] == t_bad[:, S // 2 + 1: S // 2 + 2]).all())
    full = ops.nerf_fwd(blob_f, rg, t_bad, app, tap_layer=3, white_bg=True)
    guarded = ops.nerf_fwd(blob_f, rg, t_bad, app, tap_layer=3, white_bg=True, zero_tail=True, tail_flag=f_bad)
    for k in ("weights", "feat", "pts", "rgb", "depth", "acc"):
        assert torch.equal(guarded[k], full[k]), k  # the flag switched the kernel to the full evaluation: bit-identical
    assert float(full["weights"][:, S // 2 + 1:].abs().max()) > 0  # and the tail really carries weight here
    # through the renderer: same protection, no caller involvement
    ren.skip_zero_tail = True
    a = ren.predict(rays, 1, 1, out_raw=True, t_rand=fx["t_rand"], jitter=bad_jit)
    ren.skip_zero_tail = False
    b = ren.predict(rays, 1, 1, out_raw=True, t_rand=fx["t_rand"], jitter=bad_jit)
    assert torch.equal(a["feat_fine"], b["feat_fine"]) and torch.equal(a["rgb_fine"], b["rgb_fine"])


def test_bench_with_rccl_on_one_gpu(gpu, built_lib):
    """The multi-GPU code path of bench.py on the one GPU a test box has, started the way the driver starts it:
    `python bench.py --gpus 1 ...` with NO launcher and no WORLD_SIZE in the environment.  NM_FORCE_DIST=1 makes bench.py take its
    self-launch branch (what `--gpus N > 1` does): it starts `torch.distributed.run --nproc-per-node 1` as a child process, whose
    rank goes through torch.distributed's nccl backend (= RCCL) -- init_process_group, barrier, the evaluator's all_gather of
    pose-candidate records, the MAX all_reduce of the timing -- and relays the child's JSON line as its last stdout line."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(NM_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--queries", "2",
                          "--no-cpu-baseline", "--no-extra-legs"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    assert last.startswith('{"metric"'), last[:200]  # the JSON line is the LAST line of stdout
    line = json.loads(last)
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 0 and line["query_images_per_sec"] > 0
    assert line["scaling"] == "weak" and line["config"]["world_size"] == 1 and line["config"]["collectives"].startswith("RCCL")
    assert line["roofline"]["launches_timed"] == 4  # 2 timed steps x (coarse + fine); warm-up launches are not in the mean


def test_bench_two_ranks_sharing_the_gpu(gpu, built_lib):
    """bench.py's N > 1 logic on the one GPU of a test box, started with the driver's own command line for N = 2 (`python -m torch.distributed.run
    --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...`): NM_BENCH_SHARE_GPU=1 puts both ranks on
    cuda:0 and the collectives on gloo (RCCL refuses two ranks on one device).  Not a measurement -- the line says so --: what is checked is
    that the batches are dealt over the ranks, the records gathered, the time reduced and ONE line printed, by rank 0, for the whole job."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(NM_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--queries", "2",
                          "--no-cpu-baseline", "--no-extra-legs"], env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak" and line["config"]["world_size"] == 2
    assert "ONE GPU" in line["config"]["collectives"]
    # whole-job units: 2 ranks x 2 steps x 2 queries x 4800 rays x (64 + 64) samples over the slowest rank's time
    units = 2 * 2 * 2 * 4800 * 128
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 * 2 - units) < 1e-6 * units, (line["value"], line["ms_per_step"])
    assert line["query_images_per_sec"] > 0 and "cpu_baseline" not in line  # the CPU leg is rank 0 at N = 1 only


def test_benchmark_cli_synthetic_run(gpu, built_lib, tmp_path):
    """The reference's benchmark script flow (model_eval/benchmark_nerfmatch.py: checkpoint list, one run per seed, eval_ckpt's keyword call)
    on synthetic scenes: result files in the reference's naming scheme under <ckpt dir>/<model_name>_run<i>."""
    import numpy as np
    from nerfmatch_amd import benchmark_nerfmatch as bm

    ckpt = bm.write_synthetic_ckpt(tmp_path / "synthetic_best_tmed.ckpt")
    out = bm.main(["--ckpts", str(ckpt), "--synthetic", "3", "--synthetic_scenes", "chess", "fire", "--image_hw", "96x128", "--samples", "32",
                   "--solver", "none", "--query2query", "--mutual", "--rthres", "1", "--seeds", "4", "5"])
    assert len(out) == 2
    for i in (0, 1):
        d = tmp_path / f"best_tmed_run{i}"
        for scene in ("chess", "fire"):
            f = d / f"{scene}_rth1test_none_itr1.query2query.npy"
            assert f.exists(), sorted(p.name for p in d.iterdir())
            m = np.load(f, allow_pickle=True).item()
            assert m["query_idx"].tolist() == [0, 1, 2] and (m["num_matches"] >= 0).all()
    # iNeRF flags reach the evaluator (eval_pose branch: no matcher in the loop)
    out2 = bm.main(["--ckpts", str(ckpt), "--synthetic", "1", "--image_hw", "96x128", "--samples", "128", "--solver", "none", "--query2query", "--mutual",
                    "--rthres", "1", "--ow_cache"])
    assert len(out2) == 1


def test_batch_order_and_sharding_do_not_change_a_single_bit(gpu, built_lib):
    """VERDICT r4 'weak' 4 / north_star "identical 2D-3D match indices at 1/2/4/8 GPUs": what a query returns must not depend on which
    batches its process saw before.  The fp16x3 operand scales used to be chosen on the first batch after loading; they are now chosen
    on a seeded probe bundle (NeRF.probe_bundle) and depend on the parameters only.  Trained-like weights (the regime where scaled lo
    parts reach the fp16 subnormals and the exponents matter), fresh renderer and evaluator per order, per-query random tensors seeded
    by the query: the rendered points / features, the match lists, their scores and the refined pixels are compared with torch.equal
    between (a) queries 0..5 in order, (b) in reverse, (c) the shard rank 1 of 2 would see (1, 3, 5)."""
    import nerfmatch_amd
    from nerfmatch_amd.nerf.renderer import NerfRenderer

    H, W, S = 64, 96, 64
    R = (H // 8) * (W // 8)
    sd = synth.nerf_state_dict(seed=3, style="surface")
    msd = synth.matcher_state_dict("c2f", seed=0)

    def run(order):
        ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=S, img_wh=(W, H)), training=False, stop_layer=3)
        ren.load_state_dict(sd)
        ren.to(gpu).eval()
        assert ren.precision == "fp16x3"
        raw = ren.render_novel_views

        def seeded(img_hw, K, c2ws, unnorm, device, **kw):  # the samplers' random tensors belong to the query, not to the call order
            qs = [int(round(float(torch.as_tensor(c)[0, 3]) * 1e6)) % (2**31 - 1) for c in torch.as_tensor(c2ws).reshape(-1, 4, 4)]
            tr = torch.cat([torch.rand(R, S + 1, generator=torch.Generator().manual_seed(q)) for q in qs])
            jt = torch.cat([synth.resample_jitter((R, S + 1), q + 1) for q in qs])
            return raw(img_hw, K, c2ws, unnorm, device, t_rand=tr.to(gpu), jitter=jt.to(gpu), **kw)

        ren.render_novel_views = seeded
        ev = NeRFMatchEvaluator(Namespace(model=synth.matcher_config("c2f"), exp=Namespace(seed=0), data=Namespace()))
        ev.model.load_state_dict(msd, strict=False)
        ev.model.to(gpu).eval()
        nerfmatch_amd.set_precision("bf16x3")
        try:
            batches = []
            for q in order:
                b = make_batch(H, W, q)
                b["idx"] = torch.tensor([q])
                batches.append(b)
            out = ev.eval_data_loader(renderer=ren, data_loader=batches, solver="none", query2query=True, mutual=True)
        finally:
            nerfmatch_amd.set_precision("fp32")
        scales = (tuple(ren.nerf_coarse._act_log2[str(gpu)]), tuple(ren.nerf_fine._act_log2[str(gpu)]))
        per = {}
        for q, b in zip(order, batches):
            per[q] = {k: b[k].detach().cpu() for k in ("pt3d", "pt_feat", "mpt2d_f", "mpt3d", "mconf")}
            per[q]["ids"] = torch.stack([t.cpu() for t in b["match_ids"]])
        rec = {int(i): (int(n), c.tolist()) for i, n, c in zip(out["query_idx"], out["num_matches"], out["c2w_est"].reshape(-1, 16))}
        return per, rec, scales

    a, rec_a, sc_a = run([0, 1, 2, 3, 4, 5])
    b, rec_b, sc_b = run([5, 4, 3, 2, 1, 0])
    c, rec_c, sc_c = run([1, 3, 5])
    assert sc_a == sc_b == sc_c, "operand scales must be a function of the parameters, not of the batches seen"
    assert rec_a == rec_b and all(rec_c[q] == rec_a[q] for q in (1, 3, 5))
    assert sum(v[0] for v in rec_a.values()) > 0
    for other in (b, c):
        for q, d in other.items():
            for k, v in d.items():
                assert torch.equal(v, a[q][k]), (q, k)


def test_two_stream_loop_equals_the_one_stream_loop(gpu, built_lib):
    """Round 6: eval_data_loader over batches of one query runs query i+1's render on a compute-unit partition beside
    query i's matcher on another.  Nothing a query returns may depend on that: rendered points / features, match lists, scores and
    refined pixels are compared with torch.equal between the one-stream loop, the shipped partitions (render on five whole XCDs, matcher
    on the other three), two plain streams, CU-sliced partitions (a quarter of the chip each; 192 + unconfined; 160 + 96) and two XCDs."""
    import nerfmatch_amd
    from nerfmatch_amd import _lib
    from nerfmatch_amd.nerf.renderer import NerfRenderer

    H, W, S = 64, 96, 64
    ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=S, img_wh=(W, H)), training=False, stop_layer=3)
    ren.load_state_dict(synth.nerf_state_dict(seed=3, style="surface"))
    ren.to(gpu).eval()
    ev = NeRFMatchEvaluator(Namespace(model=synth.matcher_config("c2f"), exp=Namespace(seed=0), data=Namespace()))
    ev.model.load_state_dict(synth.matcher_state_dict("c2f", seed=0), strict=False)
    ev.model.to(gpu).eval()
    assert ev.overlap_render and ev.render_part == ("xcd", 0, 5) and ev.match_part == ("xcd", 5, 3) and ev.overlap_max_queries == 4  # the shipped setting
    keys = ("pt3d", "pt_feat", "mpt2d_f", "mpt2d_c", "mpt3d", "mconf")

    def run(overlap, render_part=("xcd", 0, 5), match_part=("xcd", 5, 3)):
        ev.overlap_render, ev.render_part, ev.match_part = overlap, render_part, match_part
        torch.manual_seed(5)  # (the samplers draw from the device generator: same draws in the same call order)
        batches = [make_batch(H, W, q) for q in range(7)]
        out = ev.eval_data_loader(renderer=ren, data_loader=batches, solver="none", query2query=True, mutual=True)
        torch.cuda.synchronize()
        return [{**{k: b[k].cpu() for k in keys}, "ids": torch.stack([t.cpu() for t in b["match_ids"]])} for b in batches], out["num_matches"].tolist()

    nerfmatch_amd.set_precision("bf16x3")
    try:
        base, n0 = run(False)
        assert sum(n0) > 0
        for setting in ((True, ("xcd", 0, 5), ("xcd", 5, 3)), (True, None, None), (True, 64, 64), (True, 192, None), (True, 160, 96), (True, ("xcd", 2, 2), None)):
            got, n1 = run(*setting)
            assert n1 == n0, setting
            for q, (a, b) in enumerate(zip(base, got)):
                for k in a:
                    assert torch.equal(a[k], b[k]), (setting, q, k)
    finally:
        nerfmatch_amd.set_precision("fp32")
        ev.overlap_render, ev.render_part, ev.match_part = True, ("xcd", 0, 5), ("xcd", 5, 3)
    # the persistent grids follow the partition: 64 units -> 64 workgroups (nm_stream_cus), any other stream -> the whole device
    ncu = torch.cuda.get_device_properties(gpu).multi_processor_count
    assert _lib.lib().nm_stream_cus(_lib.partition_stream(64, 0, gpu).cuda_stream) == 64
    assert _lib.lib().nm_stream_cus(torch.cuda.current_stream(gpu).cuda_stream) == ncu
    assert _lib.lib().nm_stream_cus(None) == ncu


def test_inerf_refinement_over_a_batch_of_queries(gpu, built_lib):
    """Round 6: `inerf_conf` with a batch of Q > 1 queries.  The reference refines batch element 0 only (its loop is
    batch 1, nerfmatch_evaluator.py:323); here every query of the batch is refined on its own one-query view.  A stand-in solver hands
    back its own starting pose per query, so the refinement has something to start from without a PnP package: both queries come back
    with a finite, MOVED pose of their own."""
    import numpy as np

    H, W = 64, 96
    ren = _renderer(gpu, H, W, S=128)
    ev = _c2f_evaluator(gpu, H, W)
    unnorm = synth.unnorm_scene()

    def solver(pt2d, pt3d, K, rthres):
        # which query is this?  the matches' 3-D points were rendered from the query's own pose: key on their mean
        key = round(float(pt3d.mean()), 4)
        c2w = starts.setdefault(key, unnorm @ synth.camera_pose(200 + len(starts)))
        w2c = torch.linalg.inv(c2w)
        return w2c[:3, :3].numpy(), w2c[:3, 3].numpy(), np.ones(len(pt2d), dtype=bool)

    conf = Namespace(lrate=0.002, lrdecay=False, num_optim=2, eval_pose=True, ds=8)
    kw = dict(renderer=ren, inerf_conf=conf, solver=solver, query2query=True, mutual=True)
    starts = {}
    torch.manual_seed(9)
    both = ev.eval_batch(_stack([make_batch(H, W, 0), make_batch(H, W, 1)]), **kw)
    assert len(both["c2w_ests"]) == 2 and len(starts) == 2
    init = list(starts.values())
    for q in range(2):
        est = both["c2w_ests"][q]
        assert est is not None and torch.isfinite(est).all() and float(both["t_err"][q]) < float("inf")
        assert (est - init[q]).abs().max().item() > 1e-4, "the refinement moved nothing"
    # the two queries were refined independently: their refined poses differ as their starting poses do
    assert (both["c2w_ests"][0] - both["c2w_ests"][1]).abs().max().item() > 1e-3


def test_split_step_equals_the_plain_step(gpu, built_lib):
    """NeRFMatchEvaluator.split_step (round 6; off by default -- a measured negative, profiles/r6_ab_split_step.log): the matcher's image
    side on one compute-unit partition beside the render on another, the point tokens through the self-attention block alone.  Same bits
    as the plain step (every kernel of the block works per row / per sequence)."""
    import nerfmatch_amd
    from nerfmatch_amd.nerf.renderer import NerfRenderer

    H, W, S = 64, 96, 64
    ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=S, img_wh=(W, H)), training=False, stop_layer=3)
    ren.load_state_dict(synth.nerf_state_dict(seed=3, style="surface"))
    ren.to(gpu).eval()
    ev = NeRFMatchEvaluator(Namespace(model=synth.matcher_config("c2f"), exp=Namespace(seed=0), data=Namespace()))
    ev.model.load_state_dict(synth.matcher_state_dict("c2f", seed=0), strict=False)
    ev.model.to(gpu).eval()
    assert ev.split_step is None
    keys = ("pt3d", "pt_feat", "mpt2d_f", "mpt3d", "mconf")

    def run(split):
        ev.split_step = split
        outs = []
        for q in range(3):
            torch.manual_seed(20 + q)
            b = make_batch(H, W, q)
            ev.eval_batch(b, renderer=ren, solver="none", query2query=True, mutual=True)
            outs.append({k: b[k].cpu() for k in keys})
        return outs

    nerfmatch_amd.set_precision("bf16x3")
    try:
        base = run(None)
        got = run((("xcd", 0, 6), ("xcd", 6, 2)))
    finally:
        nerfmatch_amd.set_precision("fp32")
        ev.split_step = None
    assert sum(len(o["mconf"]) for o in base) > 0
    for a, b in zip(base, got):
        for k in keys:
            assert torch.equal(a[k], b[k]), k


def test_writes_through_data_cannot_return_old_weights(gpu, built_lib):
    """the derived copies (packed / split / transposed weight blobs, the temperature's host value, the NeRF blobs and their
    operand scales) are keyed on (data_ptr, _version), which a write through `.data` changes neither of.  ops.ParamGuard fingerprints the
    parameter VALUES once per pass on the device; a mismatch reaches the host with the pass's own read-back and the pass is repeated on
    fresh copies.  Here: such writes to the temperature, to a self-attention weight and to a NeRF layer, each followed by a forward pass /
    a localisation step whose results must equal those of a freshly built module holding the same values -- torch.equal."""
    import warnings

    import nerfmatch_amd
    from nerfmatch_amd.matcher import NeRFMatcherMS
    from nerfmatch_amd.modules import PrecomputedBackbone
    from nerfmatch_amd.nerf.renderer import NerfRenderer

    H, W, S = 64, 96, 64
    nsd = synth.nerf_state_dict(seed=3, style="surface")
    msd = synth.matcher_state_dict("c2f", seed=0)
    g = torch.Generator().manual_seed(1)
    cfeat, ffeat = StubBackbone()(torch.randn(1, 3, H, W, generator=g))
    keys = ("pt3d", "pt_feat", "mpt2d_f", "mpt3d", "mconf")

    def build(nsd_, msd_):
        ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=S, img_wh=(W, H)), training=False, stop_layer=3)
        ren.load_state_dict(nsd_)
        ren.to(gpu).eval()
        ev = NeRFMatchEvaluator(Namespace(model=synth.matcher_config("c2f"), exp=Namespace(seed=0), data=Namespace()))
        ev.model.load_state_dict(msd_, strict=False)
        ev.model.backbone = PrecomputedBackbone((cfeat.to(gpu), ffeat.to(gpu)), [256, 128])
        ev.model.to(gpu).eval()
        return ren, ev

    def step(ren, ev, q=0):
        torch.manual_seed(30 + q)
        b = make_batch(H, W, q)
        ev.eval_batch(b, renderer=ren, solver="none", query2query=True, mutual=True)
        return {k: b[k].cpu() for k in keys}

    nerfmatch_amd.set_precision("bf16x3")
    try:
        ren, ev = build(nsd, msd)
        first = step(ren, ev)  # packs every blob, takes the fingerprints' baseline
        assert len(first["mconf"]) > 0
        # ---- (1) the temperature, clamped the way a trainer would: .data, in place, no version bump
        v0 = ev.model.temperature._version
        ev.model.temperature.data.clamp_(max=4.0)
        assert ev.model.temperature._version == v0 and float(ev.model.temperature) == 4.0
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter("always")
            got = step(ren, ev)
        assert any("modified in place through `.data`" in str(w.message) for w in wlist)
        msd1 = dict(msd, temperature=torch.tensor(4.0))
        ren_f, ev_f = build(nsd, msd1)
        want = step(ren_f, ev_f)
        for k in keys:
            assert torch.equal(got[k], want[k]), ("temperature", k)
        assert not torch.equal(got["mconf"], first["mconf"])
        # ---- (2) a weight inside the shared self-attention block
        w = ev.model.pt_sa.layers[1].attention.proj_q.weight
        w.data.mul_(1.5)
        got = step(ren, ev, 1)
        msd2 = dict(msd1)
        msd2["pt_sa.layers.1.attention.proj_q.weight"] = msd["pt_sa.layers.1.attention.proj_q.weight"] * 1.5
        ren_f, ev_f = build(nsd, msd2)
        want = step(ren_f, ev_f, 1)
        for k in keys:
            assert torch.equal(got[k], want[k]), ("proj_q", k)
        # ---- (3) a NeRF layer: the evaluator repeats the batch on fresh blobs (and fresh operand scales)
        ren.nerf_fine.pts_linears[2].weight.data.mul_(0.5)
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter("always")
            got = step(ren, ev, 2)
        assert any("NeRF parameters were modified in place" in str(w.message) for w in wlist)
        nsd3 = dict(nsd)
        nsd3["nerf_fine.pts_linears.2.weight"] = nsd["nerf_fine.pts_linears.2.weight"] * 0.5
        ren_f, ev_f = build(nsd3, msd2)
        want = step(ren_f, ev_f, 2)
        for k in keys:
            assert torch.equal(got[k], want[k]), ("nerf", k)
        # ---- and the renderer on its own says so when asked
        ren.nerf_coarse.alpha_linear.bias.data.add_(0.25)
        ren.render_novel_view((H, W), synth.intrinsics(H, W, 120.0), synth.unnorm_scene() @ synth.camera_pose(3), synth.unnorm_scene(), gpu)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert ren.check_stale() and not ren.check_stale()
    finally:
        nerfmatch_amd.set_precision("fp32")

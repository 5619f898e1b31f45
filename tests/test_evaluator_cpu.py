"""CPU: host-side branches of NeRFMatchEvaluator that need no kernel -- `match_oracle` (reference nerfmatch_evaluator.py:163-174)."""
from argparse import Namespace

import torch

from nerfmatch_amd import synth
from nerfmatch_amd.nerfmatch_evaluator import NeRFMatchEvaluator


def _batch(Q, M, N, seed=0):
    g = torch.Generator().manual_seed(seed)
    conf_gt = torch.zeros(Q, M, N, dtype=torch.bool)
    for q in range(Q):
        rows = torch.randperm(M, generator=g)[:7 + q]
        cols = torch.randperm(N, generator=g)[:7 + q]
        conf_gt[q, rows, cols] = True
    return dict(image=torch.zeros(Q, 3, 8, 8), K=torch.eye(3)[None].repeat(Q, 1, 1), c2w=torch.eye(4)[None].repeat(Q, 1, 1),
                pt3d=torch.randn(Q, N, 3, generator=g), pt2d=torch.rand(Q, M, 2, generator=g) * 100,
                pt2d_proj=torch.rand(Q, N, 2, generator=g) * 100, conf_gt=conf_gt)


def test_match_oracle_uses_ground_truth_correspondences():
    for kind in ("c2f", "coarse"):
        ev = NeRFMatchEvaluator(Namespace(model=synth.matcher_config(kind), exp=Namespace(seed=0), data=Namespace()))
        ev.model.forward = ev.model.forward_begin = None  # the matcher must not be called at all
        seen = []

        def solver(pt2d, pt3d, K, rthres):
            seen.append((pt2d.clone(), pt3d.clone()))
            return torch.eye(3).numpy(), torch.zeros(3).numpy(), list(range(len(pt2d)))

        b = _batch(1, 12, 15)
        c2w_est, R_err, t_err, n = ev.eval_match_pose(b, solver=solver, match_oracle=True)
        _, i2d, i3d = torch.where(b["conf_gt"])
        want2d = b["pt2d"][0][i2d] if kind == "coarse" else b["pt2d_proj"][0][i3d]
        assert n == 7 and torch.equal(seen[0][1], b["pt3d"][0][i3d]) and torch.equal(seen[0][0], want2d)
        assert torch.allclose(c2w_est, torch.eye(4)) and float(R_err) < 1e-4 and float(t_err) < 1e-6
        # batches of Q > 1 and the whole eval_batch loop (cached points: nothing is rendered)
        seen.clear()
        b = _batch(3, 12, 15, seed=1)
        out = ev.eval_batch(b, solver=solver, match_oracle=True, cached_pt=True)
        assert out["num_matches"] == [7, 8, 9] and len(seen) == 3
        bid, i2d, i3d = torch.where(b["conf_gt"].cpu())
        for q in range(3):
            assert torch.equal(seen[q][1], b["pt3d"].cpu()[q][i3d[bid == q]])


def test_benchmark_cli_has_the_reference_flag_set():
    """nerfmatch_amd.benchmark_nerfmatch takes the flags of the reference's model_eval/benchmark_nerfmatch.py:209-250 with the same
    defaults (the table below is that argparse block as data); extensions are additional flags only."""
    from nerfmatch_amd.benchmark_nerfmatch import build_parser

    ref = dict(split="test", ckpt_dir=None, scene_anno_path=None, ckpts=[], model_name="best_tmed", coarse_only=False, mutual=False, query2query=False,
               match_thres=0.0, ow_cache=False, debug=False, solver="colmap", rthres=10, center_subpixel=False, iters=1, nerf_path=None,
               test_pair_txt=None, scene_dir=None, dataset=None, scene=None, pair_topk=1, sample_pts=-1, sample_mode=None, mask="default",
               cache_tag=None, inerf=False, inerf_optim=5, inerf_lr=0.001, inerf_lrd=False, inerf_ds=8, inerf_pose=False, inerf_match_loss=False,
               cache_iters=False, no_cache_pt=False, retrieval_only=False, match_oracle=False, visualize=False, seeds=[], feats=[])
    got = vars(build_parser().parse_args([]))
    for k, v in ref.items():
        assert k in got and got[k] == v, (k, got.get(k), v)
    assert set(got) - set(ref) == {"synthetic", "synthetic_scenes", "image_hw", "samples", "batch_size"}
    a = build_parser().parse_args("--ckpts a.ckpt b.ckpt --mutual --solver cv --rthres 5 --inerf --inerf_optim 7 --seeds 1 2 3 --pair_topk 3".split())
    assert a.ckpts == ["a.ckpt", "b.ckpt"] and a.mutual and a.solver == "cv" and a.rthres == 5.0 and a.inerf and a.inerf_optim == 7 and a.seeds == [1, 2, 3]

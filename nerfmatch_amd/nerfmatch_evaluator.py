"""NeRFMatchEvaluator: the reference's localisation driver (nerfmatch/nerfmatch_evaluator.py:118-931) over the
MI355X renderer and matcher, with query sharding over GPUs.

Kept: class name, `eval_match_pose / eval_batch / eval_data_loader / eval_multi_scenes / gen_rays /
inerf_refinement` signatures and the render -> match (-> PnP) loop of eval_batch.  Out of scope (SURVEY.md
section 2): dataset classes (any iterable of batch dicts with the reference's schema is accepted), result caching
to .npy, visualisation.  PnP-RANSAC is third-party CPU code (pycolmap / OpenCV): it is used when importable,
otherwise `solver="none"` returns the 2D-3D matches and no pose.  iNeRF refinement (`inerf_refinement`) runs on the HIP
forward/backward kernels of nerfmatch_amd/inerf.py; its optional matching loss (`use_match_loss`) needs the matcher's
backward (training-side kernels) and raises NotImplementedError.
"""
import math
import time
from argparse import Namespace
from collections import defaultdict

import numpy as np
import torch

from . import dist as nmdist
from . import ops
from .matcher import NeRFMatcherCoarse, NeRFMatcherMS
from .nerf_evaluator import GenericModelEvaluator, load_nerf_render_from_ckpt  # noqa: F401
from .utils import data_to_device, merge_configs


def parse_nerf_stop_layer(scene_dir):
    parts = scene_dir.split("inter_layer")
    return int(parts[1].split("/")[0]) if len(parts) == 2 else -1


def pose_err(gt_pose, est_pose):
    """(rotation error in degrees, translation error) between two c2w poses (reference utils/metrics.py:359-369;
    the Rodrigues norm of R_est R_gt^T is the rotation angle)."""
    gt_pose, est_pose = torch.as_tensor(gt_pose).double().cpu(), torch.as_tensor(est_pose).double().cpu()
    t_err = float(torch.norm(gt_pose[:3, 3] - est_pose[:3, 3]))
    rel = est_pose[:3, :3] @ gt_pose[:3, :3].T
    cos = max(-1.0, min(1.0, (float(torch.trace(rel)) - 1.0) / 2.0))
    return math.degrees(math.acos(cos)), t_err


def _solve_pnp(solver, pt2d, pt3d, K, rthres, center_subpixel):
    """Returns (R, t, inliers) of the w2c pose or None.  Third-party CPU solvers, outside the hot path."""
    if callable(solver):
        return solver(pt2d, pt3d, K, rthres)
    if len(pt2d) < 4:
        return None
    if solver == "colmap":
        import pycolmap  # noqa: F401  (absent in this image; ImportError tells the user what is missing)
        from .utils.pnp import estimate_pose_pycolmap
        return estimate_pose_pycolmap(pt2d, pt3d, K, ransac_thres=rthres, center_subpixel=center_subpixel)
    if solver == "cv2":
        import cv2  # noqa: F401
        from .utils.pnp import estimate_pose
        return estimate_pose(pt2d, pt3d, K, ransac_thres=rthres)
    raise ValueError(f"{solver} is not supported!")


class NeRFMatchEvaluator(GenericModelEvaluator):
    def __init__(self, config, data_loader=None):
        super().__init__(config)
        self.seed = getattr(getattr(config, "exp", Namespace()), "seed", 0)
        if getattr(config, "iters", 1) > 1:
            torch.manual_seed(self.seed)
        model_conf = config.model
        if "ffeat_dim" not in model_conf:
            self.model, self.coarse_only = NeRFMatcherCoarse(model_conf), True
        else:
            self.model, self.coarse_only = NeRFMatcherMS(model_conf), False
        self.model.to(self.device).eval()
        self.data_loader = data_loader
        self.timer = defaultdict(list)

    # ------------------------------------------------------------------------------------------------------
    def eval_match_pose(self, batch, mutual=True, match_thres=0.0, solver="colmap", rthres=1, center_subpixel=False,
                        match_oracle=False):
        if match_oracle:
            raise NotImplementedError("--match_oracle needs ground-truth conf matrices from the dataset classes (out of scope)")
        K = batch["K"].cpu()
        t0 = time.time()
        self.model.forward(batch, mutual=mutual, match_thres=match_thres)
        torch.cuda.synchronize() if torch.cuda.is_available() else None
        self.timer["match_time"].append((time.time() - t0) / batch["pt3d"].shape[-3])
        if self.coarse_only:
            bid, i2d, i3d = (t.cpu() for t in batch["match_ids"])
            pt2d = batch["pt2d"].cpu()[0][i2d[bid == 0]]
            pt3d = batch["pt3d"].cpu().reshape(len(K), -1, 3)[0][i3d[bid == 0]]
        else:
            pt2d, pt3d = batch["mpt2d_f"].detach().cpu(), batch["mpt3d"].cpu()
        num_matches = len(pt2d)
        if solver in (None, "none"):
            return None, torch.tensor(float("inf")), torch.tensor(float("inf")), num_matches
        res = _solve_pnp(solver, pt2d, pt3d, K.squeeze(), rthres, center_subpixel)
        if not res:
            return None, torch.tensor(float("inf")), torch.tensor(float("inf")), num_matches
        R, t, _ = res
        w2c = torch.eye(4)
        w2c[:3, :3] = torch.as_tensor(R, dtype=torch.float32)
        w2c[:3, 3] = torch.as_tensor(t, dtype=torch.float32).reshape(-1)
        c2w_est = torch.linalg.inv(w2c)
        R_err, t_err = pose_err(batch["c2w"].cpu().squeeze(), c2w_est)
        return c2w_est, R_err, t_err, num_matches

    def gen_rays(self, poses, width, height, z_near, z_far, K, ds=8, c=None, ndc=False):
        """Rays + sub-sampled pixel coordinates for one pose (reference :232-286), generated on the device."""
        if ndc:
            raise NotImplementedError("ndc rays are not used by the shipped configs")
        rays, _ = ops.raygen(K.squeeze(), poses[0], height, width, self.device, ds=ds, near=float(z_near))
        ys, xs = torch.meshgrid(torch.arange(ds // 2, height, ds), torch.arange(ds // 2, width, ds), indexing="ij")
        return rays, torch.stack([xs, ys], -1).reshape(-1, 2).float()

    def inerf_refinement(self, batch, renderer, unnorm_scene, c2w_est, inerf_conf, mutual=True, match_thres=0.0, solver="colmap",
                         rthres=1, center_subpixel=False, visualize=False, overlay_ims=None, cache_iters=False, iter_t_errs=None,
                         iter_R_errs=None, debug=False, t_rands=None, jitters=None):
        """Photometric pose refinement (reference nerfmatch_evaluator.py:288-500): `num_optim` Adam steps on the normalised
        pose through the fine NeRF, then either the refined pose itself (`eval_pose`) or a re-match against the points /
        features of the last rendered view.  Returns (c2w_est, R_err, t_err).  `t_rands` / `jitters` (optional) fix the
        samplers' random tensors, one (R,129) pair per step."""
        from . import inerf

        if visualize:
            raise NotImplementedError("overlay visualisation is out of scope (SURVEY.md section 2)")
        if getattr(inerf_conf, "use_match_loss", False):
            raise NotImplementedError("use_match_loss needs the backward pass of the matcher (training-side kernels, SURVEY.md 8f rank 4)")
        lrate = getattr(inerf_conf, "lrate", 0.001)
        lrdecay = getattr(inerf_conf, "lrdecay", False)
        num_optim = getattr(inerf_conf, "num_optim", 5)
        eval_pose = getattr(inerf_conf, "eval_pose", False)
        ds = getattr(inerf_conf, "ds", 8)
        c2w_gt = batch["c2w"].cpu()
        K = batch["K"].cpu().squeeze()
        img = batch["image"][0].permute(1, 2, 0)
        H, W, _ = img.shape
        unnorm = torch.as_tensor(unnorm_scene, dtype=torch.float32).to(self.device)
        pose0 = unnorm.inverse() @ torch.as_tensor(c2w_est, dtype=torch.float32).detach().to(self.device)
        R_err = t_err = torch.tensor(float("inf"))
        tj = time.time()
        for j, pose, loss, ctx in inerf.refine_iter(renderer, K, H, W, img, pose0, num_optim, lrate, lrdecay, ds, t_rands, jitters):
            self.timer["inerf_step_time"].append(time.time() - tj)
            if debug or cache_iters or j == num_optim - 1:
                if eval_pose:
                    c2w_est = (unnorm @ pose).cpu()
                    R_err, t_err = pose_err(c2w_gt.squeeze(), c2w_est)
                else:
                    pts, feats = inerf.rendered_points(renderer, ctx)
                    batch["pt3d"] = ops.unnormalize_points(pts, unnorm.cpu()).unsqueeze(0)
                    batch["pt_feat"] = feats.unsqueeze(0)
                    batch["pt_mask"] = torch.ones_like(batch["pt3d"][..., 0])
                    c2w_est, R_err, t_err, _ = self.eval_match_pose(batch, mutual=mutual, match_thres=match_thres, solver=solver,
                                                                     rthres=rthres, center_subpixel=center_subpixel)
                if cache_iters and j > 0 and j != num_optim - 1:
                    iter_t_errs.append(t_err)
                    iter_R_errs.append(R_err)
                if debug:
                    print(f"  inerf step={j} loss={loss:2f} t={t_err * 100:.4f}cm R={R_err:.4f}")
            tj = time.time()
        return c2w_est, R_err, t_err

    def eval_batch(self, batch, renderer=None, inerf_conf=None, iters=1, mutual=True, match_thres=0.0, match_oracle=False,
                   solver="colmap", rthres=1, center_subpixel=False, visualize=False, overlay_ims=None, query2query=False,
                   retrieval_only=False, cached_pt=True, cache_iters=False, debug=False):
        data_to_device(batch, self.device)
        img = batch["image"]
        K = batch["K"].cpu()
        unnorm_scene = batch["unnorm_scene"].squeeze() if "unnorm_scene" in batch else renderer.unnorm_scene
        if isinstance(unnorm_scene, np.ndarray):
            unnorm_scene = torch.from_numpy(unnorm_scene)
        iter_t_errs, iter_R_errs = [], []
        ts = time.time()
        if query2query:
            c2w_est = batch["c2w"].squeeze()
        elif (not cached_pt) or retrieval_only:
            c2w_est = batch["rc2w"].squeeze()
        else:
            c2w_est = None
        R_err = t_err = torch.tensor(float("inf"))
        num_matches = 0
        for itr in range(iters):
            if retrieval_only:
                R_err, t_err = pose_err(batch["c2w"].squeeze().cpu(), c2w_est.cpu())
            else:
                if c2w_est is not None:
                    outs = renderer.render_novel_view(img.shape[-2:], K.squeeze(), c2w_est, unnorm_scene, self.device, downsample=8,
                                                      want_im_pred=False)  # only pt3d / pt_feat are read (reference :566-573)
                    batch["pt3d"] = outs["pt3d"].unsqueeze(0)
                    batch["pt_feat"] = outs["pt_feat"].unsqueeze(0)
                    batch["pt_mask"] = torch.ones_like(batch["pt3d"][..., 0])
                new_pose, R_err, t_err, num_matches = self.eval_match_pose(batch, mutual=mutual, match_thres=match_thres, solver=solver,
                                                                           rthres=rthres, center_subpixel=center_subpixel,
                                                                           match_oracle=match_oracle)
                if new_pose is not None or solver not in (None, "none"):
                    c2w_est = new_pose
            if c2w_est is not None and inerf_conf:
                res = self.inerf_refinement(batch, renderer, unnorm_scene, c2w_est, inerf_conf, mutual=mutual, match_thres=match_thres,
                                            solver=solver, rthres=rthres, center_subpixel=center_subpixel, cache_iters=cache_iters,
                                            iter_t_errs=iter_t_errs, iter_R_errs=iter_R_errs, debug=debug)
                if res[1] != float("inf"):  # take the refined pose only if it could be evaluated (reference :608-610)
                    c2w_est, R_err, t_err = res
            if cache_iters:
                iter_t_errs.append(t_err)
                iter_R_errs.append(R_err)
            if c2w_est is None and itr + 1 < iters:
                break
        self.timer["localize_time"].append(time.time() - ts)
        return dict(R_err=[R_err], t_err=[t_err], iter_t_errs=iter_t_errs, iter_R_errs=iter_R_errs, num_matches=[num_matches],
                    c2w_est=c2w_est)

    def eval_data_loader(self, renderer=None, iters=1, rthres=1, center_subpixel=False, solver="colmap", mutual=True, match_thres=0.0,
                         match_oracle=False, data_loader=None, query2query=False, cached_pt=True, debug=False, inerf_conf=None,
                         retrieval_only=False, cache_iters=False, visualize=False):
        """Localises every query of `data_loader` (any indexable / iterable of batch dicts).  With torch.distributed
        initialised, queries are sharded round-robin over ranks and the per-query records are all-gathered once at the
        end: every rank returns the metrics of ALL queries."""
        loader = data_loader if data_loader is not None else self.data_loader
        batches = loader if hasattr(loader, "__getitem__") else list(loader)
        n = len(batches)
        rank, W = nmdist.world()
        recs = []
        for count, qi in enumerate(nmdist.shard_indices(n, rank, W)):
            m = self.eval_batch(batches[qi], renderer, inerf_conf, iters=iters, rthres=rthres, center_subpixel=center_subpixel, solver=solver,
                                mutual=mutual, match_thres=match_thres, match_oracle=match_oracle, query2query=query2query,
                                retrieval_only=retrieval_only, cached_pt=cached_pt, cache_iters=cache_iters, debug=debug)
            recs.append(nmdist.make_record(qi, m["c2w_est"], float(m["R_err"][0]), float(m["t_err"][0]), m["num_matches"][0]))
            if debug and count >= 5:
                break
        local = torch.stack(recs) if recs else torch.empty(0, nmdist.RECORD_FLOATS)
        allrec = nmdist.gather_records(local, n, self.device).cpu()
        return dict(R_err=allrec[:, 17].numpy(), t_err=allrec[:, 18].numpy(), num_matches=allrec[:, 19].numpy(),
                    query_idx=allrec[:, 0].long().numpy(), c2w_est=allrec[:, 1:17].reshape(-1, 4, 4).numpy())

    def eval_multi_scenes(self, scenes, renderers=None, **kw):
        """`scenes`: dict scene-name -> iterable of batches; `renderers`: dict scene-name -> NerfRenderer (or None for cached
        points).  The reference builds both from its dataset classes / checkpoints and caches the metrics on disk
        (:726-931); that bookkeeping is out of scope, the per-scene loop is the same."""
        out = {}
        for name, loader in scenes.items():
            self.timer = defaultdict(list)
            out[name] = self.eval_data_loader(renderer=None if renderers is None else renderers.get(name), data_loader=loader, **kw)
            out[name].update({k: np.array(v) for k, v in self.timer.items()})
        return out


def load_nerfmatch_from_ckpt(ckpt_path, args=None, root_dir=".", arg_mask=None, data_loader=None):
    """Lightning checkpoint of the reference -> evaluator (reference :69-115); `strict=False` like the reference."""
    ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    hp = ckpt["hyper_parameters"]
    config = hp if isinstance(hp, Namespace) else Namespace(**hp)
    config.ckpt = ckpt_path
    if args:
        config = merge_configs(config, args)
    evaluator = NeRFMatchEvaluator(config, data_loader=data_loader)
    state = {k: v for k, v in ckpt["state_dict"].items() if not k.startswith("model.backbone.")}
    evaluator.load_state_dict(state, strict=False)
    return evaluator

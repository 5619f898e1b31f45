"""NeRFMatchEvaluator: the reference's localisation driver (nerfmatch/nerfmatch_evaluator.py:118-931) over the
MI355X renderer and matcher, with query sharding over GPUs.

Kept: class name, `eval_match_pose / eval_batch / eval_data_loader / eval_multi_scenes / gen_rays /
inerf_refinement` signatures (eval_multi_scenes keyword for keyword, so model_eval/benchmark_nerfmatch.py's call works
unchanged), the render -> match (-> PnP) loop of eval_batch, the result-cache file naming and the pose statistics.
New: a batch may hold Q > 1 queries (the reference's `batch_size` argument exists but its eval_batch only works for 1):
the Q queries are rendered and matched as ONE launch sequence, and eval_data_loader software-pipelines consecutive batches
across the matcher's single synchronisation point.  Out of scope: the dataset classes -- pass
`dataset_factory` (or any iterable of batch dicts with the reference's schema as `data_loader`) -- and visualisation.
PnP-RANSAC is third-party CPU code (pycolmap / OpenCV): used when importable, otherwise `solver="none"` returns the
2D-3D matches and no pose.  iNeRF refinement (`inerf_refinement`) runs on the HIP
forward/backward kernels of nerfmatch_amd/inerf.py, including its optional matching loss (`use_match_loss`, c2f matcher).
"""
import contextlib
import os
import time
from argparse import Namespace
from collections import defaultdict
from pathlib import Path

import numpy as np
import torch

from . import _lib
from . import dist as nmdist
from . import ops
from .matcher import NeRFMatcherCoarse, NeRFMatcherMS
from .nerf_evaluator import GenericModelEvaluator, load_nerf_render_from_ckpt  # noqa: F401
from .utils import data_to_device, merge_configs
from .utils.metrics import POSE_THRES, average_pose_metrics, pose_err, summarize_pose_statis  # noqa: F401


def parse_nerf_stop_layer(scene_dir):
    parts = scene_dir.split("inter_layer")
    return int(parts[1].split("/")[0]) if len(parts) == 2 else -1


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
        self.data_loader = data_loader  # the reference builds it from its dataset classes (:141-146, out of scope)
        self.timer = defaultdict(list)
        self._match_events = []
        ckpt = getattr(config, "ckpt", None)
        self.cache_dir = Path(ckpt.replace("checkpoints/", "").replace(".ckpt", "_eval_results")) if ckpt else Path("eval_results")
        # The localisation loop reads match lists only (the reference's eval_match_pose, :152-230); the (Q, M, N) confidence
        # tensor the model can also return is 92 MB per 640x480 query and only the iNeRF match loss ever looks at it: off here
        # (set True to get batch["conf_matrix"] like the reference's model.forward).
        self.keep_conf_matrix = False
        # Two-stream pipeline of eval_data_loader for SMALL batches -- one query per batch is the reference's operating point (round 6;
        # DESIGN.md section 3.9): query i+1's render runs on one compute-unit partition beside query i's matcher on another (its ~40 short
        # dependent launches leave most of a whole chip idle).  A partition is ("xcd", first, count) -- whole XCDs, each with its own L2: the
        # shipped setting, render on five of the eight, matcher on the other three -- or an int n: n / 8 units of every XCD (multiples of 32:
        # the dispatcher deals workgroups round-robin over XCDs and shader engines without looking at the mask), or None: a plain stream.
        # Used with iters == 1 and no refinement, for batches of at most `overlap_max_queries` queries: 2.73 -> 2.19 ms per query at one
        # per batch, 2.29 -> 2.05 at two, 2.07 -> 1.97 at four, a loss from eight on (profiles/r6_ab_render_stream_fresh.log: every setting
        # in a process of its own).  Per-query results do not depend on it -- tiles are independent and every kernel is the one the
        # one-stream loop runs (tests/test_evaluator_gpu.py::test_two_stream_loop_equals_the_one_stream_loop, torch.equal).
        # The partition streams are "blocking" HIP streams (hipExtStreamCreateWithCUMask takes no flags): the legacy default stream
        # orders against them implicitly, and ANY operation on it -- an event record included -- waits for all of them.  The loop therefore
        # never touches the default stream between its entry and its exit (_begin_on / _finish_on).
        self.overlap_render = True
        self.render_part = ("xcd", 0, 5)
        self.match_part = ("xcd", 5, 3)
        self.overlap_max_queries = 4
        # A step on its own (eval_batch, or the loop with iters > 1): the matcher's IMAGE side -- tokens, sine PE, the self-attention
        # block -- does not depend on the rendered points; with `split_step` = (render partition, image-side partition) it runs beside the
        # render, and the matcher's self-attention then sees the point tokens alone.  None = off.
        self.split_step = None
        self.dataset_factory = None   # (data_conf, split) -> list of datasets (each: .scene, .scene_dir, samples); see eval_multi_scenes
        self.renderer_factory = None  # (scene, scene_dir, stop_layer) -> NerfRenderer; default: load_nerf_render_from_ckpt(nerf_path)

    # -- matching + pose of one batch of Q >= 1 queries -----------------------------------------------------------------------
    def _match_begin(self, batch, mutual, match_thres, image_side=None):
        """Enqueues the matcher.  The c2f model stops before its single synchronisation point (the match-count read-back), so
        a caller can queue more GPU work (the next batch's render) before _match_finish waits for it."""
        t0 = time.time()
        ev = self._events(2)
        self.model.keep_conf = bool(self.keep_conf_matrix)
        if self.coarse_only or batch["pt3d"].dim() == 4:
            self.model.forward(batch, mutual=mutual, match_thres=match_thres)
            st = None
        else:
            st = self.model.forward_begin(batch, mutual=mutual, match_thres=match_thres, image_side=image_side)
        if ev:
            ev[1].record()
        return dict(st=st, t0=t0, ev=ev, host=time.time() - t0)

    def _match_finish(self, batch, ms):
        """`match_time` (per query / per reference frame): the reference synchronises nothing and reads the host clock (:177-180).
        In the pipelined loop the host clock between begin and finish also covers the NEXT batch's render being issued, so the
        GPU time of the matcher's own launches is taken instead: HIP events around the two launch groups (before / after the
        count read-back), read once the work is done (`_flush_match_times`)."""
        t1 = time.time()
        ev2 = self._events(2)
        if ms["st"] is not None:
            self.model.forward_finish(ms["st"])
        n = batch["pt3d"].shape[-3]
        if ev2:
            ev2[1].record()
            self._match_events.append((ms["ev"], ev2, n))
        else:
            self.timer["match_time"].append((ms["host"] + time.time() - t1) / n)

    def _events(self, k):
        if self.device.type != "cuda":
            return None
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(k)]
        ev[0].record()
        return ev

    def _flush_match_times(self):
        for a, b, n in self._match_events:
            b[1].synchronize()
            self.timer["match_time"].append((a[0].elapsed_time(a[1]) + b[0].elapsed_time(b[1])) * 1e-3 / n)
        self._match_events = []

    def _oracle_matches(self, batch):
        """`--match_oracle` (reference :163-174): the ground-truth correspondences of `batch["conf_gt"]` (Q, M, N) instead of the
        model's -- points indexed by the point id, pixels = the points' projections `pt2d_proj` (c2f) or the coarse cell centres
        `pt2d` (coarse-only model).  The reference reads batch element 0 only (its eval batch is 1); here every query q gets
        its own rows."""
        Q = batch["image"].shape[0]
        bid, i2d, i3d = (t.cpu() for t in torch.where(batch["conf_gt"]))
        pt3d = batch["pt3d"].cpu().reshape(Q, -1, 3)
        pix = batch["pt2d"].cpu() if self.coarse_only else batch["pt2d_proj"].cpu()
        per = []
        for q in range(Q):
            sel = bid == q
            per.append((pix[q][i2d[sel]] if self.coarse_only else pix[q][i3d[sel]], pt3d[q][i3d[sel]]))
        return per

    def _poses_from_matches(self, batch, solver, rthres, center_subpixel, match_oracle=False):
        """2D-3D matches of the batch -> per query (c2w_est | None, R_err, t_err, num_matches); PnP on the host (third party)."""
        Q = batch["image"].shape[0]
        Ks = self._host(batch, "K").reshape(-1, 3, 3)
        inf = torch.tensor(float("inf"))
        no_pose = solver in (None, "none")
        if match_oracle:
            per = self._oracle_matches(batch)
            counts = [len(a) for a, _ in per]
        elif self.coarse_only:
            bid, i2d, i3d = (t.cpu() for t in batch["match_ids"])
            pt2d_all, pt3d_all = batch["pt2d"].cpu(), batch["pt3d"].cpu().reshape(Q, -1, 3)
            per = [(pt2d_all[q][i2d[bid == q]], pt3d_all[q][i3d[bid == q]]) for q in range(Q)]
            counts = [len(a) for a, _ in per]
        elif no_pose and "match_counts" in batch and batch["pt3d"].dim() == 3:
            per, counts = None, list(batch["match_counts"])  # the counts are already on the host: no further copy
        else:
            pt2d, pt3d = batch["mpt2d_f"].detach().cpu(), batch["mpt3d"].cpu()
            if Q == 1:
                per = [(pt2d, pt3d)]
            else:
                mb = batch["m_bids"].cpu()
                per = [(pt2d[mb == q], pt3d[mb == q]) for q in range(Q)]
            counts = [len(a) for a, _ in per]
        out = []
        for q in range(Q):
            res = None if no_pose else _solve_pnp(solver, per[q][0], per[q][1], Ks[min(q, len(Ks) - 1)], rthres, center_subpixel)
            if not res:
                out.append((None, inf, inf, counts[q]))
                continue
            R, t, _ = res
            w2c = torch.eye(4)
            w2c[:3, :3] = torch.as_tensor(R, dtype=torch.float32)
            w2c[:3, 3] = torch.as_tensor(t, dtype=torch.float32).reshape(-1)
            c2w_est = torch.linalg.inv(w2c)
            R_err, t_err = pose_err(self._host(batch, "c2w").reshape(-1, 4, 4)[q], c2w_est)
            out.append((c2w_est, R_err, t_err, counts[q]))
        return out

    def eval_match_pose(self, batch, mutual=True, match_thres=0.0, solver="colmap", rthres=1, center_subpixel=False,
                        match_oracle=False):
        """reference :152-230: matcher forward, then PnP.  Returns (c2w_est, R_err, t_err, num_matches) for a batch of one
        query (the reference's case) and a list of such tuples for Q > 1."""
        if not match_oracle:
            self._match_finish(batch, self._match_begin(batch, mutual, match_thres))
            self._flush_match_times()
        res = self._poses_from_matches(batch, solver, rthres, center_subpixel, match_oracle=match_oracle)
        return res[0] if len(res) == 1 else res

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
        use_match_loss = getattr(inerf_conf, "use_match_loss", False)  # (either model class: _MatcherBase.match_loss)
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
        match = None
        if use_match_loss:  # reference :429-437: the query image against the view rendered in this step
            on_dev = lambda k: batch[k].to(self.device) if batch.get(k) is not None else None
            match = dict(model=self.model, image=on_dev("image"), im_mask=on_dev("im_mask"), pt_mask=on_dev("pt_mask"), unnorm=unnorm)
        tj = time.time()
        for j, pose, loss, ctx in inerf.refine_iter(renderer, K, H, W, img, pose0, num_optim, lrate, lrdecay, ds, t_rands, jitters,
                                                    match=match):
            self.timer["inerf_step_time"].append(time.time() - tj)
            if debug or cache_iters or j == num_optim - 1:
                if eval_pose:
                    c2w_est = (unnorm @ pose).cpu()
                    R_err, t_err = pose_err(c2w_gt.squeeze(), c2w_est)
                else:
                    pts, feats = inerf.rendered_points(renderer, ctx)
                    batch["pt3d"] = ops.unnormalize_points(pts, unnorm.cpu()).unsqueeze(0)
                    batch["pt_feat"] = feats.unsqueeze(0)
                    batch["pt_mask"] = self._ones_mask(batch["pt3d"])
                    c2w_est, R_err, t_err, _ = self.eval_match_pose(batch, mutual=mutual, match_thres=match_thres, solver=solver,
                                                                     rthres=rthres, center_subpixel=center_subpixel)
                if cache_iters and j > 0 and j != num_optim - 1:
                    iter_t_errs.append(t_err)
                    iter_R_errs.append(R_err)
                if debug:
                    print(f"  inerf step={j} loss={loss:2f} t={t_err * 100:.4f}cm R={R_err:.4f}")
            tj = time.time()
        return c2w_est, R_err, t_err

    # -- localisation of one batch ------------------------------------------------------------------------------------------
    def _render_into(self, batch, renderer, poses, unnorm_scene, side=None):
        """Render the points / features seen from `poses` (Q world poses) into the batch (reference :556-574).  Only pt3d and
        pt_feat are read afterwards, so the render skips the colour heads."""
        hw = batch["image"].shape[-2:]
        Ks = self._host(batch, "K").reshape(-1, 3, 3)
        poses = torch.stack([torch.as_tensor(p).detach().float().cpu() for p in poses])
        if side is not None:
            # the pipelined loop: this render goes to the render stream; the caller's stream waits for its event (below) and for nothing
            # else of that stream, the render stream for nothing of the caller's -- its inputs are host tensors and the packed weights
            cur = torch.cuda.current_stream(self.device)
            with torch.cuda.stream(side):
                self._render_into(batch, renderer, poses, unnorm_scene)
                done = torch.cuda.Event()
                done.record(side)
            cur.wait_event(done)
            if cur != torch.cuda.default_stream(self.device):  # (the legacy default stream orders against the partition streams by itself, _finish_on)
                for k in ("pt3d", "pt_feat", "pt_mask"):  # allocated on the render stream's pool, read by the caller's stream
                    batch[k].record_stream(cur)
            return
        if len(poses) == 1 or bool((Ks == Ks[:1]).all()):
            outs = renderer.render_novel_views(hw, Ks[0], poses, unnorm_scene, self.device, downsample=8, want_im_pred=False)
            pt3d, pt_feat = outs["pt3d"], outs["pt_feat"]
        else:  # different intrinsics per query: one ray-generation launch each
            outs = [renderer.render_novel_view(hw, Ks[q], poses[q], unnorm_scene, self.device, downsample=8, want_im_pred=False)
                    for q in range(len(poses))]
            pt3d, pt_feat = torch.stack([o["pt3d"] for o in outs]), torch.stack([o["pt_feat"] for o in outs])
        batch["pt3d"], batch["pt_feat"] = pt3d, pt_feat
        batch["pt_mask"] = self._ones_mask(pt3d)
        batch["_render_tok"] = getattr(renderer, "__dict__", {}).get("_stale_last")  # (ParamGuard's flag copy behind this render, see _localize_finish)

    def _render_beside_image_side(self, batch, renderer, poses, unnorm_scene):
        """The render on one compute-unit partition and the matcher's image side on another, side by side; the caller's stream continues
        behind both.  -> (im_cfeat, im_ffeat) for forward_begin(image_side=...)."""
        def stream_of(spec):
            return (_lib.partition_stream(0, 0, self.device, xcds=(int(spec[1]), int(spec[2]))) if isinstance(spec, tuple)
                    else _lib.partition_stream(int(spec[0]), int(spec[1]), self.device))

        rs, ims = stream_of(self.split_step[0]), stream_of(self.split_step[1])
        cur = torch.cuda.current_stream(self.device)
        rs.wait_stream(cur)
        ims.wait_stream(cur)
        self.model.keep_conf = bool(self.keep_conf_matrix)
        with torch.cuda.stream(ims):
            image_side = self.model.forward_image_side(batch["image"])
            done_i = torch.cuda.Event()
            done_i.record(ims)
        self._render_into(batch, renderer, poses, unnorm_scene, side=rs)  # (the caller's stream waits for the render's event in there)
        cur.wait_event(done_i)
        if cur != torch.cuda.default_stream(self.device):
            for t in image_side:
                t.record_stream(cur)
        return image_side

    def _ones_mask(self, pt3d):
        """The all-valid point mask of rendered points.  The reference builds a fresh float `ones_like(pt3d[..., 0])` per batch
        (:568, :476); here it is ONE bool tensor per shape, shared by every batch of that shape (a bool mask is what the kernels read
        without a conversion launch, and one fill launch per shape instead of one per batch).  READ-ONLY for consumers: an in-place edit
        would show up in every later batch -- clone it first (`batch["pt_mask"] = batch["pt_mask"].clone()`) to mask points out."""
        key = (tuple(pt3d.shape[:-1]), str(pt3d.device))
        ones = self.__dict__.setdefault("_ones_masks", {})
        if key not in ones:
            ones.clear()
            ones[key] = torch.ones(pt3d.shape[:-1], dtype=torch.bool, device=pt3d.device)
        return ones[key]

    _HOST_KEYS = ("K", "c2w", "rc2w", "unnorm_scene")

    def _host(self, batch, key):
        """Host copy of a small per-query tensor (intrinsics, poses, scene normalisation).  Taken once per batch BEFORE the
        batch moves to the device: a `.cpu()` of a device tensor is a full stream synchronisation, and one of those between
        two batches would drain the work the pipelined loop has queued (the reference pays it: `batch["K"].cpu()`, :529).
        Batches that arrive with these tensors already on the device cost one synchronisation here."""
        host = batch.setdefault("_host", {})
        if key not in host:
            host[key] = batch[key].detach().cpu()
        return host[key]

    def _localize_begin(self, batch, renderer, o):
        """Iteration 0 up to the matcher's synchronisation point: everything enqueued, nothing read back."""
        for k in self._HOST_KEYS:
            if k in batch and isinstance(batch[k], torch.Tensor):
                self._host(batch, k)
        # The small per-query tensors stay where they are (they are only read on the host); everything else moves to the GPU.
        # Moving them too (the reference does) costs a pageable host-to-device copy each, and such a copy waits for ALL queued
        # GPU work: the host would resume with an empty queue and the GPU would idle while the next batch is being issued.
        host_only = {k: batch.pop(k) for k in self._HOST_KEYS if k in batch and isinstance(batch[k], torch.Tensor) and not batch[k].is_cuda}
        data_to_device(batch, self.device)
        batch.update(host_only)
        Q = batch["image"].shape[0]
        unnorm_scene = self._host(batch, "unnorm_scene").reshape(-1, 4, 4)[0] if "unnorm_scene" in batch else getattr(renderer, "unnorm_scene", None)
        if isinstance(unnorm_scene, np.ndarray):
            unnorm_scene = torch.from_numpy(unnorm_scene)
        if o["query2query"]:
            poses = list(self._host(batch, "c2w").reshape(-1, 4, 4))
        elif (not o["cached_pt"]) or o["retrieval_only"]:
            poses = list(self._host(batch, "rc2w").reshape(-1, 4, 4))
        else:
            poses = [None] * Q
        st = dict(batch=batch, renderer=renderer, o=o, Q=Q, unnorm_scene=unnorm_scene, poses=poses, ts=time.time(), ms=None)
        if not o["retrieval_only"]:
            image_side = None
            if all(p is not None for p in poses):
                side = o.get("render_stream") if Q <= self.overlap_max_queries else None
                split = (side is None and self.split_step is not None and Q <= self.overlap_max_queries and not self.coarse_only
                         and not o["match_oracle"] and self.device.type == "cuda")
                if split:
                    image_side = self._render_beside_image_side(batch, renderer, poses, unnorm_scene)
                else:
                    self._render_into(batch, renderer, poses, unnorm_scene, side=side)
            if not o["match_oracle"]:
                st["ms"] = self._match_begin(batch, o["mutual"], o["match_thres"], image_side=image_side)
        return st

    def _localize_finish(self, st):
        """Completes iteration 0 (count read-back, fine stage, PnP), then runs the remaining iterations / the refinement."""
        batch, renderer, o, Q, unnorm_scene, poses = st["batch"], st["renderer"], st["o"], st["Q"], st["unnorm_scene"], st["poses"]
        inf = torch.tensor(float("inf"))
        R_errs, t_errs, nums = [inf] * Q, [inf] * Q, [0] * Q
        iter_t_errs, iter_R_errs = [], []
        last_pose = list(poses)
        for itr in range(o["iters"]):
            if o["retrieval_only"]:
                for q in range(Q):
                    R_errs[q], t_errs[q] = pose_err(self._host(batch, "c2w").reshape(-1, 4, 4)[q], poses[q].cpu())
            else:
                if itr > 0:
                    # Q == 1 is the reference's loop: no pose -> no re-render, the old points are matched again (:556).  In a
                    # batch, a query whose PnP failed is re-rendered from its last valid pose so that the others can proceed.
                    have = poses if Q == 1 else [p if p is not None else lp for p, lp in zip(poses, last_pose)]
                    if all(p is not None for p in have):
                        self._render_into(batch, renderer, have, unnorm_scene)
                    if not o["match_oracle"]:
                        st["ms"] = self._match_begin(batch, o["mutual"], o["match_thres"])
                if not o["match_oracle"]:
                    self._match_finish(batch, st["ms"])
                if itr == 0 and renderer is not None and not st.get("redone") and batch.get("_render_tok") is not None:
                    # the matcher's read-back has just synchronised behind this batch's render: did that render run on blobs older than
                    # the parameters (a write through `.data`, ops.ParamGuard)?  Then the blobs are fresh by now and the batch is repeated.
                    if renderer._token_stale(batch.pop("_render_tok"), wait=True):
                        again = self._localize_begin(batch, renderer, o)
                        again["redone"] = True
                        return self._localize_finish(again)
                res = self._poses_from_matches(batch, o["solver"], o["rthres"], o["center_subpixel"], match_oracle=o["match_oracle"])
                for q, (pose, R_err, t_err, n) in enumerate(res):
                    R_errs[q], t_errs[q], nums[q] = R_err, t_err, n
                    if pose is not None or o["solver"] not in (None, "none"):
                        poses[q] = pose  # solver "none": keep the pose the points were rendered from
                    if pose is not None:
                        last_pose[q] = pose
                if o["inerf_conf"] and o["cache_iters"]:
                    iter_t_errs.append(t_errs[0])
                    iter_R_errs.append(R_errs[0])
            if o["inerf_conf"]:
                # The reference refines batch element 0 only (its loop is batch 1: `batch["image"].clone()[0]`, :323); a batch of Q queries
                # is refined query by query, each on its own one-query view of the batch.
                for q in range(Q):
                    if poses[q] is None:
                        continue
                    sub = batch if Q == 1 else self._query_view(batch, q, Q)
                    res = self.inerf_refinement(sub, renderer, unnorm_scene, poses[q], o["inerf_conf"], mutual=o["mutual"],
                                                match_thres=o["match_thres"], solver=o["solver"], rthres=o["rthres"],
                                                center_subpixel=o["center_subpixel"], cache_iters=o["cache_iters"] and q == 0,
                                                iter_t_errs=iter_t_errs, iter_R_errs=iter_R_errs, debug=o["debug"])
                    if res[1] != float("inf"):  # take the refined pose only if it could be evaluated (reference :608-610)
                        poses[q], R_errs[q], t_errs[q] = res
            if o["cache_iters"]:
                iter_t_errs.append(t_errs[0] if Q == 1 else list(t_errs))
                iter_R_errs.append(R_errs[0] if Q == 1 else list(R_errs))
            if o["debug"]:
                print(f">> iter={itr} matches={nums} t={[float(t) * 100 for t in t_errs]}cm R={[float(r) for r in R_errs]}")
            if all(p is None for p in poses) and all(p is None for p in last_pose):
                break  # nothing to render from: further iterations would repeat this one
        # per-query wall time of the step; in the pipelined loop it starts when the previous batch finished (st["ts"] is
        # moved there by eval_data_loader), i.e. it is the steady-state time per query, not begin-to-finish across the overlap
        self.timer["localize_time"].append((time.time() - st["ts"]) / Q)
        return dict(R_err=list(R_errs), t_err=list(t_errs), iter_t_errs=iter_t_errs, iter_R_errs=iter_R_errs, num_matches=list(nums),
                    c2w_est=poses[0] if Q == 1 else list(poses), c2w_ests=list(poses))

    _PER_QUERY = ("image", "im_mask", "K", "c2w", "rc2w", "pt2d", "pt3d", "pt_feat", "pt_mask", "unnorm_scene", "pt2d_proj", "conf_gt", "idx")

    @classmethod
    def _query_view(cls, batch, q, Q):
        """One-query view of a batch of Q queries: the per-query inputs of the reference's batch schema (nerfmatch_dataset.py:311-325) are
        sliced to [q:q+1] (views, no copies); what an earlier matcher pass left in the batch (match lists of ALL queries) is not carried over."""
        sub = {}
        for k in cls._PER_QUERY:
            v = batch.get(k)
            if isinstance(v, torch.Tensor) and v.dim() and v.shape[0] == Q:
                sub[k] = v[q:q + 1]
            elif v is not None or k in batch:
                sub[k] = v
        if "_host" in batch:
            sub["_host"] = {hk: (hv[q:q + 1] if isinstance(hv, torch.Tensor) and hv.dim() and hv.shape[0] == Q else hv) for hk, hv in batch["_host"].items()}
        return sub

    @staticmethod
    def _opts(**kw):
        return kw

    def _pipeline_streams(self, renderer, o):
        """(render stream, matcher stream | None, the caller's stream) of the pipelined loop, or None when it runs on one stream (no renderer, CPU, iterated
        localisation / refinement -- their re-renders depend on the matcher's result -- or `overlap_render` off)."""
        if (not self.overlap_render or renderer is None or self.device.type != "cuda" or o["iters"] != 1 or o["inerf_conf"] or o["retrieval_only"]
                or (o["cached_pt"] and not o["query2query"])):
            return None
        def plain(tag):  # a non-blocking stream of torch's own (never the legacy default stream: it orders against every partition stream)
            st = self.__dict__.get(tag)
            if st is None:
                st = self.__dict__[tag] = torch.cuda.Stream(device=self.device)
            return st

        ncu = torch.cuda.get_device_properties(self.device).multi_processor_count

        def part(spec, first_free, tag):
            if spec is None:
                return plain(tag), first_free
            if isinstance(spec, tuple):  # ("xcd", first, count): whole XCDs
                return _lib.partition_stream(0, 0, self.device, xcds=(int(spec[1]), int(spec[2]))), first_free
            n = max(32, min(int(spec), ncu - first_free) // 32 * 32)  # (whole multiples of 32: the same number of units in every XCD and engine)
            return _lib.partition_stream(n, first_free, self.device), first_free + n

        rs, used = part(self.render_part, 0, "_plain_render_stream")
        ms, _ = part(self.match_part, used, "_plain_match_stream")
        return rs, ms, torch.cuda.current_stream(self.device)

    def _begin_on(self, batch, renderer, o, streams):
        """_localize_begin; a batch small enough for the two-stream pipeline has its matcher issued on the matcher's partition (its render
        goes to the render stream inside _localize_begin), every other batch runs on the caller's stream as before."""
        ms = None
        if streams is not None and streams[1] is not None and batch["image"].shape[0] <= self.overlap_max_queries:
            ms = streams[1]
            # A loader that builds device tensors does so on the caller's stream: the matcher's stream is ordered behind it.  NOT when that
            # is the legacy default stream: the partition streams are "blocking" streams (hipExtStreamCreateWithCUMask takes no flags), which
            # the null stream orders against implicitly -- and an event recorded on the null stream waits for ALL of them, i.e. the render
            # and the matcher of consecutive queries would run one after the other again (measured: 4.5 instead of 2.4 ms per query).
            if streams[2] != torch.cuda.default_stream(self.device):
                ms.wait_stream(streams[2])
        with (torch.cuda.stream(ms) if ms is not None else contextlib.nullcontext()):
            st = self._localize_begin(batch, renderer, o)
        st["stream"] = ms
        return st

    def _finish_on(self, st, streams):
        """_localize_finish on the stream the batch was begun on; when that is the matcher's partition, the tensors it left in the batch
        dict are made known to the caller's stream (the allocator must not hand their memory out again while the caller still reads them)."""
        ms = st.get("stream")
        with (torch.cuda.stream(ms) if ms is not None else contextlib.nullcontext()):
            m = self._localize_finish(st)
        if ms is not None and streams[2] != torch.cuda.default_stream(self.device):
            # (not for the legacy default stream: it is ordered against the partition streams implicitly, and an allocator event on it -- one
            # per freed tensor -- is a barrier over all of them: 4.5 instead of 2.5 ms per query, scripts/ab_loop_context.py)
            cur = streams[2]
            for v in st["batch"].values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(cur)
        return m

    @contextlib.contextmanager
    def _on_stream(self, streams):
        """The loop body with the render stream (and the matcher's partition) ordered behind the caller's stream at entry and the caller's
        stream behind both at exit."""
        if streams is None:
            yield
            return
        rs, ms, cur = streams
        rs.wait_stream(cur)  # packed weights, calibration: whatever the caller's stream has written so far
        if ms is not None:
            ms.wait_stream(cur)
        try:
            yield
        finally:
            cur.wait_stream(rs)
            if ms is not None:
                cur.wait_stream(ms)

    def eval_batch(self, batch, renderer=None, inerf_conf=None, iters=1, mutual=True, match_thres=0.0, match_oracle=False,
                   solver="colmap", rthres=1, center_subpixel=False, visualize=False, overlay_ims=None, query2query=False,
                   retrieval_only=False, cached_pt=True, cache_iters=False, debug=False):
        """reference :502-629.  The batch may hold Q >= 1 queries; per-query lists come back (`R_err`, `t_err`, `num_matches`,
        `c2w_ests`; `c2w_est` is the pose itself when Q == 1)."""
        if visualize:
            raise NotImplementedError("overlay visualisation is out of scope (SURVEY.md section 2)")
        o = self._opts(inerf_conf=inerf_conf, iters=iters, mutual=mutual, match_thres=match_thres, solver=solver, rthres=rthres,
                       center_subpixel=center_subpixel, query2query=query2query, retrieval_only=retrieval_only, cached_pt=cached_pt,
                       cache_iters=cache_iters, debug=debug, match_oracle=match_oracle)
        out = self._localize_finish(self._localize_begin(batch, renderer, o))
        self._flush_match_times()
        return out

    def eval_data_loader(self, renderer=None, iters=1, rthres=1, center_subpixel=False, solver="colmap", mutual=True, match_thres=0.0,
                         match_oracle=False, data_loader=None, query2query=False, cached_pt=True, debug=False, inerf_conf=None,
                         retrieval_only=False, cache_iters=False, visualize=False):
        """reference :630-724 over any iterable of batch dicts (a torch DataLoader, a list, ...; a batch holds Q >= 1 queries).

        * Pipelining: batch i+1's render and matcher are enqueued BEFORE batch i's match counts are read back, so the GPU
          stays busy across the one synchronisation point of a localisation step (what bench.py measures).
        * Multi-GPU: with torch.distributed initialised, batches are dealt round-robin over ranks and the per-query records
          [idx, c2w_est(16), R_err, t_err, num_matches] are all-gathered once at the end (RCCL over xGMI): every rank
          returns the metrics of ALL queries, ordered by query index."""
        if visualize:
            raise NotImplementedError("overlay visualisation is out of scope (SURVEY.md section 2)")
        loader = data_loader if data_loader is not None else self.data_loader
        rank, W = nmdist.world()
        o = self._opts(inerf_conf=inerf_conf, iters=iters, mutual=mutual, match_thres=match_thres, solver=solver, rthres=rthres,
                       center_subpixel=center_subpixel, query2query=query2query, retrieval_only=retrieval_only, cached_pt=cached_pt,
                       cache_iters=cache_iters, debug=debug, match_oracle=match_oracle)
        if W > 1 and renderer is not None:
            nmdist.agree_calibration(renderer, self.device)  # identical fp16x3 operand scales on every rank: identical bits per query
        streams = self._pipeline_streams(renderer, o)
        if streams is not None:
            o["render_stream"] = streams[0]
        full_bs = getattr(loader, "batch_size", None)
        recs, iter_t, iter_R = [], [], []
        # Global query index of a batch's first query: `batch["idx"]` when the dataset provides it, else bi * batch_size with
        # the loader's declared batch size; a loader without one (list, generator) must deliver equal-size batches except the
        # last -- the first batch each rank sees fixes the size and any larger / non-final smaller batch is refused, because
        # indices would collide and gather_records would attribute metrics to the wrong queries.
        n_batches = len(loader) if hasattr(loader, "__len__") else None
        # what this rank saw of un-indexed batches: [index of its short batch (-1: none), index of its last batch (-1: none)]
        seen = dict(short=-1, last=-1, bad=None)  # bad: a violation THIS rank detected (raised only after the collective, see below)

        def emit(bi, Q, m, idx=None):
            q0 = bi * full_bs
            if idx is None:
                err = None
                if Q > full_bs or seen["short"] >= 0:
                    # a larger batch, or ANY batch behind a short one: q0 = bi * batch_size would collide with another batch's indices
                    err = (f"batch {bi} holds {Q} queries, batch size {full_bs}, after a short batch at {seen['short']}: only the LAST "
                           "batch may be short (give the loader a `batch_size` attribute or put a per-query `idx` tensor into the batches)")
                elif Q < full_bs and n_batches is not None and bi != n_batches - 1:
                    err = f"batch {bi} of {n_batches} holds {Q} queries but the loader's batch size is {full_bs}: only the LAST batch may be short"
                if err is not None:
                    # In a sharded run the other ranks are on their way to the collectives below: raising HERE would leave them
                    # blocked there until the process-group timeout.  The violation is recorded, this rank's remaining
                    # batches are dropped, and every rank raises after the all-reduce that tells them all.
                    if W == 1:
                        raise ValueError(err)
                    seen["bad"] = seen["bad"] or err
                    return
                if Q < full_bs:
                    seen["short"] = bi
                seen["last"] = max(seen["last"], bi)
            for q in range(Q):
                if idx is not None:
                    q0 = int(idx[q]) - q
                recs.append(nmdist.make_record(q0 + q, m["c2w_ests"][q], float(m["R_err"][q]), float(m["t_err"][q]), m["num_matches"][q]))
            if cache_iters:
                iter_t.append(m["iter_t_errs"])
                iter_R.append(m["iter_R_errs"])

        pending = None
        done = 0
        last_done = 0.0
        first_size = {}
        if hasattr(loader, "__getitem__") and hasattr(loader, "__len__"):
            mine = ((bi, loader[bi]) for bi in nmdist.shard_indices(len(loader), rank, W))
        else:  # a DataLoader-like iterable: every rank walks ALL of it (and so sees the global first batch) and keeps its share
            def walk():
                for bi, b in enumerate(loader):
                    if bi == 0:
                        first_size["n"] = b["image"].shape[0]
                    if bi % W == rank:
                        yield bi, b
            mine = walk()
        with _lib.steady_gc(), self._on_stream(streams):  # (the resident objects are exempt from the cyclic collector while the loop runs: no 80 ms pauses)
            for bi, batch in mine:
                if full_bs is None:
                    # no declared batch size: all ranks must agree on it, and a short last batch must not define it -- the size of the
                    # first batch of the GLOBAL sequence: this very batch when bi == 0, the one every rank walked past in a stream, else
                    # loader[0] of an indexable loader (materialised once, only on ranks whose first batch is not batch 0)
                    if bi == 0:
                        full_bs = batch["image"].shape[0]
                    elif "n" in first_size:
                        full_bs = first_size["n"]
                    else:
                        full_bs = loader[0]["image"].shape[0]
                idx = batch.get("idx") if isinstance(batch, dict) else None
                idx = None if idx is None else torch.as_tensor(idx).reshape(-1).cpu()
                st = self._begin_on(batch, renderer, o, streams)
                st["idx"] = idx
                if pending is not None:
                    pending[1]["ts"] = max(pending[1].get("ts", 0.0), last_done)
                    emit(pending[0], pending[1]["Q"], self._finish_on(pending[1], streams), pending[1]["idx"])
                    last_done = time.time()
                pending = (bi, st)
                done += 1
                if debug and done > 5:
                    break
            if pending is not None:
                pending[1]["ts"] = max(pending[1].get("ts", 0.0), last_done)
                emit(pending[0], pending[1]["Q"], self._finish_on(pending[1], streams), pending[1]["idx"])
        self._flush_match_times()
        if W > 1:
            # One 24-byte MAX all-reduce settles two things for everybody: (1) a stream without length: "only the last batch may be
            # short" cannot be checked by one rank alone (the short batch and the batches behind it may sit on different ranks);
            # (2) a violation some rank detected locally -- every rank raises, none is left waiting in a collective.
            import torch.distributed as dist
            chk = torch.tensor([seen["short"], seen["last"], 1 if seen["bad"] else 0], device=self.device, dtype=torch.int64)
            dist.all_reduce(chk, op=dist.ReduceOp.MAX)
            if int(chk[2]):
                raise ValueError(seen["bad"] or "another rank found a batch whose size breaks the query indexing (only the LAST batch may be short)")
            if n_batches is None and int(chk[0]) >= 0 and int(chk[0]) != int(chk[1]):
                raise ValueError(f"batch {int(chk[0])} was short but batch {int(chk[1])} followed it: query indices collide (give the batches a per-query `idx`)")
        local = torch.stack(recs) if recs else torch.empty(0, nmdist.RECORD_FLOATS)
        allrec = nmdist.gather_records(local, None, self.device).cpu()
        out = dict(R_err=allrec[:, 17].numpy(), t_err=allrec[:, 18].numpy(), num_matches=allrec[:, 19].numpy(),
                   query_idx=allrec[:, 0].long().numpy(), c2w_est=allrec[:, 1:17].reshape(-1, 4, 4).numpy())
        if cache_iters:  # rank-local (like the timers): the iteration traces are diagnostics, not part of the gathered record
            out.update(iter_t_errs=iter_t, iter_R_errs=iter_R)
        return out

    # -- all scenes of a benchmark -----------------------------------------------------------------------------------------
    def _result_cache_path(self, scene, split, rthres, mutual, match_thres, solver, center_subpixel, retrieval_only, iters, inerf_conf,
                           conf, test_pair_txt, cached_pt, query2query, cache_iters, match_oracle, debug):
        """File name of a scene's cached metrics: the reference's scheme (:782-850), so that result files are interchangeable."""
        tags = [f"{scene}_rth{rthres:.0f}{split}"]
        tags += ["_coarse"] if self.coarse_only else []
        tags += [] if mutual else ["_no_mutual"]
        tags += [f"_sc{match_thres:.2f}"] if match_thres > 0 else []
        tags += [f"_{solver}"] if solver != "cv" else []
        tags += ["_subpx"] if center_subpixel else []
        tags += ["_IR"] if retrieval_only else []
        if inerf_conf:
            t = f"_itr{iters}ds{getattr(inerf_conf, 'ds', 8)}inerf{getattr(inerf_conf, 'num_optim', 5)}lr{getattr(inerf_conf, 'lrate', 0.001)}"
            t += "lrdcos" if getattr(inerf_conf, "lrdecay", False) > 0 else ""
            t += "pose" if getattr(inerf_conf, "eval_pose", False) else "match"
            tags.append(t)
        else:
            tags.append(f"_itr{iters}")
        if getattr(conf, "dataset", None) == "NeRFMatchMultiPair":
            tags.append(f"_top{conf.pair_topk}pt{conf.sample_pts}")
            tags += [f"_{conf.sample_mode}"] if getattr(conf, "sample_mode", None) else []
        tags += ["." + test_pair_txt.split("netvlad10-")[1].replace(".txt", "_pairs")] if test_pair_txt else []
        tags += [] if cached_pt else ["_nocache"]
        tags += [".query2query"] if query2query else []
        tags += [".itercache"] if cache_iters else []
        tags += [".match_oracle"] if match_oracle else []
        tags += [".debug"] if debug else []
        return str(self.cache_dir / ("".join(tags) + ".npy"))

    def eval_multi_scenes(self, split="test", batch_size=1, rthres=1, center_subpixel=False, solver="colmap", mutual=True,
                          match_thres=0.0, iters=1, nerf_path=None, inerf_conf=None, test_pair_txt=None, scene_dir=None, ow_cache=False,
                          data_conf=None, query2query=False, cached_pt=True, stop_layer=-1, debug=False, visualize=False, cache_dir=None,
                          cache_iters=False, retrieval_only=False, match_oracle=False, seed=None, dataset_factory=None,
                          renderer_factory=None):
        """reference :726-931, keyword for keyword (the call in model_eval/benchmark_nerfmatch.py:126-151 works unchanged).

        The reference builds one dataset per scene from its dataset classes (`init_mixed_dataset` / `init_multiscene_dataset`,
        out of scope): here `dataset_factory(data_conf, split)` -- argument or `self.dataset_factory` -- returns that list; each
        dataset needs `.scene`, `.scene_dir` and the reference's per-sample dict schema (a `torch.utils.data.Dataset` or any
        sequence).  Renderers come from `renderer_factory(scene, scene_dir, stop_layer)` or, like the reference, from
        `load_nerf_render_from_ckpt(nerf_path with $scene / #scene replaced)`.  Per scene: result-cache lookup (`ow_cache`),
        `eval_data_loader` in batches of `batch_size` queries, timers, np.save of the metrics, pose statistics; returns the
        list of per-scene summaries (the reference prints their average)."""
        from torch.utils.data import DataLoader

        if cache_dir:
            self.cache_dir = Path(cache_dir)
        self.cache_dir.mkdir(parents=True, exist_ok=True)
        conf = getattr(self.config, "data", Namespace())
        if data_conf is not None:
            conf = merge_configs(conf, data_conf)
        if test_pair_txt:
            conf.test_pair_txt = test_pair_txt
        if scene_dir:
            conf.scene_dir = scene_dir
        factory = dataset_factory or self.dataset_factory
        if factory is None:
            raise NotImplementedError("the reference's dataset classes are out of scope (SURVEY.md section 2): pass dataset_factory="
                                      "(data_conf, split) -> [dataset with .scene / .scene_dir, ...] or set evaluator.dataset_factory")
        make_renderer = renderer_factory or self.renderer_factory
        metr_all = []
        for dataset in factory(conf, split):
            if seed:
                torch.manual_seed(seed)
                np.random.seed(seed)
            self.timer = defaultdict(list)
            scene = dataset.scene
            cache_path = self._result_cache_path(scene, split, rthres, mutual, match_thres, solver, center_subpixel, retrieval_only, iters,
                                                 inerf_conf, conf, test_pair_txt, cached_pt, query2query, cache_iters, match_oracle, debug)
            pose_thres = POSE_THRES.get(scene, [(5, 5)])
            rank = nmdist.world()[0]
            if os.path.exists(cache_path) and not ow_cache:
                metrics = np.load(cache_path, allow_pickle=True).item()
                metr_all.append(summarize_pose_statis(metrics, pose_thres=pose_thres, t_unit="cm", t_scale=1e2, print_out=rank == 0))
                continue
            loader = dataset if hasattr(dataset, "batch_size") else DataLoader(dataset, shuffle=False, batch_size=batch_size, pin_memory=False)
            renderer = None
            if (not cached_pt) or query2query or (iters > 1) or inerf_conf:
                sl = stop_layer if stop_layer > 0 else parse_nerf_stop_layer(getattr(dataset, "scene_dir", "") or "")
                if make_renderer is not None:
                    renderer = make_renderer(scene, getattr(dataset, "scene_dir", None), sl)
                else:
                    renderer = load_nerf_render_from_ckpt(nerf_path.replace("$scene", scene).replace("#scene", scene), self.device, stop_layer=sl)
            metrics = self.eval_data_loader(renderer=renderer, iters=iters, rthres=rthres, center_subpixel=center_subpixel, solver=solver,
                                            mutual=mutual, match_thres=match_thres, match_oracle=match_oracle, data_loader=loader,
                                            query2query=query2query, cached_pt=cached_pt, debug=debug, inerf_conf=inerf_conf,
                                            retrieval_only=retrieval_only, cache_iters=cache_iters, visualize=visualize)
            for k, v in self.timer.items():
                metrics[k] = np.array(v)
            if rank == 0:
                np.save(cache_path, metrics)
            metr_all.append(summarize_pose_statis(metrics, pose_thres=pose_thres, t_unit="cm", t_scale=1e2, print_out=rank == 0))
        if metr_all:
            average_pose_metrics(metr_all, print_out=nmdist.world()[0] == 0)
        return metr_all


def load_nerfmatch_from_ckpt(ckpt_path, args=None, root_dir=".", arg_mask=None, data_loader=None, backbone=None):
    """Lightning checkpoint of the reference -> evaluator (reference :69-115).

    The image backbone is timm's ConvFormer (third party, out of scope): pass it as `backbone` (a module with the contract of
    nerfmatch_amd.modules: .feat_dim and forward(img) -> (cfeat, ffeat) | cfeat) and its `model.backbone.*` tensors are loaded
    into it; without one the checkpoint's backbone tensors are dropped and the stub backbone stays (synthetic benchmarks).
    Every other key must match: missing or unexpected matcher weights raise (the reference's strict=False would let a
    renamed weight pass silently)."""
    ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    hp = ckpt["hyper_parameters"]
    config = hp if isinstance(hp, Namespace) else Namespace(**hp)
    config.ckpt = ckpt_path
    if args:
        config = merge_configs(config, args)
    if backbone is not None:
        config.model.backbone = "stub"  # placeholder while the model is built; replaced below
    evaluator = NeRFMatchEvaluator(config, data_loader=data_loader)
    state = dict(ckpt["state_dict"])
    if backbone is not None:
        evaluator.model.backbone = backbone.to(evaluator.device)
    if not any(True for _ in evaluator.model.backbone.parameters()):
        state = {k: v for k, v in state.items() if not k.startswith("model.backbone.")}
    res = evaluator.load_state_dict(state, strict=False)
    # im_sa.* aliases pt_sa.* when the self-attention block is shared (same module object): either spelling may be absent
    alias = lambda k: k.startswith("model.im_sa.") and getattr(evaluator.model, "im_sa", None) is getattr(evaluator.model, "pt_sa", None)
    missing = [k for k in res.missing_keys if not alias(k)]
    unexpected = [k for k in res.unexpected_keys if not alias(k)]
    if missing or unexpected:
        raise RuntimeError(f"{ckpt_path}: checkpoint does not fit the model -- missing {missing[:8]}{'...' if len(missing) > 8 else ''}, "
                           f"unexpected {unexpected[:8]}{'...' if len(unexpected) > 8 else ''}")
    return evaluator

"""Checkpoint loading for the renderer (API of nerfmatch/nerf_evaluator.py:119-156).

`load_nerf_render_from_ckpt(ckpt_path, device, stop_layer)` reads a Lightning checkpoint written by the reference
(`hyper_parameters` = nested Namespace, `state_dict` with a `model.` prefix) and returns a NerfRenderer whose
weights are packed for the HIP kernels.  torch >= 2.6 defaults to weights_only=True, which rejects these pickles,
hence weights_only=False.

`NerfEvaluator` (reference :159-402) is the render-side evaluator behind model_eval/eval_nerf.py (BASELINE config 1):
`eval_batch` -> `NerfRenderer.predict` on a frame's ray bundle, the PSNR loop, and `cache_scene_pts`, the writer of the
scene-feature cache the matcher's datasets read.  The dataset classes are out of scope: pass any
iterable of the reference's batch dicts as `data_loader`."""
import os
from argparse import Namespace
from collections import defaultdict
from pathlib import Path

import numpy as np
import torch

from . import dist as nmdist
from . import ops
from .nerf.renderer import NerfRenderer
from .utils.metrics import compute_nerf_metrics


def local_device():
    """The GPU of this process: LOCAL_RANK when a launcher set it (one process per GPU; the device is made current, because
    the kernels run on the current device's stream), else the current device.  The reference pins cuda:0
    (nerf_evaluator.py:153) -- it never runs more than one evaluation process."""
    import os

    if not torch.cuda.is_available():
        return torch.device("cpu")
    lr = os.environ.get("LOCAL_RANK")
    if lr is not None and int(lr) < torch.cuda.device_count():
        torch.cuda.set_device(int(lr))
    return torch.device("cuda", torch.cuda.current_device())


class GenericModelEvaluator(torch.nn.Module):
    """reference: nerf_evaluator.py:149-156 (device selection, grad globally off)."""

    def __init__(self, config):
        super().__init__()
        self.device = local_device()
        torch.set_grad_enabled(False)
        self.config = config


def load_nerf_render_from_ckpt(ckpt_path, device, stop_layer=-1, unnorm_scene=None):
    ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    state = ckpt["state_dict"]
    vocab = state["model.embedding_a.weight"].shape[0] if "model.embedding_a.weight" in state else -1
    hp = ckpt["hyper_parameters"]
    config = hp if isinstance(hp, Namespace) else Namespace(**hp)
    render = NerfRenderer(config, num_frames=vocab, training=False, stop_layer=stop_layer)
    new_state = {k[len("model."):]: v for k, v in state.items() if k.startswith("model.")}
    missing = render.load_state_dict(new_state, strict=False)
    if [k for k in missing.missing_keys if "scales" not in k]:
        raise RuntimeError(f"checkpoint lacks NeRF weights: {missing.missing_keys}")
    render.to(device).eval()
    # The reference recomputes the scene normalisation from the training transforms json
    # (nerf_evaluator.py:99-116): a one-off CPU/JSON step outside the hot path; pass it in (or store it in the ckpt).
    render.unnorm_scene = unnorm_scene if unnorm_scene is not None else ckpt.get("unnorm_scene")
    return render


def save_nerf_ckpt(path, config, state_dict, unnorm_scene=None, epoch=0, global_step=0):
    """Writes a checkpoint with the reference's Lightning layout (used by tests / synthetic benchmarks)."""
    ckpt = dict(state_dict={f"model.{k}": v for k, v in state_dict.items()}, hyper_parameters=vars(config), epoch=epoch,
                global_step=global_step)
    if unnorm_scene is not None:
        ckpt["unnorm_scene"] = unnorm_scene
    torch.save(ckpt, path)


class NerfEvaluator(GenericModelEvaluator):
    """reference: nerf_evaluator.py:159-402.  Same constructor arguments plus `data_loader` (the reference builds its
    loader from the out-of-scope dataset classes, :186-190)."""

    def __init__(self, config, mask=False, frame_num=-1, vocab_num=100, stop_layer=-1, data_loader=None):
        super().__init__(config)
        self.seed = getattr(getattr(config, "exp", Namespace()), "seed", 0)
        self.config.data.mask_transient = bool(mask)
        self.config.data.white_bg = bool(mask)
        if frame_num > 0:
            self.config.data.max_sample_num = frame_num
        self.model = NerfRenderer(self.config, num_frames=vocab_num, training=False, stop_layer=stop_layer)
        self.model.to(self.device).eval()
        self.comp_radii = self.model.embed_type == "mip"
        self.data_loader = data_loader
        self.split = getattr(config, "split", "test")
        ckpt = getattr(config, "ckpt", None)
        self.cache_dir = None
        if ckpt:
            wh = config.data.img_wh
            self.cache_dir = Path(ckpt.replace("checkpoints/", "").replace(".ckpt", f"_rendered_{wh[0]}-{wh[1]}_{self.split}"))
            if self.model.mip_var_scale > -1:
                self.cache_dir = self.cache_dir / f"mip_var{self.model.mip_var_scale}"

    # -- one frame -----------------------------------------------------------------------------------------------------
    def _parse(self, batch):
        w, h = (int(v) for v in batch["img_wh"].reshape(-1)[:2])
        rays = batch["rays"].reshape(-1, batch["rays"].shape[-1]).to(self.device)
        ts = batch["ts"].reshape(-1) if "ts" in batch else None  # stays where it is: constant ids are detected on the host
        return w, h, rays, ts

    def eval_batch(self, batch, comp_metric=True, **kw):
        """reference :200-232: rays of one frame -> predict (image-shaped rgb / depth, per-ray feats and points);
        with comp_metric the validation MSE / PSNR against batch["rgbs"]."""
        w, h, rays, ts = self._parse(batch)
        preds = self.model.predict(rays, w, h, ray_id=ts, **kw)
        if not comp_metric:
            return preds
        rgb_gt = batch["rgbs"].reshape(h, w, -1).to(self.device)
        masks = batch["mask"].reshape(h, w, -1).to(self.device) if "mask" in batch else None
        return preds, compute_nerf_metrics(preds, rgb_gt, mask_loss=masks)

    def unnorm(self, unnorm_scene, org_mat):
        """Normalised points -> world (reference :234-238 does this on the host; here nm_unnormalize_points)."""
        pts = org_mat.reshape(-1, 3).to(self.device, torch.float32).contiguous()
        return ops.unnormalize_points(pts, torch.as_tensor(unnorm_scene)).reshape(org_mat.shape)

    def eval_data_loader(self, data_loader=None, save_depth=False, cache_dir=None, debug=False):
        """PSNR over the frames of the loader (reference :240-306).  The reference also writes every rendering as PNG through
        imageio (third party, absent here): done when imageio is importable and a cache_dir is known, skipped otherwise."""
        loader = data_loader if data_loader is not None else self.data_loader
        cache_dir = Path(cache_dir) if cache_dir else self.cache_dir
        try:
            import imageio
        except ImportError:
            imageio = None
        if cache_dir is not None and imageio is not None:
            cache_dir = cache_dir / "debug" if debug else cache_dir
            (cache_dir / "rgb").mkdir(parents=True, exist_ok=True)
        results = defaultdict(list)
        for i, batch in enumerate(loader):
            preds, metrics = self.eval_batch(batch)
            results["psnr"].append(float(metrics["rgb_fine_psnr"]))
            if cache_dir is not None and imageio is not None:
                rgb = preds["rgb_fine"] if "rgb_fine" in preds else preds["rgb_coarse"]
                imageio.imwrite(cache_dir / "rgb" / f"{batch['img_idx'][0]}.png", (255 * rgb.clamp(0, 1)).byte().cpu().numpy())
            if debug and i > 10:
                break
        if cache_dir is not None and imageio is not None:
            np.save(cache_dir / "results.npy", dict(results))
        return results

    # -- scene-feature cache ---------------------------------------------------------------
    def cache_scene_pts(self, feat_comb="lin", debug=False, cache_dir=None, frames_per_launch=4, **predict_kw):
        """Writes one pickled-dict .npy per frame of the loader in the reference's format (reference :308-372):
        {pt3d (N,3) world, unnorm_scene (4,4), pt_feat (N,256), pt_color (N,3) in [0,1][, sky_mask, mask]}, read back by
        NeRFMatchPair (datasets/data_loading.py:36-80).  Frames are independent: they shard over ranks like queries
        (nerfmatch_amd.dist.shard_indices) and `frames_per_launch` of them share one launch per kernel.  Returns the files
        written by this rank.  `predict_kw` (t_rand / jitter: the samplers' random tensors, explicit
        inputs in this code base) is handed to predict()."""
        self.model.ret_pfeat, self.model.feat_comb = True, feat_comb
        if cache_dir is None:
            if self.cache_dir is None:
                raise ValueError("cache_dir is needed (no config.ckpt to derive it from)")
            parts = list(self.cache_dir.parts)  # reference :317-330: <root>/scene_dirs/.../scene_msk/ds<ds><comb>
            parts[1] = "scene_dirs"
            del parts[-2]
            base = Path(os.path.join(*parts))
            base = base / "debug" if debug else base
            scene_dir = base / "scene_msk" / f"ds{self.config.downsample}{feat_comb}"
        else:
            scene_dir = Path(cache_dir) / "ds8lin"
        scene_dir.mkdir(parents=True, exist_ok=True)
        loader = self.data_loader
        frames = loader if hasattr(loader, "__getitem__") else list(loader)
        mine = nmdist.shard_indices(len(frames))
        written = []
        # Round 6: the files of launch group g are pickled and written by a worker thread while group g + 1 renders (the loop was serial:
        # render, synchronous read-back, np.save -- the GPU idle for the ~1.3 ms per frame the host spent on the last two).  The read-back goes
        # into pinned staging buffers (`cache_writers` + 1 sets: a set is reused when its writer is done), ordered by an event the writer waits for.
        from concurrent.futures import ThreadPoolExecutor

        writers = max(1, int(getattr(self, "cache_writers", 1)))  # (2 or 3 measured no faster: scripts/perf_cache_frames.py)
        pool = ThreadPoolExecutor(max_workers=writers)
        sets = [dict(fut=None, bufs={}) for _ in range(writers + 1)]

        def staging(st, key, like, rows):
            b = st["bufs"].get(key)
            if b is None or b.shape[0] < rows or b.shape[1:] != like.shape[1:]:
                b = st["bufs"][key] = torch.empty((max(rows, 1),) + tuple(like.shape[1:]), dtype=like.dtype, pin_memory=like.is_cuda)
            return b

        try:
            for gi, a in enumerate(range(0, len(mine), frames_per_launch)):
                group = [frames[i] for i in mine[a:a + frames_per_launch]]
                parsed = [self._parse(b) for b in group]
                counts = [p[2].shape[0] for p in parsed]
                ts = None
                if parsed[0][3] is not None:
                    ts = torch.cat([p[3].to("cpu") for p in parsed])
                # a frame dict holds fine-pass outputs only (pts_fine, feat_fine, rgb_fine): the lean render (coarse pass reduced to the
                # weights that place the fine samples) computes exactly those
                preds = self.model.predict(torch.cat([p[2] for p in parsed]), 1, 1, out_raw=True, ray_id=ts, **{"lean": True, **predict_kw})
                pts, feat, rgb = preds["pts_fine"], preds["feat_fine"], preds["rgb_fine"].reshape(-1, 3).clamp(0, 1)
                st = sets[gi % len(sets)]
                if st["fut"] is not None:
                    st["fut"].result()  # (its staging buffers are free again; a writer's exception surfaces here)
                total = sum(counts)
                h_pts, h_feat, h_rgb = staging(st, "pts", pts, total), staging(st, "feat", feat, total), staging(st, "rgb", rgb, total)
                jobs, off = [], 0
                for b, n in zip(group, counts):
                    unnorm_scene = torch.eye(4)
                    p3 = pts[off:off + n]
                    if "unnorm_scene" in b:
                        unnorm_scene = torch.as_tensor(b["unnorm_scene"][0]).cpu().float()
                        p3 = self.unnorm(unnorm_scene, p3)
                    h_pts[off:off + n].copy_(p3, non_blocking=True)
                    # ((buffer, offset, rows): the writer slices the staging buffers; a pickled slice carries its own rows only)
                    scene_pts = dict(pt3d=(h_pts, off, n), unnorm_scene=unnorm_scene.numpy(), pt_feat=(h_feat, off, n), pt_color=(h_rgb, off, n))
                    if "sky_mask" in b:
                        scene_pts["sky_mask"] = torch.as_tensor(b["sky_mask"]).cpu().numpy()
                    if "valid_mask" in b:
                        scene_pts["mask"] = torch.as_tensor(b["valid_mask"]).squeeze().cpu().numpy()
                    path = scene_dir / f"{b['img_idx'][0]}.npy"
                    jobs.append((path, scene_pts))
                    written.append(path)
                    off += n
                h_feat[:total].copy_(feat[:total], non_blocking=True)
                h_rgb[:total].copy_(rgb[:total], non_blocking=True)
                ev = torch.cuda.Event() if pts.is_cuda else None
                if ev is not None:
                    ev.record()

                def run(ev=ev, jobs=jobs):
                    if ev is not None:
                        ev.synchronize()
                    for path, d in jobs:
                        np.save(path, {k: (v[0][v[1]:v[1] + v[2]].numpy() if isinstance(v, tuple) else v) for k, v in d.items()})

                st["fut"] = pool.submit(run)
                if debug and a > 10:
                    break
            for st in sets:
                if st["fut"] is not None:
                    st["fut"].result()
        finally:
            pool.shutdown(wait=True)
        self.model.ret_pfeat = False
        return written

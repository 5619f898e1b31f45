"""Scene-feature cache: render every reference frame once with the HIP renderer and
store per-frame pickled-dict .npy files in the reference's on-disk format, so its NeRFMatchPair dataset reader
(`load_frame_3d`, nerfmatch/datasets/data_loading.py:36-80) can consume them unchanged.

Writer side of NerfEvaluator.cache_scene_pts (nerfmatch/nerf_evaluator.py:308-402):
    {pt3d (N,3) world, unnorm_scene (4,4), pt_feat (N,256), pt_color (N,3) clamped to [0,1]}   + optional masks
Frames are independent, so they shard over ranks exactly like queries (nerfmatch_amd.dist.shard_indices) and are
rendered `batch` at a time (one launch per kernel over batch*R rays)."""
from pathlib import Path

import numpy as np
import torch

from . import dist as nmdist


def cache_scene_pts(renderer, frames, K, img_hw, unnorm_scene, out_dir, device, feat_comb="lin", downsample=8, batch=4,
                    rank=None, world_size=None):
    """frames: list of (name, c2w 4x4 world pose).  Returns the list of files written by THIS rank."""
    out_dir = Path(out_dir) / f"ds{downsample}{feat_comb}"
    out_dir.mkdir(parents=True, exist_ok=True)
    renderer.ret_pfeat, renderer.feat_comb = True, feat_comb
    unnorm = torch.as_tensor(unnorm_scene).detach().to("cpu", torch.float32)
    mine = nmdist.shard_indices(len(frames), rank, world_size)
    written = []
    for a in range(0, len(mine), batch):
        idx = mine[a:a + batch]
        poses = torch.stack([torch.as_tensor(frames[i][1]).float() for i in idx])
        out = renderer.render_novel_views(img_hw, K, poses, unnorm, device, downsample=downsample, lean=True)
        pt3d, feat = out["pt3d"].cpu().numpy(), out["pt_feat"].cpu().numpy()
        color = out["im_pred"].reshape(len(idx), -1, 3).clamp(0, 1).cpu().numpy()
        for j, i in enumerate(idx):
            path = out_dir / f"{frames[i][0]}.npy"
            np.save(path, dict(pt3d=pt3d[j], unnorm_scene=unnorm.numpy(), pt_feat=feat[j], pt_color=color[j]))
            written.append(path)
    renderer.ret_pfeat = False
    return written


def load_frame_3d(path):
    """Reader with the reference's semantics (data_loading.py:36-80, no masks): (pt3d, pt_feat, mask, unnorm_scene)."""
    d = np.load(path, allow_pickle=True).item()
    return d["pt3d"], d["pt_feat"], np.ones(len(d["pt3d"]), dtype=np.bool_), d["unnorm_scene"]

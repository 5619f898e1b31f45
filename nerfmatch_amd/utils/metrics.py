"""Host-side evaluation statistics used by the evaluators (numpy / tiny torch ops on K-element lists; not on the hot path).

Restated from the reference's definitions (nerfmatch/utils/metrics.py): scene-dependent pose thresholds :27-42, validation
PSNR of `compute_nerf_metrics` :59-96, `cal_error_auc` :340-352, `pose_recall` :355-356, `pose_err` :359-369,
`summarize_pose_statis` :545-595, `average_pose_metrics` :598-606 -- the evaluators return / print the same quantities, so
result files written here and by the reference are interchangeable."""
import math
from argparse import Namespace

import numpy as np
import torch

# (translation cm, rotation deg) thresholds following DSAC* (reference :27-42)
POSE_THRES = {
    "GreatCourt": [(5, 45)], "KingsCollege": [(5, 38)], "OldHospital": [(5, 22)], "ShopFacade": [(5, 15)], "StMarysChurch": [(5, 35)],
    "chess": [(5, 5)], "fire": [(5, 5)], "heads": [(5, 5)], "office": [(5, 5)], "pumpkin": [(5, 5)], "redkitchen": [(5, 5)], "stairs": [(5, 5)],
}


def pose_err(gt_pose, est_pose):
    """(rotation error in degrees, translation error) between two c2w poses; the Rodrigues norm of R_est R_gt^T the
    reference takes (cv2.Rodrigues) is the rotation angle."""
    gt_pose, est_pose = torch.as_tensor(gt_pose).double().cpu(), torch.as_tensor(est_pose).double().cpu()
    t_err = float(torch.norm(gt_pose[:3, 3] - est_pose[:3, 3]))
    rel = est_pose[:3, :3] @ gt_pose[:3, :3].T
    cos = max(-1.0, min(1.0, (float(torch.trace(rel)) - 1.0) / 2.0))
    return math.degrees(math.acos(cos)), t_err


def pose_recall(r_errs, t_errs, r_thres, t_thres):
    return ((np.array(r_errs) < r_thres) & (np.array(t_errs) < t_thres)).mean() * 100


def cal_error_auc(errors, thresholds):
    if len(errors) == 0:
        return np.zeros(len(thresholds))
    n = len(errors)
    errors = np.append([0.0], np.sort(errors))
    recalls = np.arange(n + 1) / n
    aucs = []
    for thres in thresholds:
        last = np.searchsorted(errors, thres)
        rcs = np.append(recalls[:last], recalls[last - 1])
        ers = np.append(errors[:last], thres)
        aucs.append(np.sum((ers[1:] - ers[:-1]) * (rcs[1:] + rcs[:-1]) / 2.0) / thres)  # trapezoid rule (np.trapz)
    return np.array(aucs) * 100


def summarize_pose_statis(statis, pose_thres=(1, 2, 5, 10), auc_thres=(1, 2, 5, 10), t_unit="?", t_scale=1, print_out=True):
    printf = print if print_out else (lambda *_: None)
    if isinstance(statis, dict):
        statis = Namespace(**statis)
    pose_thres = [(th, th) if np.isscalar(th) else tuple(th) for th in pose_thres]
    r_errs, t_errs = np.asarray(statis.R_err, dtype=np.float64), t_scale * np.asarray(statis.t_err, dtype=np.float64)
    printf(f"\nSamples: {len(r_errs)} t_unit={t_unit} t_scale={t_scale}")
    if "num_matches" in statis:
        printf(f"Mean matches: {np.mean(statis.num_matches):.0f}")
    t_med, r_med = np.median(t_errs), np.median(r_errs)
    printf(f"Median Error: {t_med:.1f}/{r_med:.1f} {t_unit}/deg")
    pose_rec = np.array([pose_recall(r_errs, t_errs, rth, tth) for rth, tth in pose_thres])
    printf(f"Recall@{pose_thres}{t_unit}/deg: {pose_rec}%")
    printf(f"AUC@{list(auc_thres)}{t_unit}/deg: {cal_error_auc(np.maximum(t_errs, r_errs), auc_thres)}%")
    summary = {"t_med": t_med, "r_med": r_med, "recall": pose_rec[0]}
    if "match_time" in statis:
        summary["match_time"] = float(np.mean(statis.match_time) * 1000)
        printf(f"Avg match time: {summary['match_time']:.1f}ms")
    return summary


def average_pose_metrics(metr_all, print_out=True):
    avg = {k: float(np.mean([m[k] for m in metr_all])) for k in metr_all[0]}
    if print_out:
        print(f"\nAverage metrics of {len(metr_all)} (scene) caches:")
        print(f"Median pose error(cm/deg): {avg['t_med']:.1f}/{avg['r_med']:.1f}")
        print(f"Recall(%): {avg['recall']:.1f}")
    return avg


def compute_nerf_metrics(preds, rgb_gt, mask_loss=None):
    """Validation branch of the reference's compute_nerf_metrics: 0.5 * mean(mask * (rgb - gt)^2) and its PSNR for the
    coarse and fine images (the 0.5 is the reference's)."""
    m = torch.round(mask_loss) if mask_loss is not None else 1
    out = {}
    for key in ("coarse", "fine"):
        if f"rgb_{key}" in preds:
            mse = 0.5 * (m * (preds[f"rgb_{key}"] - rgb_gt) ** 2).mean()
            out[f"rgb_{key}_mse"], out[f"rgb_{key}_psnr"] = mse, -10 * torch.log10(mse)
    if "rgb_fine_mse" not in out:
        out["rgb_fine_mse"], out["rgb_fine_psnr"] = out["rgb_coarse_mse"], out["rgb_coarse_psnr"]
    return out

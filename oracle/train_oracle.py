"""ORACLE (test infrastructure, NOT product code) -- training step of the coarse-to-fine matcher head.

CPU restatement (torch-CPU fp32, gradients by torch autograd over the restated forward) of SURVEY.md section 8f rank 4:
the losses of NeRFMatcherMS.forward_with_metrics.  Only tests/ may import this.

Parity status: PINNED against tests/golden/matcher_train.npz (loss values, sampled match lists and gradients produced by
importing the reference, tests/golden/make_golden.py::train_fixture); the backbone and kornia caveats of
matcher_oracle.py apply.
"""
import torch

from . import matcher_oracle as mo


def matching_loss(conf, conf_gt, alpha=0.25, gamma=2.0):
    """Focal loss on the dual-softmax confidence.  nerfmatch/utils/metrics.py:372-380."""
    conf = torch.clamp(conf, 1e-6, 1 - 1e-6)
    pos, neg = conf_gt == 1, conf_gt == 0
    loss_pos = -alpha * torch.pow(1 - conf[pos], gamma) * conf[pos].log()
    loss_neg = -alpha * torch.pow(conf[neg], gamma) * (1 - conf[neg]).log()
    return loss_pos.mean() + loss_neg.mean()


def fine_match_loss_l2_std(mpt2d_f, mpt2d_f_gt, std, mask=None):
    """nerfmatch/utils/metrics.py:425-451: squared pixel distance weighted by the (detached) normalised inverse std."""
    inverse_std = 1.0 / torch.clamp(std, min=1e-10)
    weight = (inverse_std / torch.mean(inverse_std)).detach()
    if mask is None:
        mask = torch.ones_like(weight)
    if mask.sum() == 0:
        mask = mask.clone()
        mask[0] = True
        weight[0] = 0.0
    flow_l2 = ((mpt2d_f - mpt2d_f_gt) ** 2).sum(-1)
    return (flow_l2 * weight * mask).mean()


def fine_loss_l2_std(expec_f, expec_f_gt):
    """LoFTR's window-level loss.  nerfmatch/utils/metrics.py:393-422."""
    correct = torch.linalg.norm(expec_f_gt, ord=float("inf"), dim=1) < 1
    inverse_std = 1.0 / torch.clamp(expec_f[:, 2], min=1e-10)
    weight = (inverse_std / torch.mean(inverse_std)).detach()
    if not correct.any():
        correct = correct.clone()
        correct[0] = True
        weight[0] = 0.0
    flow_l2 = ((expec_f_gt[correct] - expec_f[correct, :2]) ** 2).sum(-1)
    return (flow_l2 * weight[correct]).mean()


def feat_l2(im_feat, pt_feat, conf_gt):
    """nerfmatch/utils/metrics.py:383-390."""
    out = []
    for i, iconf in enumerate(conf_gt):
        im_ids, pt_ids = torch.where(iconf)
        out.append((im_feat[i][im_ids] - pt_feat[i][pt_ids]).norm(dim=-1).mean())
    return torch.stack(out).mean()


def c2f_train_step(p, cfg, cfeat_map, ffeat_map, pt_feat, pt3d, pt2d, pt2d_proj, conf_gt, im_mask=None, pt_mask=None,
                   coarse_only=False, fine_loss="match"):
    """forward(training=True) + the loss statements of forward_with_metrics (nerfmatch/nerfmatch_c2f_trainer.py:490-551,
    pose metrics omitted).  Returns dict(coarse_loss, fine_loss, loss, preds)."""
    preds = mo.c2f_forward_match(p, cfg, cfeat_map, ffeat_map, pt_feat, pt3d, im_mask, pt_mask, mutual=False, match_thres=0.0,
                                 conf_gt=conf_gt)
    coarse_loss = matching_loss(preds["conf_matrix"], conf_gt)
    out = dict(coarse_loss=coarse_loss, preds=preds, feat_l2=feat_l2(preds["im_cfeat"], preds["pt_cfeat"], conf_gt))
    b_ids, i_ids, j_ids = preds["match_ids"]
    if len(i_ids) == 0 or coarse_only:
        out["loss"] = coarse_loss
        return out
    mpt2d_c = pt2d[b_ids, i_ids]
    mpt2d_f = mpt2d_c + preds["expec_f"][:, :2] * getattr(cfg, "win_sz", 5) / 2 * 2
    mpt2d_f_gt = pt2d_proj[b_ids, j_ids]
    coarse_pos = (mpt2d_f_gt - mpt2d_c).norm(dim=-1) < getattr(cfg, "coarse_dthres", 20)
    if fine_loss == "match":
        fine_loss = fine_match_loss_l2_std(mpt2d_f, mpt2d_f_gt, preds["expec_f"][:, 2], mask=coarse_pos)
    else:  # "exp": radius = fine_ds * win_sz // 2  (c2f_trainer.py:545-547)
        fine_loss = fine_loss_l2_std(preds["expec_f"], (mpt2d_f_gt - mpt2d_c) / (2 * getattr(cfg, "win_sz", 5) // 2))
    out.update(fine_loss=fine_loss, loss=coarse_loss + fine_loss, coarse_pos=coarse_pos)
    return out

"""CPU: the trainer's config-driven optimiser / schedule (a reference `optim:` block has no `lr`)."""
from argparse import Namespace

import pytest
import torch

from nerfmatch_amd import synth
from nerfmatch_amd.trainer import NeRFMatchMSTrainer


def make(optim, **top):
    cfg = Namespace(model=synth.matcher_config("c2f"), optim=Namespace(coarse_only_epochs=0, **optim), **top)
    return NeRFMatchMSTrainer(cfg, device="cpu")


def test_reference_yaml_optim_block():
    """configs/nerfmatch/nerfmatch_7scenes_sfm_c2f.yaml:21-27 + exp.batch_size 2 on 8 GPUs: lr = clr * 8 * 2 / cbs, Adam, cosine."""
    tr = make(dict(optimizer="adam", adapt_lr=True, clr=0.0004, cbs=16, weight_decay=0.0, lr_scheduler="cosine"),
              gpu_num=8, exp=Namespace(batch_size=2, max_epochs=30))
    opt, sch = tr.configure_optimizers()
    assert isinstance(opt, torch.optim.Adam) and abs(opt.param_groups[0]["lr"] - 0.0004) < 1e-12
    assert isinstance(sch, torch.optim.lr_scheduler.CosineAnnealingLR) and sch.T_max == 30
    tr = make(dict(optimizer="adamw", adapt_lr=False, clr=0.001, cbs=16, weight_decay=0.01, lr_scheduler="steplr", decay_step=[3, 6], decay_gamma=0.5,
                   max_epochs=9), gpu_num=4, exp=Namespace(batch_size=3))
    opt, sch = tr.configure_optimizers()
    assert isinstance(opt, torch.optim.AdamW) and opt.param_groups[0]["lr"] == 0.001 and opt.param_groups[0]["weight_decay"] == 0.01
    assert isinstance(sch, torch.optim.lr_scheduler.MultiStepLR)


def test_explicit_lr_and_errors():
    tr = make(dict(lr=0.002))
    opt, sch = tr.configure_optimizers()
    assert opt.param_groups[0]["lr"] == 0.002 and sch is None
    with pytest.raises(ValueError, match="optimizer_factory"):
        make(dict(optimizer="ranger", lr=0.1)).configure_optimizers()
    with pytest.raises(ValueError, match="scheduler_factory"):
        make(dict(lr=0.1, lr_scheduler="poly")).configure_optimizers()
    with pytest.raises(ValueError, match="clr"):
        make(dict()).configure_optimizers()


def test_gpu_num_defaults_to_world_size_and_steplr_needs_epochs(monkeypatch):
    """the reference's yamls carry no gpu_num (train() sets it from the device count) -- under an N-rank launch the adaptive
    rate must use N, not 1; steplr with decay_per_step and no epoch count raises the descriptive error, not int(None)."""
    from nerfmatch_amd import dist as nmdist

    monkeypatch.setattr(nmdist, "world", lambda: (0, 4))
    monkeypatch.setattr(nmdist, "broadcast_module", lambda *a, **k: None)
    monkeypatch.setattr(nmdist.GradBuckets, "__init__", lambda self, *a, **k: None)
    tr = make(dict(optimizer="adam", adapt_lr=True, clr=0.0004, cbs=16), exp=Namespace(batch_size=2))
    assert tr.gpu_num == 4 and abs(tr.learning_rate() - 0.0004 * 4 * 2 / 16) < 1e-15
    assert make(dict(lr=0.1), gpu_num=2).gpu_num == 2
    with pytest.raises(ValueError, match="max_epochs"):
        make(dict(lr=0.1, lr_scheduler="steplr", decay_per_step=3, decay_gamma=0.5)).configure_optimizers()

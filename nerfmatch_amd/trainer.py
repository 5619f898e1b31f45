"""Training loop of the matcher models without pytorch-lightning (the reference wraps the same calls in LightningModules:
NeRFMatchMSTrainer nerfmatch_c2f_trainer.py:554-650, NeRFMatchCoarseTrainer nerfmatch_coarse_trainer.py:390-470, launched by
train() :793-860 with Lightning's DDP plugin).  Host-side plumbing only: the model's forward_with_metrics builds the graph whose
forward and backward passes are the HIP kernels (nerfmatch_amd.autograd); data parallelism is one process per GPU with
nerfmatch_amd.dist.GradBuckets (initial weights broadcast from rank 0, RCCL all-reduce of flat gradient buckets overlapped
with the backward pass).

Optimiser / schedule: `optimizer_factory(params)` / `scheduler_factory(optimizer)` when given; otherwise the `optim:` block of
the reference's yamls is honoured (configs/nerfmatch/*.yaml:21-27 carry `optimizer`, `adapt_lr`, `clr`, `cbs`, `weight_decay`,
`lr_scheduler` and NO `lr`): the rate is `optim.lr` if present, else clr * gpu_num * batch_size / cbs when `adapt_lr` (the
default; nerfmatch_c2f_trainer.py:666-671, 778-783) and `clr` otherwise; optimizer sgd / adam / adamw with `weight_decay`
(utils/optim.py:25-58); `lr_scheduler` cosine (T_max = max_epochs, floor 1e-8) or steplr (utils/optim.py:61-75).  Any other
value raises with the name of the factory argument to use instead (the reference's remaining table -- rmsprop, radam, ranger,
poly, chained, warm-up -- is host-side configuration outside SURVEY section 8)."""
import torch

from . import dist as nmdist
from .matcher import NeRFMatcherCoarse, NeRFMatcherMS


class _TrainerBase:
    model_cls = None

    def __init__(self, config, device="cuda", bucket_mb=64, optimizer_factory=None, scheduler_factory=None):
        self.config = config
        self.model = self.model_cls(config.model).to(device)
        self.rthres = getattr(config.model, "rthres", 1)
        # the reference's train() sets config.gpu_num from the device count before config_adaptive_lr runs and its yamls carry no
        # gpu_num (nerfmatch_c2f_trainer.py:793-800): under an N-rank launch the default is the world size, not 1
        self.gpu_num = getattr(config, "gpu_num", None) or nmdist.world()[1]
        self.current_epoch = 0
        self.optimizer = self.scheduler = None
        self._opt_factory, self._sched_factory = optimizer_factory, scheduler_factory
        # replicas start from rank 0's weights (what Lightning's DDP plugin does in the reference); buffers included
        nmdist.broadcast_module(self.model, src=0)
        self.buckets = nmdist.GradBuckets(self.model.parameters(), bucket_mb=bucket_mb)

    def learning_rate(self):
        """`optim.lr` if set; else the reference's batch-size-adaptive rule on (clr, cbs)."""
        o = self.config.optim
        if getattr(o, "lr", None) is not None:
            return float(o.lr)
        if not hasattr(o, "clr"):
            raise ValueError("config.optim has neither `lr` nor `clr`: set one, or pass optimizer_factory=")
        if not getattr(o, "adapt_lr", True):
            return float(o.clr)
        exp = getattr(self.config, "exp", None)
        batch = getattr(exp, "batch_size", None)
        if batch is None or not hasattr(o, "cbs"):
            raise ValueError("optim.adapt_lr needs exp.batch_size and optim.cbs (lr = clr * gpu_num * batch_size / cbs); set optim.lr or adapt_lr: False otherwise")
        return float(o.clr) * self.gpu_num * batch / float(o.cbs)

    def configure_optimizers(self):
        params = self.model.parameters()
        o = self.config.optim
        if self._opt_factory:
            self.optimizer = self._opt_factory(params)
        else:
            lr, wd, eps = self.learning_rate(), float(getattr(o, "weight_decay", 0.0)), float(getattr(o, "eps", 1e-8))
            kind = getattr(o, "optimizer", "adam")
            if kind == "adam":
                self.optimizer = torch.optim.Adam(params, lr=lr, eps=eps, weight_decay=wd)
            elif kind == "adamw":
                self.optimizer = torch.optim.AdamW(params, lr=lr, eps=eps, weight_decay=wd)
            elif kind == "sgd":
                self.optimizer = torch.optim.SGD(params, lr=lr, momentum=float(getattr(o, "momentum", 0.0)), weight_decay=wd)
            else:
                raise ValueError(f"optim.optimizer = {kind!r} is not built in (adam, adamw, sgd are): pass optimizer_factory=")
        if self._sched_factory:
            self.scheduler = self._sched_factory(self.optimizer)
        elif not self._opt_factory and getattr(o, "lr_scheduler", None) is not None:
            sch = o.lr_scheduler
            epochs = getattr(o, "max_epochs", None) or getattr(getattr(self.config, "exp", None), "max_epochs", None)
            if sch == "cosine":
                if epochs is None:
                    raise ValueError("lr_scheduler 'cosine' needs optim.max_epochs (or exp.max_epochs)")
                self.scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=int(epochs), eta_min=1e-8)
            elif sch == "steplr":
                step = getattr(o, "decay_per_step", None)
                if step and epochs is None:
                    raise ValueError("lr_scheduler 'steplr' with optim.decay_per_step needs optim.max_epochs (or exp.max_epochs); give optim.decay_step otherwise")
                miles = list(range(step, int(epochs), step)) if step else list(o.decay_step)
                self.scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=miles, gamma=float(o.decay_gamma))
            else:
                raise ValueError(f"optim.lr_scheduler = {sch!r} is not built in (cosine, steplr are): pass scheduler_factory=")
        return self.optimizer, self.scheduler

    def model_forward(self, batch, training=False):
        raise NotImplementedError

    def training_step(self, data, batch_idx=0):
        """forward + backward + gradient all-reduce + optimiser step; returns the metrics of the step (detached scalars)."""
        if self.optimizer is None:
            self.configure_optimizers()
        with torch.enable_grad():
            metrics = self.model_forward(data, training=True)
            self.optimizer.zero_grad(set_to_none=True)
            self.buckets.start()
            metrics["loss"].backward()
        self.buckets.finish()
        self.optimizer.step()
        return {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in metrics.items()}

    def validation_step(self, data, batch_idx=0):
        with torch.no_grad():  # the modules leave the autograd path when gradients are disabled
            return self.model_forward(data, training=False)

    def on_epoch_end(self):
        self.current_epoch += 1
        if self.scheduler is not None:
            self.scheduler.step()

    def fit(self, loader, max_epochs=1, log=None):
        from ._lib import steady_gc

        for _ in range(max_epochs):
            self.model.train()
            # (the process's resident objects frozen for the epoch, as in the evaluator's loops: a step allocates ~10^3 short-lived objects and,
            # unfrozen, one step in a few pays a full pass of the cyclic collector over everything torch has loaded -- 20 ms on a 10 ms step)
            with steady_gc():
                for i, batch in enumerate(loader):
                    m = self.training_step(batch, i)
                    if log is not None:
                        log(self.current_epoch, i, m)
            self.on_epoch_end()


class NeRFMatchMSTrainer(_TrainerBase):
    model_cls = NeRFMatcherMS

    def __init__(self, config, device="cuda", bucket_mb=64, **kw):
        super().__init__(config, device, bucket_mb, **kw)
        self.coarse_only_epochs = getattr(config.optim, "coarse_only_epochs", 0)

    def model_forward(self, batch, training=False, oracle=False):
        coarse_only = self.current_epoch < self.coarse_only_epochs
        return self.model.forward_with_metrics(batch, rthres=self.rthres, training=training, coarse_only=coarse_only, oracle=oracle)


class NeRFMatchCoarseTrainer(_TrainerBase):
    model_cls = NeRFMatcherCoarse

    def model_forward(self, batch, training=False):
        return self.model.forward_with_metrics(batch, rthres=self.rthres)

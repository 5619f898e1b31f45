"""Training loop of the matcher models without pytorch-lightning (the reference wraps the same calls in LightningModules:
NeRFMatchMSTrainer nerfmatch_c2f_trainer.py:554-650, NeRFMatchCoarseTrainer nerfmatch_coarse_trainer.py:390-470, launched by
train() :793-860 with Lightning's DDP plugin).  Host-side plumbing only: the model's forward_with_metrics builds the graph whose
forward and backward passes are the HIP kernels (nerfmatch_amd.autograd); data parallelism is one process per GPU with
nerfmatch_amd.dist.GradBuckets (initial weights broadcast from rank 0, RCCL all-reduce of flat gradient buckets overlapped
with the backward pass).  The optimiser / learning-rate schedule table of the reference (utils/optim.py) is out of scope
(SURVEY.md section 2 row 20): pass `optimizer_factory(params) -> torch.optim.Optimizer` (and optionally
`scheduler_factory(optimizer)`); the default is Adam at `config.optim.lr`."""
import torch

from . import dist as nmdist
from .matcher import NeRFMatcherCoarse, NeRFMatcherMS


class _TrainerBase:
    model_cls = None

    def __init__(self, config, device="cuda", bucket_mb=64, optimizer_factory=None, scheduler_factory=None):
        self.config = config
        self.model = self.model_cls(config.model).to(device)
        self.rthres = getattr(config.model, "rthres", 1)
        self.gpu_num = getattr(config, "gpu_num", 1)
        self.current_epoch = 0
        self.optimizer = self.scheduler = None
        self._opt_factory, self._sched_factory = optimizer_factory, scheduler_factory
        # replicas start from rank 0's weights (what Lightning's DDP plugin does in the reference); buffers included
        nmdist.broadcast_module(self.model, src=0)
        self.buckets = nmdist.GradBuckets(self.model.parameters(), bucket_mb=bucket_mb)

    def configure_optimizers(self):
        params = self.model.parameters()
        self.optimizer = self._opt_factory(params) if self._opt_factory else torch.optim.Adam(params, lr=self.config.optim.lr)
        if self._sched_factory:
            self.scheduler = self._sched_factory(self.optimizer)
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
        for _ in range(max_epochs):
            self.model.train()
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

"""Training loop of the matcher models without pytorch-lightning (the reference wraps the same calls in LightningModules:
NeRFMatchMSTrainer nerfmatch_c2f_trainer.py:554-650, NeRFMatchCoarseTrainer nerfmatch_coarse_trainer.py:390-470, launched by
train() :793-860 with Lightning's DDP plugin).  Host-side plumbing only: the model's forward_with_metrics builds the graph whose
forward and backward passes are the HIP kernels (nerfmatch_amd.autograd); the optimiser and the learning-rate schedule are
torch.optim objects configured from the reference's `optim:` block (utils/optim.py:25-100); data parallelism is one process
per GPU with nerfmatch_amd.dist.GradBuckets (RCCL all-reduce of flat gradient buckets overlapped with the backward pass)."""
import torch
from torch.optim.lr_scheduler import CosineAnnealingLR, MultiStepLR

from . import dist as nmdist
from .matcher import NeRFMatcherCoarse, NeRFMatcherMS


def config_adaptive_lr(optim_conf, batch_size, gpu_num=1):
    """lr = clr * (gpu_num * batch_size) / cbs   (config_adaptive_lr, nerfmatch_c2f_trainer.py:666-671)."""
    true_batch = gpu_num * batch_size
    return optim_conf.clr * true_batch / optim_conf.cbs, true_batch


def init_optimizer(config, parameters, eps=1e-8):
    """utils/optim.py:25-59 (sgd / adam / adamw / rmsprop / radam)."""
    eps = float(getattr(config, "eps", eps))
    kind = config.optimizer
    if kind == "sgd":
        return torch.optim.SGD(parameters, lr=config.lr, momentum=config.momentum, weight_decay=config.weight_decay)
    table = dict(adam=torch.optim.Adam, adamw=torch.optim.AdamW, rmsprop=torch.optim.RMSprop, radam=torch.optim.RAdam)
    if kind not in table:
        raise ValueError("optimizer not recognized!")
    return table[kind](parameters, lr=config.lr, eps=eps, weight_decay=config.weight_decay)


def init_scheduler(config, optimizer):
    """utils/optim.py:62-100: per-epoch 'steplr' or 'cosine' (the shipped configs use cosine)."""
    if config.lr_scheduler == "steplr":
        step = getattr(config, "decay_per_step", None)
        milestones = list(range(step, config.max_epochs, step)) if step else config.decay_step
        return MultiStepLR(optimizer, milestones=milestones, gamma=config.decay_gamma)
    if config.lr_scheduler == "cosine":
        return CosineAnnealingLR(optimizer, T_max=config.max_epochs, eta_min=1e-8)
    raise ValueError("scheduler not recognized!")


class _TrainerBase:
    model_cls = None

    def __init__(self, config, device="cuda", bucket_mb=64):
        self.config = config
        self.model = self.model_cls(config.model).to(device)
        self.rthres = getattr(config.model, "rthres", 1)
        self.gpu_num = getattr(config, "gpu_num", 1)
        self.current_epoch = 0
        self.optimizer = self.scheduler = None
        self.buckets = nmdist.GradBuckets(self.model.parameters(), bucket_mb=bucket_mb)

    def configure_optimizers(self):
        conf = self.config.optim
        self.optimizer = init_optimizer(conf, self.model.parameters())
        if getattr(conf, "lr_scheduler", None) is not None:
            self.scheduler = init_scheduler(conf, self.optimizer)
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

    def __init__(self, config, device="cuda", bucket_mb=64):
        super().__init__(config, device, bucket_mb)
        self.coarse_only_epochs = getattr(config.optim, "coarse_only_epochs", 0)

    def model_forward(self, batch, training=False, oracle=False):
        coarse_only = self.current_epoch < self.coarse_only_epochs
        return self.model.forward_with_metrics(batch, rthres=self.rthres, training=training, coarse_only=coarse_only, oracle=oracle)


class NeRFMatchCoarseTrainer(_TrainerBase):
    model_cls = NeRFMatcherCoarse

    def model_forward(self, batch, training=False):
        return self.model.forward_with_metrics(batch, rthres=self.rthres)

"""Import-path compatibility: the reference keeps NeRFMatcherMS and its LightningModule in nerfmatch/nerfmatch_c2f_trainer.py
(:77, :554).  The trainer here is the Lightning-free loop of nerfmatch_amd/trainer.py."""
from .matcher import NeRFMatcherMS  # noqa: F401
from .trainer import NeRFMatchMSTrainer  # noqa: F401

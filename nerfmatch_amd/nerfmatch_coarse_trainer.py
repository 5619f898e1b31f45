"""Import-path compatibility: the reference keeps NeRFMatcherCoarse and its LightningModule in
nerfmatch/nerfmatch_coarse_trainer.py (:50, :390)."""
from .matcher import NeRFMatcherCoarse  # noqa: F401
from .trainer import NeRFMatchCoarseTrainer  # noqa: F401

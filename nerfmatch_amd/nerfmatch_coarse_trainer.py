"""Import-path compatibility: the reference keeps NeRFMatcherCoarse in nerfmatch/nerfmatch_coarse_trainer.py:50."""
from .matcher import NeRFMatcherCoarse  # noqa: F401

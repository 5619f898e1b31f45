"""Import-path compatibility: the reference keeps NeRFMatcherMS in nerfmatch/nerfmatch_c2f_trainer.py:77.
Only the model class is provided (training is out of scope)."""
from .matcher import NeRFMatcherMS  # noqa: F401

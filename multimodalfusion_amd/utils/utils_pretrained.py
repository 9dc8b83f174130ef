"""Helpers of the stage-2 pipeline under the reference's names (utils/utils_pretrained.py)."""
from .utils import initialize_weights  # noqa: F401  (utils/utils_pretrained.py:145-154 is the same initialiser)

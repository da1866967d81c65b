"""Test stand-in for ``torchvision`` (absent from this image); the reference's dataset modules import
``torchvision.transforms`` at module level (src/datasets/uavid.py:13)."""
from . import transforms  # noqa: F401

"""`from pytorch_modules.backbones.mobilenet import InvertedResidual` (reference models/unet.py:8)."""
from pytorch_segmentation_amd.backbones.mobilenet import InvertedResidual  # noqa: F401

"""`from models import DeepLabV3Plus, HRNet, UNet` -- the import the reference's scripts use (models/__init__.py:1-3)."""
from pytorch_segmentation_amd.models import ASPP, ASPPPooling, DeepLabV3Plus, HRNet, UNet  # noqa: F401

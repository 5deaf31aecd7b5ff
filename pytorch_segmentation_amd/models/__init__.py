"""`from models import DeepLabV3Plus, HRNet, UNet` of the reference (models/__init__.py:1-3)."""
from .aspp import ASPP, ASPPPooling
from .deeplabv3plus import DeepLabV3Plus
from .hrnet import HRNet
from .unet import UNet

__all__ = ['ASPP', 'ASPPPooling', 'DeepLabV3Plus', 'HRNet', 'UNet']

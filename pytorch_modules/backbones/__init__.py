"""`from pytorch_modules.backbones import resnet50, mobilenet_v2` (reference models/deeplabv3plus.py:7, unet.py:7)."""
from pytorch_segmentation_amd.backbones import mobilenet_v2, resnet50  # noqa: F401


def resnet34(*args, **kwargs):
    raise NotImplementedError('resnet34 is imported by reference models/unet.py:7 but not used by any model there')

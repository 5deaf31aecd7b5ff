"""`from pytorch_modules.nn import ConvNormAct, ...` (reference models/aspp.py:5, deeplabv3plus.py:8, unet.py:9,
hrnet.py:11, utils/utils.py:7)."""
import torch.nn as nn

from pytorch_segmentation_amd.nn import BatchNorm2d, Conv2d, ConvNormAct  # noqa: F401


class SeparableConvNormAct(nn.Module):
    """Imported by reference models/aspp.py:5 but never instantiated there."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError('SeparableConvNormAct is not used by the reference models and has no HIP block; '
                                  'compose pytorch_segmentation_amd.nn.Conv2d (depthwise) + ConvNormAct')


class FocalBCELoss(nn.Module):
    """Instantiated at import time by reference utils/utils.py:14 and never called."""

    def forward(self, *args, **kwargs):
        raise NotImplementedError('FocalBCELoss is dead code in the reference (utils/utils.py:14); the loss on the hot '
                                  'path is pytorch_segmentation_amd.utils.compute_loss')

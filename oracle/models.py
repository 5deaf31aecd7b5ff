"""Oracle restatement of the reference's model composition (CPU torch fp32).

Test infrastructure only -- see oracle/__init__.py.  Attribute names match the
reference so state-dicts are interchangeable with the product modules and with
the reference's own classes (checked by tests/test_oracle_golden.py against
fixtures generated from the reference's files).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .backbones import mobilenet_v2, resnet50
from .blocks import ConvNormAct, initialize_weights


def _up(x, factor):
    # every decoder upsample in the reference is bilinear, align_corners=True
    # (models/deeplabv3plus.py:34-37,40-43; models/unet.py:30-55)
    return F.interpolate(x, scale_factor=factor, mode='bilinear', align_corners=True)


class ASPPPooling(nn.Module):
    """Image-level branch: GAP -> 1x1 ConvNormAct -> broadcast back to HxW.
    Follows reference models/aspp.py:8-19 (bilinear, align_corners=False from a
    1x1 source is a constant broadcast)."""

    def __init__(self, inplanes, planes):
        super().__init__()
        self.gap = nn.Sequential(nn.AdaptiveAvgPool2d(1), ConvNormAct(inplanes, planes, 1))

    def forward(self, x):
        h, w = x.shape[-2:]
        return F.interpolate(self.gap(x), size=(h, w), mode='bilinear', align_corners=False)


class ASPP(nn.Module):
    """Five parallel branches [pool, 1x1, 3x3 d=r ...] -> channel concat -> 1x1 project.
    Follows reference models/aspp.py:22-37."""

    def __init__(self, inplanes, planes, atrous_rates=(12, 24, 36)):
        super().__init__()
        branches = [ASPPPooling(inplanes, planes), ConvNormAct(inplanes, planes, 1)]
        branches += [ConvNormAct(inplanes, planes, dilation=r) for r in atrous_rates]
        self.blocks = nn.ModuleList(branches)
        self.project = ConvNormAct(planes * len(branches), planes, 1)

    def forward(self, x):
        return self.project(torch.cat([b(x) for b in self.blocks], dim=1))


class DeepLabV3Plus(nn.Module):
    """Follows reference models/deeplabv3plus.py:14-44 (R50 OS16, rates 6/12/18)."""

    def __init__(self, num_classes, backbone=None):
        super().__init__()
        self.backbone = backbone if backbone is not None else resnet50(
            replace_stride_with_dilation=[False, False, True])
        self.project = ConvNormAct(256, 128, 1)
        self.aspp = ASPP(2048, 256, [6, 12, 18])
        self.cls_conv = nn.Conv2d(384, num_classes, 3, padding=1)
        for m in (self.aspp, self.project, self.cls_conv):
            initialize_weights(m)

    def head(self, features):
        low = self.project(features[1])
        x = _up(self.aspp(features[-1]), 4)
        x = self.cls_conv(torch.cat([x, low], 1))
        return _up(x, 4)

    def forward(self, x):
        return self.head(self.backbone(x))


class UNet(nn.Module):
    """Follows reference models/unet.py:13-56 (MobileNetV2 encoder)."""

    def __init__(self, num_classes, backbone=None):
        super().__init__()
        self.backbone = backbone if backbone is not None else mobilenet_v2()
        self.up_convs = nn.ModuleList([ConvNormAct(1280, 256), ConvNormAct(352, 128),
                                       ConvNormAct(160, 64)])
        self.cls_conv = nn.Conv2d(88, num_classes, 3, padding=1)
        initialize_weights(self.up_convs)
        initialize_weights(self.cls_conv)

    def head(self, features):
        _, x2, x3, x4, x = features  # the stride-2 map is unused (models/unet.py:28)
        for conv, skip in zip(self.up_convs, (x4, x3, x2)):
            x = torch.cat([_up(conv(x), 2), skip], 1)
        x = self.cls_conv(_up(x, 2))
        return _up(x, 2)

    def forward(self, x):
        return self.head(self.backbone(x))

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


# ------------------------------------------------------------------------------------------------- HRNet
class HRBasicBlock(nn.Module):
    """Two 3x3 conv+BN with an identity (or projected) residual.  Follows reference models/hrnet.py:27-56."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        return self.relu(self.bn2(self.conv2(y)) + skip)


def _projection(cin, cout, stride):
    return nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))


class HRModule(nn.Module):
    """Parallel-resolution branches followed by the all-to-all fusion.  Follows reference models/hrnet.py:107-229:
    branch i = num_blocks[i] residual blocks; output i = relu(sum_j f_ij(x_j)) with f_ij = identity (j == i),
    1x1 ConvNormAct + x2^(j-i) bilinear align_corners=False (j > i), or (i-j) stride-2 3x3 ConvNormActs of which the
    last has no activation (j < i; the non-final ones sit in their own nn.Sequential, :216-220)."""

    def __init__(self, num_branches, block, num_blocks, num_inchannels, num_channels, multi_scale_output=True):
        super().__init__()
        if not (num_branches == len(num_blocks) == len(num_channels) == len(num_inchannels)):
            raise ValueError('branch / block / channel list lengths differ')
        self.num_branches = num_branches
        self.num_inchannels = num_inchannels
        branches = []
        for i in range(num_branches):
            width = num_channels[i] * block.expansion
            proj = _projection(num_inchannels[i], width, 1) if num_inchannels[i] != width else None
            seq = [block(num_inchannels[i], num_channels[i], 1, proj)]
            num_inchannels[i] = width
            seq += [block(width, num_channels[i]) for _ in range(1, num_blocks[i])]
            branches.append(nn.Sequential(*seq))
        self.branches = nn.ModuleList(branches)
        self.fuse_layers = None
        if num_branches > 1:
            ch, rows = num_inchannels, []
            for i in range(num_branches if multi_scale_output else 1):
                row = []
                for j in range(num_branches):
                    if j == i:
                        row.append(None)
                    elif j > i:
                        row.append(nn.Sequential(ConvNormAct(ch[j], ch[i], 1),
                                                 nn.Upsample(scale_factor=2 ** (j - i), mode='bilinear',
                                                             align_corners=False)))
                    else:
                        steps = [nn.Sequential(ConvNormAct(ch[j], ch[j], 3, 2)) for _ in range(i - j - 1)]
                        steps.append(ConvNormAct(ch[j], ch[i], 3, 2, activate=None))
                        row.append(nn.Sequential(*steps))
                rows.append(nn.ModuleList(row))
            self.fuse_layers = nn.ModuleList(rows)
        self.relu = nn.ReLU(True)
        initialize_weights(self)

    def forward(self, xs):
        xs = [branch(x) for branch, x in zip(self.branches, xs)]
        if self.fuse_layers is None:
            return xs[:1]
        outs = []
        for i, row in enumerate(self.fuse_layers):
            total = None
            for j, x in enumerate(xs):
                term = x if j == i else row[j](x)
                total = term if total is None else total + term
            outs.append(self.relu(total))
        return outs


class HRNet(nn.Module):
    """Follows reference models/hrnet.py:231-406: stem (two stride-2 ConvNormActs, the first without activation,
    then four 64->256 bottlenecks), three stages of one HRModule with 2/3/4 branches of widths 32*2^i and four basic
    blocks each, a transition before every stage that spawns the new half-resolution branch from the LAST previous
    branch (:374-398), a 1x1 classifier on branch 0 and a x4 bilinear align_corners=False upsample."""

    def __init__(self, num_classes=2, num_branches_list=(2, 3, 4)):
        super().__init__()
        from .backbones import Bottleneck
        blocks = [Bottleneck(64, 64, 1, _projection(64, 256, 1))] + [Bottleneck(256, 64) for _ in range(3)]
        self.stem = nn.Sequential(ConvNormAct(3, 64, 3, 2, activate=None), ConvNormAct(64, 64, 3, 2),
                                  nn.Sequential(*blocks))
        pre = [256]
        for k, nb in enumerate(num_branches_list):
            widths = [32 * 2 ** i for i in range(nb)]
            trans = []
            for i, w in enumerate(widths):
                if i < len(pre):
                    trans.append(None if pre[i] == w else ConvNormAct(pre[i], w, 3))
                else:
                    hops = i + 1 - len(pre)
                    trans.append(nn.Sequential(*[ConvNormAct(pre[-1], w if h == hops - 1 else pre[-1], 3, 2)
                                                 for h in range(hops)]))
            setattr(self, 'transition%d' % (k + 1), nn.ModuleList(trans))
            module = HRModule(nb, HRBasicBlock, [4] * nb, list(widths), widths,
                              multi_scale_output=k < len(num_branches_list) - 1)
            setattr(self, 'stage%d' % (k + 2), nn.Sequential(module))
            pre = module.num_inchannels
        self.num_stages = len(num_branches_list)
        self.final_layer = nn.Conv2d(pre[0], num_classes, 1)

    def forward(self, x):
        ys = [self.stem(x)]
        for k in range(self.num_stages):
            trans = getattr(self, 'transition%d' % (k + 1))
            xs = [ys[i] if t is None else t(ys[-1]) for i, t in enumerate(trans)]
            ys = getattr(self, 'stage%d' % (k + 2))(xs)
        return F.interpolate(self.final_layer(ys[0]), scale_factor=(4, 4), mode='bilinear', align_corners=False)

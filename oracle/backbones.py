"""Oracle backbones (CPU torch).  Test infrastructure only.

``pytorch_modules.backbones.{resnet50,mobilenet_v2}`` are external to the
reference tree.  Their contract, from the call sites:

* both return a LIST OF 5 FEATURE MAPS              models/unet.py:28, models/deeplabv3plus.py:29-32
* resnet50(replace_stride_with_dilation=[F,F,T]):  channels 64/256/512/1024/2048 at strides
  2/4/8/16/16 (features[1] feeds a 256->128 1x1, features[-1] feeds ASPP(2048,...);
  models/deeplabv3plus.py:17-21,30-38)
* mobilenet_v2: channels 16/24/32/96/1280 at strides 2/4/8/16/32 (decoder input widths
  1280, 352=256+96, 160=128+32, 88=64+24; models/unet.py:19-23,28-46)

The architectures are the standard torchvision ones (He et al. 2015 v1.5 with the stride
on the 3x3; Sandler et al. 2018), which is what pytorch_modules wraps.  ``pretrained``
weights cannot be fetched offline and are ignored.
"""
import torch
import torch.nn as nn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation,
                               dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNet50(nn.Module):
    def __init__(self, replace_stride_with_dilation=(False, False, False),
                 layers=(3, 4, 6, 3), width=64):
        super().__init__()
        self.inplanes = width
        self.dilation = 1
        self.conv1 = nn.Conv2d(3, width, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(width, layers[0], 1, False)
        self.layer2 = self._make_layer(width * 2, layers[1], 2, replace_stride_with_dilation[0])
        self.layer3 = self._make_layer(width * 4, layers[2], 2, replace_stride_with_dilation[1])
        self.layer4 = self._make_layer(width * 8, layers[3], 2, replace_stride_with_dilation[2])

    def _make_layer(self, planes, blocks, stride, dilate):
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample, previous_dilation)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes, dilation=self.dilation))
        return nn.Sequential(*layers)

    def forward(self, x):
        f0 = self.relu(self.bn1(self.conv1(x)))
        f1 = self.layer1(self.maxpool(f0))
        f2 = self.layer2(f1)
        f3 = self.layer3(f2)
        f4 = self.layer4(f3)
        return [f0, f1, f2, f3, f4]


def resnet50(pretrained=False, replace_stride_with_dilation=(False, False, False), **kw):
    return ResNet50(replace_stride_with_dilation, **kw)


class ConvBNReLU6(nn.Sequential):
    def __init__(self, cin, cout, k=3, stride=1, groups=1):
        super().__init__(
            nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False),
            nn.BatchNorm2d(cout), nn.ReLU6(inplace=True))


class InvertedResidual(nn.Module):
    def __init__(self, inp, oup, stride, expand_ratio):
        super().__init__()
        hidden = int(round(inp * expand_ratio))
        self.use_res_connect = stride == 1 and inp == oup
        layers = []
        if expand_ratio != 1:
            layers.append(ConvBNReLU6(inp, hidden, 1))
        layers += [ConvBNReLU6(hidden, hidden, 3, stride, groups=hidden),
                   nn.Conv2d(hidden, oup, 1, bias=False), nn.BatchNorm2d(oup)]
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        return x + self.conv(x) if self.use_res_connect else self.conv(x)


class MobileNetV2(nn.Module):
    # t, c, n, s  (Sandler et al. 2018, table 2)
    CFG = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2),
           (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]
    TAPS = (1, 3, 6, 13, 18)  # features[i] outputs returned: 16@2, 24@4, 32@8, 96@16, 1280@32

    def __init__(self):
        super().__init__()
        feats = [ConvBNReLU6(3, 32, 3, 2)]
        cin = 32
        for t, c, n, s in self.CFG:
            for i in range(n):
                feats.append(InvertedResidual(cin, c, s if i == 0 else 1, t))
                cin = c
        feats.append(ConvBNReLU6(cin, 1280, 1))
        self.features = nn.Sequential(*feats)

    def forward(self, x):
        outs = []
        for i, f in enumerate(self.features):
            x = f(x)
            if i in self.TAPS:
                outs.append(x)
        return outs


def mobilenet_v2(pretrained=False, **kw):
    return MobileNetV2()

"""Oracle restatement of the third-party blocks the reference builds on.

``pytorch_modules.nn.ConvNormAct`` and ``pytorch_modules.utils.initialize_weights``
(pytorch-modules>=0.3.0, reference requirements.txt:5) are NOT in the reference
tree; their contract is fixed by the reference's call sites:

* positional ``(cin, cout, 1)`` -> 1x1            models/aspp.py:12,27,30, models/deeplabv3plus.py:20
* ``(cin, cout)`` -> 3x3 default                  models/unet.py:19-21
* ``dilation=rate`` keyword                        models/aspp.py:29
* ``(cin, cout, 3, 2)`` -> stride as 4th arg       models/hrnet.py:261,324
* ``activate=None`` -> no activation               models/hrnet.py:213-217,260
* output H x W equals input H x W at stride 1 for every dilation (the branches are
  ``torch.cat``-ed, models/aspp.py:36) => padding = (k-1)//2 * dilation.
* it is an ``nn.Sequential`` of conv -> BatchNorm2d -> activation (state-dict
  keys ``<name>.0.weight``, ``<name>.1.{weight,bias,running_mean,running_var}``);
  the conv has no bias (a bias would be cancelled by train-mode BN anyway).

Test infrastructure only -- see oracle/__init__.py.
"""
import torch
import torch.nn as nn


class ConvNormAct(nn.Sequential):
    def __init__(self, in_channels, out_channels, ksize=3, stride=1, groups=1,
                 dilation=1, activate=True):
        padding = (ksize - 1) // 2 * dilation
        layers = [
            nn.Conv2d(in_channels, out_channels, ksize, stride=stride,
                      padding=padding, dilation=dilation, groups=groups,
                      bias=False),
            nn.BatchNorm2d(out_channels),  # eps 1e-5, momentum 0.1 (models/hrnet.py:14)
        ]
        if activate is True:
            layers.append(nn.ReLU(inplace=True))
        elif activate is not None and activate is not False:
            layers.append(activate)
        super().__init__(*layers)


def initialize_weights(module):
    """Kaiming-normal conv weights (fan_out, relu), zero biases, BN gamma=1 beta=0.

    Restated from the role the call sites give it (models/deeplabv3plus.py:24-26,
    models/unet.py:24-25, models/hrnet.py:127): the standard torchvision-style
    init.  Parity tests never depend on it (they fill parameters explicitly).
    """
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, 0, 0.01)
            if m.bias is not None:
                nn.init.zeros_(m.bias)

"""MobileNetV2 encoder on the HIP kernels -- the `pytorch_modules.backbones.mobilenet_v2` the reference imports
(models/unet.py:7,16-17).  Contract from the call sites (models/unet.py:19-23,28-46): a list of 5 feature maps,
channels 16/24/32/96/1280 at strides 2/4/8/16/32.  torchvision architecture and parameter names
(`features.N...`), as restated in oracle/backbones.py.  1x1 convs run on the MFMA implicit GEMM, the 3x3
depthwise convs on the direct HBM-bound kernels.
"""
import torch.nn as nn

from .. import ops
from ..nn import ACT_NONE, BatchNorm2d, Conv2d, ConvNormAct


class ConvBNReLU6(ConvNormAct):
    def __init__(self, cin, cout, k=3, stride=1, groups=1):
        super().__init__(cin, cout, k, stride, groups, 1, activate=nn.ReLU6(inplace=True))


class InvertedResidual(nn.Module):
    def __init__(self, inp, oup, stride, expand_ratio):
        super().__init__()
        hidden = int(round(inp * expand_ratio))
        self.use_res_connect = stride == 1 and inp == oup
        layers = []
        if expand_ratio != 1:
            layers.append(ConvBNReLU6(inp, hidden, 1))
        layers += [ConvBNReLU6(hidden, hidden, 3, stride, groups=hidden), Conv2d(hidden, oup, 1, bias=False),
                   BatchNorm2d(oup)]
        self.conv = nn.Sequential(*layers)

    def fwd(self, x, env):
        cur, saved = x, []
        n = len(self.conv)
        for m in list(self.conv)[:n - 2]:
            cur, s = m.fwd(cur, env)
            saved.append(s)
        pconv, pbn = self.conv[n - 2], self.conv[n - 1]
        y, st, sc = pconv.fwd(cur, env, want_stats=pbn.training)
        out, sb = pbn.fwd(y, st, env, act=ACT_NONE, residual=x if self.use_res_connect else None)
        return out, (saved, sc, sb)

    def bwd(self, dout, saved_all, env):
        saved, sc, sb = saved_all
        n = len(self.conv)
        pconv, pbn = self.conv[n - 2], self.conv[n - 1]
        dy = pbn.bwd(dout, sb, env)          # no activation: the residual gradient is dout itself
        d = pconv.bwd(dy, sc, env)
        mods = list(self.conv)[:n - 2]
        for i in range(len(mods) - 1, -1, -1):
            last = i == 0 and self.use_res_connect
            if last:
                # dx = dout (skip path) + main path, merged in the dgrad epilogue when the first layer is a dense 1x1
                dx = dout.like()
                ops.copy2d(dout, dx)
                d = mods[i].bwd(d, saved[i], env, dx_out=dx, dx_accumulate=True)
            else:
                d = mods[i].bwd(d, saved[i], env)
        return d


class MobileNetV2(nn.Module):
    CFG = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]
    TAPS = (1, 3, 6, 13, 18)

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
        self.out_channels = (16, 24, 32, 96, 1280)

    def fwd(self, x, env):
        outs, saved, cur = [], [], x
        for i, f in enumerate(self.features):
            cur, s = f.fwd(cur, env)
            saved.append(s)
            if i in self.TAPS:
                outs.append(cur)
        return outs, saved

    def bwd(self, dfeats, saved, env):
        grads = dict(zip(self.TAPS, dfeats))
        d = None
        mods = list(self.features)
        for i in range(len(mods) - 1, -1, -1):
            g = grads.get(i)
            if g is not None:
                if d is None:
                    d = g
                else:
                    ops.copy2d(g, d, accumulate=True)
            if d is None:
                continue
            if i == 0:
                mods[i].bwd(d, saved[i], env, need_dx=False)
            else:
                d = mods[i].bwd(d, saved[i], env)

    def forward(self, x):
        """NCHW image -> list of the five NCHW feature maps (the backbone contract the reference's model files use)."""
        from ..bridge import run_backbone
        return run_backbone(self, x)


def mobilenet_v2(pretrained=False, **kw):
    m = MobileNetV2()
    if pretrained:
        from .resnet import load_pretrained
        load_pretrained(m, 'mobilenet_v2', 'PSEG_PRETRAINED_MOBILENET_V2')
    return m

"""ResNet-50 encoder on the HIP kernels -- the `pytorch_modules.backbones.resnet50` the reference imports
(models/deeplabv3plus.py:7,17-19).  Contract from the call sites: returns a list of 5 feature maps, channels
64/256/512/1024/2048 at strides 2/4/8/16/16 with replace_stride_with_dilation=[False, False, True].
Architecture and parameter names are torchvision's (v1.5: stride on the 3x3), as restated in oracle/backbones.py.
`pretrained` weights cannot be fetched offline; load a state-dict instead.
"""
import os
import warnings

import torch
import torch.nn as nn

from .. import ops
from ..nn import ACT_NONE, ACT_RELU, BatchNorm2d, Conv2d
from ..ops import Act


def load_pretrained(module, arch, env_var):
    """`pretrained=True` at the reference's call sites (models/deeplabv3plus.py:17-19, models/unet.py:16-17) downloads
    ImageNet weights; there is no network here.  The parameter names are torchvision's, so a torchvision-format
    state-dict file named by ``env_var`` is loaded (fc.* / classifier.* entries are ignored); without it the encoder
    stays RANDOM-INIT and this says so loudly -- training from scratch is a different experiment from the reference's
    fine-tuning."""
    path = os.environ.get(env_var, '')
    if path:
        sd = torch.load(path, map_location='cpu')
        sd = sd.get('state_dict', sd.get('model', sd)) if isinstance(sd, dict) else sd
        own = module.state_dict()
        sd = {k: v for k, v in sd.items() if k in own}
        missing = [k for k in own if k not in sd and 'num_batches_tracked' not in k]
        if missing:
            raise RuntimeError('%s=%s lacks %d of the %s encoder tensors (first: %s)' % (env_var, path, len(missing), arch, missing[0]))
        module.load_state_dict(sd, strict=False)
        return True
    warnings.warn('%s(pretrained=True): no ImageNet weights available offline -- the encoder is RANDOM-INIT. Point %s at a '
                  'torchvision-format state-dict to fine-tune as the reference does, or pass --weights to train.py.'
                  % (arch, env_var), RuntimeWarning, stacklevel=3)
    return False


STEM_WGRAD_HINT = os.environ.get('PSEG_STEM_WGRAD_HINT', '0') == '1'
STEM_WGRAD_ON_MAIN = os.environ.get('PSEG_STEM_WGRAD_AUX', '0') != '1'


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def fwd(self, x, env):
        y1, st1, s1 = self.conv1.fwd(x, env, want_stats=self.bn1.training)
        z1, b1 = self.bn1.fwd(y1, st1, env, act=ACT_RELU)
        y2, st2, s2 = self.conv2.fwd(z1, env, want_stats=self.bn2.training)
        z2, b2 = self.bn2.fwd(y2, st2, env, act=ACT_RELU)
        y3, st3, s3 = self.conv3.fwd(z2, env, want_stats=self.bn3.training)
        sd = bd = None
        identity = x
        if self.downsample is not None:
            dconv, dbn = self.downsample[0], self.downsample[1]
            yd, std, sd = dconv.fwd(x, env, want_stats=dbn.training)
            identity, bd = dbn.fwd(yd, std, env, act=ACT_NONE)
        out, b3 = self.bn3.fwd(y3, st3, env, act=ACT_RELU, residual=identity)  # relu(bn3 + identity), one pass
        return out, (s1, b1, s2, b2, s3, b3, sd, bd)

    def bwd(self, dout, saved, env):
        s1, b1, s2, b2, s3, b3, sd, bd = saved
        d_id = dout.like()                                  # gradient of the identity branch = relu-masked dout
        dy3 = self.bn3.bwd(dout, b3, env, dres=d_id)
        # (conv3 / conv2 are the only consumers of bn2's / bn1's output: their data gradients carry those layers' backward sums)
        dz2 = self.conv3.bwd(dy3, s3, env, bn_prev=b2)
        dy2 = self.bn2.bwd(dz2, b2, env, want_planes=self.conv2.wants_dy_planes(s2, env))
        dz1 = self.conv2.bwd(dy2, s2, env, bn_prev=b1)
        dy1 = self.bn1.bwd(dz1, b1, env)
        if self.downsample is not None:
            dconv, dbn = self.downsample[0], self.downsample[1]
            dyd = dbn.bwd(d_id, bd, env)
            dx = dconv.bwd(dyd, sd, env)
            self.conv1.bwd(dy1, s1, env, dx_out=dx, dx_accumulate=True)   # merge the two paths in the dgrad epilogue
            return dx
        self.conv1.bwd(dy1, s1, env, dx_out=d_id, dx_accumulate=True)
        return d_id


class ResNet50(nn.Module):
    def __init__(self, replace_stride_with_dilation=(False, False, False), layers=(3, 4, 6, 3), width=64):
        super().__init__()
        self.inplanes = width
        self.dilation = 1
        self.conv1 = Conv2d(3, width, 7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(width, layers[0], 1, False)
        self.layer2 = self._make_layer(width * 2, layers[1], 2, replace_stride_with_dilation[0])
        self.layer3 = self._make_layer(width * 4, layers[2], 2, replace_stride_with_dilation[1])
        self.layer4 = self._make_layer(width * 8, layers[3], 2, replace_stride_with_dilation[2])
        self.out_channels = (width, width * 4, width * 8, width * 16, width * 32)

    def _make_layer(self, planes, blocks, stride, dilate):
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                       BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample, previous_dilation)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes, dilation=self.dilation))
        return nn.Sequential(*layers)

    optional_f0 = True      # fwd(x, env, want_f0=False) may skip the stride-2 feature map (see fwd)

    def fwd(self, x, env, want_f0=True):
        """x: Act [B,H,W,4] (RGB + one zero channel).  -> ([f0..f4], saved)
        want_f0=False (a caller that reads no stride-2 feature: DeepLabV3+): the stem's BatchNorm + ReLU + max-pool run as ONE pass
        where they can (training-mode statistics) and f0 -- 268 MB at the benchmark shape -- is neither written nor read back;
        feats[0] is then None."""
        y0, st0, s0 = self.conv1.fwd(x, env, want_stats=self.bn1.training)
        fused = None if want_f0 else self.bn1.fwd_pooled(y0, st0, env, ACT_RELU, 3, 2, 1)
        if fused is not None:
            p, arg, b0 = fused
            f0 = None
        else:
            f0, b0 = self.bn1.fwd(y0, st0, env, act=ACT_RELU)
            Hp, Wp = ops.conv_out_size(f0.H, 3, 2, 1, 1), ops.conv_out_size(f0.W, 3, 2, 1, 1)
            p = f0.new(f0.B, Hp, Wp, f0.C)
            p.amax = f0.amax   # max-pooling cannot exceed its input's max
            arg = ops.maxpool_fwd(f0, p, 3, 2, 1, want_argmax=env.save)
        f0_shape = y0           # (backward only needs f0's geometry: the conv output has it)
        feats, saved_layers, cur = [f0], [], p
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            sl = []
            for blk in layer:
                cur, sb = blk.fwd(cur, env)
                sl.append(sb)
            saved_layers.append(sl)
            feats.append(cur)
        return feats, (s0, b0, f0_shape, arg, saved_layers)

    def bwd(self, dfeats, saved, env):
        """dfeats: list of 5 (Act or None) gradients of the returned features."""
        s0, b0, f0, arg, saved_layers = saved
        layers = (self.layer1, self.layer2, self.layer3, self.layer4)
        d = None
        for li in (3, 2, 1, 0):
            g = dfeats[li + 1]
            if g is not None:
                if d is None:
                    d = g
                else:
                    ops.copy2d(g, d, accumulate=True)
            if d is None:
                continue
            for blk, sb in zip(reversed(list(layers[li])), reversed(saved_layers[li])):
                d = blk.bwd(d, sb, env)
        if d is None and dfeats[0] is None:
            return
        df0 = f0.like(zero=(d is None))
        if d is not None:
            ops.maxpool_bwd(d, arg, df0, 3, 2, 1)
        if dfeats[0] is not None:
            ops.copy2d(dfeats[0], df0, accumulate=True)
        dy0 = self.bn1.bwd(df0, b0, env)
        # The stem's weight gradient is the LAST kernel of a backward pass and nothing follows it on this stream: enqueued HERE it
        # runs beside the weight gradients of layer 1 that still wait on the auxiliary stream, instead of behind them
        # (the step's tail, where one stream ran alone: 0.41 ms -> see profiles/EXPERIMENTS.md 5.12).  PSEG_STEM_WGRAD_AUX=1: as before.
        if STEM_WGRAD_ON_MAIN and env.overlap_wgrad:
            # planned as a launch that runs ALONE (two resident blocks per CU) unless PSEG_STEM_WGRAD_HINT=1: measured both ways
            # in round 6 (profiles/r06_ab_stem_hint.txt: 42.88 alone vs 42.95 ms concurrent, three alternations -- no difference)
            self.conv1.bwd(dy0, s0, env, need_dx=False, wgrad_on_main=True, wgrad_concurrent=STEM_WGRAD_HINT)
        else:
            self.conv1.bwd(dy0, s0, env, need_dx=False)   # image gradient is never needed

    def forward(self, x):
        """NCHW image -> list of the five NCHW feature maps (the backbone contract the reference's model files use)."""
        from ..bridge import run_backbone
        return run_backbone(self, x)


def resnet50(pretrained=False, replace_stride_with_dilation=(False, False, False), **kw):
    m = ResNet50(replace_stride_with_dilation, **kw)
    if pretrained:
        load_pretrained(m, 'resnet50', 'PSEG_PRETRAINED_RESNET50')
    return m

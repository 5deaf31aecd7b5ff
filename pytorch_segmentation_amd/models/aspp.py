"""Atrous spatial pyramid pooling head on the HIP kernels.

Same constructor, attribute names (`gap`, `blocks`, `project`) and arithmetic as the reference's models/aspp.py:8-37,
restructured for the hardware: the five branches write their BN+ReLU outputs directly into channel slices of one
[B,H,W,5*planes] buffer (the reference's torch.cat at models/aspp.py:36 becomes free), the image-level branch is a
pixel-sum kernel + a B-row GEMM + a broadcast kernel (bilinear resize from 1x1 is a constant broadcast,
models/aspp.py:16-19), and in backward the five input gradients merge in the dgrad epilogues (accumulate flag).
"""
import torch.nn as nn

from .. import ops
from ..nn import ConvNormAct
from ..ops import Act


class ASPPPooling(nn.Module):
    def __init__(self, inplanes, planes):
        super().__init__()
        self.gap = nn.Sequential(nn.AdaptiveAvgPool2d(1), ConvNormAct(inplanes, planes, 1))

    def fwd(self, x, env, out):
        pooled = x.new(x.B, 1, 1, x.C)
        pooled.amax = x.amax                     # a mean cannot exceed the max
        ops.pool_sum(x, pooled, 1.0 / (x.H * x.W))
        z, saved = self.gap[1].fwd(pooled, env)
        ops.broadcast(z, out)
        ops.raise_amax(out, z)
        return out, saved

    def bwd(self, dout, saved, env, dx_out, dx_accumulate):
        dz = dout.new(dout.B, 1, 1, dout.C)
        ops.pool_sum(dout, dz, 1.0)                       # backward of the broadcast: sum over pixels
        dpooled = self.gap[1].bwd(dz, saved, env)
        ops.broadcast(dpooled, dx_out, scale=1.0 / (dx_out.H * dx_out.W), accumulate=dx_accumulate)
        return dx_out

    def forward(self, x):
        from ..bridge import run_module
        return run_module(_PoolAdapter(self), x)


class _PoolAdapter:
    """block protocol for a standalone ASPPPooling call"""

    def __init__(self, m):
        self.m = m
        self.block_out_channels = m.gap[1].block_out_channels

    def modules(self):
        return self.m.modules()

    def block_fwd(self, x, env):
        out = x.new(x.B, x.H, x.W, self.block_out_channels)
        return self.m.fwd(x, env, out)

    def block_bwd(self, dy, saved, env, need_dx=True):
        dx = dy.new(dy.B, dy.H, dy.W, self.m.gap[1].conv.cin_p)
        return self.m.bwd(dy, saved, env, dx, False)


class ASPP(nn.Module):
    def __init__(self, inplanes, planes, atrous_rates=(12, 24, 36)):
        super().__init__()
        branches = [ASPPPooling(inplanes, planes), ConvNormAct(inplanes, planes, 1)]
        branches += [ConvNormAct(inplanes, planes, dilation=r) for r in atrous_rates]
        self.blocks = nn.ModuleList(branches)
        self.project = ConvNormAct(planes * len(branches), planes, 1)
        self.planes = planes

    def fwd(self, x, env, out=None):
        n, P = len(self.blocks), self.planes
        assert P % 4 == 0
        cat = x.new(x.B, x.H, x.W, n * P, amax=env.track_amax)
        saved = []
        for i, blk in enumerate(self.blocks):
            _, s = blk.fwd(x, env, out=cat.slice(i * P, (i + 1) * P))
            saved.append(s)
        z, sp = self.project.fwd(cat, env, out=out)
        return z, (saved, sp, (x.B, x.H, x.W, x.C))

    def bwd(self, dout, saved_all, env, need_dx=True):
        saved, sp, (B, H, W, C) = saved_all
        P = self.planes
        dcat = self.project.bwd(dout, sp, env)
        if not need_dx:
            dx = None
        else:
            dx = dout.new(B, H, W, C)
        # dilated / 1x1 branches first (the first one overwrites dx, the rest accumulate), pooled branch last
        first = True
        for i in range(len(self.blocks) - 1, 0, -1):
            self.blocks[i].bwd(dcat.slice(i * P, (i + 1) * P), saved[i], env, need_dx=need_dx, dx_out=dx,
                               dx_accumulate=not first)
            first = False
        if need_dx:
            self.blocks[0].bwd(dcat.slice(0, P), saved[0], env, dx, not first)
        else:
            self.blocks[0].bwd(dcat.slice(0, P), saved[0], env, dout.new(B, H, W, C), False)
        return dx

    def forward(self, x):
        from ..bridge import run_module
        return run_module(self, x)

    def block_fwd(self, x, env):
        return self.fwd(x, env)

    def block_bwd(self, dy, saved, env, need_dx=True):
        return self.bwd(dy, saved, env, need_dx=need_dx)

    @property
    def block_out_channels(self):
        return self.planes

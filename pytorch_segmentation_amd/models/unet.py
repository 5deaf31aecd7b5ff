"""UNet (MobileNetV2 encoder) on the HIP kernels.

Constructor, attribute names (`backbone`, `up_convs`, `cls_conv`) and arithmetic follow the reference's
models/unet.py:13-56: three decoder ConvNormAct 3x3 stages, each followed by x2 bilinear (align_corners=True)
and a skip concat; one more x2; cls_conv 3x3 (88 -> nc, bias) at H/2; final x2.  Each upsample writes straight
into the channel slice of the next stage's concat buffer and the skip is a strided copy into the other slice, so
no torch.cat pass exists; the last upsample stores the logits NCHW.
"""
import torch.nn as nn

from .. import ops
from ..backbones import mobilenet_v2
from ..nn import Conv2d, ConvNormAct, initialize_weights, loss_grad_in
from ..ops import Act


class UNet(nn.Module):
    def __init__(self, num_classes, backbone=None):
        super().__init__()
        self.backbone = backbone if backbone is not None else mobilenet_v2(pretrained=True)
        self.up_convs = nn.ModuleList([ConvNormAct(1280, 256), ConvNormAct(352, 128), ConvNormAct(160, 64)])
        self.cls_conv = Conv2d(88, num_classes, 3, padding=1)
        self.num_classes = num_classes
        initialize_weights(self.up_convs)
        initialize_weights(self.cls_conv)

    def head_fwd(self, feats, env):
        """feats: [_, x2 (24ch, /4), x3 (32ch, /8), x4 (96ch, /16), x (1280ch, /32)]"""
        _, x2, x3, x4, x = feats
        saved, cur = [], x
        for conv, skip in zip(self.up_convs, (x4, x3, x2)):
            z, s = conv.fwd(cur, env)
            assert (z.H * 2, z.W * 2) == (skip.H, skip.W)
            cat = z.new(z.B, skip.H, skip.W, z.C + skip.C, amax=env.track_amax)
            ops.bilinear_fwd(z, cat.slice(0, z.C), True)
            ops.copy2d(skip, cat.slice(z.C, z.C + skip.C))
            if env.track_amax:
                ops.raise_amax(cat, z)
                if skip.amax is not None:
                    ops.raise_amax(cat, skip)
                else:
                    cat.amax = None              # unknown bound: the conv falls back to the generic amax pass
            saved.append((s, (z.B, z.H, z.W, z.C)))
            cur = cat
        up = cur.new(cur.B, cur.H * 2, cur.W * 2, cur.C)
        up.amax = cur.amax
        ops.bilinear_fwd(cur, up, True)
        lr, _, s_cls = self.cls_conv.fwd(up, env, out_f32=True)      # (half policy: the logits leave in fp32)
        out = ops.bilinear_fwd_nchw(lr, self.num_classes, lr.H * 2, lr.W * 2, True)
        return out, (saved, s_cls, (cur.B, cur.H, cur.W, cur.C), (lr.B, lr.H, lr.W, lr.C))

    def head_bwd(self, dout, saved_all, env, need_dfeats=True):
        saved, s_cls, cshape, lshape = saved_all
        dlr = Act.empty(*lshape, dout.device, zero=True)
        ops.bilinear_bwd_nchw(dout, dlr, self.num_classes, True)
        dup = self.cls_conv.bwd(loss_grad_in(dlr, env), s_cls, env)
        dcat = dup.new(*cshape)
        ops.bilinear_bwd(dup, dcat, True)
        dskips = []
        for conv, (s, zshape) in zip(reversed(list(self.up_convs)), reversed(saved)):
            zc = zshape[3]
            dz = dcat.new(*zshape)
            ops.bilinear_bwd(dcat.slice(0, zc), dz, True)
            dskips.append(dcat.slice(zc, dcat.C))          # gradient of the skip feature: a view, no copy
            dcat = conv.bwd(dz, s, env, need_dx=need_dfeats)
        dx2, dx3, dx4 = dskips
        return [None, dx2, dx3, dx4, dcat]

    def model_fwd(self, x, env):
        xa = Act.from_nchw(x, 8 if env.half else 4, dtype=env.act_dtype)
        feats, s_bb = self.backbone.fwd(xa, env)
        out, s_head = self.head_fwd(feats, env)
        return out, (s_bb, s_head)

    def model_bwd(self, dout, saved, env):
        s_bb, s_head = saved
        dfeats = self.head_bwd(dout, s_head, env)
        self.backbone.bwd(dfeats, s_bb, env)

    def forward(self, x):
        from ..bridge import run_model
        return run_model(self, x)

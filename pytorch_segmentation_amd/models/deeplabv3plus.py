"""DeepLabV3+ (ResNet-50, output stride 16, ASPP rates 6/12/18) on the HIP kernels.

Constructor, attribute names (`backbone`, `project`, `aspp`, `cls_conv`) and arithmetic follow the reference's
models/deeplabv3plus.py:14-44.  Data flow on the GPU:

  image NCHW --layout kernel--> NHWC(4ch) --backbone--> f1 [B,H/4,W/4,256], f4 [B,H/16,W/16,2048]
  cat = [B,H/4,W/4,384]:  channels 0..255  <- x4 bilinear(ASPP(f4))      (written by the resize kernel)
                          channels 256..383 <- project(f1)               (written by its BN+ReLU pass)
  cls_conv(cat) -> [B,H/4,W/4,nc padded to 4] --x4 bilinear, NCHW store--> logits [B,nc,H,W]

so neither torch.cat (reference :38) nor an NHWC->NCHW conversion of the logits exists as a separate pass.
"""
import torch.nn as nn

from .. import ops
from ..backbones import resnet50
from ..nn import Conv2d, ConvNormAct, initialize_weights, loss_grad_in
from ..ops import Act
from .aspp import ASPP


class DeepLabV3Plus(nn.Module):
    def __init__(self, num_classes, backbone=None):
        super().__init__()
        self.backbone = backbone if backbone is not None else resnet50(
            pretrained=True, replace_stride_with_dilation=[False, False, True])
        self.project = ConvNormAct(256, 128, 1)
        self.aspp = ASPP(2048, 256, [6, 12, 18])
        self.cls_conv = Conv2d(384, num_classes, 3, padding=1)
        self.num_classes = num_classes
        for m in (self.aspp, self.project, self.cls_conv):
            initialize_weights(m)

    # ---- head on explicit feature maps (also the unit the parity fixtures pin)
    # the class logits leave the head at stride 4 and are up-sampled x4 (reference :40-43); with lowres=True the head
    # hands out the stride-4 logits themselves (ops.ce_upsampled_fwd_bwd takes the loss from them)
    lowres_loss = (4, True)      # (scale factor, align_corners) of the final up-sampling

    def head_fwd(self, low_in, high_in, env, lowres=False):
        B, H4, W4 = low_in.B, low_in.H, low_in.W
        cat = low_in.new(B, H4, W4, 384, amax=env.track_amax)
        _, s_proj = self.project.fwd(low_in, env, out=cat.slice(256, 384))
        a, s_aspp = self.aspp.fwd(high_in, env)
        assert (a.H * 4, a.W * 4) == (H4, W4), 'ASPP map x4 must match the stride-4 map'
        ops.bilinear_fwd(a, cat.slice(0, 256), True)
        ops.raise_amax(cat, a)                   # bilinear interpolation is a convex combination
        lr, _, s_cls = self.cls_conv.fwd(cat, env, out_f32=True)     # (half policy: the logits leave in fp32)
        out = lr if lowres else ops.bilinear_fwd_nchw(lr, self.num_classes, H4 * 4, W4 * 4, True)
        return out, (s_proj, s_aspp, s_cls, (a.B, a.H, a.W, a.C), (lr.B, lr.H, lr.W, lr.C))

    def head_bwd(self, dout, saved, env, need_dlow=True, need_dhigh=True, lowres=False):
        """dout: gradient of the NCHW logits, or (lowres=True) of the stride-4 NHWC logits (an Act)."""
        s_proj, s_aspp, s_cls, ashape, lshape = saved
        if lowres:
            dlr = dout
        else:
            dlr = Act.empty(*lshape, dout.device, zero=True)         # padded class channels stay zero
            ops.bilinear_bwd_nchw(dout, dlr, self.num_classes, True)
        dcat = self.cls_conv.bwd(loss_grad_in(dlr, env), s_cls, env)
        da = dcat.new(*ashape)
        ops.bilinear_bwd(dcat.slice(0, 256), da, True)
        dhigh = self.aspp.bwd(da, s_aspp, env, need_dx=need_dhigh)
        dlow = self.project.bwd(dcat.slice(256, 384), s_proj, env, need_dx=need_dlow)
        return dlow, dhigh

    def model_fwd(self, x, env, lowres=False):
        xa = Act.from_nchw(x, 8 if env.half else 4, dtype=env.act_dtype)
        # (only the stride-4 and stride-16 features are read: a ResNet-50 stem then never materialises its activated stride-2 map)
        if getattr(self.backbone, 'optional_f0', False):
            feats, s_bb = self.backbone.fwd(xa, env, want_f0=False)
        else:                       # (the callable-backbone contract: fwd(x, env))
            feats, s_bb = self.backbone.fwd(xa, env)
        out, s_head = self.head_fwd(feats[1], feats[-1], env, lowres=lowres)
        return out, (s_bb, s_head)

    def model_bwd(self, dout, saved, env, lowres=False):
        s_bb, s_head = saved
        dlow, dhigh = self.head_bwd(dout, s_head, env, lowres=lowres)
        self.backbone.bwd([None, dlow, None, None, dhigh], s_bb, env)

    def forward(self, x):
        from ..bridge import run_model
        return run_model(self, x)

"""Where does the half-precision forward drift from the fp32 one?  DeepLabV3+ features stage by stage (max-norm and
relative L2 of half vs fp32 on the same weights / batch).  usage: python tools/half_diverge.py [batch] [size]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fill  # noqa: E402
from pytorch_segmentation_amd import models, prepare  # noqa: E402
from pytorch_segmentation_amd.nn import Env  # noqa: E402
from pytorch_segmentation_amd.ops import Act  # noqa: E402


def cmp(name, a, b):
    a, b = a.to_nchw().double(), b.to_nchw().double()
    print('%-28s max-norm %.2e  rel-L2 %.2e  (peak %.3g)' % (name, ((a - b).abs().max() / b.abs().max()).item(),
                                                           ((a - b).norm() / b.norm()).item(), b.abs().max().item()))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    torch.manual_seed(0)
    m = models.DeepLabV3Plus(21)
    ar = prepare(m, 'cuda')
    m.train()
    x = fill.images('div/x', (B, 3, S, S)).cuda()
    outs = {}
    for pol in ('fp32', 'half'):
        env = Env(save=False, policy=pol)
        if env.half:
            ar.prepare_half()
        xa = Act.from_nchw(x, 8 if env.half else 4, dtype=env.act_dtype)
        bb = m.backbone
        y0, st0, _ = bb.conv1.fwd(xa, env, want_stats=True)
        f0, _ = bb.bn1.fwd(y0, st0, env, act=1)
        feats, _ = bb.fwd(xa, env)
        a, _ = m.aspp.fwd(feats[-1], env)
        out, _ = m.head_fwd(feats[1], feats[-1], env)
        outs[pol] = dict(stem_conv=y0, stem_bn=f0, f1=feats[1], f2=feats[2], f3=feats[3], f4=feats[4], aspp=a, logits=out)
    for k in outs['fp32']:
        a, b = outs['half'][k], outs['fp32'][k]
        if torch.is_tensor(a):
            print('%-28s max-norm %.2e' % (k, ((a.double() - b.double()).abs().max() / b.abs().max()).item()))
        else:
            cmp(k, a, b)


if __name__ == '__main__':
    main()

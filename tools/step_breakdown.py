"""Per-conv-call timing inside a real DeepLabV3+ training step (B=16, 512x512): where the conv time goes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pytorch_segmentation_amd import ops  # noqa: E402
from pytorch_segmentation_amd.models import DeepLabV3Plus  # noqa: E402
from pytorch_segmentation_amd.utils import Trainer, compute_loss  # noqa: E402


def main():
    ops.OVERLAP_WGRAD = False   # one stream: per-launch durations are not disturbed by a concurrent weight gradient
    dev = torch.device('cuda', 0)
    model = DeepLabV3Plus(21)
    tr = Trainer(model, None, loss_fn=compute_loss, accumulate=1, lr=1e-3, device=dev)
    model.train()
    x, t = bench.synthetic_batch(16, 512, 21, dev, 1)
    for _ in range(2):
        tr.train_batch(x, t)
    shapes = []
    orig = {}

    def spy(name, desc):
        o = getattr(ops, name)
        orig[name] = o

        def w(*a, **k):
            shapes.append(desc(*a))
            return o(*a, **k)
        setattr(ops, name, w)

    spy('conv2d_fwd', lambda x, w, b, y, kh, kw, s, p, d: 'fwd   x%dx%dx%d->%d k%d s%d d%d' % (x.H, x.W, x.C, y.C, kh, s, d))
    spy('conv2d_dgrad', lambda dy, wT, dx, kh, kw, s, p, d: 'dgrad x%dx%dx%d->%d k%d s%d d%d' % (dx.H, dx.W, dx.C, dy.C, kh, s, d))
    spy('conv2d_wgrad', lambda x, dy, dw, kh, kw, s, p, d: 'wgrad x%dx%dx%d->%d k%d s%d d%d' % (x.H, x.W, x.C, dy.C, kh, s, d))
    with bench.ConvMeter(ops) as meter:
        tr.train_batch(x, t)
        torch.cuda.synchronize()
        recs = [(e0.elapsed_time(e1), dn, us) for _, e0, e1, dn, us in meter.records]
    agg = {}
    for desc, (ms, dn, us) in zip(shapes, recs):
        a = agg.setdefault(desc, [0.0, 0.0, 0.0, 0])
        a[0] += ms; a[1] += dn; a[2] += us; a[3] += 1
    tot = sum(v[0] for v in agg.values())
    print('total conv ms %.2f' % tot)
    for desc, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
        print('%-46s n=%2d  %6.2f ms  %5.1f%%  exec %6.1f TF  useful %6.1f TF' % (desc, v[3], v[0], 100 * v[0] / tot, v[1] / v[0] / 1e9, v[2] / v[0] / 1e9))


if __name__ == '__main__':
    main()

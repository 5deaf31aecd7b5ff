"""Per-call table of the conv kernels of one DeepLabV3+ training step (B=16, 512x512): shape, ms, executed / useful TF.
usage: python tools/step_convs.py [policy]   (one stream, HIP events around every conv call)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from pytorch_segmentation_amd import ops
from pytorch_segmentation_amd.models import DeepLabV3Plus
from pytorch_segmentation_amd.utils import Trainer, compute_loss
pol = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
torch.manual_seed(0)
m = DeepLabV3Plus(21)
tr = Trainer(m, None, loss_fn=compute_loss, lr=1e-3)
tr.env.policy = pol
m.train()
x, t = bench.synthetic_batch(16, 512, 21, 'cuda', 1)
for _ in range(3):
    tr.train_batch(x, t)
ops.OVERLAP_WGRAD = False
shapes = []
orig = {}
def wrap(name):
    o = getattr(ops, name); orig[name] = o
    def f(*a, **kw):
        if name == 'conv2d_fwd': xx, yy, k, s, p, d = a[0], a[3], a[4], a[6], a[7], a[8]; desc = (xx.C, xx.H, yy.C, k, s, d)
        elif name == 'conv2d_dgrad': dy, dx, k, s, p, d = a[0], a[2], a[3], a[5], a[6], a[7]; desc = (dx.C, dx.H, dy.C, k, s, d)
        else: xx, dy, k, s, p, d = a[0], a[1], a[3], a[5], a[6], a[7]; desc = (xx.C, xx.H, dy.C, k, s, d)
        shapes.append(desc)
        return o(*a, **kw)
    setattr(ops, name, f)
with bench.ConvMeter(ops) as meter:
    for n in ('conv2d_fwd', 'conv2d_dgrad', 'conv2d_wgrad'):
        wrap(n)
    tr.train_batch(x, t)
    meter.summary()
agg = {}
for (name, e0, e1, dn, us), desc in zip(meter.records, shapes):
    k = (name, desc)
    a = agg.setdefault(k, [0, 0.0, 0.0, 0.0])
    a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += dn; a[3] += us
tot = {}
for (name, desc), a in agg.items():
    tot[name] = tot.get(name, 0.0) + a[1]
print('policy', pol, {k: round(v, 2) for k, v in tot.items()})
for name in ('conv2d_fwd', 'conv2d_dgrad', 'conv2d_wgrad'):
    print('----', name, '(Cin, H, Cout, k, stride, dil)  calls  ms  executed TF  useful TF')
    rows = sorted([(a[1], desc, a) for (n, desc), a in agg.items() if n == name], reverse=True)
    for ms, desc, a in rows[:40]:
        print('  %-28s x%d %7.3f ms  %6.1f  %6.1f' % (desc, a[0], ms, a[2] / ms / 1e9, a[3] / ms / 1e9))

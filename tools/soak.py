"""300 training steps (fixed synthetic batch, SGD lr 2e-2) under each conv policy: loss trend, finiteness and -- for the
half-precision policy -- the loss-scale state and the distance of its loss curve from the fp32 one.
usage: python tools/soak.py [deeplabv3plus|hrnet] [policies, comma separated]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pytorch_segmentation_amd import models  # noqa: E402
from pytorch_segmentation_amd.utils import Trainer, compute_loss  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'deeplabv3plus'
    pols = (sys.argv[2] if len(sys.argv) > 2 else 'fp32,mixed,limb,half').split(',')
    cls, B, S = {'deeplabv3plus': (models.DeepLabV3Plus, 16, 512), 'hrnet': (models.HRNet, 8, 512)}[name]
    curves = {}
    for pol in pols:
        torch.manual_seed(0)
        m = cls(21)
        tr = Trainer(m, None, loss_fn=compute_loss, lr=2e-2)
        tr.env.policy = pol
        m.train()
        x, t = bench.synthetic_batch(B, S, 21, 'cuda', 7)
        ls = []
        for i in range(300):
            ls.append(tr.train_batch(x, t))
        ls = [v.item() for v in ls]
        curves[pol] = ls
        ok = all(torch.isfinite(p).all().item() for p in m.parameters())
        print(name, pol, [round(ls[i], 4) for i in (0, 50, 100, 150, 200, 250, 299)], 'finite params:', ok,
              'scaler:', tr.loss_scale_state(), flush=True)
    if 'fp32' in curves:
        for pol in pols:
            if pol != 'fp32':
                rel = [abs(a - b) / abs(b) for a, b in zip(curves[pol], curves['fp32'])]
                d = max(rel)
                late = max(rel[50:])
                print('%s vs fp32: largest relative loss difference over 300 steps %.3f%% (at step %d; %.3f%% from step 50 on; first steps %s vs %s)' % (
                    pol, 100 * d, rel.index(d), 100 * late, [round(v, 3) for v in curves[pol][:6]], [round(v, 3) for v in curves['fp32'][:6]]), flush=True)


if __name__ == '__main__':
    main()

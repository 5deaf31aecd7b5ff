"""How many data gradients of one DeepLabV3+ training step run on the pre-split LDS-DMA limb kernel (debug aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pytorch_segmentation_amd import models, ops  # noqa: E402
from pytorch_segmentation_amd.utils import Trainer, compute_loss  # noqa: E402

dev = torch.device('cuda', 0)
model = models.DeepLabV3Plus(21)
tr = Trainer(model, None, loss_fn=compute_loss, accumulate=1, lr=1e-3, device=dev, graph=False)
model.train()
x, t = bench.synthetic_batch(16, 512, 21, dev, 1)
n = {'planes': 0, 'plain': 0}
o1, o2 = ops.conv2d_dgrad_planes, ops.conv2d_dgrad


def a(*k, **kw):
    n['planes'] += 1
    dy, dx = k[1], k[3]
    print('planes dy', (dy.B, dy.H, dy.W, dy.C), '-> dx C', dx.C, 'k', k[4], 'd', k[8])
    return o1(*k, **kw)


def b(*k, **kw):
    n['plain'] += 1
    return o2(*k, **kw)


ops.conv2d_dgrad_planes, ops.conv2d_dgrad = a, b
tr.train_batch(x, t)
torch.cuda.synchronize()
print(ops.POLICY_NAME, n)

"""Long -mp run that crosses the loss scaler's growth interval (2000 clean steps) twice: UNet 256x256 B=8, 4500 steps over
eight different synthetic batches (replayed captured steps), scaler state and loss every 500 steps.
usage: python tools/soak_scaler.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pytorch_segmentation_amd import models  # noqa: E402
from pytorch_segmentation_amd.utils import Trainer, compute_loss  # noqa: E402

torch.manual_seed(0)
m = models.UNet(2)
tr = Trainer(m, None, loss_fn=compute_loss, lr=5e-3, mixed_precision=True, graph=True)
m.train()
batches = [bench.synthetic_batch(8, 256, 2, 'cuda', 100 + i) for i in range(8)]
for step in range(4500):
    x, t = batches[step % 8]
    loss = tr.train_batch(x, t)
    if step % 500 == 499 or step in (0, 1999, 2000, 2001, 3999, 4000, 4001):
        print('step %4d loss %.4f scaler %s' % (step, loss.item(), tr.loss_scale_state()), flush=True)
ok = all(torch.isfinite(p).all().item() for p in m.parameters())
print('finite parameters:', ok)

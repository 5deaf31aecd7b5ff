"""300 training steps of DeepLabV3+ (B=16, 512x512, fixed synthetic batch) under each conv policy: loss trend and finiteness."""
import os, sys, torch
sys.path.insert(0, '/root/repo')
import bench
from pytorch_segmentation_amd.models import DeepLabV3Plus
from pytorch_segmentation_amd.utils import Trainer, compute_loss
for pol in ('fp32', 'mixed', 'limb'):
    torch.manual_seed(0)
    m = DeepLabV3Plus(21)
    tr = Trainer(m, None, loss_fn=compute_loss, lr=2e-2)
    tr.env.policy = pol
    m.train()
    x, t = bench.synthetic_batch(16, 512, 21, 'cuda', 7)
    ls = []
    for i in range(300):
        l = tr.train_batch(x, t)
        if i % 50 == 0 or i == 299:
            ls.append(round(l.item(), 4))
    ok = all(torch.isfinite(p).all().item() for p in m.parameters())
    print(pol, ls, 'finite params:', ok, flush=True)

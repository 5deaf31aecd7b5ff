"""Which ops issue the ~68 __amd_rocclr_copyBuffer launches of an eager DeepLabV3+ training step?  torch.profiler view.
usage: python tools/find_copies.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pytorch_segmentation_amd import models  # noqa: E402
from pytorch_segmentation_amd.utils import Trainer, compute_loss  # noqa: E402

dev = torch.device('cuda', 0)
model = models.DeepLabV3Plus(21)
tr = Trainer(model, None, loss_fn=compute_loss, lr=1e-3, device=dev, graph=False)
model.train()
x, t = bench.synthetic_batch(int(os.environ.get("FC_B", "16")), int(os.environ.get("FC_S", "512")), 21, dev, 1)
for _ in range(2):
    tr.train_batch(x, t)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_batch(x, t)
    torch.cuda.synchronize()
ev = prof.events()
names = {}
for e in ev:
    n = e.name
    if 'emcpy' in n or 'copy' in n.lower() or 'emset' in n or 'fill' in n.lower():
        key = (n[:60], str(e.device_type), tuple(e.stack[:3]) if e.stack else ())
        names[key] = names.get(key, 0) + 1
for k, c in sorted(names.items(), key=lambda kv: -kv[1])[:20]:
    print(c, k)
print(prof.key_averages().table(sort_by="count", row_limit=12, max_name_column_width=60))

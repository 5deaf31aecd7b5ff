"""Are there device-to-device / host-to-device copies (or fills) inside a training step, and from where?  torch.profiler (ROCTracer
activities + Python stacks) over two eager DeepLabV3+ steps: prints copy / fill-like events with their package frames, then the 40 most
frequent event names.  (Answer, round 5: none -- the ~480 `__amd_rocclr_copyBuffer` launches of a profiled bench run are the model's
parameters going to the device once, not per-step work.)
usage: python tools/find_memcpy.py [policy]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pytorch_segmentation_amd import models  # noqa: E402
from pytorch_segmentation_amd.utils import Trainer, compute_loss  # noqa: E402

pol = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
m = models.DeepLabV3Plus(21)
tr = Trainer(m, None, loss_fn=compute_loss, lr=1e-3, graph=False)
tr.env.policy = pol
m.train()
x, t = bench.synthetic_batch(16, 512, 21, 'cuda', 7)
for _ in range(3):
    tr.train_batch(x, t)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(2):
        tr.train_batch(x, t)
    torch.cuda.synchronize()
names = collections.Counter()
allnames = collections.Counter()
stacks = collections.Counter()
for ev in prof.events():
    n = ev.name
    allnames[n] += 1
    if 'emcpy' in n or 'emset' in n or 'copy' in n.lower() or 'fill' in n.lower() or 'zero_' in n:
        names[n] += 1
        st = [s for s in (ev.stack or []) if 'pytorch_segmentation_amd' in s or 'bench' in s]
        stacks[(n, tuple(st[:3]))] += 1
print(names.most_common(12))
print([(k[:60], v) for k, v in allnames.most_common(40)])
for (n, st), c in stacks.most_common(12):
    print(c, n, st)

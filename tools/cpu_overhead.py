"""Host-side enqueue time of one training step (no device sync inside) vs the device time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pytorch_segmentation_amd import ops
from pytorch_segmentation_amd.models import DeepLabV3Plus
from pytorch_segmentation_amd.utils import Trainer, compute_loss
pol = sys.argv[1] if len(sys.argv) > 1 else 'mixed'
ops.set_conv_precision(pol)
dev = torch.device('cuda', 0)
import pytorch_segmentation_amd.models as _M
name = sys.argv[2] if len(sys.argv) > 2 else "deeplab"
model = {"deeplab": _M.DeepLabV3Plus, "hrnet": _M.HRNet, "unet": _M.UNet}[name](21)
tr = Trainer(model, None, loss_fn=compute_loss, accumulate=1, lr=1e-3, device=dev)
model.train()
x, t = bench.synthetic_batch(8 if name != "deeplab" else 16, 512, 21, dev, 1)
for _ in range(3):
    tr.train_batch(x, t)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    tr.train_batch(x, t)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('policy %s: host enqueue %.2f ms/step, wall %.2f ms/step' % (pol, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    tr.train_batch(x, t)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)

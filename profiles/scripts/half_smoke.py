"""Shake-down of the half-precision (-mp) policy: the same model trained a few steps under fp32 and under half from one
initialisation; prints the loss curves, the loss-scale state and the step time.
usage: python tools/half_smoke.py [deeplabv3plus|unet|hrnet] [batch] [size] [classes] [steps]"""
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pytorch_segmentation_amd import models  # noqa: E402
from pytorch_segmentation_amd.utils import Trainer  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'deeplabv3plus'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    nc = int(sys.argv[4]) if len(sys.argv) > 4 else 21
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 8
    cls = {'hrnet': models.HRNet, 'unet': models.UNet, 'deeplabv3plus': models.DeepLabV3Plus}[name]
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    base = cls(nc)
    sd = copy.deepcopy(base.state_dict())
    x, t = bench.synthetic_batch(B, S, nc, dev, 1)
    for mp in (False, True):
        model = cls(nc)
        model.load_state_dict(sd)
        tr = Trainer(model, None, accumulate=1, lr=1e-3, device=dev, mixed_precision=mp)
        model.train()
        losses = []
        for _ in range(steps):
            losses.append(tr.train_batch(x, t))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.train_batch(x, t)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print('%s %s: %.2f ms/step; losses %s; scaler %s' % (name, 'half' if mp else 'fp32', dt * 1e3,
                                                          ' '.join('%.4f' % float(l) for l in losses), tr.loss_scale_state()),
              flush=True)


if __name__ == '__main__':
    main()

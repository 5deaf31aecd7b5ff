"""Training-step time of any of the three models (synthetic data) -- the non-headline BASELINE.json configs.
usage: python tools/bench_model.py hrnet|unet|deeplabv3plus [batch] [size] [classes] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pytorch_segmentation_amd import models, ops  # noqa: E402
from pytorch_segmentation_amd.utils import Trainer, compute_loss  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'hrnet'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    nc = int(sys.argv[4]) if len(sys.argv) > 4 else 21
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
    cls = {'hrnet': models.HRNet, 'unet': models.UNet, 'deeplabv3plus': models.DeepLabV3Plus}[name]
    dev = torch.device('cuda', 0)
    model = cls(nc)
    tr = Trainer(model, None, loss_fn=compute_loss, accumulate=1, lr=1e-3, device=dev)
    model.train()
    x, t = bench.synthetic_batch(B, S, nc, dev, 1)
    for _ in range(6):        # (AUTO graph mode: eager, judged, eager under the capture's configuration, captured; then replays)
        tr.train_batch(x, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_batch(x, t)
    host = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # enqueue cost of ONE step into empty queues (no back-pressure from the device)
    one = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        tr.train_batch(x, t)
        one = min(one, time.perf_counter() - t1)
    torch.cuda.synchronize()
    for sg in tr._graphs.values():
        if sg is not None and getattr(sg, 'lanes', 0):
            print('lane executor:', sg.lane_info)
    print('graph mode %s, decisions %s' % (tr.graph, tr.graph_decisions()))
    print('%s B=%d %dx%d nc=%d policy=%s: %.2f ms/step  %.1f img/s  (host enqueue %.2f ms/step, %.2f into empty queues)  peak mem %.1f GB' % (
        name, B, S, S, nc, ops.POLICY_NAME, dt * 1e3, B / dt, host * 1e3, one * 1e3, torch.cuda.max_memory_allocated() / 2 ** 30))


if __name__ == '__main__':
    main()

"""The only hardware number available for the gradient-exchange path on a one-GPU box: the DeepLabV3+ training step
(B=16, 512x512) with the bucketed side-stream reducer forced on over ONE RCCL rank (the all-reduce is the identity, but
its launches, events, side stream and the join before the fused optimiser all execute) against the same step with the
reducer off -- under the fp32 policy and under `half` (-mp).  Not a scaling measurement: no scaling curve exists.
usage: python tools/bench_forced_reducer.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29577')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
import bench  # noqa: E402
from pytorch_segmentation_amd.models import DeepLabV3Plus  # noqa: E402
from pytorch_segmentation_amd.utils import Trainer, compute_loss  # noqa: E402

x, t = bench.synthetic_batch(16, 512, 21, torch.device('cuda', 0), 1)
for mp in (False, True):
    for force in ('1', '0'):
        os.environ['PSEG_FORCE_REDUCER'] = force
        torch.manual_seed(0)
        m = DeepLabV3Plus(21)
        tr = Trainer(m, None, loss_fn=compute_loss, accumulate=1, lr=1e-3, mixed_precision=mp)
        assert tr.reducer.enabled == (force == '1')
        m.train()
        for _ in range(3):
            tr.train_batch(x, t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            tr.train_batch(x, t)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print('%s, reducer %s (1-rank RCCL, %d buckets of <= 32 MiB over %.1f MB of fp32 gradients): %.2f ms/step  %.1f img/s'
              % ('half (-mp)' if mp else 'fp32', 'ON ' if force == '1' else 'off', len(tr.reducer.buckets),
                 tr.arena.numel * 4 / 1e6, dt * 1e3, 16 / dt), flush=True)
        del tr, m
dist.destroy_process_group()

"""The only hardware number available for the gradient-exchange path on a one-GPU box: the DeepLabV3+ training step
(B=16, 512x512) with the bucketed side-stream reducer forced on over ONE RCCL rank (the all-reduce is the identity, but
its launches, events, side stream and the join before the fused optimiser all execute) against the same step with the
reducer off -- under the fp32 policy and under `half` (-mp).  Not a scaling measurement: no scaling curve exists.
usage: python tools/bench_forced_reducer.py"""
import os
import subprocess
import sys
import time

if len(sys.argv) == 1:
    # every (case, reducer on/off) pair in a fresh process: a second Trainer in one process inherits allocator / pool state
    # from the first (measured: up to 2 ms of difference that belongs to neither)
    for case in range(7):
        for force in ('1', '0'):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), str(case), force], capture_output=True, text=True)
            out = [ln for ln in r.stdout.splitlines() if 'reducer' in ln]
            print(out[-1] if out else 'case %d force %s FAILED: %s' % (case, force, r.stderr[-400:]), flush=True)
    sys.exit(0)


import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', str(29577 + 2 * int(sys.argv[1]) + int(sys.argv[2])))
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
import bench  # noqa: E402
from pytorch_segmentation_amd.models import DeepLabV3Plus  # noqa: E402
from pytorch_segmentation_amd.utils import Trainer, compute_loss  # noqa: E402

from pytorch_segmentation_amd.models import HRNet  # noqa: E402

CASES = [('DeepLabV3+ B=16', DeepLabV3Plus, 16, False, False), ('DeepLabV3+ B=16', DeepLabV3Plus, 16, True, False),
         # captured + replayed steps: the bucket all-reduces hang behind the replay's markers (csrc/lanes.hip)
         ('DeepLabV3+ B=16', DeepLabV3Plus, 16, False, True), ('DeepLabV3+ B=16', DeepLabV3Plus, 16, True, True),
         ('HRNet B=8', HRNet, 8, True, True),
         # the library's own RCCL binding (pseg_allreduce_bucket) instead of torch.distributed's all_reduce
         ('DeepLabV3+ B=16 native', DeepLabV3Plus, 16, False, False), ('DeepLabV3+ B=16 native', DeepLabV3Plus, 16, True, True)]
if 'native' in CASES[int(sys.argv[1])][0]:
    os.environ['PSEG_NATIVE_ALLREDUCE'] = '1'
for label, cls, B, mp, graph in [CASES[int(sys.argv[1])]]:
    x, t = bench.synthetic_batch(B, 512, 21, torch.device('cuda', 0), 1)
    for force in (sys.argv[2],):
        os.environ['PSEG_FORCE_REDUCER'] = force
        torch.manual_seed(0)
        m = cls(21)
        tr = Trainer(m, None, loss_fn=compute_loss, accumulate=1, lr=1e-3, mixed_precision=mp, graph=graph)
        assert tr.reducer.enabled == (force == '1')
        m.train()
        for _ in range(6):
            tr.train_batch(x, t)
        torch.cuda.synchronize()
        dt = 1e9
        for _rep in range(3):          # best of three blocks of ten steps (the boxes are noisy at the 1 ms level)
            t0 = time.perf_counter()
            for _ in range(10):
                tr.train_batch(x, t)
            torch.cuda.synchronize()
            dt = min(dt, (time.perf_counter() - t0) / 10)
        marks = [sg.lane_info.get('markers', 0) for sg in tr._graphs.values() if sg is not None]
        print('%s %s %s, reducer %s (1-rank RCCL, %d buckets of <= 32 MiB over %.1f MB of fp32 gradients%s): %.2f ms/step  %.1f img/s'
              % (label, 'half (-mp)' if mp else 'fp32', 'replayed' if graph else 'eager', 'ON ' if force == '1' else 'off',
                 len(tr.reducer.buckets), tr.arena.numel * 4 / 1e6,
                 (', %d markers' % marks[0]) if (marks and force == '1') else '', dt * 1e3, B / dt), flush=True)
        del tr, m
dist.destroy_process_group()

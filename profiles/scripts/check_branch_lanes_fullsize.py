"""HRNet 512x512 B=8 at full size: N optimiser steps replayed with the branch lanes (twice) and eagerly on one stream must
leave bit-identical parameters -- a race between lanes would show as a difference.  usage: python tools/check_branch_lanes_fullsize.py [steps] [half|fp32] [model] [B] [S] [classes]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pytorch_segmentation_amd import models, ops  # noqa: E402
from pytorch_segmentation_amd.utils import Trainer, compute_loss  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
mp = (sys.argv[2] if len(sys.argv) > 2 else 'half') == 'half'
name = sys.argv[3] if len(sys.argv) > 3 else 'hrnet'
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
S = int(sys.argv[5]) if len(sys.argv) > 5 else 512
nc = int(sys.argv[6]) if len(sys.argv) > 6 else 21
cls = {'hrnet': models.HRNet, 'unet': models.UNet, 'deeplabv3plus': models.DeepLabV3Plus}[name]
dev = torch.device('cuda', 0)
torch.manual_seed(0)
state = {k: v.clone() for k, v in cls(nc).state_dict().items()}
batches = [bench.synthetic_batch(B, S, nc, dev, seed) for seed in range(4)]
res = []
for graph, overlap in ((True, True), (True, True), (False, True), (False, False)):
    ops.OVERLAP_WGRAD = overlap          # (the last run: everything on ONE stream, slab reductions batched)
    m = cls(nc)
    m.load_state_dict(state)
    tr = Trainer(m, None, loss_fn=compute_loss, lr=1e-3, graph=graph, mixed_precision=mp, device=dev)
    m.train()
    losses = []
    for s in range(steps):
        x, t = batches[s % 4]
        losses.append(tr.train_batch(x, t))
    torch.cuda.synchronize()
    info = [sg.lane_info for sg in tr._graphs.values() if sg is not None] if graph else None
    res.append(([l.item() for l in losses], tr.arena.params.clone(), tr.loss_scale_state() if mp else None))
    print('graph=%s two-stream=%s lanes=%s last losses %s' % (graph, overlap, info, ['%.6f' % v for v in res[-1][0][-3:]]), flush=True)
    del tr
ok = True
for k in (1, 2, 3):
    same_l = res[0][0] == res[k][0]
    same_p = torch.equal(res[0][1], res[k][1])
    first = next((i for i, (a, b) in enumerate(zip(res[0][0], res[k][0])) if a != b), None)
    print('run 0 vs run %d: losses identical %s (first difference at step %s), parameters identical %s; loss-scale states %s / %s' % (k, same_l, first, same_p, res[0][2], res[k][2]))
    ok = ok and same_l and same_p
sys.exit(0 if ok else 1)

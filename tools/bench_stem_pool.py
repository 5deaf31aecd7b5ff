"""bn_act_fwd + maxpool_fwd against the fused pseg_bn_act_maxpool_fwd at the ResNet stem's shape.  usage: python tools/bench_stem_pool.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_segmentation_amd import ops  # noqa: E402

for dtype in (torch.float32, torch.float16):
    B, C, H, W = 16, 64, 256, 256
    x = ops.Act(torch.randn(B * H * W * C, device='cuda').to(dtype), B, H, W, C, C)
    co = ops.bn_finalize(ops.col_stats(x), x.M, torch.ones(C).cuda(), torch.zeros(C).cuda(), None, None, 0.0, 1e-5)
    z = x.like()
    p = x.new(B, 128, 128, C)

    def sep():
        ops.bn_act_fwd(x, co, 1, z)
        ops.maxpool_fwd(z, p, 3, 2, 1)

    def fused():
        ops.bn_act_maxpool_fwd(x, co, 1, p, 3, 2, 1)

    for name, fn in (('separate', sep), ('fused', fused)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        e1.synchronize()
        print(dtype, name, '%.1f us' % (e0.elapsed_time(e1) / 50 * 1e3))

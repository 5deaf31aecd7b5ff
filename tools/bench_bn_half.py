"""Streaming rate of the fp16 BatchNorm passes at the DeepLabV3+ (B = 16, 512x512) tensor shapes, beside a plain device copy
of the same bytes.  usage: python tools/bench_bn_half.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_segmentation_amd import ops  # noqa: E402
from tools.bench_conv import timeit  # noqa: E402

SHAPES = [('l1 64ch', 16, 128, 128, 64), ('l1 256ch', 16, 128, 128, 256), ('l2 128ch', 16, 64, 64, 128),
          ('l2 512ch', 16, 64, 64, 512), ('l3 256ch', 16, 32, 32, 256), ('l3 1024ch', 16, 32, 32, 1024),
          ('l4 512ch', 16, 32, 32, 512), ('l4 2048ch', 16, 32, 32, 2048), ('stem 64ch', 16, 256, 256, 64)]


def main():
    dev = 'cuda'
    h = torch.float16
    print('%-10s %8s | %-22s | %-22s | %-22s | %-22s | copy' % ('shape', 'MB', 'fwd (r+w)', 'fwd+res (2r+w)', 'bwd reduce (2r)', 'bwd apply (2r+w)'))
    for name, B, H, W, C in SHAPES:
        n = B * H * W * C
        y = ops.Act(torch.randn(n, device=dev).to(h), B, H, W, C, C)
        res = ops.Act(torch.randn(n, device=dev).to(h), B, H, W, C, C)
        z, dz, dy = y.like(), ops.Act(torch.randn(n, device=dev).to(h), B, H, W, C, C), y.like()
        g = torch.ones(C, device=dev)
        b = torch.zeros(C, device=dev)
        st = ops.col_stats(y)
        co = ops.bn_finalize(st, y.M, g, b, None, None, 0.1, 1e-5)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        mb = n * 2 / 1e6
        t1 = timeit(lambda: ops.bn_act_fwd(y, co, ops.ACT_RELU, z), 30)
        t2 = timeit(lambda: ops.bn_act_fwd(y, co, ops.ACT_RELU, z, residual=res), 30)
        # backward = reduce + finalize + apply: time the whole, then the finalize-free estimate through the two big passes
        t3 = timeit(lambda: ops.bn_act_bwd(dz, None, y, co, ops.ACT_RELU, dy, dg, db), 30)
        a, c = y.t, z.t
        t4 = timeit(lambda: c.copy_(a), 30)

        def rate(t, passes):
            return '%6.1f us %5.2f TB/s' % (t * 1e3, passes * mb / t / 1e6 * 1e3 / 1e3)
        print('%-10s %8.1f | %s | %s | %-22s | %s | %s' % (name, mb, rate(t1, 2), rate(t2, 3), 'bwd all: ' + rate(t3, 5), '', rate(t4, 2)))


if __name__ == '__main__':
    main()

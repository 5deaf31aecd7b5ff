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
    h = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == 'fp32') else torch.float16
    es = 4 if h == torch.float32 else 2
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
        mb = n * es / 1e6
        t1 = timeit(lambda: ops.bn_act_fwd(y, co, ops.ACT_RELU, z), 30)
        t2 = timeit(lambda: ops.bn_act_fwd(y, co, ops.ACT_RELU, z, residual=res), 30)
        # backward = reduce + finalize + apply: time the whole, then the finalize-free estimate through the two big passes
        t3 = timeit(lambda: ops.bn_act_bwd(dz, None, y, co, ops.ACT_RELU, dy, dg, db), 30)
        t5 = timeit(lambda: ops.bn_act_fwd(y, None, ops.ACT_RELU, z), 30)          # no coefficients: plain activation pass
        t6 = timeit(lambda: ops.bn_act_fwd(y, None, ops.ACT_RELU, z, residual=res), 30)
        a, c = y.t, z.t
        t4 = timeit(lambda: c.copy_(a), 30)

        def rate(t, passes):
            return '%6.1f us %5.2f TB/s' % (t * 1e3, passes * mb / t / 1e6 * 1e3 / 1e3)
        print('%-10s %7.1f MB | fwd %6.1f us %4.2f TB/s | fwd+res %6.1f us %4.2f | act only %6.1f us %4.2f | act+res %6.1f us %4.2f | bwd %6.1f us %4.2f | copy %6.1f us %4.2f' % (
            name, mb, t1 * 1e3, 2 * mb / t1 / 1e3, t2 * 1e3, 3 * mb / t2 / 1e3, t5 * 1e3, 2 * mb / t5 / 1e3, t6 * 1e3, 3 * mb / t6 / 1e3,
            t3 * 1e3, 5 * mb / t3 / 1e3, t4 * 1e3, 2 * mb / t4 / 1e3))


if __name__ == '__main__':
    main()

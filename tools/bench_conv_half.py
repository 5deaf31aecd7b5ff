"""Micro-benchmark of the fp16 (half-precision policy) conv kernels at the DeepLabV3+ / ResNet-50 / HRNet shapes.
Prints time, dense TFLOP/s and the HBM floor (operands + result once, at 5.5 TB/s) per kernel.
usage: python tools/bench_conv_half.py [shape names...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_segmentation_amd import ops  # noqa: E402

SHAPES = [
    # name, B, Cin, H, W, Cout, k, stride, pad, dil
    ('aspp_d6', 16, 2048, 32, 32, 256, 3, 1, 6, 6),
    ('aspp_d12', 16, 2048, 32, 32, 256, 3, 1, 12, 12),
    ('aspp_d18', 16, 2048, 32, 32, 256, 3, 1, 18, 18),
    ('aspp_1x1', 16, 2048, 32, 32, 256, 1, 1, 0, 1),
    ('aspp_proj', 16, 1280, 32, 32, 256, 1, 1, 0, 1),
    ('low_proj', 16, 256, 128, 128, 128, 1, 1, 0, 1),
    ('cls_conv', 16, 384, 128, 128, 24, 3, 1, 1, 1),
    ('stem7x7', 16, 8, 512, 512, 64, 7, 2, 3, 1),
    ('l1_1x1a', 16, 64, 128, 128, 64, 1, 1, 0, 1),
    ('l1_3x3', 16, 64, 128, 128, 64, 3, 1, 1, 1),
    ('l1_1x1b', 16, 64, 128, 128, 256, 1, 1, 0, 1),
    ('l1_1x1c', 16, 256, 128, 128, 64, 1, 1, 0, 1),
    ('l2_3x3', 16, 128, 64, 64, 128, 3, 1, 1, 1),
    ('l2_1x1b', 16, 128, 64, 64, 512, 1, 1, 0, 1),
    ('l2_1x1c', 16, 512, 64, 64, 128, 1, 1, 0, 1),
    ('l3_3x3', 16, 256, 32, 32, 256, 3, 1, 1, 1),
    ('l3_1x1b', 16, 256, 32, 32, 1024, 1, 1, 0, 1),
    ('l3_1x1c', 16, 1024, 32, 32, 256, 1, 1, 0, 1),
    ('l4_3x3d2', 16, 512, 32, 32, 512, 3, 1, 2, 2),
    ('l4_1x1b', 16, 512, 32, 32, 2048, 1, 1, 0, 1),
    ('l4_1x1a', 16, 2048, 32, 32, 512, 1, 1, 0, 1),
    # pure GEMMs through the same kernels (1x1 convs): the contraction of layer 4's 3x3 conv without its taps, and a shape with
    # eight tiles per CU -- what the kernel structure itself reaches when nothing conv-specific is in the way
    ('gemm_l4', 16, 4608, 32, 32, 512, 1, 1, 0, 1),
    ('gemm_big', 16, 4096, 64, 64, 1024, 1, 1, 0, 1),
    # the same weight-gradient tiles over 1/4, 1x and 4x the pixels: fixed cost against cost per K-step
    ('l3_1x1b_q', 4, 256, 32, 32, 1024, 1, 1, 0, 1),
    ('l3_1x1b_4x', 64, 256, 32, 32, 1024, 1, 1, 0, 1),
    ('l3_3x3_q', 4, 256, 32, 32, 256, 3, 1, 1, 1),
    ('l3_3x3_4x', 64, 256, 32, 32, 256, 3, 1, 1, 1),
    ('hr_32', 8, 32, 128, 128, 32, 3, 1, 1, 1),
    ('hr_64', 8, 64, 64, 64, 64, 3, 1, 1, 1),
    ('hr_128', 8, 128, 32, 32, 128, 3, 1, 1, 1),
    ('hr_256', 8, 256, 16, 16, 256, 3, 1, 1, 1),
    ('hr_32s2', 8, 32, 128, 128, 64, 3, 2, 1, 1),
]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    only = sys.argv[1:] or None
    skip_default = ('gemm_l4', 'gemm_big', 'l3_1x1b_q', 'l3_1x1b_4x', 'l3_3x3_q', 'l3_3x3_4x')
    tot = [0.0, 0.0, 0.0]
    for name, B, Cin, H, W, Cout, k, s, p, d in SHAPES:
        if (only and name not in only) or (not only and name in skip_default):
            continue
        Ho, Wo = ops.conv_out_size(H, k, s, p, d), ops.conv_out_size(W, k, s, p, d)
        h = torch.float16
        x = ops.Act(torch.randn(B * H * W * Cin, device='cuda').to(h), B, H, W, Cin, Cin)
        w = (torch.randn(Cout * k * k * Cin, device='cuda') * 0.02).to(h)
        wT = w.view(Cout, k * k, Cin).permute(2, 1, 0).contiguous().view(-1)
        y = ops.Act.empty(B, Ho, Wo, Cout, 'cuda', dtype=h)
        dy = ops.Act(torch.randn(B * Ho * Wo * Cout, device='cuda').to(h), B, Ho, Wo, Cout, Cout)
        dx = ops.Act.empty(B, H, W, Cin, 'cuda', dtype=h)
        dw = torch.empty(Cout * k * k * Cin, device='cuda')
        flop = 2.0 * B * Ho * Wo * Cout * Cin * k * k
        iters = 20
        tf = timeit(lambda: ops.conv2d_fwd(x, w, None, y, k, k, s, p, d, want_stats=True), iters)
        td = timeit(lambda: ops.conv2d_dgrad(dy, wT, dx, k, k, s, p, d), iters)
        tw = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, k, k, s, p, d), iters)
        byt = 2.0 * (x.M * Cin + y.M * Cout + w.numel())
        floor = byt / 5.5e12 * 1e3
        tot[0] += tf
        tot[1] += td
        tot[2] += tw
        print('%-10s %7.1f GF floor %6.3f ms | fwd %7.3f ms %6.0f TF | dgrad %7.3f ms %6.0f TF | wgrad %7.3f ms %6.0f TF' % (
            name, flop / 1e9, floor, tf, flop / tf / 1e9, td, flop / td / 1e9, tw, flop / tw / 1e9), flush=True)
    print('total fwd %.2f ms dgrad %.2f ms wgrad %.2f ms' % tuple(tot))


if __name__ == '__main__':
    main()

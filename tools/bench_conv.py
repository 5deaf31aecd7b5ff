"""Micro-benchmark of the implicit-GEMM conv kernels at the DeepLabV3+ / ResNet-50 shapes (B=16, 512x512).
Prints achieved dense TFLOP/s per kernel; fp32-MFMA peak on MI355X is 157.3 TF."""
import json
import os
import sys
import time

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
    ('stem7x7', 16, 4, 512, 512, 64, 7, 2, 3, 1),
    ('l1_1x1a', 16, 64, 128, 128, 64, 1, 1, 0, 1),
    ('l1_3x3', 16, 64, 128, 128, 64, 3, 1, 1, 1),
    ('l1_1x1b', 16, 64, 128, 128, 256, 1, 1, 0, 1),
    ('l1_1x1c', 16, 256, 128, 128, 64, 1, 1, 0, 1),
    ('l2_1x1a', 16, 256, 128, 128, 128, 1, 1, 0, 1),
    ('l2_1x1b', 16, 128, 64, 64, 512, 1, 1, 0, 1),
    ('l2_1x1c', 16, 512, 64, 64, 128, 1, 1, 0, 1),
    ('l3_1x1c', 16, 1024, 32, 32, 256, 1, 1, 0, 1),
    ('l2_3x3', 16, 128, 64, 64, 128, 3, 1, 1, 1),
    ('l3_3x3', 16, 256, 32, 32, 256, 3, 1, 1, 1),
    ('l3_1x1b', 16, 256, 32, 32, 1024, 1, 1, 0, 1),
    ('l4_3x3d2', 16, 512, 32, 32, 512, 3, 1, 2, 2),
    ('l4_1x1b', 16, 512, 32, 32, 2048, 1, 1, 0, 1),
    ('l4_1x1a', 16, 2048, 32, 32, 512, 1, 1, 0, 1),
    # HRNet (configs[4]: B = 8, 512x512): the four resolution branches' 3x3 convs and a stride-2 fusion conv
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
    prec = 'fp32'
    argv = sys.argv[1:]
    fp16 = False
    if argv and argv[0] == 'fp16x3':
        fp16 = True
        argv.pop(0)
    if argv and argv[0] in ('fp32', 'mixed', 'limb', 'bf16x3', 'bf16x6'):
        prec = argv.pop(0)
    ops.set_conv_precision(prec)
    print('precision', prec)
    only = argv or None
    rows = []
    for name, B, Cin, H, W, Cout, k, s, p, d in SHAPES:
        if only and name not in only:
            continue
        Ho, Wo = ops.conv_out_size(H, k, s, p, d), ops.conv_out_size(W, k, s, p, d)
        x = ops.Act(torch.randn(B * H * W * Cin, device='cuda'), B, H, W, Cin, Cin)
        w = torch.randn(Cout * k * k * Cin, device='cuda') * 0.02
        y = ops.Act.empty(B, Ho, Wo, Cout, 'cuda')
        dy = ops.Act(torch.randn(B * Ho * Wo * Cout, device='cuda'), B, Ho, Wo, Cout, Cout)
        dx = ops.Act.empty(B, H, W, Cin, 'cuda')
        dw = torch.empty_like(w)
        wT = ops.filter_transpose(w, Cout, k * k, Cin)
        flop = 2.0 * B * Ho * Wo * Cout * Cin * k * k
        iters = 10 if flop > 5e10 else 30
        kwf, kwd = {}, {}
        if fp16:
            kwf = dict(precision=ops.PREC_FP16X3, amax_x=ops.amax_of(x), amax_w=ops.amax_of(w))
            kwd = dict(precision=ops.PREC_FP16X3, amax_dy=ops.amax_of(dy), amax_w=ops.amax_of(wT))
        tf = timeit(lambda: ops.conv2d_fwd(x, w, None, y, k, k, s, p, d, want_stats=True, **kwf), iters)
        td = timeit(lambda: ops.conv2d_dgrad(dy, wT, dx, k, k, s, p, d, **kwd), iters)
        tw = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, k, k, s, p, d), iters)
        row = dict(name=name, gflop=flop / 1e9, fwd_ms=tf, dgrad_ms=td, wgrad_ms=tw, fwd_tf=flop / tf / 1e9,
                   dgrad_tf=flop / td / 1e9, wgrad_tf=flop / tw / 1e9)
        rows.append(row)
        print('%-10s %8.1f GF  fwd %7.3f ms %6.1f TF | dgrad %7.3f ms %6.1f TF | wgrad %7.3f ms %6.1f TF' % (
            name, row['gflop'], tf, row['fwd_tf'], td, row['dgrad_tf'], tw, row['wgrad_tf']), flush=True)
    tot = sum(r['gflop'] for r in rows)
    print('total fwd %.2f ms dgrad %.2f ms wgrad %.2f ms' % (sum(r['fwd_ms'] for r in rows), sum(r['dgrad_ms'] for r in rows),
                                                           sum(r['wgrad_ms'] for r in rows)))
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(rows, open('gpurun_out/bench_conv.json', 'w'), indent=1)


if __name__ == '__main__':
    main()

"""fp32 vs limb kernels on narrow / small-map shapes (HRNet, UNet): where does limb arithmetic pay?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_segmentation_amd import ops
from bench_conv import timeit
SHAPES = [(8, 32, 128, 32, 3), (8, 64, 64, 64, 3), (8, 128, 32, 128, 3), (8, 256, 16, 256, 3), (8, 64, 128, 64, 3),
          (8, 64, 128, 256, 1), (8, 256, 128, 64, 1), (16, 64, 128, 64, 3), (16, 128, 64, 128, 3), (16, 256, 32, 256, 3),
          (8, 96, 16, 576, 1), (8, 576, 16, 96, 1), (8, 144, 64, 24, 1), (8, 32, 32, 192, 1), (8, 1280, 8, 256, 3), (8, 352, 16, 128, 3)]
for B, Cin, S, Cout, k in SHAPES:
    p = k // 2
    x = ops.Act(torch.randn(B * S * S * Cin, device='cuda'), B, S, S, Cin, Cin)
    w = torch.randn(Cout * k * k * Cin, device='cuda') * 0.02
    y = ops.Act.empty(B, S, S, Cout, 'cuda')
    dy = ops.Act(torch.randn(B * S * S * Cout, device='cuda'), B, S, S, Cout, Cout)
    dx = ops.Act.empty(B, S, S, Cin, 'cuda')
    dw = torch.empty_like(w)
    wT = ops.filter_transpose(w, Cout, k * k, Cin)
    ax, aw, ady, awt = ops.amax_of(x), ops.amax_of(w), ops.amax_of(dy), ops.amax_of(wT)
    r = {}
    for name, P in (('fp32', ops.PREC_FP32), ('bf16x3', ops.PREC_BF16X3)):
        r[name] = (timeit(lambda: ops.conv2d_fwd(x, w, None, y, k, k, 1, p, 1, want_stats=True, precision=P), 20),
                   timeit(lambda: ops.conv2d_dgrad(dy, wT, dx, k, k, 1, p, 1, precision=P), 20),
                   timeit(lambda: ops.conv2d_wgrad(x, dy, dw, k, k, 1, p, 1, precision=P), 20))
    f16 = timeit(lambda: ops.conv2d_fwd(x, w, None, y, k, k, 1, p, 1, want_stats=True, precision=ops.PREC_FP16X3, amax_x=ax, amax_w=aw), 20)
    print('B%d Cin%4d S%3d Cout%4d k%d  M=%6d | fwd fp32 %.3f fp16x3 %.3f | dgrad fp32 %.3f bf16x3 %.3f | wgrad fp32 %.3f bf16x3 %.3f' % (
        B, Cin, S, Cout, k, B * S * S, r['fp32'][0], f16, r['fp32'][1], r['bf16x3'][1], r['fp32'][2], r['bf16x3'][2]), flush=True)

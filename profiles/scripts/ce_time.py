import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
from pytorch_segmentation_amd import ops
from bench_conv import timeit
B, C, h, w = 16, 21, 128, 128
lr = ops.Act(torch.randn(B * h * w * 24, device='cuda'), B, h, w, 24, 24)
t = torch.randint(0, C, (B, 4 * h, 4 * w), device='cuda')
def three():
    full = ops.bilinear_fwd_nchw(lr, C, 4 * h, 4 * w, True)
    o, d = ops.ce_fwd_bwd(full, t)
    d3 = ops.Act.empty(B, h, w, 24, 'cuda', zero=True)
    ops.bilinear_bwd_nchw(d, d3, C, True)
    return o
def fused():
    return ops.ce_upsampled_fwd_bwd(lr, C, t, True)[0]
print('three-pass %.3f ms   fused %.3f ms' % (timeit(three, 20), timeit(fused, 20)))

"""Standalone timings of the small HBM-bound kernels of a DeepLabV3+ step (B=16, 512x512)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_segmentation_amd import ops
from tools.bench_conv import timeit
dev = 'cuda'
f0 = ops.Act(torch.randn(16 * 256 * 256 * 64, device=dev), 16, 256, 256, 64, 64)
p = ops.Act.empty(16, 128, 128, 64, dev)
arg = ops.maxpool_fwd(f0, p, 3, 2, 1, want_argmax=True)
print('maxpool_fwd %.3f ms' % timeit(lambda: ops.maxpool_fwd(f0, p, 3, 2, 1, want_argmax=True), 20))
d = ops.Act(torch.randn(16 * 128 * 128 * 64, device=dev), 16, 128, 128, 64, 64)
df0 = f0.like()
print('maxpool_bwd %.3f ms  (ideal %.3f at 5 TB/s)' % (timeit(lambda: ops.maxpool_bwd(d, arg, df0, 3, 2, 1), 20), (67 + 17 + 268) / 5e3))
logits = torch.randn(16, 21, 512, 512, device=dev)
tgt = torch.randint(0, 21, (16, 512, 512), device=dev)
print('ce_fwd_bwd %.3f ms  (ideal %.3f)' % (timeit(lambda: ops.ce_fwd_bwd(logits, tgt), 20), (88 + 88 + 34) / 5e3))
dcat = ops.Act(torch.randn(16 * 128 * 128 * 384, device=dev), 16, 128, 128, 384, 384)
da = ops.Act.empty(16, 32, 32, 256, dev)
print('bilinear_bwd_nhwc %.3f ms (ideal %.3f)' % (timeit(lambda: ops.bilinear_bwd(dcat.slice(0, 256), da, True), 20), 268 / 5e3))
lr = ops.Act(torch.randn(16 * 128 * 128 * 24, device=dev), 16, 128, 128, 24, 24)
print('bilinear_fwd_nchw %.3f ms (ideal %.3f)' % (timeit(lambda: ops.bilinear_fwd_nchw(lr, 21, 512, 512, True), 20), 88 / 5e3))
dout = torch.randn(16, 21, 512, 512, device=dev)
dlr = ops.Act.empty(16, 128, 128, 24, dev, zero=True)
print('bilinear_bwd_nchw %.3f ms (ideal %.3f)' % (timeit(lambda: ops.bilinear_bwd_nchw(dout, dlr, 21, True), 20), 88 / 5e3))

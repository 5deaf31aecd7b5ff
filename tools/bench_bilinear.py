import sys, torch
sys.path.insert(0, '/root/repo')
from pytorch_segmentation_amd import ops
from tools.bench_conv import timeit
lr = ops.Act(torch.randn(16*128*128*24, device='cuda'), 16, 128, 128, 24, 24)
t = timeit(lambda: ops.bilinear_fwd_nchw(lr, 21, 512, 512, True), 20)
dout = torch.randn(16, 21, 512, 512, device='cuda')
dlr = ops.Act.empty(16, 128, 128, 24, 'cuda', zero=True)
t2 = timeit(lambda: ops.bilinear_bwd_nchw(dout, dlr, 21, True), 20)
print('fwd_nchw %.3f ms   bwd_nchw %.3f ms' % (t, t2))

"""ASPP atrous convs (16 x 2048 x 32 x 32 -> 256, 3x3, rates 6 / 12 / 18), fp32: fwd / dgrad / wgrad time under planner
overrides given as NAME=VALUE,... groups on the command line (e.g. PSEG_WGRAD_SPLITS=4 PSEG_CONV_NOBAND=1)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_segmentation_amd import ops, _lib
from bench_conv import timeit
B, Cin, S, Cout, k = 16, 2048, 32, 256, 3
x = ops.Act(torch.randn(B * S * S * Cin, device='cuda'), B, S, S, Cin, Cin)
w = torch.randn(Cout * k * k * Cin, device='cuda') * 0.02
y = ops.Act.empty(B, S, S, Cout, 'cuda')
dy = ops.Act(torch.randn(B * S * S * Cout, device='cuda'), B, S, S, Cout, Cout)
dx = ops.Act.empty(B, S, S, Cin, 'cuda')
dw = torch.empty_like(w)
wT = ops.filter_transpose(w, Cout, k * k, Cin)
for group in ([''] + sys.argv[1:]):
    keys = []
    for kv in filter(None, group.split(',')):
        n, v = kv.split('=')
        os.environ[n] = v
        keys.append(n)
    _lib.clear_query_cache()
    for d in (6, 12, 18):
        f = timeit(lambda: ops.conv2d_fwd(x, w, None, y, k, k, 1, d, d, want_stats=True, precision=ops.PREC_FP32), 10)
        g = timeit(lambda: ops.conv2d_dgrad(dy, wT, dx, k, k, 1, d, d, precision=ops.PREC_FP32), 10)
        h = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, k, k, 1, d, d, precision=ops.PREC_FP32), 10)
        print('%-40s d=%2d  fwd %.3f  dgrad %.3f  wgrad %.3f ms' % (group or 'default', d, f, g, h), flush=True)
    for n in keys:
        del os.environ[n]

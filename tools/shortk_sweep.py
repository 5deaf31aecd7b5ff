"""Short-K / skinny convs of the ResNet-50 encoder at B=16, 512x512 (fp32): fwd / dgrad / wgrad time under planner
overrides given as NAME=VALUE,... groups (e.g. PSEG_CONV_BM=128,PSEG_CONV_BN=64)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_segmentation_amd import ops, _lib
from bench_conv import timeit
SHAPES = [('l1_1x1b 64->256 @128', 16, 64, 128, 256, 1), ('l1_1x1c 256->64 @128', 16, 256, 128, 64, 1),
          ('l1_1x1a 64->64 @128', 16, 64, 128, 64, 1), ('l2_1x1b 128->512 @64', 16, 128, 64, 512, 1),
          ('l2_1x1c 512->128 @64', 16, 512, 64, 128, 1), ('low_proj 256->128 @128', 16, 256, 128, 128, 1),
          ('l1_3x3 64->64 @128', 16, 64, 128, 64, 3), ('l3_1x1c 1024->256 @32', 16, 1024, 32, 256, 1)]
groups = [''] + sys.argv[1:]
for name, B, Cin, S, Cout, k in SHAPES:
    p = k // 2
    x = ops.Act(torch.randn(B * S * S * Cin, device='cuda'), B, S, S, Cin, Cin)
    w = torch.randn(Cout * k * k * Cin, device='cuda') * 0.02
    y = ops.Act.empty(B, S, S, Cout, 'cuda')
    dy = ops.Act(torch.randn(B * S * S * Cout, device='cuda'), B, S, S, Cout, Cout)
    dx = ops.Act.empty(B, S, S, Cin, 'cuda')
    dw = torch.empty_like(w)
    wT = ops.filter_transpose(w, Cout, k * k, Cin)
    gf = 2.0 * B * S * S * Cin * Cout * k * k / 1e9
    for group in groups:
        keys = []
        for kv in filter(None, group.split(',')):
            n, v = kv.split('=')
            os.environ[n] = v
            keys.append(n)
        _lib.clear_query_cache()
        f = timeit(lambda: ops.conv2d_fwd(x, w, None, y, k, k, 1, p, 1, want_stats=True, precision=ops.PREC_FP32), 20)
        g = timeit(lambda: ops.conv2d_dgrad(dy, wT, dx, k, k, 1, p, 1, precision=ops.PREC_FP32), 20)
        h = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, k, k, 1, p, 1, precision=ops.PREC_FP32), 20)
        print('%-24s %-34s fwd %.3f (%5.1f TF)  dgrad %.3f (%5.1f)  wgrad %.3f (%5.1f)' % (
            name, group or 'default', f, gf / f, g, gf / g, h, gf / h), flush=True)
        for n in keys:
            del os.environ[n]
    _lib.clear_query_cache()

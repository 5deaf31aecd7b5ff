import sys, torch
sys.path.insert(0, '/root/repo')
from pytorch_segmentation_amd import ops
from tools.bench_conv import timeit
ops.set_conv_precision('mixed')
for B in (2, 4, 8, 16, 32):
    Cin, H, W, Cout, k, d = 2048, 32, 32, 256, 3, 6
    x = ops.Act(torch.randn(B * H * W * Cin, device='cuda'), B, H, W, Cin, Cin)
    w = torch.randn(Cout * k * k * Cin, device='cuda') * 0.02
    y = ops.Act.empty(B, H, W, Cout, 'cuda')
    flop = 2.0 * B * H * W * Cout * Cin * k * k
    t = timeit(lambda: ops.conv2d_fwd(x, w, None, y, k, k, 1, d, d, want_stats=True), 20)
    print('ASPP d6 B=%2d: %.3f ms  %.1f TF (input %d MB)' % (B, t, flop / t / 1e9, B * H * W * Cin * 4 >> 20))

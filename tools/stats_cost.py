import sys, torch
sys.path.insert(0, '/root/repo')
from pytorch_segmentation_amd import ops
from tools.bench_conv import timeit
ops.set_conv_precision('mixed')
for name, B, Cin, H, W, Cout, k, s, p, d in [('l1_1x1b', 16, 64, 128, 128, 256, 1, 1, 0, 1), ('l1_1x1a', 16, 64, 128, 128, 64, 1, 1, 0, 1), ('l1_3x3', 16, 64, 128, 128, 64, 3, 1, 1, 1), ('low_proj', 16, 256, 128, 128, 128, 1, 1, 0, 1), ('l2_1x1b', 16, 128, 64, 64, 512, 1, 1, 0, 1)]:
    x = ops.Act(torch.randn(B * H * W * Cin, device='cuda'), B, H, W, Cin, Cin)
    w = torch.randn(Cout * k * k * Cin, device='cuda') * 0.02
    y = ops.Act.empty(B, H, W, Cout, 'cuda')
    t1 = timeit(lambda: ops.conv2d_fwd(x, w, None, y, k, k, s, p, d, want_stats=True), 30)
    t0 = timeit(lambda: ops.conv2d_fwd(x, w, None, y, k, k, s, p, d, want_stats=False), 30)
    gb = (x.t.numel() + y.t.numel()) * 4 / 1e9
    print('%-9s stats %.3f ms  nostats %.3f ms   %.2f GB -> %.2f TB/s (nostats)' % (name, t1, t0, gb, gb / t0))

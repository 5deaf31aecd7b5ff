import os, sys, torch
sys.path.insert(0, '/root/repo')
from pytorch_segmentation_amd import ops
B, Cin, S, Cout, k = 16, 2048, 32, 256, 3
x = ops.Act(torch.randn(B * S * S * Cin, device='cuda'), B, S, S, Cin, Cin)
w = torch.randn(Cout * k * k * Cin, device='cuda') * 0.02
y = ops.Act.empty(B, S, S, Cout, 'cuda')
for d in (6, 12):
    print('--- fwd d', d, flush=True)
    ops.conv2d_fwd(x, w, None, y, k, k, 1, d, d, want_stats=True, precision=ops.PREC_FP32)
    torch.cuda.synchronize()

"""Parity + speed of the pre-split LDS-DMA limb data gradient against the in-kernel-split one."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_segmentation_amd import ops

def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

CASES = [  # B, Cin, H, W, Cout, k, stride, pad, dil
    (16, 512, 32, 32, 512, 3, 1, 2, 2), (16, 2048, 32, 32, 256, 3, 1, 6, 6), (16, 2048, 32, 32, 256, 3, 1, 12, 12),
    (16, 2048, 32, 32, 256, 3, 1, 18, 18), (16, 512, 32, 32, 2048, 1, 1, 0, 1), (16, 256, 32, 32, 256, 3, 1, 1, 1),
    (16, 1024, 32, 32, 256, 1, 1, 0, 1), (16, 256, 128, 128, 64, 1, 1, 0, 1), (16, 256, 64, 64, 256, 3, 2, 1, 1),
    (16, 128, 64, 64, 128, 3, 1, 1, 1), (16, 256, 128, 128, 128, 1, 1, 0, 1), (2, 128, 24, 24, 64, 3, 1, 6, 6),
]
if len(sys.argv) > 1:
    CASES = CASES[:int(sys.argv[1])]
torch.manual_seed(0)
for (B, Cin, H, W, Cout, k, s, p, d) in CASES:
    Ho, Wo = ops.conv_out_size(H, k, s, p, d), ops.conv_out_size(W, k, s, p, d)
    dy = ops.Act(torch.randn(B * Ho * Wo * Cout, device='cuda'), B, Ho, Wo, Cout, Cout)
    w = torch.randn(Cout * k * k * Cin, device='cuda') * 0.02
    wT = ops.filter_transpose(w, Cout, k * k, Cin)
    dx0 = ops.Act.empty(B, H, W, Cin, 'cuda'); dx1 = ops.Act.empty(B, H, W, Cin, 'cuda'); dx2 = ops.Act.empty(B, H, W, Cin, 'cuda')
    ok = ops.dgrad_planes_ok(dy, dx0, k, k, s, p, d)
    ops.conv2d_dgrad(dy, wT, dx0, k, k, s, p, d, precision=ops.PREC_FP32)
    t_old = timeit(lambda: ops.conv2d_dgrad(dy, wT, dx1, k, k, s, p, d, precision=ops.PREC_BF16X3))
    flop = 2.0 * B * Ho * Wo * Cout * Cin * k * k
    line = '%-44s ok=%d  old %.3f ms %6.1f TF' % ((B, Cin, H, W, Cout, k, s, p, d), ok, t_old, flop / t_old / 1e9)
    if ok:
        wp = ops.split_planes(wT.view(Cin, k * k * Cout))
        t_split = timeit(lambda: ops.split_planes(dy))
        dp = ops.split_planes(dy)
        t_new = timeit(lambda: ops.conv2d_dgrad_planes(dp, dy, wp, dx2, k, k, s, p, d))
        ref = dx0.t.double()
        e_old = ((dx1.t.double() - ref).abs().max() / ref.abs().max()).item()
        e_new = ((dx2.t.double() - ref).abs().max() / ref.abs().max()).item()
        e_on = ((dx2.t.double() - dx1.t.double()).abs().max() / ref.abs().max()).item()
        # accumulate
        ops.conv2d_dgrad_planes(dp, dy, wp, dx2, k, k, s, p, d, accumulate=True)
        e_acc = ((dx2.t.double() - 2 * ref).abs().max() / (2 * ref.abs().max())).item()
        line += ' | new %.3f ms %6.1f TF (split pass %.3f ms) | err vs fp32: old %.1e new %.1e, new-old %.1e, acc %.1e' % (
            t_new, flop / t_new / 1e9, t_split, e_old, e_new, e_on, e_acc)
    print(line, flush=True)

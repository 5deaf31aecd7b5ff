"""Feasibility of forward pipelining over batch halves (round 6): in the forward pass conv -> statistics -> finalize -> BatchNorm apply is a
serial chain and the matrix pipe idles during the apply passes (1.94 ms of a 14.5 ms forward pass).  With the batch split in halves A | B the
apply of B can run (second stream) beside the NEXT layer's conv of A: conv_l(A) conv_l(B) finalize_l apply_l(A) { apply_l(B) || conv_{l+1}(A) } ...
This script times a chain of L pointwise conv + BatchNorm + ReLU layers at a ResNet shape both ways with the production ops."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_segmentation_amd import ops
from pytorch_segmentation_amd.ops import Act

ops.set_conv_precision('fp32')
dev = torch.device('cuda', 0)


def run(B, H, W, C1, C2, L, mode, iters=20):
    g = torch.Generator(device='cuda').manual_seed(0)
    chans = [C1 if i % 2 == 0 else C2 for i in range(L + 1)]
    ws = [torch.empty(chans[i + 1] * chans[i], device=dev).uniform_(-0.05, 0.05, generator=g) for i in range(L)]
    gam = [torch.ones(chans[i + 1], device=dev) for i in range(L)]
    bet = [torch.zeros(chans[i + 1], device=dev) for i in range(L)]
    x0 = Act.empty(B, H, W, chans[0], dev)
    x0.t.uniform_(-1, 1, generator=g)
    ys = [Act.empty(B, H, W, chans[i + 1], dev) for i in range(L)]
    zs = [Act.empty(B, H, W, chans[i + 1], dev) for i in range(L)]
    side = torch.cuda.Stream()
    evs = [torch.cuda.Event() for _ in range(2 * L + 2)]
    nb = B // 2

    def full():
        x = x0
        for i in range(L):
            st = ops.conv2d_fwd(x, ws[i], None, ys[i], 1, 1, 1, 0, 1, want_stats=True)
            co = ops.bn_finalize(st, ys[i].M, gam[i], bet[i], None, None, 0.0, 1e-5)
            ops.bn_act_fwd(ys[i], co, 1, zs[i])
            x = zs[i]

    def halves():
        cur = torch.cuda.current_stream()
        x = x0
        pendB = None          # event: apply of half B of the previous layer done (on the side stream)
        for i in range(L):
            sa = ops.conv2d_fwd(ops._sub(x, 0, nb), ws[i], None, ops._sub(ys[i], 0, nb), 1, 1, 1, 0, 1, want_stats=True)
            if pendB is not None:
                cur.wait_event(pendB)
            sb = ops.conv2d_fwd(ops._sub(x, nb, nb), ws[i], None, ops._sub(ys[i], nb, nb), 1, 1, 1, 0, 1, want_stats=True)
            st = (torch.cat([sa[0], sb[0]], dim=1), sa[1] + sb[1], sa[2])
            co = ops.bn_finalize(st, ys[i].M, gam[i], bet[i], None, None, 0.0, 1e-5)
            ops.bn_act_fwd(ops._sub(ys[i], 0, nb), co, 1, ops._sub(zs[i], 0, nb))
            evs[2 * i].record(cur)
            side.wait_event(evs[2 * i])
            with torch.cuda.stream(side):
                ops.bn_act_fwd(ops._sub(ys[i], nb, nb), co, 1, ops._sub(zs[i], nb, nb))
                evs[2 * i + 1].record(side)
            pendB = evs[2 * i + 1]
            x = zs[i]
        cur.wait_event(pendB)

    fn = full if mode == 'full' else halves
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, zs[-1].t.double().abs().sum().item()


for name, (B, H, W, C1, C2) in {'layer3 1x1 256<->1024 @32x32': (16, 32, 32, 256, 1024), 'layer2 1x1 128<->512 @64x64': (16, 64, 64, 128, 512),
                                'layer1 1x1 64<->256 @128x128': (16, 128, 128, 64, 256), 'layer4 1x1 512<->2048 @32x32': (16, 32, 32, 512, 2048)}.items():
    tf, cf = run(B, H, W, C1, C2, 8, 'full')
    th, ch = run(B, H, W, C1, C2, 8, 'halves')
    print('%-34s 8 layers: full batch %.3f ms, halves pipelined %.3f ms (%+.1f %%)   checksum rel diff %.1e' % (name, tf, th, 100 * (th / tf - 1), abs(cf - ch) / cf))

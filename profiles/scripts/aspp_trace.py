"""Per-block timing of the class-sorted ASPP forward conv (16 x 2048 x 32 x 32 -> 256, 3x3, rate d) from the trace build:
block duration by live-tap class (all 512 blocks are resident from t = 0: two per CU).
    PSEG_BUILD_TRACE=1 python -m pytorch_segmentation_amd.csrc.build --force
usage: python tools/aspp_trace.py d"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_segmentation_amd import ops, _lib
d = int(sys.argv[1])
B, Cin, S, Cout, k = 16, 2048, 32, 256, 3
x = ops.Act(torch.randn(B * S * S * Cin, device='cuda'), B, S, S, Cin, Cin)
w = torch.randn(Cout * k * k * Cin, device='cuda') * 0.02
y = ops.Act.empty(B, S, S, Cout, 'cuda')
run = lambda: ops.conv2d_fwd(x, w, None, y, k, k, 1, d, d, want_stats=True, precision=ops.PREC_FP32)
for _ in range(3):
    run()
buf = torch.zeros(5 * 65536, dtype=torch.int64, device='cuda')
torch.cuda.synchronize()
_lib.call('pseg_debug_conv_trace', buf.data_ptr())
run()
torch.cuda.synchronize()
_lib.call('pseg_debug_conv_trace', 0)
t = buf.view(-1, 5).cpu()
n = int((t[:, 3] != 0).sum())
t = t[:n]
tick = 1e-2
t0 = int(t[:, 0].min())
print('d=%d: %d blocks, launch span %.1f us' % (d, n, (int(t[:, 3].max()) - t0) * tick))
# live taps of a pixel class per axis
def live(h):
    return sum(1 for tt in (-1, 0, 1) if 0 <= h + tt * d < S)
cls = sorted(((live(h) * live(ww)) for h in range(S) for ww in range(S)), reverse=True)   # per image, descending
# class-sorted GEMM rows: class by class, inside a class image by image -> taps of row r
import collections
cnt = collections.Counter(cls)
rows = []
for taps in sorted(cnt, reverse=True):
    rows += [taps] * (cnt[taps] * B)
gridN = 4 if n == 512 else (n * 128 // (B * S * S) if n else 1)
BM = B * S * S * gridN // n
def tile_taps(tm):
    return max(rows[tm * BM:(tm + 1) * BM])
dur = {}
for bid in range(n):
    tile = (bid & ~255) + (bid & 7) * 32 + ((bid & 255) >> 3)
    tp = tile_taps(tile // gridN)
    dt = (int(t[bid, 3]) - int(t[bid, 0])) * tick
    dur.setdefault(tp, []).append(dt)
for tp in sorted(dur, reverse=True):
    v = torch.tensor(dur[tp])
    print('  %d-tap tiles: %4d blocks, duration mean %.1f us (min %.1f max %.1f) = %.1f us per live tap' % (
        tp, len(v), v.mean(), v.min(), v.max(), v.mean() / tp))

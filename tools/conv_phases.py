"""Phase timing inside the exact-fp32 LDS-DMA gather kernel (pseg_debug_conv_trace): per block, time from entry to the
first tile landed (DMA latency), to the last MFMA (main loop), to the stores drained (epilogue); plus blocks per CU and
the launch's span.  Needs a trace build of the library:
    PSEG_BUILD_TRACE=1 python -m pytorch_segmentation_amd.csrc.build --force
usage: python tools/conv_phases.py B Cin S Cout k [dgrad]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_segmentation_amd import ops, _lib
B, Cin, S, Cout, k = [int(a) for a in sys.argv[1:6]]
dgrad = len(sys.argv) > 6
p = k // 2
x = ops.Act(torch.randn(B * S * S * Cin, device='cuda'), B, S, S, Cin, Cin)
w = torch.randn(Cout * k * k * Cin, device='cuda') * 0.02
y = ops.Act.empty(B, S, S, Cout, 'cuda')
dy = ops.Act(torch.randn(B * S * S * Cout, device='cuda'), B, S, S, Cout, Cout)
dx = ops.Act.empty(B, S, S, Cin, 'cuda')
wT = ops.filter_transpose(w, Cout, k * k, Cin)
def run():
    if dgrad:
        ops.conv2d_dgrad(dy, wT, dx, k, k, 1, p, 1, precision=ops.PREC_FP32)
    else:
        ops.conv2d_fwd(x, w, None, y, k, k, 1, p, 1, want_stats=True, precision=ops.PREC_FP32)
for _ in range(3):
    run()
buf = torch.zeros(5 * 65536, dtype=torch.int64, device='cuda')
torch.cuda.synchronize()
_lib.call('pseg_debug_conv_trace', buf.data_ptr())
run()
torch.cuda.synchronize()
_lib.call('pseg_debug_conv_trace', 0)
t = buf.view(-1, 5).cpu()
t = t[t[:, 3] != 0]
n = t.shape[0]
t0 = t[:, 0].min()
tick = 1e-2    # wall_clock64: 100 MHz -> 10 ns
ld, mm, ep, tot = [(t[:, a] - t[:, b]).double() * tick for a, b in ((1, 0), (2, 1), (3, 2), (3, 0))]
print('%d blocks; launch span %.1f us' % (n, float(t[:, 3].max() - t0) * tick))
for name, v in (('entry -> first tile landed', ld), ('main loop', mm), ('epilogue (stores drained)', ep), ('block total', tot)):
    print('  %-28s mean %6.2f us  p10 %6.2f  p50 %6.2f  p90 %6.2f' % (name, v.mean(), v.quantile(0.1), v.quantile(0.5), v.quantile(0.9)))
cu = {}
for i in range(n):
    cu.setdefault(int(t[i, 4]) & 0xFF00 | ((int(t[i, 4]) >> 13) & 7) << 16, []).append(i)
print('  (SE, CU) slots seen per XCC-less id: %d; blocks per slot ~%.1f' % (len(cu), n / max(len(cu), 1)))
start = ((t[:, 0] - t0).double() * tick)
print('  block start times: p10 %.1f p50 %.1f p90 %.1f us' % (start.quantile(0.1), start.quantile(0.5), start.quantile(0.9)))

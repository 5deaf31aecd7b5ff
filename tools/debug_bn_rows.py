"""The shape at which the gfx950 store-data hazard showed (csrc/half_io.h, profiles/EXPERIMENTS.md 0.13): fp16 BatchNorm forward,
M = 65536 pixels, C = 64 channels (8 x 32 thread blocks of bn_act_fwd_rows_kernel).  Compares the kernel with torch and prints where
the results differ.  With the offset of the 128-bit buffer store in an SGPR ~1700 elements were wrong, differently on every run;
now: bad 0.  usage: python tools/debug_bn_rows.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_segmentation_amd import ops
B, C, H, W = 4, 64, 128, 128
torch.manual_seed(0)
M = B * H * W
y = ops.Act((torch.randn(M * C, device='cuda') * 2 + 1).half(), B, H, W, C, C)
g = torch.ones(C, device='cuda'); b = torch.zeros(C, device='cuda')
co = ops.bn_finalize(ops.col_stats(y), M, g, b, None, None, 0.1, 1e-5)
z = y.like(); z.t.fill_(-7.0)
ops.bn_act_fwd(y, co, ops.ACT_RELU, z)
torch.cuda.synchronize()
yy = y.t.float().view(M, C)
ref = torch.relu((yy - co[0]) * co[2] + co[3]).half().float()
got = z.t.float().view(M, C)
bad = (got - ref).abs() > 2e-3 * ref.abs() + 1e-3
print('bad', int(bad.sum()), 'unwritten', int((got == -7.0).sum()), 'nan', int(torch.isnan(got).sum()))
idx = bad.nonzero()
if len(idx):
    r, c = idx[:, 0], idx[:, 1]
    print('rows mod 128:', sorted(set((r % 128).tolist()))[:40])
    print('row blocks:', sorted(set((r // 128).tolist()))[:20], '... count', len(set((r // 128).tolist())))
    print('channels:', sorted(set(c.tolist())))
    print(idx[:10].tolist(), got[r[0], c[0]].item(), ref[r[0], c[0]].item())
if len(idx):
    r0 = int(r[0])
    torch.set_printoptions(linewidth=200, precision=4, sci_mode=False)
    print('row', r0, 'got ', got[r0, 28:64])
    print('row', r0, 'ref ', ref[r0, 28:64])
    pre = ((yy - co[0]) * co[2] + co[3])
    print('row', r0, 'pre ', pre[r0, 28:64])
    print('row', r0 - 1, 'ref ', ref[r0 - 1, 28:64])
    print('row', r0 + 1, 'ref ', ref[r0 + 1, 28:64])
    # does the wrong pair equal some other pair of the same row block?
    blk = (r0 // 128) * 128
    tgt = got[r0, 32:34]
    cand = ((ref[blk:blk + 128].view(128, 32, 2) - tgt).abs().sum(-1) < 1e-6).nonzero()
    print('wrong pair', tgt.tolist(), 'found as correct pair at (row-in-block, pair):', cand[:6].tolist())
    cand2 = ((pre[blk:blk + 128].half().float().view(128, 32, 2) - tgt).abs().sum(-1) < 1e-6).nonzero()
    print('... as pre-activation pair at:', cand2[:6].tolist())

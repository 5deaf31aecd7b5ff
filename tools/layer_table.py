"""Per-layer table of the conv launches of ONE training step.

    python tools/layer_table.py [fp32|half] [deeplabv3plus|hrnet|unet] [--md out.md]

Every conv2d_fwd / conv2d_dgrad / conv2d_wgrad call of a real step of the benchmark configuration (DeepLabV3+ R50, 21 classes,
512x512, B = 16 by default) is timed with a HIP-event pair on the launch stream, one stream, averaged over three steps; calls
are grouped by (kind, shape).  Columns: launches per step, us per launch, algorithmic TF (in-bounds taps, logical channels: the
work the model asked for) and DENSE-EQUIVALENT TF (padded channels, every tap of the gather form -- for strided data gradients and
dilated convs that is work the kernels skip, so the figure can exceed the 157.3 TF peak; it is a comparison aid, not a rate
the hardware ran at), what bounds the layer, its ideal time and the gap.  Round 6: the ideal time is taken
from the ALGORITHMIC work (live taps, live output pixels) at the MEASURED hardware ceilings of MI355X_MICROARCH.md -- 155 TF exact
fp32 MFMA, 6.29 TB/s streaming HBM -- so that no row can show a negative gap and the sum of gaps means "time above what the
hardware could do"; round 5 used dense FLOPs at a 131 TF "sustained" yardstick and showed -0.3 .. -0.6 ms on the dilated / strided
rows.  Sorted by gap: the top of the list is where the step's time is.
(weight-gradient rows include the slab reduction launched by the same call.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

CEILING_TF = {'fp32': 155.0, 'half': 2500.0}       # MI355X_MICROARCH.md: measured exact-fp32 MFMA rate (99 % of 157.3); dense fp16 peak
PEAK_TF = {'fp32': 157.3, 'half': 2500.0}
HBM_TBS = 6.29                                     # MI355X_MICROARCH.md: measured streaming copy (79 % of the 8 TB/s spec)


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith('--')]
    policy = argv[0] if argv else 'fp32'
    which = argv[1] if len(argv) > 1 else 'deeplabv3plus'
    md = sys.argv[sys.argv.index('--md') + 1] if '--md' in sys.argv else None
    from pytorch_segmentation_amd import models as zoo
    from pytorch_segmentation_amd import ops
    from pytorch_segmentation_amd.utils import Trainer, compute_loss
    dev = torch.device('cuda', 0)
    cls, B, S, nc = {'deeplabv3plus': (zoo.DeepLabV3Plus, 16, 512, 21), 'hrnet': (zoo.HRNet, 8, 512, 21),
                     'unet': (zoo.UNet, 8, 256, 2)}[which]
    torch.manual_seed(0)
    model = cls(nc)
    tr = Trainer(model, None, loss_fn=compute_loss, lr=1e-3, device=dev, graph=False)
    tr.env.policy = policy
    model.train()
    x, t = bench.synthetic_batch(B, S, nc, dev, 1234)
    ops.OVERLAP_WGRAD = False
    for _ in range(3):
        tr.train_batch(x, t)
    torch.cuda.synchronize()
    esz = 2.0 if policy == 'half' else 4.0
    rows = {}
    STEPS = 3

    def shape_of(kind, a):
        if kind == 'conv2d_fwd':
            xx, yy, kh, kw, s, p, d = a[0], a[3], a[4], a[5], a[6], a[7], a[8]
        elif kind == 'conv2d_dgrad':
            yy, xx, kh, kw, s, p, d = a[0], a[2], a[3], a[4], a[5], a[6], a[7]
        else:
            xx, yy, kh, kw, s, p, d = a[0], a[1], a[3], a[4], a[5], a[6], a[7]
        return (xx.B, xx.H, xx.W, xx.C, yy.H, yy.W, yy.C, kh, kw, s, p, d)

    with bench.ConvMeter(ops, model) as meter:
        # wrap again to capture the shapes next to the meter's records
        shapes = []
        for kind in ('conv2d_fwd', 'conv2d_dgrad', 'conv2d_wgrad'):
            inner = getattr(ops, kind)

            def wrap(*a, _inner=inner, _kind=kind, **kw):
                shapes.append((_kind, shape_of(_kind, a)))
                return _inner(*a, **kw)
            setattr(ops, kind, wrap)
        for _ in range(STEPS):
            tr.train_batch(x, t)
        torch.cuda.synchronize()
    recs = [r for r in meter.records if r[0].startswith('conv2d')]
    assert len(recs) == len(shapes), (len(recs), len(shapes))
    for (name, e0, e1, dense, useful), (kind, shp) in zip(recs, shapes):
        assert name == kind
        r = rows.setdefault((kind, shp), {'n': 0, 'ms': 0.0, 'dense': dense, 'useful': useful})
        r['n'] += 1
        r['ms'] += e0.elapsed_time(e1)
    out = []
    for (kind, shp), r in rows.items():
        Bn, H, W, Cin, Ho, Wo, Cout, kh, kw, s, p, d = shp
        per_step = r['n'] / STEPS
        us = 1e3 * r['ms'] / r['n']
        tf_alg = r['useful'] / (us * 1e-6) / 1e12
        tf_exe = r['dense'] / (us * 1e-6) / 1e12
        # HBM floor: operands + result once (weights included)
        act_in, act_out, wbytes = Bn * H * W * Cin * esz, Bn * Ho * Wo * Cout * esz, Cout * kh * kw * Cin * (4.0 if kind == 'conv2d_wgrad' else esz)
        floor_us = (act_in + act_out + wbytes) / (HBM_TBS * 1e12) * 1e6
        mfma_us = r['useful'] / (CEILING_TF[policy] * 1e12) * 1e6      # live taps / live pixels / logical channels
        ideal = max(floor_us, mfma_us)
        bound = 'MFMA' if mfma_us >= floor_us else 'HBM'
        if us > 3 * ideal and us < 40:
            bound += ' (latency)'
        gap_ms = per_step * (us - ideal) * 1e-3
        out.append((gap_ms, kind[7:], shp, per_step, us, tf_alg, tf_exe, bound, ideal))
    out.sort(key=lambda r: -r[0])
    tot = sum(r[3] * r[4] for r in out) * 1e-3
    lines = ['| # | kind | B x H x W x Cin -> Ho x Wo x Cout, k / s / p / d | launches / step | us / launch | ms / step | TF algorithmic | '
             'TF dense-equivalent | bound | ideal us | gap ms / step |', '|---|---|---|---|---|---|---|---|---|---|---|']
    for i, (gap, kind, shp, n, us, ta, te, bound, ideal) in enumerate(out):
        Bn, H, W, Cin, Ho, Wo, Cout, kh, kw, s, p, d = shp
        lines.append('| %d | %s | %dx%dx%dx%d -> %dx%dx%d, %dx%d / %d / %d / %d | %.0f | %.1f | %.3f | %.1f | %.1f | %s | %.1f | %.3f |'
                     % (i + 1, kind, Bn, H, W, Cin, Ho, Wo, Cout, kh, kw, s, p, d, n, us, n * us * 1e-3, ta, te, bound, ideal, gap))
    head = ('conv launches of one %s step under `%s` (%s, B=%d, %dx%d, %d classes), one stream, HIP events per call, mean of %d steps: '
            '%.2f ms in %d launches; sum of gaps to the measured ceilings (%.0f TF on the algorithmic FLOPs / %.2f TB/s) %.2f ms\n'
            % (which, policy, torch.cuda.get_device_name(0), B, S, S, nc, STEPS, tot, sum(r[3] for r in out), CEILING_TF[policy], HBM_TBS,
               sum(r[0] for r in out)))
    text = head + '\n' + '\n'.join(lines) + '\n'
    print(text)
    if md:
        open(md, 'w').write(text)


if __name__ == '__main__':
    main()

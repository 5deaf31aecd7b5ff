"""The code path bench.py times, tested at the size it is timed at (BASELINE.json configs[2]: DeepLabV3+ R50, 21 classes,
512x512, batch 16): the fused loss on the stride-4 logits at 16 x 21 x 128 x 128 -> 512 x 512, and one
Trainer.train_batch -- fused low-resolution loss, filter transposes on the second stream, forked weight gradients, with
and without the captured-step replay -- against the bridge path (model(x); compute_loss; backward) from the same state.
Reference: models/deeplabv3plus.py:40-43 (x4 bilinear up-sampling, align_corners=True), utils/utils.py:18-21 (loss)."""
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import fill
from oracle import models as omodels

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def test_ce_upsampled_at_the_benchmark_shape():
    """pseg_ce_upsampled_fwd_bwd at 16 x 21 x 128 x 128 -> 512 x 512 (16384 blocks with halo tiles), align_corners=True,
    with ignored and out-of-range labels: loss and gradient against the library's three-pass path on all 16 images
    (bilinear_fwd_nchw -> ce_fwd_bwd -> bilinear_bwd_nchw) and against float64 CPU autograd through F.interpolate +
    F.cross_entropy on two of them; two runs bit-identical."""
    from pytorch_segmentation_amd import ops
    B, C, h, w, H, W = 16, 21, 128, 128, 512, 512
    x = fill.uniform('ceup_c2/x', (B, C, h, w), 3.0)
    t = fill.labels('ceup_c2/t', (B, H, W), C, block=16)
    t[0, :40, :72] = -100
    t[5, 100:228, 300:] = -100
    t[15, 500:, :] = -100
    t[3, 17:29, 200:260] = C + 2          # out of range: ignored and reported
    lr = ops.Act.from_nchw(x.cuda(), 24)
    tg = t.cuda()
    assert ops.ce_upsampled_ok(lr, C, H, W, True)
    out, dlr = ops.ce_upsampled_fwd_bwd(lr, C, tg, True)
    out2, dlr2 = ops.ce_upsampled_fwd_bwd(lr, C, tg, True)
    assert torch.equal(out, out2) and torch.equal(dlr.t, dlr2.t)
    # the three-pass path on all 16 images
    full = ops.bilinear_fwd_nchw(lr, C, H, W, True)
    o3, dfull = ops.ce_fwd_bwd(full, tg)
    d3 = ops.Act.empty(B, h, w, 24, 'cuda', zero=True)
    ops.bilinear_bwd_nchw(dfull, d3, C, True)
    assert abs(out[0].item() - o3[0].item()) < 2e-6 * abs(o3[0].item())
    assert out[1].item() == o3[1].item() and out[2].item() == o3[2].item() == float((t >= C).sum())
    assert rel(dlr.t, d3.t) < 2e-5
    assert torch.count_nonzero(dlr.view4()[..., C:]).item() == 0
    # float64 autograd on images 0 and 3 (the ones with ignored / out-of-range labels at a corner and mid-image): the
    # batch-of-16 gradient of an image is its own mean-CE gradient times (its valid pixels / all valid pixels)
    n_all = float(out[1].item())
    got = dlr.to_nchw(C).cpu().double()
    for b in (0, 3):
        xr = x[b:b + 1].double().requires_grad_()
        tt = t[b:b + 1].clone()
        tt[tt >= C] = -100
        up = F.interpolate(xr, size=(H, W), mode='bilinear', align_corners=True)
        loss = F.cross_entropy(up, tt, ignore_index=-100, reduction='sum')
        loss.backward()
        assert rel(got[b:b + 1], xr.grad / n_all) < 2e-5, b
    # the loss itself against float64 on the whole batch (forward only: cheap)
    with torch.no_grad():
        tt = t.clone()
        tt[tt >= C] = -100
        ref = F.cross_entropy(F.interpolate(x.double(), size=(H, W), mode='bilinear', align_corners=True), tt, ignore_index=-100)
    assert abs(out[0].item() - ref.item()) < 1e-5 * abs(ref.item())


@pytest.mark.parametrize('policy', ['fp32', 'half'])
def test_trainer_step_at_config2_matches_the_bridge_path(policy):
    """One Trainer.train_batch at configs[2] size with the defaults bench.py runs (fused low-resolution loss, early filter
    transposes on the second stream, weight gradients forked to it) against the bridge path from the same state: loss to
    2e-6, gradient arena to 1e-4 of its peak, two runs bit-identical, and the same through the captured-step replay
    (graph=True: eager, capture, replay -- all three with identical gradients).  Under `half` (the -mp path bench.py
    reports in other_policies) the two paths take the loss through different kernels on the same fp32 logits: loss 1e-5,
    loss-scaled gradient arena 5e-3 of its peak (measured 2.2e-3: the fp16 roundings of dlogits that differ by 1e-5)."""
    from pytorch_segmentation_amd.models import DeepLabV3Plus
    from pytorch_segmentation_amd.utils import Trainer, compute_loss
    ref = omodels.DeepLabV3Plus(21)
    fill.fill_module_(ref, 'cfg2b')
    x = fill.images('cfg2b/x', (16, 3, 512, 512)).cuda()
    tgt = fill.labels('cfg2b/t', (16, 512, 512), 21, block=16).cuda()
    dev = torch.device('cuda', 0)

    def fresh(graph):
        m = DeepLabV3Plus(21)
        m.load_state_dict(ref.state_dict())
        tr = Trainer(m, None, lr=0.0, momentum=0.0, device=dev, graph=graph, mixed_precision=(policy == 'half'))
        m.train()
        return m, tr

    m, tr = fresh(False)
    # A: the explicit path of train_batch (what bench.py times)
    lo = tr._fwd_loss_bwd(x, tgt)
    loss_a, g_a = lo[0].item(), tr.arena.grads.clone()
    lo = tr._fwd_loss_bwd(x, tgt)
    assert lo[0].item() == loss_a and torch.equal(tr.arena.grads, g_a)            # bit-reproducible
    assert torch.isfinite(g_a).all()
    # B: the bridge path from the same parameters (autograd Function around the model, full-resolution logits, stock loss)
    tr.arena.zero_grad()
    tr.env.accumulate = False
    out = m(x)
    loss_b = compute_loss(out, tgt, m)
    loss_b.backward()
    torch.cuda.synchronize()
    g_b = tr.arena.grads.clone()
    l_tol, g_tol = (2e-6, 1e-4) if policy == 'fp32' else (1e-5, 5e-3)
    assert abs(loss_a - loss_b.item()) < l_tol * abs(loss_b.item()), (loss_a, loss_b.item())
    assert rel(g_a, g_b) < g_tol, rel(g_a, g_b)
    del out, loss_b, g_b
    # C: the whole train_batch, eager and replayed from the captured step (lr = 0: the parameters stay put)
    for graph in (False, True):
        m2, tr2 = fresh(graph)
        for i in range(3):                       # graph=True: first sight eager, then capture, then replay
            l = tr2.train_batch(x, tgt)
            torch.cuda.synchronize()
            assert l.item() == loss_a, (graph, i, l.item(), loss_a)
            assert torch.equal(tr2.arena.grads, g_a), (graph, i)
        if graph:
            assert any(sg is not None for sg in tr2._graphs.values())
        if policy == 'half':
            assert tr2.loss_scale_state()['steps_applied'] == 3
        del m2, tr2
        torch.cuda.empty_cache()


def test_bench_line_carries_the_other_configurations():
    """`python bench.py` as the driver runs it at N = 1 (two steps here): ONE JSON line with the contract's keys, the roofline
    object, and `other_configs` -- BASELINE.json configs[4] (HRNet) and configs[1] (UNet) under fp32 and -mp, each timed in a
    process of its own with a default-constructed Trainer that must have chosen the replay by itself."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--steps', '2', '--warmup', '1', '--no-cpu-baseline',
                        '--also', 'half'], capture_output=True, text=True, timeout=600, cwd=repo)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['dtype'] == 'f32' and d['value'] > 0
    assert d['roofline']['bound'] == 'mfma' and 0 < d['roofline']['frac'] < 1
    assert set(d['other_policies']) == {'half'}
    oc = d['other_configs']
    assert set(oc) == {'hrnet', 'unet'}
    for name in oc:
        for pol in ('fp32', 'half'):
            e = oc[name][pol]
            assert 'error' not in e, e
            assert e['ms_per_step'] > 0 and e['replayed'] is True and e['lane_executor']['lanes'] >= 2, (name, pol, e)
    assert oc['hrnet']['half']['lane_executor']['lanes'] >= 4

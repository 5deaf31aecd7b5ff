"""The oracle restatement (oracle/) must reproduce the fixtures that oracle/gen_golden.py produced by
running the REFERENCE's own models/aspp.py, models/deeplabv3plus.py, models/unet.py, utils/utils.py.
CPU only; this is what pins the oracle."""
import os

import numpy as np
import torch

from oracle import fill, loss as oloss, margins, models as omodels

RTOL, ATOL = 1e-5, 1e-6  # same torch ops in a different composition order: fp32 round-off only


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + '.npz')))


def _close(a, b, rtol=RTOL, atol=ATOL):
    a = (a.detach() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a))).double()
    b = (b.detach() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b))).double()
    scale = b.abs().max().item() + 1e-30
    err = (a - b).abs().max().item()
    assert err <= atol + rtol * scale, 'max err %g vs scale %g' % (err, scale)


def _check_grads(module, g):
    for name, p in module.named_parameters():
        if 'grad/' + name in g:
            _close(p.grad, g['grad/' + name], 2e-5, 1e-6)
        elif 'gsum/' + name in g:
            flat = p.grad.double().reshape(-1)
            step = flat.numel() // 64
            _close(flat[:64], g['ghead/' + name], 2e-5, 1e-6)
            _close(flat[::step][:64], g['gstride/' + name], 2e-5, 1e-6)
            s, a = g['gsum/' + name]
            assert abs(flat.abs().sum().item() - a) <= 1e-4 * a
        else:
            assert p.grad is None, name


def _check_buffers(module, g):
    for name, b in module.named_buffers():
        if 'buf/' + name in g:
            _close(b, g['buf/' + name])


def _check_margin(m, forward, g, prepare=None, floor=5e-6):
    """Flip-free fixtures (oracle/margins.py): the stored min_margin -- smallest |ReLU pre-activation| / layer peak -- is
    what the restatement measures on itself in fp32, it clears the floor, and every ReLU call's margin exceeds its own
    fp32-vs-fp64 noise (site_noise32, measured at generation) at least tenfold."""
    mn, ms = margins.measure(m, forward, prepare)
    want = float(g['min_margin'])
    assert want > floor
    assert mn > 0.8 * want, (mn, want)
    sm, sn = g['site_margins'], g['site_noise32']
    assert len(ms) == len(sm)
    fin = np.isfinite(sm)
    assert (sm[fin] > 10 * sn[fin]).all(), float((sm[fin] / np.maximum(sn[fin], 1e-30)).min())


def test_aspp_small(golden_dir):
    g = _load(golden_dir, 'aspp_small')
    m = omodels.ASPP(64, 16, [6, 12, 18])
    fill.fill_module_(m, 'aspp_small')
    m.train()
    x = fill.uniform('aspp_small/x', (2, 64, 24, 24), 1.0).requires_grad_()
    y = m(x)
    gy = fill.uniform('aspp_small/gy', tuple(y.shape), 1.0)
    (y * gy).sum().backward()
    _close(y, g['y'])
    _close(x.grad, g['dx'], 2e-5, 1e-6)
    _check_grads(m, g)
    _check_buffers(m, g)
    m.eval()
    with torch.no_grad():
        _close(m(x), g['y_eval'])


def _sub_out(t):
    return t[:, :, ::7, ::7]


def _sub_df1(t):
    return t[:, ::4, ::3, ::3]


def _sub_df4(t):
    return t[:, ::16]


def _sums_close(t, ref, rtol=1e-4):
    t = t.detach().double()
    assert abs(t.sum().item() - ref[0]) <= rtol * ref[1] and abs(t.abs().sum().item() - ref[1]) <= rtol * ref[1]


def test_deeplab_head(golden_dir):
    """Head at the real widths on the pyramid of a 384x384 image (24x24 ASPP maps, every dilated tap live); the fixture
    stores strided sub-samples + whole-tensor sums of the big arrays (oracle/gen_golden.py::gen_deeplab_head)."""
    g = _load(golden_dir, 'deeplab_head')
    S = int(g['size'])
    assert S >= 384
    m = omodels.DeepLabV3Plus(21, backbone=torch.nn.Identity())
    fill.fill_module_(m, 'deeplab_head')
    assert margins.apply(m, g) >= 7
    m.train()
    chans, strides = (64, 256, 512, 1024, 2048), (2, 4, 8, 16, 16)
    feats = [fill.uniform('deeplab_head/f%d' % i, (4, c, S // s, S // s), 1.0).abs_().requires_grad_()
             for i, (c, s) in enumerate(zip(chans, strides))]
    _check_margin(m, lambda: m.head([f.detach() for f in feats]), g)
    out = m.head(feats)
    tgt = fill.labels('deeplab_head/target', (4, S, S), 21, block=8)
    loss = oloss.compute_loss(out, tgt)
    loss.backward()
    _close(_sub_out(out), g['out_sub'])
    _sums_close(out, g['out_sums'])
    assert abs(loss.item() - float(g['loss'])) <= 1e-5 * abs(float(g['loss']))
    _close(_sub_df1(feats[1].grad), g['df1_sub'], 2e-5, 1e-7)
    _close(_sub_df4(feats[4].grad), g['df4_sub'], 2e-5, 1e-7)
    _sums_close(feats[1].grad, g['df1_sums'])
    _sums_close(feats[4].grad, g['df4_sums'])
    safe = torch.from_numpy(np.unpackbits(g['margin_ok'])[:4 * S * S].reshape(4, S, S).astype(bool))
    assert safe.float().mean() > 0.95
    assert np.array_equal(oloss.predict_mask(out)[safe].numpy(), g['mask'].astype(np.int64)[safe.numpy()])
    _check_grads(m, g)
    _check_buffers(m, g)


def test_unet_head(golden_dir):
    g = _load(golden_dir, 'unet_head')
    m = omodels.UNet(2, backbone=torch.nn.Identity())
    fill.fill_module_(m, 'unet_head')
    m.train()
    chans, strides = (16, 24, 32, 96, 1280), (2, 4, 8, 16, 32)
    feats = [fill.uniform('unet_head/f%d' % i, (2, c, 64 // s, 64 // s), 1.0).abs_().requires_grad_()
             for i, (c, s) in enumerate(zip(chans, strides))]
    out = m.head(feats)
    tgt = fill.labels('unet_head/target', (2, 64, 64), 2, block=8)
    loss = oloss.compute_loss(out, tgt)
    loss.backward()
    _close(out, g['out'])
    assert abs(loss.item() - float(g['loss'])) <= 1e-5 * abs(float(g['loss']))
    for i in (1, 2, 3, 4):
        _close(feats[i].grad, g['df%d' % i], 2e-5, 1e-7)
    assert np.array_equal(oloss.predict_mask(out).numpy(), g['mask'])
    _check_grads(m, g)
    _check_buffers(m, g)


def test_hrnet_small(golden_dir):
    """Whole reference HRNet (models/hrnet.py) vs the oracle restatement: same state-dict keys in the same order,
    same logits / loss / running statistics / parameter gradients on the seeded batch."""
    g = _load(golden_dir, 'hrnet_small')
    m = omodels.HRNet(5)
    assert list(m.state_dict().keys()) == [str(k) for k in g['keys']]
    fill.fill_module_(m, 'hrnet_small')
    assert margins.apply(m, g) >= 90
    m.train()
    x = fill.images('hrnet_small/x', (4, 3, 64, 64))
    tgt = fill.labels('hrnet_small/target', (4, 64, 64), 5, block=8)
    _check_margin(m, lambda: m(x), g)
    out = m(x)
    loss = oloss.compute_loss(out, tgt)
    loss.backward()
    _close(out, g['out'])
    assert abs(loss.item() - float(g['loss'])) <= 1e-5 * abs(float(g['loss']))
    assert np.array_equal(oloss.predict_mask(out).numpy(), g['mask'])
    _check_grads(m, g)
    _check_buffers(m, g)
    m.eval()
    with torch.no_grad():
        _close(m(x), g['out_eval'])


def test_loss_argmax_metrics(golden_dir):
    g = _load(golden_dir, 'loss_metrics')
    logits = fill.uniform('loss/logits', (2, 21, 32, 32), 4.0).requires_grad_()
    tgt = fill.labels('loss/target', (2, 32, 32), 21, block=4)
    loss = oloss.compute_loss(logits, tgt)
    loss.backward()
    assert abs(loss.item() - float(g['ce_loss'])) <= 1e-6 * abs(float(g['ce_loss']))
    _close(logits.grad, g['ce_dlogits'])
    assert np.array_equal(oloss.predict_mask(logits).numpy(), g['ce_mask'])
    lg2 = fill.uniform('loss/logits2', (2, 5, 16, 16), 3.0).requires_grad_()
    tgt2 = fill.labels('loss/target2', (2, 40, 24), 5, block=4)
    loss2 = oloss.compute_loss(lg2, tgt2)
    loss2.backward()
    assert abs(loss2.item() - float(g['ce2_loss'])) <= 1e-6 * abs(float(g['ce2_loss']))
    _close(lg2.grad, g['ce2_dlogits'])
    assert np.array_equal(oloss.predict_mask(torch.from_numpy(g['tie_logits'])).numpy(), g['tie_mask'])
    T, P, R, miou, F1 = oloss.compute_metrics(torch.from_numpy(g['m_tp']), torch.from_numpy(g['m_fn']),
                                              torch.from_numpy(g['m_fp']))
    for got, key in ((T, 'm_T'), (P, 'm_P'), (R, 'm_R'), (miou, 'm_miou'), (F1, 'm_F1')):
        assert np.array_equal(got.numpy(), g[key]), key


def test_class_counts_matches_bruteforce():
    pred = fill.labels('cc/pred', (2, 16, 16), 5)
    tgt = fill.labels('cc/tgt', (2, 16, 16), 5)
    tp, fn, fp = oloss.class_counts(pred, tgt, 5)
    for c in range(5):
        assert tp[c] == ((pred == c) & (tgt == c)).sum()
        assert fn[c] == ((pred != c) & (tgt == c)).sum()
        assert fp[c] == ((pred == c) & (tgt != c)).sum()


def test_fill_is_stable():
    # closed-form fill must never drift: fixtures depend on it
    u = fill.uniform('stability', (4,), 1.0).tolist()
    assert u == fill.uniform('stability', (4,), 1.0).tolist()
    assert all(-1.0 <= v < 1.0 for v in u)
    lab = fill.labels('stability', (1, 4, 4), 7)
    assert lab.min() >= 0 and lab.max() < 7


MARGIN_CASES = [('full_dl', lambda: omodels.DeepLabV3Plus(21), 128, 4, False),
                ('full_dl16', lambda: omodels.DeepLabV3Plus(21), 128, 16, False),
                ('full_unet', lambda: omodels.UNet(2), 128, 4, False),
                ('full_hrnet', lambda: omodels.HRNet(5), 64, 4, False),
                ('cfg1_unet', lambda: omodels.UNet(2), 256, 8, False),
                ('frozen_deeplabv3plus', lambda: omodels.DeepLabV3Plus(21), 128, 4, True),
                ('frozen_unet', lambda: omodels.UNet(2), 128, 4, True),
                ('frozen_hrnet', lambda: omodels.HRNet(5), 64, 4, True)]


def case_arrays(golden_dir, key):
    """entries of tests/golden/margins.npz that belong to one whole-model case, with the case prefix removed"""
    z = np.load(os.path.join(golden_dir, 'margins.npz'))
    return {k[len(key) + 1:]: z[k] for k in z.files if k.startswith(key + '/')}


import pytest  # noqa: E402


@pytest.mark.parametrize('case', MARGIN_CASES, ids=[c[0] for c in MARGIN_CASES])
def test_whole_model_cases_are_flip_free(golden_dir, case):
    """The whole-model parity cases of tests/test_models_gpu.py run on betas nudged away from every ReLU kink
    (tests/golden/margins.npz): re-measure the margin on the oracle."""
    key, make, S, B, frozen = case
    g = case_arrays(golden_dir, key)
    m = make()
    fill.fill_module_(m, key)
    assert margins.apply(m, g) > 20
    x = fill.images(key + '/x', (B, 3, S, S))
    if frozen:
        _check_margin(m, lambda: m(x), g, prepare=lambda: margins.freeze_stats(m, x))
    else:
        m.train()
        _check_margin(m, lambda: m(x), g)



def test_gradnoise_allowance_is_pinned(golden_dir):
    """tests/test_models_gpu.py::grad_bound relaxes the 1e-3 gradient contract only for tensors whose stored fp32-oracle
    distance from fp64 ('gradnoise/<parameter>') exceeds 5e-4.  That set is pinned here, so a regenerated fixture cannot
    widen it silently: the image-pool branch of DeepLabV3+ at batch 4 (training and frozen-statistics case, two tensors
    each, all below 1e-3) and NOTHING at the reference's batch of 16."""
    z = np.load(os.path.join(golden_dir, 'margins.npz'))
    big = sorted((k, float(z[k])) for k in z.files if '/gradnoise/' in k and float(z[k]) > 5e-4)
    assert [k for k, _ in big] == [
        'frozen_deeplabv3plus/gradnoise/aspp.blocks.0.gap.1.0.weight',
        'frozen_deeplabv3plus/gradnoise/aspp.blocks.0.gap.1.1.weight',
        'full_dl/gradnoise/aspp.blocks.0.gap.1.0.weight',
        'full_dl/gradnoise/aspp.project.0.weight'], big
    assert all(v < 1e-3 for _, v in big), big
    assert not [k for k in z.files if k.startswith('full_dl16/gradnoise/')]

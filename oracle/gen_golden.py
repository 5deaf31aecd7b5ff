#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE'S OWN FILES.

Run in the build container only (the reference tree is absent on the GPU box):

    python oracle/gen_golden.py            # writes tests/golden/*.npz

How: the reference's ``models/aspp.py``, ``models/deeplabv3plus.py``, ``models/unet.py``, ``models/hrnet.py``
and ``utils/utils.py`` are imported unmodified from /root/reference.  Their absent third-party
imports are satisfied by in-memory stand-ins:

* ``pytorch_modules.nn.ConvNormAct`` / ``pytorch_modules.utils.initialize_weights`` -> the call-site
  restatements in oracle/blocks.py (that package is not vendored);
* ``pytorch_modules.backbones.{resnet50,mobilenet_v2}`` -> a stub that returns preset feature maps, so
  the fixture pins the reference's HEAD composition at the real channel widths;
* ``cv2`` / ``imgaug`` (only touched by the reference's data/visualisation code, never by
  ``compute_loss`` / ``compute_metrics``) -> ``MagicMock`` modules.

Inputs and parameters are closed-form (oracle/fill.py), so fixtures hold only the expected outputs
(plus small inputs where cheap) and tests regenerate the rest from keys.

Test infrastructure only -- see oracle/__init__.py.
"""
import os
import sys
import types
from unittest import mock

sys.dont_write_bytecode = True  # never drop __pycache__ into the read-only reference tree

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('PSEG_REFERENCE', '/root/reference')
sys.path.insert(0, REPO)

from oracle import blocks as oblocks  # noqa: E402
from oracle import fill  # noqa: E402
from oracle import margins  # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')


class FeatureStub(nn.Module):
    """Backbone stand-in: ignores the image, returns the preset feature list."""

    def __init__(self):
        super().__init__()
        self.features = None

    def forward(self, x):
        return self.features


def install_standins():
    pm = types.ModuleType('pytorch_modules')
    pm_nn = types.ModuleType('pytorch_modules.nn')
    pm_nn.ConvNormAct = oblocks.ConvNormAct
    pm_nn.SeparableConvNormAct = None  # imported but unused (models/aspp.py:5)
    pm_nn.FocalBCELoss = lambda *a, **k: None  # instantiated but unused (utils/utils.py:14)
    pm_bb = types.ModuleType('pytorch_modules.backbones')
    pm_bb.resnet50 = lambda *a, **k: FeatureStub()
    pm_bb.resnet34 = lambda *a, **k: FeatureStub()
    pm_bb.mobilenet_v2 = lambda *a, **k: FeatureStub()
    pm_mb = types.ModuleType('pytorch_modules.backbones.mobilenet')
    pm_mb.InvertedResidual = None
    pm_ut = types.ModuleType('pytorch_modules.utils')
    pm_ut.initialize_weights = oblocks.initialize_weights
    pm_ut.IMG_EXT = ['.jpg', '.png']
    pm_ut.device = torch.device('cpu')
    pm_ut.Fetcher = None
    pm_ut.Trainer = None
    for name, m in [('pytorch_modules', pm), ('pytorch_modules.nn', pm_nn),
                    ('pytorch_modules.backbones', pm_bb),
                    ('pytorch_modules.backbones.mobilenet', pm_mb),
                    ('pytorch_modules.utils', pm_ut)]:
        sys.modules[name] = m
    for name in ['cv2', 'imgaug', 'imgaug.augmenters', 'imgaug.augmentables',
                 'imgaug.augmentables.segmaps', 'imgaug.augmentables.polys']:
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock(name=name)
    sys.path.insert(0, REF)


def np_(t):
    return t.detach().cpu().numpy()


def grads_digest(module, full_below=80000):
    """Per-parameter gradient digests: full tensor when small, else (sum, abs-sum, first 64, strided 64)."""
    out = {}
    for name, p in module.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().double().reshape(-1)
        if g.numel() <= full_below:
            out['grad/' + name] = np_(p.grad)
        else:
            step = g.numel() // 64
            out['gsum/' + name] = np.array([g.sum().item(), g.abs().sum().item()])
            out['ghead/' + name] = g[:64].float().numpy()
            out['gstride/' + name] = g[::step][:64].float().numpy()
    return out


def bn_buffers(module):
    return {'buf/' + n: np_(b) for n, b in module.named_buffers() if 'num_batches' not in n}


def flip_free(m, make_forward, prepare=None, verbose=False, reach=0.02):
    """Nudge the BatchNorm betas of the float model `m` until no ReLU pre-activation of the case sits near its kink
    (oracle/margins.py; done on a float64 copy, `make_forward(model, dtype)` -> closure that runs the case).
    -> fixture entries: nudge/<parameter>, min_margin, site_margins, site_noise32 (fp32-vs-fp64 distance per ReLU call,
    the quantity the margin has to exceed)."""
    import copy
    m64 = copy.deepcopy(m).double()
    prep64 = (lambda: prepare(m64, torch.float64)) if prepare else None
    nd, mn, ms = margins.nudge(m64, make_forward(m64, torch.float64), prepare=prep64, verbose=verbose, reach=reach)
    margins.apply(m, {'nudge/' + k: v.numpy() for k, v in nd.items()})
    prep32 = (lambda: prepare(m, torch.float32)) if prepare else None
    noise = margins.noise32(m64, make_forward(m64, torch.float64), m, make_forward(m, torch.float32), prep64, prep32)
    ms = np.array(ms)
    fin = np.isfinite(ms)
    print('    flip-free: %d ReLU calls (%d one-sided), min margin %.2e, fp32 noise max %.2e, min margin/noise %.0f'
          % (len(ms), (~fin).sum(), mn, max(noise), (ms[fin] / np.maximum(np.array(noise)[fin], 1e-30)).min()))
    d = {'nudge/' + k: v.numpy() for k, v in nd.items()}
    d['min_margin'] = np.array(mn)
    d['site_margins'] = ms
    d['site_noise32'] = np.array(noise)
    return d


def gen_aspp_small():
    """reference models/aspp.py ASPP(64, 16, [6,12,18]) on [2,64,24,24]: d=18 exceeds H/2, so most
    taps of the third branch fall in the zero padding -- the regime of the real 32x32 map."""
    from models.aspp import ASPP  # the reference's file
    torch.manual_seed(0)
    m = ASPP(64, 16, [6, 12, 18])
    fill.fill_module_(m, 'aspp_small')
    m.train()
    x = fill.uniform('aspp_small/x', (2, 64, 24, 24), 1.0).requires_grad_()
    y = m(x)
    gy = fill.uniform('aspp_small/gy', tuple(y.shape), 1.0)
    (y * gy).sum().backward()
    d = {'y': np_(y), 'dx': np_(x.grad)}
    d.update(grads_digest(m))
    d.update(bn_buffers(m))
    m.eval()
    with torch.no_grad():
        d['y_eval'] = np_(m(x))
    np.savez_compressed(os.path.join(OUT, 'aspp_small.npz'), **d)
    return d


def deeplab_features(key, B, S):
    """R50-OS16 feature pyramid for an S x S image (post-ReLU maps => non-negative)."""
    chans = (64, 256, 512, 1024, 2048)
    strides = (2, 4, 8, 16, 16)
    return [fill.uniform('%s/f%d' % (key, i), (B, c, S // s, S // s), 1.0).abs_()
            for i, (c, s) in enumerate(zip(chans, strides))]


HEAD_S = 384   # image side of the DeepLabV3+ head fixture: stride-16 maps are 24x24, so every tap of the rate-6/12/18
               # convs is live somewhere on the map (on a 48x48 image they were 3x3 and only the centre tap ever was)


def sub_out(t):
    """Logits [B,nc,S,S] -> every 7th row / column (7 is coprime to the x4 resize: all interpolation phases occur)."""
    return t[:, :, ::7, ::7]


def sub_df1(t):
    return t[:, ::4, ::3, ::3]


def sub_df4(t):
    return t[:, ::16]


def sums(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item()])


def gen_deeplab_head():
    """reference models/deeplabv3plus.py DeepLabV3Plus(21) at the REAL channel widths (ASPP 2048->256,
    K = 18432 contractions) on the feature pyramid of a 384x384 image (24x24 ASPP maps: all 27 dilated taps live),
    B=4, + compute_loss + backward.  The big tensors are stored as strided sub-samples plus (sum, abs-sum) of the whole.
    (B=4: with B=2 the image-pool branch's BatchNorm sees two samples per channel, xhat = +-1 exactly and its
    input gradient is pure cancellation noise -- not a meaningful parity target.)"""
    from models.deeplabv3plus import DeepLabV3Plus  # the reference's file
    from utils.utils import compute_loss  # the reference's file (cv2/imgaug mocked)
    torch.manual_seed(0)
    m = DeepLabV3Plus(21)
    assert isinstance(m.backbone, FeatureStub)
    fill.fill_module_(m, 'deeplab_head')
    m.train()
    S = HEAD_S
    feats = deeplab_features('deeplab_head', 4, S)

    def make_forward(model, dtype):
        def run():
            model.backbone.features = [f.to(dtype) for f in feats]
            model(torch.zeros(4, 3, S, S, dtype=dtype))
        return run
    nd = flip_free(m, make_forward)
    for f in feats:
        f.requires_grad_()
    m.backbone.features = feats
    out = m(torch.zeros(4, 3, S, S))
    tgt = fill.labels('deeplab_head/target', (4, S, S), 21, block=8)
    loss = compute_loss(out, tgt, m)
    loss.backward()
    d = {'size': np.array(S), 'out_sub': np_(sub_out(out)), 'out_sums': sums(out), 'out_absmax': np.array(out.abs().max().item()),
         'loss': np.array(loss.item()),
         'df1_sub': np_(sub_df1(feats[1].grad)), 'df1_sums': sums(feats[1].grad),
         'df1_absmax': np.array(feats[1].grad.abs().max().item()),
         'df4_sub': np_(sub_df4(feats[4].grad)), 'df4_sums': sums(feats[4].grad),
         'df4_absmax': np.array(feats[4].grad.abs().max().item()),
         'mask': np_(out.max(1)[1]).astype(np.uint8)}
    top2 = out.detach().topk(2, dim=1).values
    d['margin_ok'] = np.packbits(np_((top2[:, 0] - top2[:, 1]) > 1e-3 * out.abs().max()))
    d.update(nd)
    d.update(grads_digest(m))
    d.update(bn_buffers(m))
    np.savez_compressed(os.path.join(OUT, 'deeplab_head.npz'), **d)
    return d


def unet_features(key, B, S):
    chans = (16, 24, 32, 96, 1280)
    strides = (2, 4, 8, 16, 32)
    return [fill.uniform('%s/f%d' % (key, i), (B, c, S // s, S // s), 1.0).abs_()
            for i, (c, s) in enumerate(zip(chans, strides))]


def gen_unet_head():
    """reference models/unet.py UNet(2) decoder at the real widths on a 64x64 image's pyramid, B=2."""
    from models.unet import UNet  # the reference's file
    from utils.utils import compute_loss
    torch.manual_seed(0)
    m = UNet(2)
    fill.fill_module_(m, 'unet_head')
    m.train()
    feats = unet_features('unet_head', 2, 64)
    for f in feats:
        f.requires_grad_()
    m.backbone.features = feats
    out = m(torch.zeros(2, 3, 64, 64))
    tgt = fill.labels('unet_head/target', (2, 64, 64), 2, block=8)
    loss = compute_loss(out, tgt, m)
    loss.backward()
    d = {'out': np_(out), 'loss': np.array(loss.item()), 'target': np_(tgt),
         'mask': np_(out.max(1)[1])}
    for i in (1, 2, 3, 4):
        d['df%d' % i] = np_(feats[i].grad)
    d.update(grads_digest(m))
    d.update(bn_buffers(m))
    np.savez_compressed(os.path.join(OUT, 'unet_head.npz'), **d)
    return d


def gen_hrnet_small():
    """reference models/hrnet.py HRNet(5) -- the WHOLE network (it has no external backbone) -- on a [4,3,64,64]
    image batch: logits in train and eval mode, loss, running statistics and parameter-gradient digests.
    The branch maps are 16/8/4/2 px, so the deepest BatchNorms see 16 values per channel: late gradients are
    conditioned like the other whole-model cases (tests judge them against the fp64 oracle, not this fp32 run)."""
    from models.hrnet import HRNet  # the reference's file
    from utils.utils import compute_loss
    torch.manual_seed(0)
    m = HRNet(5)
    fill.fill_module_(m, 'hrnet_small')
    m.train()
    x = fill.images('hrnet_small/x', (4, 3, 64, 64))
    tgt = fill.labels('hrnet_small/target', (4, 64, 64), 5, block=8)
    nd = flip_free(m, lambda model, dtype: (lambda: model(x.to(dtype))))
    out = m(x)
    loss = compute_loss(out, tgt, m)
    loss.backward()
    d = {'out': np_(out), 'loss': np.array(loss.item()), 'mask': np_(out.max(1)[1]).astype(np.uint8)}
    d.update(nd)
    d.update(grads_digest(m, full_below=1024))
    d.update(bn_buffers(m))
    d['keys'] = np.array(list(m.state_dict().keys()))
    m.eval()
    with torch.no_grad():
        d['out_eval'] = np_(m(x))
    np.savez_compressed(os.path.join(OUT, 'hrnet_small.npz'), **d)
    return d


def gen_loss_metrics():
    """reference utils/utils.py compute_loss (equal-size and resized) and compute_metrics;
    reference test.py:31 argmax."""
    from utils.utils import compute_loss, compute_metrics
    d = {}
    logits = fill.uniform('loss/logits', (2, 21, 32, 32), 4.0).requires_grad_()
    tgt = fill.labels('loss/target', (2, 32, 32), 21, block=4)
    loss = compute_loss(logits, tgt, None)
    loss.backward()
    d['ce_loss'] = np.array(loss.item())
    d['ce_dlogits'] = np_(logits.grad)
    d['ce_mask'] = np_(logits.max(1)[1])
    # resized: logits at 16x16 against 40x24 targets (the --multi-scale path)
    lg2 = fill.uniform('loss/logits2', (2, 5, 16, 16), 3.0).requires_grad_()
    tgt2 = fill.labels('loss/target2', (2, 40, 24), 5, block=4)
    loss2 = compute_loss(lg2, tgt2, None)
    loss2.backward()
    d['ce2_loss'] = np.array(loss2.item())
    d['ce2_dlogits'] = np_(lg2.grad)
    # ties: argmax must return the FIRST maximal index
    tie = torch.zeros(1, 4, 2, 2)
    tie[0, 1, 0, 0] = 1.0
    tie[0, 3, 0, 0] = 1.0
    tie[0, 2, 1, 1] = -0.0
    d['tie_logits'] = np_(tie)
    d['tie_mask'] = np_(tie.max(1)[1])
    # metrics incl. the zero guards (class 3 never appears anywhere)
    tp = torch.tensor([10., 0., 5., 0.])
    fn = torch.tensor([2., 3., 0., 0.])
    fp = torch.tensor([1., 0., 7., 0.])
    T, P, R, miou, F1 = compute_metrics(tp.clone(), fn.clone(), fp.clone())
    d.update(m_tp=np_(tp), m_fn=np_(fn), m_fp=np_(fp), m_T=np_(T), m_P=np_(P), m_R=np_(R),
             m_miou=np_(miou), m_F1=np_(F1))
    np.savez_compressed(os.path.join(OUT, 'loss_metrics.npz'), **d)
    return d


# whole-model parity cases that the tests evaluate with the oracle at run time (tests/test_models_gpu.py): only their
# flip-free betas are fixture data.  (case key, oracle model factory, classes, image side, batch, frozen statistics)
MARGIN_CASES = [
    ('full_dl', lambda om: om.DeepLabV3Plus(21), 21, 128, 4, False),
    # the reference's own batch (BASELINE.json configs[2]: 16 per GPU): the image-pool branch of the ASPP head
    # (reference models/aspp.py:11-12) then normalises ONE value per image over 16 samples instead of 4
    ('full_dl16', lambda om: om.DeepLabV3Plus(21), 21, 128, 16, False),
    ('full_unet', lambda om: om.UNet(2), 2, 128, 4, False),
    ('full_hrnet', lambda om: om.HRNet(5), 5, 64, 4, False),
    ('cfg1_unet', lambda om: om.UNet(2), 2, 256, 8, False),
    ('frozen_deeplabv3plus', lambda om: om.DeepLabV3Plus(21), 21, 128, 4, True),
    ('frozen_unet', lambda om: om.UNet(2), 2, 128, 4, True),
    ('frozen_hrnet', lambda om: om.HRNet(5), 5, 64, 4, True),
]


def gen_margins():
    """Flip-free betas of the whole-model cases (oracle restatement incl. the external backbones' stand-ins, the same
    modules the tests compare against): tests/golden/margins.npz, entries '<case>/nudge/<parameter>' etc."""
    from oracle import models as omodels
    out = {}
    # PSEG_MARGIN_CASES=a,b: (re)generate only those cases and keep the other entries of the existing file as they are
    only = [c for c in os.environ.get('PSEG_MARGIN_CASES', '').split(',') if c]
    path = os.path.join(OUT, 'margins.npz')
    if only and os.path.exists(path):
        z = np.load(path)
        out = {k: z[k] for k in z.files if k.split('/')[0] not in only}
    for key, make, nc, S, B, frozen in MARGIN_CASES:
        if only and key not in only:
            continue
        print('  case %s' % key, flush=True)
        m = make(omodels)
        fill.fill_module_(m, key)
        x = fill.images(key + '/x', (B, 3, S, S))
        if frozen:
            m.eval()
            d = flip_free(m, lambda model, dtype: (lambda: model(x.to(dtype))),
                          prepare=lambda model, dtype: margins.freeze_stats(model, x.to(dtype)))
        else:
            m.train()
            # (16 images: four times the pre-activations per channel of the batch-4 cases, so the widest gap near 0 is four
            # times narrower -- the betas may move up to 6 % of the layer's peak instead of 2 % to reach 10 x the fp32 noise)
            d = flip_free(m, lambda model, dtype: (lambda: model(x.to(dtype))), reach=0.06 if B >= 16 else 0.02)
        d.update(grad_noise(m, x, fill.labels(key + '/t', (B, S, S), nc, block=8), frozen))
        for k, v in d.items():
            out[key + '/' + k] = v
    np.savez_compressed(os.path.join(OUT, 'margins.npz'), **out)
    return out


def grad_noise(m, x, tgt, frozen, floor=2e-4):
    """Per parameter: the distance (max-norm over the tensor's peak) of the fp32 CPU gradient from the fp64 one on this
    case -- what ANY fp32 evaluation is uncertain by.  Stored for the tensors where it exceeds `floor` (the image-pool
    branch of DeepLabV3+ normalises over FOUR samples: 8e-4), as 'gradnoise/<parameter>': the tests bound the HIP gradient
    by max(1e-3, 3 x this), a number fixed in the fixture rather than re-measured on whatever CPU runs the test."""
    import copy
    from oracle import loss as oloss
    m64 = copy.deepcopy(m).double()
    if frozen:
        margins.freeze_stats(m, x)
        margins.freeze_stats(m64, x.double())
    else:
        m.train(), m64.train()
    for mod, inp in ((m, x), (m64, x.double())):
        mod.zero_grad()
        oloss.compute_loss(mod(inp), tgt).backward()
    out = {}
    g64 = dict((n, p.grad) for n, p in m64.named_parameters())
    gmax = max(v.abs().max().item() for v in g64.values())
    for n, p in m.named_parameters():
        ref = g64[n]
        if ref.abs().max().item() < 1e-9 * gmax:
            continue
        e = ((p.grad.double() - ref).abs().max() / ref.abs().max()).item()
        if e > floor:
            out['gradnoise/' + n] = np.array(e)
    print('    fp32 gradient noise above %.0e on %d parameter tensors%s' % (
        floor, len(out), (': worst %.2e' % max(float(v) for v in out.values())) if out else ''))
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    install_standins()
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    only = sys.argv[1:]
    for fn in (gen_aspp_small, gen_loss_metrics, gen_unet_head, gen_deeplab_head, gen_hrnet_small, gen_margins):
        if only and fn.__name__ not in only:
            continue
        d = fn()
        print('%-18s %d arrays, %.1f KB' % (fn.__name__, len(d),
                                            sum(v.nbytes for v in d.values()) / 1024))


if __name__ == '__main__':
    main()

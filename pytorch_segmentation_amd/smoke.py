"""One tiny forward + loss + backward of DeepLabV3+ on cuda:0, checked against the CPU oracle.
(`oracle/` is imported here only as the checker -- see oracle/__init__.py.)"""
import torch


CASE = 'full_dl16'      # tests/golden/margins.npz: DeepLabV3+ (21 classes), 128x128, 16 images -- the reference's batch


def run(size=128, batch=16, num_classes=21, tol=1e-3, verbose=True):
    """The case is the flip-free whole-model fixture the parity tests use (`full_dl16`: BatchNorm betas nudged so that no ReLU
    pre-activation lies within rounding of 0 -- oracle/margins.py, re-measured by tests/test_oracle_golden.py): on it two
    correct fp32 implementations differ by rounding noise only, so EVERY parameter gradient is held to the plain 1e-3
    (max-norm over the tensor's peak) against the oracle evaluated in fp64 -- no median, no yardstick relative to the fp32
    oracle's own error (round 4 accepted med_hip < 5 x med_ref on un-nudged weights, where single flipped ReLUs dominate)."""
    import copy
    import os

    import numpy as np
    from oracle import fill
    from oracle import loss as oloss
    from oracle import margins
    from oracle import models as omodels
    from . import _lib
    from .models import DeepLabV3Plus
    from .utils import compute_loss, predict_mask

    _lib.load()
    torch.manual_seed(0)
    if (size, batch, num_classes) != (128, 128 // 8, 21):
        raise ValueError('the smoke case is the fixture case %s: 128x128, 16 images, 21 classes' % CASE)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z = np.load(os.path.join(root, 'tests', 'golden', 'margins.npz'))
    fx = {k[len(CASE) + 1:]: z[k] for k in z.files if k.startswith(CASE + '/')}
    ref = omodels.DeepLabV3Plus(num_classes)
    fill.fill_module_(ref, CASE)
    assert margins.apply(ref, fx) > 20 and float(fx['min_margin']) > 5e-6, 'flip-free fixture missing or degenerate'
    assert not [k for k in fx if k.startswith('gradnoise/')], 'the smoke case must carry no per-tensor allowance'
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    ref.train()
    x = fill.images(CASE + '/x', (batch, 3, size, size))
    tgt = fill.labels(CASE + '/t', (batch, size, size), num_classes, block=8)
    ref64 = copy.deepcopy(ref).double()
    out_ref = ref(x)
    loss_ref = oloss.compute_loss(out_ref, tgt)
    loss_ref.backward()
    oloss.compute_loss(ref64(x.double()), tgt).backward()
    g64 = dict((n, p.grad) for n, p in ref64.named_parameters())

    dev = torch.device('cuda', 0)
    model = DeepLabV3Plus(num_classes)
    model.load_state_dict(state)
    model.to(dev).train()
    out = model(x.to(dev))
    loss = compute_loss(out, tgt.to(dev), model)
    loss.backward()
    torch.cuda.synchronize()

    def rel(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()

    e_out = rel(out, out_ref)
    e_loss = abs(loss.item() - loss_ref.item()) / abs(loss_ref.item())
    gmax = max(v.abs().max().item() for v in g64.values())
    worst, worst_ref, bad, checked = (0.0, None), 0.0, [], 0
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        if g64[n].abs().max().item() < 1e-9 * gmax:
            continue        # exactly zero in exact arithmetic (a BatchNorm bias in front of conv + BatchNorm)
        e = rel(p.grad, g64[n])
        checked += 1
        worst_ref = max(worst_ref, rel(q.grad, g64[n]))
        if e > worst[0]:
            worst = (e, n)
        if not e < tol:
            bad.append((n, e))
    top2 = out_ref.detach().topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-3 * out_ref.abs().max()
    mask_ok = torch.equal(predict_mask(out).cpu()[safe], oloss.predict_mask(out_ref)[safe])
    if verbose:
        print('smoke: logits rel err %.2e, loss rel err %.2e, worst of %d parameter gradients vs fp64 %.2e at %s '
              '(fp32 CPU oracle worst: %.2e), masks exact on %.1f%% safe pixels: %s'
              % (e_out, e_loss, checked, worst[0], worst[1], worst_ref, 100 * safe.float().mean().item(), mask_ok))
    assert e_out < tol and e_loss < tol and mask_ok, 'HIP path deviates from the CPU oracle'
    assert checked > 150 and not bad, 'parameter gradients beyond %.0e of the fp64 oracle: %s' % (tol, bad[:8])
    fresh_state = state
    # the -mp path (fp16 storage, fp32 master weights, loss scaling) on the same model and batch: its loss against the oracle's
    # at the half policy's own tolerance, one optimiser step applied on the device
    from .utils import Trainer
    model.load_state_dict(fresh_state)
    tr = Trainer(model, None, lr=1e-3, mixed_precision=True, device=dev)
    model.train()
    l_half = tr.train_batch(x.to(dev), tgt.to(dev)).item()
    st = tr.loss_scale_state()
    e_half = abs(l_half - loss_ref.item()) / abs(loss_ref.item())
    if verbose:
        print('smoke (-mp, fp16 storage): loss rel err vs the fp32 oracle %.2e, loss-scale state %s' % (e_half, st))
    assert e_half < 5e-3 and st['steps_applied'] + st['steps_skipped'] == 1, 'half-precision path deviates from the CPU oracle'
    return dict(logits=e_out, loss=e_loss, worst_grad=worst[0], worst_grad_ref=worst_ref, mask_exact=mask_ok)

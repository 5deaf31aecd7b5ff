"""Flip-free fixtures: keep every ReLU pre-activation of a parity case away from its kink.

Two fp32 implementations of the same network disagree by rounding (1e-6 of a layer's peak early on, 1e-4 after fifty
layers).  A ReLU pre-activation closer to 0 than that takes a different mask in each, and ONE flipped element moves the
gradients behind it by percents -- which is why gradient comparisons of whole heads / models used to need statistical
allowances.  This module removes the cause instead: for every ReLU / ReLU6 call of a forward pass (in execution order)
it moves the BatchNorm beta that feeds the call, channel by channel, to the middle of the widest gap between
pre-activations within reach, so that NO element lies within ``margin * peak`` of a kink.  The nudged betas are data
(stored in tests/golden/*.npz as ``nudge/<parameter name>``); the margin actually reached is stored next to them
(``min_margin``, ``site_margins``) and re-measured by tests/test_oracle_golden.py.

Nothing here knows the architectures: sites are found with hooks (every ``nn.ReLU`` / ``nn.ReLU6`` call), the knob of a
site is the last ``nn.BatchNorm2d`` that ran before it with the same channel count (for ``relu(bn(x) + skip)`` and the
HRNet fusion sums any member of the sum shifts the sum; bilinear resizing passes a per-channel constant through), and
every nudge is VERIFIED by the next forward pass rather than assumed.

Test infrastructure only -- see oracle/__init__.py.
"""
import numpy as np
import torch
import torch.nn as nn


class _Trace:
    """Hooks that log, per forward pass, every ReLU call: (pre-activation, its sum as a graph node, upper kink or None)
    and the BatchNorm modules in execution order."""

    def __init__(self, model):
        self.handles = []
        self.bns = []
        self.sites = []
        self.stop_after = None
        self.paused = False
        for m in model.modules():
            if isinstance(m, nn.BatchNorm2d):
                self.handles.append(m.register_forward_hook(self._bn))
            elif isinstance(m, (nn.ReLU, nn.ReLU6)):
                self.handles.append(m.register_forward_pre_hook(self._relu))

    def _bn(self, mod, inp, out):
        if not self.paused:
            self.bns.append(mod)

    def _relu(self, mod, inp):
        if self.paused:
            return
        z = inp[0]
        # (the sum is taken NOW: an in-place ReLU overwrites z right after this hook; the sum's graph node survives)
        self.sites.append((z.detach().clone(), z.sum() if z.requires_grad else None,
                           6.0 if isinstance(mod, nn.ReLU6) else None))
        if self.stop_after is not None and len(self.sites) > self.stop_after:
            raise _Stop()

    def run(self, forward, stop_after=None, prepare=None, grad=False):
        self.sites, self.bns, self.stop_after = [], [], stop_after
        try:
            if prepare is not None:      # e.g. "freeze the statistics of the current parameters": not traced
                self.paused = True
                with torch.no_grad():
                    prepare()
                self.paused = False
            with torch.enable_grad() if grad else torch.no_grad():
                forward()
        except _Stop:
            pass
        finally:
            self.paused = False
        return self.sites

    def close(self):
        for h in self.handles:
            h.remove()


class _Stop(Exception):
    pass


def _find_knobs(tr, forward, prepare):
    """For every ReLU call: the BatchNorm whose beta TRANSLATES the call's pre-activation channel by channel
    (d z[b,c,h,w] / d beta[c'] = [c == c'] everywhere), found by differentiation instead of by knowing the architecture:
    the gradient of sum(z) with respect to a candidate beta then equals the number of elements per channel, exactly
    (residual sums, the HRNet fusion sums and bilinear resizing keep it; a ReLU or a train-mode BatchNorm in between
    does not).  Of several (bn3 and the shortcut's BatchNorm of a bottleneck) the one that ran last is taken.  None: the
    call has no such knob (sums of non-negative terms -- see one_sided())."""
    sites = tr.run(forward, prepare=prepare, grad=True)
    bns = list(tr.bns)
    order = {id(b): i for i, b in enumerate(bns)}       # (a module that runs twice keeps its last position)
    uniq = sorted({id(b): b for b in bns}.values(), key=lambda b: order[id(b)])
    biases = [b.bias for b in uniq]
    knobs = []
    for z, s, _ in sites:
        knob = None
        if s is not None and z.dim() == 4:
            count = float(z.numel() // z.shape[1])
            gs = torch.autograd.grad(s, biases, retain_graph=True, allow_unused=True)
            for b, g in zip(uniq, gs):
                if g is not None and g.numel() == z.shape[1] and torch.allclose(
                        g, torch.full_like(g, count), rtol=1e-9, atol=0.0):
                    knob = b
        knobs.append(knob)
    return knobs


def one_sided(z, upper):
    """A pre-activation tensor without a negative element in front of a plain ReLU (HRNet: relu(x_0 + up(relu(..))), a
    sum of ReLU outputs) cannot flip by rounding: its mask is [z > 0], and z == 0 exactly iff every term is an exact
    zero -- decided by the masks upstream, which have margins of their own."""
    return upper is None and z.min().item() >= 0.0


def _channel_points(z, c, upper):
    """Signed distances of channel c's pre-activations from the kink(s), as one sorted float64 array."""
    v = z[:, c].reshape(-1).double().numpy()
    if upper is not None:
        v = np.concatenate([v, v - upper])
    return np.sort(v)


def _site_margin(z, upper):
    """min |distance to a kink| / peak of the site's tensor"""
    peak = z.abs().max().item()
    if one_sided(z, upper):
        return float('inf'), peak
    d = z.abs().min().item()
    if upper is not None:
        d = min(d, (z - upper).abs().min().item())
    return d / (peak + 1e-300), peak


def _best_shift(points, reach):
    """Shift d with |d| <= reach that maximises min |p + d| over the sorted points: the kink lands at t = -d, mid-gap
    (or at an end of the reach when the points near it are few).  -> (d, the clearance reached)"""
    lo = np.searchsorted(points, -reach)
    hi = np.searchsorted(points, reach)
    inner = points[max(lo - 1, 0):min(hi + 1, len(points))]
    mids = 0.5 * (inner[1:] + inner[:-1]) if len(inner) > 1 else np.zeros(0)
    t = np.concatenate([mids[np.abs(mids) <= reach], [-reach, 0.0, reach]])
    i = np.searchsorted(points, t)
    left = np.where(i > 0, t - points[np.maximum(i - 1, 0)], np.inf)
    right = np.where(i < len(points), points[np.minimum(i, len(points) - 1)] - t, np.inf)
    clear = np.minimum(left, right)
    k = int(np.argmax(clear))
    return float(-t[k]), float(clear[k])


def _buffers(model):
    return {n: b.clone() for n, b in model.named_buffers()}


def _restore(model, saved):
    with torch.no_grad():
        for n, b in model.named_buffers():
            b.copy_(saved[n])


def trace(model, forward, prepare=None):
    """-> [(pre-activation tensor, upper kink or None)] for every ReLU call of forward(); buffers are left as found."""
    saved = _buffers(model)
    tr = _Trace(model)
    try:
        sites = tr.run(forward, prepare=prepare)
    finally:
        tr.close()
        _restore(model, saved)
    return [(z, up) for z, _, up in sites]


def measure(model, forward, prepare=None):
    """-> (min margin over all ReLU calls, [margin per call]).  Margins are relative to each call's tensor peak."""
    ms = [_site_margin(z, up)[0] for z, up in trace(model, forward, prepare)]
    return (min(ms) if ms else float('inf')), ms


def nudge(model, forward, reach=0.02, passes=1, verbose=False, prepare=None):
    """Move BatchNorm betas so that the ReLU pre-activations of ``forward()`` (a closure that runs ``model`` on the
    case's input, train / eval mode as the case wants) keep clear of the kinks.  ``reach``: largest shift, as a fraction
    of the site's peak.  Works on the model's own dtype (use a .double() copy); betas are kept fp32-representable so
    that a float model loads exactly the values that were verified.
    -> ({parameter name: new beta (fp32 tensor)}, min margin, [margin per site])"""
    names = {id(m): n for n, m in model.named_modules()}
    saved = _buffers(model)
    tr = _Trace(model)
    touched = {}
    try:
        knobs = _find_knobs(tr, forward, prepare)
        nsites = len(knobs)
        for _ in range(passes):
            for k in range(nsites):
                _restore(model, saved)
                sites = tr.run(forward, stop_after=k, prepare=prepare)
                z, _, upper = sites[k]
                knob = knobs[k]
                if verbose:
                    print('  site %3d/%d  knob %-44s margin before %.1e' % (k, nsites, names.get(id(knob)),
                                                                            _site_margin(z, upper)[0]), flush=True)
                if knob is None:
                    if not one_sided(z, upper):
                        raise RuntimeError('ReLU call %d has no translating BatchNorm and is not one-sided' % k)
                    continue
                peak = z.abs().max().item()
                beta = knob.bias.data
                for c in range(z.shape[1]):
                    d, _ = _best_shift(_channel_points(z, c, upper), reach * peak)
                    if d != 0.0:
                        beta[c] = torch.tensor(float(beta[c]) + d, dtype=torch.float32).to(beta.dtype)
                touched[names[id(knob)] + '.bias'] = knob
    finally:
        tr.close()
        _restore(model, saved)
    mn, ms = measure(model, forward, prepare)
    return {n: m.bias.detach().float().clone() for n, m in touched.items()}, mn, ms


def apply(module, arrays, prefix='nudge/'):
    """Load the nudged betas of a fixture (``nudge/<parameter name>`` arrays) into ``module``."""
    params = dict(module.named_parameters())
    n = 0
    with torch.no_grad():
        for k, v in arrays.items():
            if k.startswith(prefix):
                p = params[k[len(prefix):]]
                p.copy_(torch.as_tensor(np.asarray(v)).to(p.dtype))
                n += 1
    return n


def freeze_stats(model, x):
    """Give every BatchNorm meaningful running statistics (one training forward with momentum 1: running = batch
    statistics of x), then switch to eval mode: activations stay O(1) through the whole depth."""
    for mod in model.modules():
        if isinstance(mod, nn.BatchNorm2d):
            mod.momentum = 1.0
    model.train()
    with torch.no_grad():
        model(x)
    for mod in model.modules():
        if isinstance(mod, nn.BatchNorm2d):
            mod.momentum = 0.1
    model.eval()


def noise32(model64, forward64, model32, forward32, prepare64=None, prepare32=None):
    """Per ReLU call: max |fp32 pre-activation - fp64 pre-activation| / peak -- what a margin has to exceed."""
    a = trace(model64, forward64, prepare64)
    b = trace(model32, forward32, prepare32)
    return [((za - zb.double()).abs().max() / (za.abs().max() + 1e-300)).item() for (za, _), (zb, _) in zip(a, b)]


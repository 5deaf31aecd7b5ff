"""compute_loss / masks / metrics on the HIP kernels (reference utils/utils.py:12-24,51-65, test.py:31-46)."""
import os

import torch

from .. import ops
from ..ops import Act


# PSEG_CHECK_LABELS=1: raise like torch on a target outside [0, C) (costs one host synchronisation per loss call).
# Without it such pixels are treated as ignored -- consistently in the divisor, the sum and the gradient -- and their
# number is available as element [2] of ops.ce_fwd_bwd's first result.
CHECK_LABELS = os.environ.get('PSEG_CHECK_LABELS', '0') == '1'


class _CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss() defaults; forward and backward arithmetic happen in ONE pass over the logits
    (pseg_ce_fwd_bwd); backward only rescales by the incoming scalar (a device-side no-op when it is 1)."""

    @staticmethod
    def forward(ctx, logits, targets):
        need = logits.requires_grad
        out, dl = ops.ce_fwd_bwd(logits, targets, want_grad=need)
        if CHECK_LABELS and out[2].item() != 0:     # (host sync: debugging aid, off by default)
            raise IndexError('%d target values are outside [0, %d) and are not ignore_index (torch raises "Target ... is '
                             'out of bounds" here)' % (int(out[2].item()), logits.shape[1]))
        ctx.dl = dl
        ctx.mark_non_differentiable()
        return out[0]

    @staticmethod
    def backward(ctx, g):
        dl = ctx.dl
        ctx.dl = None
        if dl is None:
            return None, None
        g = g.reshape(1) if g.dtype == torch.float32 else g.float().reshape(1)
        ops.scale_inplace(dl, g.contiguous())
        return dl, None


class _ResizeFn(torch.autograd.Function):
    """F.interpolate(outputs, size, mode='bilinear', align_corners=True) on NCHW logits
    (reference utils/utils.py:18-20; only taken when the sizes differ, i.e. --multi-scale)."""

    @staticmethod
    def forward(ctx, x, Ho, Wo):
        B, C, H, W = x.shape
        xa = Act.from_nchw(x)
        ctx.shape = (B, C, H, W, xa.C)
        return ops.bilinear_fwd_nchw(xa, C, Ho, Wo, True)

    @staticmethod
    def backward(ctx, g):
        B, C, H, W, Cp = ctx.shape
        dx = Act.empty(B, H, W, Cp, g.device, zero=True)
        ops.bilinear_bwd_nchw(g.contiguous(), dx, C, True)
        return dx.to_nchw(C), None, None


def compute_loss(outputs, targets, model=None):
    """Same signature and result as the reference's compute_loss(outputs, targets, model) (utils/utils.py:17-24)."""
    if not outputs.is_cuda:
        raise RuntimeError('compute_loss runs on the HIP path only (got %s)' % outputs.device)
    targets = targets.to(device=outputs.device, dtype=torch.int64).contiguous()
    th, tw = targets.size(1), targets.size(2)
    if (outputs.size(2), outputs.size(3)) != (th, tw):
        outputs = _ResizeFn.apply(outputs, th, tw)
    # equal sizes: bilinear align_corners=True is the exact identity, so nothing is launched
    return _CrossEntropyFn.apply(outputs.contiguous(), targets)


def predict_mask(outputs):
    """``outputs.max(1)[1]`` (reference test.py:31): int64 [B,H,W], first index on ties."""
    return ops.argmax(outputs.contiguous())


def update_class_counts(counters, predicted, targets):
    """Accumulate per-class tp / fn / fp (reference test.py:34-46) into the int64 device tensor counters[3][C]
    with one kernel instead of 3*C host synchronisations."""
    ops.confusion(predicted.contiguous().view(-1), targets.to(torch.int64).contiguous().view(-1), counters)
    return counters


def compute_metrics(tp, fn, fp):
    """reference utils/utils.py:51-65 -- host-side, [num_classes]-sized tensors (not on the hot path)."""
    tp, fn, fp = tp.clone().float(), fn.clone().float(), fp.clone().float()

    def guarded(den):
        den = den.clone()
        den[den <= 0] = 1
        return den

    miou = tp / guarded(tp + fp + fn)
    T = tp + fn
    P = tp / guarded(tp + fp)
    R = tp / guarded(tp + fn)
    F1 = 2 * tp / guarded(2 * tp + fp + fn)
    return T, P, R, miou, F1

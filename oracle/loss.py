"""Oracle restatement of the reference's loss / mask / metric arithmetic.

Test infrastructure only -- see oracle/__init__.py.
"""
import torch
import torch.nn.functional as F


def compute_loss(outputs, targets, model=None):
    """reference utils/utils.py:12,17-24 -- bilinear-resize the logits to the target's
    H x W (align_corners=True; exact identity at equal size) then
    nn.CrossEntropyLoss() defaults (mean over pixels, ignore_index=-100)."""
    outputs = F.interpolate(outputs, (targets.size(1), targets.size(2)),
                            mode='bilinear', align_corners=True)
    return F.cross_entropy(outputs, targets)


def predict_mask(outputs):
    """reference test.py:31 -- ``outputs.max(1)[1]`` (ties -> first index)."""
    return outputs.max(1)[1]


def class_counts(predicted, targets, num_classes):
    """reference test.py:34-46 -- per-class tp / fn / fp over a batch."""
    predicted = predicted.reshape(-1)
    targets = targets.reshape(-1)
    eq = predicted.eq(targets)
    tp = torch.zeros(num_classes)
    fn = torch.zeros(num_classes)
    fp = torch.zeros(num_classes)
    for c in range(num_classes):
        sel = targets.eq(c)
        positive = sel.sum().item()
        tpi = eq[sel].sum().item()
        tp[c] = tpi
        fn[c] = positive - tpi
        fp[c] = predicted.eq(c).sum().item() - tpi
    return tp, fn, fp


def compute_metrics(tp, fn, fp):
    """reference utils/utils.py:51-65 -- T, precision, recall, IoU, F1 with zero guards."""
    tp, fn, fp = tp.clone(), fn.clone(), fp.clone()

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

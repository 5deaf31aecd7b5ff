"""One tiny forward + loss + backward of DeepLabV3+ on cuda:0, checked against the CPU oracle.
(`oracle/` is imported here only as the checker -- see oracle/__init__.py.)"""
import torch


def run(size=96, batch=4, num_classes=21, tol=1e-3, verbose=True):
    from oracle import fill
    from oracle import loss as oloss
    from oracle import models as omodels
    from . import _lib
    from .models import DeepLabV3Plus
    from .utils import compute_loss, predict_mask

    _lib.load()
    torch.manual_seed(0)
    ref = omodels.DeepLabV3Plus(num_classes)
    fill.fill_module_(ref, 'smoke')
    ref.train()
    x = fill.images('smoke/x', (batch, 3, size, size))
    tgt = fill.labels('smoke/t', (batch, size, size), num_classes, block=8)
    import copy
    ref64 = copy.deepcopy(ref).double()   # exact-arithmetic yardstick for the (ill-conditioned) deep gradients
    out_ref = ref(x)
    loss_ref = oloss.compute_loss(out_ref, tgt)
    loss_ref.backward()
    oloss.compute_loss(ref64(x.double()), tgt).backward()
    g64 = dict((n, p.grad) for n, p in ref64.named_parameters())

    dev = torch.device('cuda', 0)
    model = DeepLabV3Plus(num_classes)
    model.load_state_dict(ref.state_dict())
    model.to(dev).train()
    # reset the running statistics the oracle's forward just updated
    fresh = omodels.DeepLabV3Plus(num_classes)
    fill.fill_module_(fresh, 'smoke')
    model.load_state_dict(fresh.state_dict())
    out = model(x.to(dev))
    loss = compute_loss(out, tgt.to(dev), model)
    loss.backward()
    torch.cuda.synchronize()

    def rel(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()

    e_out = rel(out, out_ref)
    e_loss = abs(loss.item() - loss_ref.item()) / abs(loss_ref.item())
    worst, worst_name = 0.0, ''
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        # excess of the HIP gradient's distance to the exact (fp64) gradient over 3x the fp32 oracle's own distance
        e = rel(p.grad, g64[n]) - 3 * rel(q.grad, g64[n])
        if e > worst:
            worst, worst_name = e, n
    mask_ok = torch.equal(predict_mask(out).cpu(), oloss.predict_mask(out_ref))
    if verbose:
        print('smoke: logits rel err %.2e, loss rel err %.2e, worst grad excess err %.2e (%s), mask exact %s'
              % (e_out, e_loss, worst, worst_name, mask_ok))
    assert e_out < tol and e_loss < tol and worst < tol, 'HIP path deviates from the CPU oracle'
    return dict(logits=e_out, loss=e_loss, grad=worst, mask_exact=mask_ok)

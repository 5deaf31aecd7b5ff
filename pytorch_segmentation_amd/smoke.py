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
    # gradients: the classifier's (directly under the loss, well conditioned) must meet the contract outright;
    # the deep ones are judged as a population against the exact (fp64) gradients, next to the fp32 oracle's own
    # distance from them (tiny batches make individual late-layer gradients chaotic in ANY fp32 implementation)
    e_cls = max(rel(model.cls_conv.weight.grad, ref.cls_conv.weight.grad), rel(model.cls_conv.bias.grad, ref.cls_conv.bias.grad))
    e_hip, e_ref = [], []
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        if g64[n].abs().max().item() < 1e-12:
            continue
        e_hip.append(rel(p.grad, g64[n]))
        e_ref.append(rel(q.grad, g64[n]))
    med_hip = sorted(e_hip)[len(e_hip) // 2]
    med_ref = sorted(e_ref)[len(e_ref) // 2]
    top2 = out_ref.detach().topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-3 * out_ref.abs().max()
    mask_ok = torch.equal(predict_mask(out).cpu()[safe], oloss.predict_mask(out_ref)[safe])
    if verbose:
        print('smoke: logits rel err %.2e, loss rel err %.2e, classifier grad rel err %.2e, median grad err vs fp64 '
              '%.2e (fp32 CPU oracle: %.2e), masks exact on %.1f%% safe pixels: %s'
              % (e_out, e_loss, e_cls, med_hip, med_ref, 100 * safe.float().mean().item(), mask_ok))
    assert e_out < tol and e_loss < tol and e_cls < tol and mask_ok, 'HIP path deviates from the CPU oracle'
    assert med_hip < max(tol, 5 * med_ref), 'HIP gradients deviate from the exact gradients more than the CPU oracle does'
    # the -mp path (fp16 storage, fp32 master weights, loss scaling) on the same model and batch: its loss against the oracle's
    # at the half policy's own tolerance, one optimiser step applied on the device
    from .utils import Trainer
    model.load_state_dict(fresh.state_dict())
    tr = Trainer(model, None, lr=1e-3, mixed_precision=True, device=dev)
    model.train()
    l_half = tr.train_batch(x.to(dev), tgt.to(dev)).item()
    st = tr.loss_scale_state()
    e_half = abs(l_half - loss_ref.item()) / abs(loss_ref.item())
    if verbose:
        print('smoke (-mp, fp16 storage): loss rel err vs the fp32 oracle %.2e, loss-scale state %s' % (e_half, st))
    assert e_half < 5e-3 and st['steps_applied'] + st['steps_skipped'] == 1, 'half-precision path deviates from the CPU oracle'
    return dict(logits=e_out, loss=e_loss, cls_grad=e_cls, median_grad=med_hip, median_grad_ref=med_ref, mask_exact=mask_ok)

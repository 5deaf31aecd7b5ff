"""Per-parameter gradient error listing of the full models vs the CPU oracle (debug aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fill, loss as oloss, models as omodels  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def run(name, hip_cls, ref, nc, S, B):
    from pytorch_segmentation_amd.utils import compute_loss
    fill.fill_module_(ref, name)
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    ref.train()
    x = fill.images(name + '/x', (B, 3, S, S))
    tgt = fill.labels(name + '/t', (B, S, S), nc, block=8)
    out_ref = ref(x)
    loss_ref = oloss.compute_loss(out_ref, tgt)
    loss_ref.backward()
    # the oracle's own conditioning: same graph in fp64
    import copy
    ref64 = copy.deepcopy(ref).double()
    ref64.load_state_dict({k: v.double() for k, v in state.items()})
    ref64.zero_grad()
    ref64.train()
    out64 = ref64(x.double())
    oloss.compute_loss(out64, tgt).backward()
    g64 = {n: p.grad for n, p in ref64.named_parameters()}
    m = hip_cls(nc)
    m.load_state_dict(state)
    m.cuda().train()
    out = m(x.cuda())
    loss = compute_loss(out, tgt.cuda(), m)
    loss.backward()
    print(name, 'out', rel(out, out_ref), 'loss', loss.item(), loss_ref.item())
    for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        e = rel(p.grad, q.grad)
        e64 = rel(p.grad, g64[n])
        r64 = rel(q.grad, g64[n])
        flag = ' <<<<' if e64 > max(1e-3, 3 * r64) else ''
        print('%-46s %-20s hip-vs-ref32 %.1e  hip-vs-ref64 %.1e  ref32-vs-ref64 %.1e%s' % (n, tuple(p.shape), e, e64, r64, flag))


if __name__ == '__main__':
    from pytorch_segmentation_amd.models import DeepLabV3Plus, HRNet, UNet
    which = sys.argv[1] if len(sys.argv) > 1 else 'both'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    if which in ('dl', 'both'):
        run('full_dl', DeepLabV3Plus, omodels.DeepLabV3Plus(21), 21, S, B)
    if which in ('unet', 'both'):
        run('full_unet', UNet, omodels.UNet(2), 2, S, B)
    if which == 'hrnet':
        run(os.environ.get('KEY', 'full_hrnet'), HRNet, omodels.HRNet(5), 5, S, B)

import sys, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from oracle import fill
from pytorch_segmentation_amd.nn import ConvNormAct
from pytorch_segmentation_amd.utils import Trainer
def build():
    torch.manual_seed(0)
    return torch.nn.Sequential(ConvNormAct(8, 16, 3), ConvNormAct(16, 16, 1), ConvNormAct(16, 8, 3, activate=None))
x = fill.uniform('hfb/x', (4, 8, 32, 32)).cuda()
t = fill.labels('hfb/t', (4, 32, 32), 8, block=4).cuda()
loss_fn = lambda out, tgt, model: F.cross_entropy(out, tgt)
g = {}
for mp in (False, True):
    m = build()
    tr = Trainer(m, None, loss_fn=loss_fn, lr=1e-2, mixed_precision=mp, device=torch.device('cuda', 0))
    m.train()
    out = m(x)
    loss = loss_fn(out, t, m)
    if mp:
        tr._bridge_half(); out = m(x); loss = loss_fn(out, t, m)
        (loss * tr.mp_state[0]).backward()
        sc = tr.mp_state[0].item()
    else:
        loss.backward(); sc = 1.0
    tr.env.half_fresh = False
    g[mp] = {n: (p.grad / sc).clone() for n, p in m.named_parameters()}
    print(mp, loss.item(), out.abs().max().item())
for n in g[False]:
    a, b = g[True][n].double(), g[False][n].double()
    print('%-20s rel %.3e  l2 %.3e peak %.3e' % (n, ((a-b).abs().max()/b.abs().max()).item(), ((a-b).norm()/b.norm()).item(), b.abs().max().item()))
print('---- train_batch updates')
upd = {}
for mp in (False, True):
    m = build()
    tr = Trainer(m, None, loss_fn=loss_fn, lr=1e-2, mixed_precision=mp, device=torch.device('cuda', 0))
    m.train()
    p0 = {n: p.detach().clone() for n, p in m.named_parameters()}
    res = []
    for step in range(2):
        tr.train_batch(x, t)
        res.append({n: (p.detach() - p0[n]).clone() for n, p in m.named_parameters()})
    upd[mp] = res
    print(mp, tr.loss_scale_state() if mp else None)
for step in range(2):
    for n in upd[False][step]:
        a, b = upd[True][step][n].double(), upd[False][step][n].double()
        print('step %d %-12s rel %.3e peak %.3e / %.3e' % (step, n, ((a-b).abs().max()/b.abs().max()).item(), b.abs().max().item(), a.abs().max().item()))

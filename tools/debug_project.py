import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fill, loss as oloss, models as omodels
from pytorch_segmentation_amd import ops
from pytorch_segmentation_amd.models import DeepLabV3Plus
from pytorch_segmentation_amd.utils import compute_loss

def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()

B, S, nc = 4, 128, 21
ref = omodels.DeepLabV3Plus(nc).double()
fill.fill_module_(ref, 'full_dl')
state = {k: v.clone().float() for k, v in ref.state_dict().items()}
ref.train()
cap = {}
ref.project[0].register_forward_hook(lambda m, i, o: cap.__setitem__('x', i[0].detach()))
ref.project[0].register_full_backward_hook(lambda m, gi, go: cap.__setitem__('dy', go[0].detach()))
ref.project.register_full_backward_hook(lambda m, gi, go: cap.__setitem__('dz', go[0].detach()))
x = fill.images('full_dl/x', (B, 3, S, S))
tgt = fill.labels('full_dl/t', (B, S, S), nc, block=8)
out_ref = ref(x.double())
oloss.compute_loss(out_ref, tgt).backward()

m = DeepLabV3Plus(nc); m.load_state_dict(state); m.cuda().train()
rec = {}
orig_w = ops.conv2d_wgrad
def spy(x_, dy_, dw, kh, kw, s, p, d, accumulate=False):
    if x_.C == 256 and dy_.C == 128 and kh == 1 and 'x' not in rec:
        rec['x'] = x_.to_nchw(); rec['dy'] = dy_.to_nchw()
    return orig_w(x_, dy_, dw, kh, kw, s, p, d, accumulate)
ops.conv2d_wgrad = spy
orig_b = ops.bn_act_bwd
def spyb(dz, z, y, co, act, dy, gg, bg, accumulate=False, dres=None, res_accumulate=False):
    if y.C == 128 and dz.ld == 384 and 'dz' not in rec:
        rec['dz'] = dz.to_nchw(); rec['z'] = z.to_nchw(); rec['y'] = y.to_nchw(); rec['co'] = co.clone()
    return orig_b(dz, z, y, co, act, dy, gg, bg, accumulate, dres, res_accumulate)
ops.bn_act_bwd = spyb
out = m(x.cuda()); loss = compute_loss(out, tgt.cuda(), m); loss.backward()
print('out', rel(out, out_ref))
print('x   ', rel(rec['x'], cap['x']))
print('dz  ', rel(rec['dz'], cap['dz']), 'max', cap['dz'].abs().max().item())
print('dy  ', rel(rec['dy'], cap['dy']), 'max', cap['dy'].abs().max().item())
print('dw  ', rel(m.project[0].weight.grad if hasattr(m.project, '__getitem__') else m.project.conv.weight.grad, ref.project[0].weight.grad))
d = (rec['dy'].cpu().double() - cap['dy']).abs()
print('dy err per-channel max (top5):', d.amax((0, 2, 3)).topk(5))
print('dz err per-channel max (top5):', (rec['dz'].cpu().double() - cap['dz']).abs().amax((0,2,3)).topk(5))

ch = 126
y64 = rec['y'].cpu().double()
print('channel', ch, 'ours mean/invstd/scale/beta', rec['co'][:, ch].tolist())
print('from our y in fp64: mean', y64[:, ch].mean().item(), 'var', y64[:, ch].var(unbiased=False).item(), 'invstd', (1.0 / (y64[:, ch].var(unbiased=False) + 1e-5).sqrt()).item())
print('y range ch', y64[:, ch].min().item(), y64[:, ch].max().item())
cap2 = {}
zz = rec['z'].cpu().double()[:, ch]
print('z>0 frac', (zz > 0).double().mean().item(), 'z max', zz.max().item())
dzc = rec['dz'].cpu().double()[:, ch]; print('dz ch abs max', dzc.abs().max().item())
dyo = rec['dy'].cpu().double()[:, ch]; dyr = cap['dy'][:, ch]
print('dy ours absmax', dyo.abs().max().item(), 'ref absmax', dyr.abs().max().item(), 'diff max', (dyo - dyr).abs().max().item())
print('gamma', m.project._modules['1'].weight[ch].item(), 'beta', m.project._modules['1'].bias[ch].item())

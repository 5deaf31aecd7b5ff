"""Diagnostic for tests/test_fullsize_parity_gpu.py: per-tensor distance of the HIP parameter gradients from the CPU oracle's at
configs[2] full size, with norm ratio and cosine, the feature gradients, and the oracle's own sensitivity."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import fill, loss as oloss, models as omodels
from pytorch_segmentation_amd import ops, prepare
from pytorch_segmentation_amd.models import DeepLabV3Plus
from pytorch_segmentation_amd.nn import Env
from pytorch_segmentation_amd.utils import compute_loss

B = int(os.environ.get('DIAG_B', 16)); S = int(os.environ.get('DIAG_S', 512)); NC = 21
torch.set_num_threads(16)
ops.set_conv_precision('fp32')
ref = omodels.DeepLabV3Plus(NC)
fill.fill_module_(ref, 'cfg2t')
state = {k: v.clone() for k, v in ref.state_dict().items()}
x = fill.images('cfg2t/x', (B, 3, S, S))
tgt = fill.labels('cfg2t/t', (B, S, S), NC, block=16)


def oracle(xx, dtype=torch.float32):
    r = omodels.DeepLabV3Plus(NC)
    r.load_state_dict(state)
    r = r.to(dtype).train()
    kept = {}
    def hook(_m, _i, out):
        for i in (1, 4):
            out[i].retain_grad(); kept[i] = out[i]
    r.backbone.register_forward_hook(hook)
    t0 = time.time()
    out = r(xx.to(dtype))
    loss = oloss.compute_loss(out, tgt)
    loss.backward()
    print('oracle %s step %.1f s, loss %.7f' % (dtype, time.time() - t0, loss.item()), flush=True)
    return out.detach(), {n: p.grad.detach().clone() for n, p in r.named_parameters()}, {i: t.grad.detach().clone() for i, t in kept.items()}


def cmp(tag, a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    l2 = ((a - b).norm() / b.norm()).item()
    mx = ((a - b).abs().max() / b.abs().max()).item()
    ratio = (a.norm() / b.norm()).item()
    cos = (a @ b / (a.norm() * b.norm())).item()
    print('%-44s l2 %.2e  max %.2e  |a|/|b| %.5f  1-cos %.2e' % (tag, l2, mx, ratio, 1 - cos), flush=True)


out_ref, g_ref, f_ref = oracle(x)
m = DeepLabV3Plus(NC)
m.load_state_dict(state)
prepare(m, 'cuda')
m.train()
xg, tg = x.cuda(), tgt.cuda()
env = Env(save=True, accumulate=False)
m._pseg_arena.transpose_filters(); env.wT_fresh = True
out2, (s_bb, s_head) = m.model_fwd(xg, env)
cmp('logits', out2, out_ref)
_, dl = ops.ce_fwd_bwd(out2, tg)
dl_ref = torch.autograd.grad(oloss.compute_loss(out_ref.clone().requires_grad_(), tgt), [], allow_unused=True) if False else None
dlow, dhigh = m.head_bwd(dl, s_head, env)
cmp('dlow (stride 4)', dlow.to_nchw(256), f_ref[1])
cmp('dhigh (stride 16)', dhigh.to_nchw(2048), f_ref[4])
m.backbone.bwd([None, dlow, None, None, dhigh], s_bb, env)
ops.join_aux(xg.device)
torch.cuda.synchronize()
names = [n for n, _ in m.named_parameters()]
show = [n for n in names if not n.startswith('backbone.layer') or n.endswith('conv1.weight') or n.endswith('conv2.weight')]
for n, p in m.named_parameters():
    if n in show:
        cmp(n, p.grad, g_ref[n])
if os.environ.get('DIAG_SENS', '1') == '1':
    # the oracle against itself: the image perturbed in the last bit
    xp = x * (1 + 2e-7)
    out_p, g_p, f_p = oracle(xp)
    cmp('ORACLE self: logits', out_p, out_ref)
    cmp('ORACLE self: dhigh', f_p[4], f_ref[4])
    for n in ('backbone.conv1.weight', 'backbone.layer1.0.conv1.weight', 'backbone.layer4.1.conv1.weight', 'aspp.project.0.weight', 'cls_conv.weight'):
        cmp('ORACLE self: ' + n, g_p[n], g_ref[n])
if os.environ.get('DIAG_F64', '0') == '1':
    out64, g64, f64 = oracle(x, torch.float64)
    cmp('fp32 oracle vs fp64: logits', out_ref, out64)
    cmp('HIP vs fp64: logits', out2, out64)
    cmp('fp32 oracle vs fp64: dhigh', f_ref[4], f64[4])
    cmp('HIP vs fp64: dhigh', dhigh.to_nchw(2048), f64[4])
    for n, p in m.named_parameters():
        if n in show:
            cmp('HIP vs fp64 ' + n, p.grad, g64[n])
            cmp('o32 vs fp64 ' + n, g_ref[n], g64[n])

"""Debug: DeepLabV3+ head at the 384x384 pyramid -- per-call check + where the df4 error sits."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
from oracle import fill, models as omodels, loss as oloss
from opcheck import OpCheck
import pytorch_segmentation_amd as pseg
from pytorch_segmentation_amd import ops
from pytorch_segmentation_amd.models import DeepLabV3Plus
from pytorch_segmentation_amd.nn import Env
from pytorch_segmentation_amd.ops import Act
ops.set_conv_precision(sys.argv[1] if len(sys.argv) > 1 else 'fp32')
S = 384
ref = omodels.DeepLabV3Plus(21, backbone=torch.nn.Identity()); fill.fill_module_(ref, 'deeplab_head')
m = DeepLabV3Plus(21, backbone=torch.nn.Identity()); m.load_state_dict(ref.state_dict()); pseg.prepare(m, 'cuda'); m.train()
chans, strides = (64, 256, 512, 1024, 2048), (2, 4, 8, 16, 16)
feats = [fill.uniform('deeplab_head/f%d' % i, (4, c, S // s, S // s), 1.0).abs_() for i, (c, s) in enumerate(zip(chans, strides))]
tgt = fill.labels('deeplab_head/target', (4, S, S), 21, block=8)
env = Env(save=True, accumulate=False)
with OpCheck(verbose=True) as oc:
    low, high = Act.from_nchw(feats[1].cuda()), Act.from_nchw(feats[4].cuda())
    out, saved = m.head_fwd(low, high, env)
    lo, dl = ops.ce_fwd_bwd(out, tgt.cuda())
    dlow, dhigh = m.head_bwd(dl, saved, env)
# oracle in fp32 and fp64
import copy
ref.train()
for dt in (torch.float32, torch.float64):
    r = copy.deepcopy(ref).to(dt)
    fs = [f.detach().clone().to(dt).requires_grad_() for f in feats]
    o = r.head(fs); l = oloss.compute_loss(o, tgt); l.backward()
    g = fs[4].grad.double()
    d = (dhigh.to_nchw().cpu().double() - g)
    print(dt, 'df4 max err rel', (d.abs().max() / g.abs().max()).item(), 'per-batch', [(d[b].abs().max() / g.abs().max()).item() for b in range(4)])
    # structure: is the error constant over pixels (pool branch)?
    dm = d.mean((2, 3), keepdim=True)
    print('   after removing per-(b,c) mean:', ((d - dm).abs().max() / g.abs().max()).item(), ' mean part:', (dm.abs().max() / g.abs().max()).item())
    if dt == torch.float64: g64 = g
    else: g32 = g
print('oracle fp32 vs fp64:', ((g32 - g64).abs().max() / g64.abs().max()).item())

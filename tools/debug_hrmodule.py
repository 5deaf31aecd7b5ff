"""HRModule-level forward/backward parity vs the CPU oracle over several seeds (debug aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fill, models as omodels  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def main():
    import pytorch_segmentation_amd as pseg
    from pytorch_segmentation_amd.models.hrnet import BasicBlock, HRModule
    from pytorch_segmentation_amd.nn import Env
    from pytorch_segmentation_amd.ops import Act
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    mso = (sys.argv[3] != '0') if len(sys.argv) > 3 else True
    widths = [32 * 2 ** i for i in range(nb)]
    for seed in range(int(os.environ.get("SEEDS", "4"))):
        key = 'hrmod%d' % seed
        ref = omodels.HRModule(nb, omodels.HRBasicBlock, [4] * nb, list(widths), widths, mso).double()
        fill.fill_module_(ref, key)
        ref.train()
        xs = [fill.uniform('%s/x%d' % (key, i), (4, w, S >> i, S >> i), 1.0).abs_().double().requires_grad_()
              for i, w in enumerate(widths)]
        outs = ref(list(xs))
        gys = [fill.uniform('%s/g%d' % (key, i), tuple(o.shape), 1.0).double() for i, o in enumerate(outs)]
        sum((o * g).sum() for o, g in zip(outs, gys)).backward()
        m = HRModule(nb, BasicBlock, [4] * nb, list(widths), widths, mso)
        m.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
        pseg.prepare(m, 'cuda')
        m.train()
        env = Env(save=True, accumulate=False)
        xa = [Act.from_nchw(x.detach().float().cuda()) for x in xs]
        oa, saved = m.fwd(xa, env)
        da = [Act.from_nchw(g.float().cuda()) for g in gys]
        dxa = m.bwd(da, saved, env)
        print(key, 'out', ['%.1e' % rel(o.to_nchw(), r) for o, r in zip(oa, outs)],
              'dx', ['%.1e' % rel(d.to_nchw(), x.grad) for d, x in zip(dxa, xs)])
        worst = sorted(((rel(p.grad, q.grad), n) for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters())),
                       reverse=True)[:6]
        for e, n in worst:
            print('   %.1e %s' % (e, n))


if __name__ == "__main__" and not os.environ.get("PROBE"):
    main()


def probe(seed=1, nb=3, S=16, i=1, j=2):
    """Compare the ReLU mask of fuse term (i, j>i) between the fp64 oracle and the HIP path."""
    import pytorch_segmentation_amd as pseg
    from pytorch_segmentation_amd.models.hrnet import BasicBlock, HRModule
    from pytorch_segmentation_amd.nn import Env
    from pytorch_segmentation_amd.ops import Act
    widths = [32 * 2 ** k for k in range(nb)]
    key = 'hrmod%d' % seed
    ref = omodels.HRModule(nb, omodels.HRBasicBlock, [4] * nb, list(widths), widths, True).double()
    fill.fill_module_(ref, key)
    ref.train()
    xs = [fill.uniform('%s/x%d' % (key, k), (4, w, S >> k, S >> k), 1.0).abs_().double() for k, w in enumerate(widths)]
    grab = {}
    ref.fuse_layers[i][j][0].register_forward_hook(lambda m, a, o: grab.__setitem__('t', o.detach()))
    ref.fuse_layers[i][j][0][1].register_forward_hook(lambda m, a, o: grab.__setitem__('bn', o.detach().clone()))
    ref(list(xs))
    import copy
    ref32 = copy.deepcopy(ref).float()
    g32 = {}
    names = {}
    for n, mod in ref.named_modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.register_forward_hook(lambda m_, a, o, n=n: grab.__setitem__('bn64/' + n, o.detach().clone()))
    for n, mod in ref32.named_modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.register_forward_hook(lambda m_, a, o, n=n: g32.__setitem__('bn64/' + n, o.detach().clone()))
    ref(list(xs))
    ref32([x.float() for x in xs])
    for n in ('branches.2.0.bn1', 'branches.2.0.bn2', 'branches.2.1.bn2', 'branches.2.3.bn2', 'fuse_layers.1.2.0.1'):
        print('oracle32 vs oracle64 at', n, (g32['bn64/' + n].double() - grab['bn64/' + n]).abs().max().item())
    m = HRModule(nb, BasicBlock, [4] * nb, list(widths), widths, True)
    m.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    pseg.prepare(m, 'cuda')
    m.train()
    env = Env(save=True, accumulate=False)
    oa, saved = m.fwd([Act.from_nchw(x.float().cuda()) for x in xs], env)
    for bi, blk in ((0, 'bn1'), (0, 'bn2'), (1, 'bn2'), (3, 'bn2')):
        sbk = saved[0][2][bi]
        yb, zb, cob = (sbk[1] if blk == 'bn1' else sbk[3])[:3]
        bn = getattr(m.branches[2][bi], blk)
        preb = ((yb.view4() - cob[0]) * cob[2] + bn.bias.detach()).permute(0, 3, 1, 2).double().cpu()
        print('hip vs oracle64 at branches.2.%d.%s' % (bi, blk), (preb - grab['bn64/branches.2.%d.%s' % (bi, blk)]).abs().max().item())
    (sc, sb), tshape = saved[1][i][j]
    y, z, co, act, ub = sb
    mean, invstd, scale, shift = co
    pre = (y.view4() - mean) * scale + m.fuse_layers[i][j][0].bn.bias.detach()
    pre = pre.permute(0, 3, 1, 2).double().cpu()
    print('z saved:', z is not None, 'act', act, 'pre-activation err', (pre - grab['bn']).abs().max().item())
    diff = (pre > 0) != (grab['bn'] > 0)
    print('mask mismatches', diff.sum().item(), 'of', diff.numel(), 'oracle', grab['bn'][diff][:8].tolist(), 'hip', pre[diff][:8].tolist())
    idx = diff.nonzero()
    print('where', idx.tolist(), 'mean/scale/beta at channel', [(mean[c].item(), scale[c].item(), invstd[c].item()) for c in idx[:, 1].tolist()])
    c = idx[0, 1].item()
    print('oracle channel values', grab['bn'][:, c].flatten()[:16].tolist())
    print('hip y channel', y.view4()[..., c].flatten()[:16].tolist())
    print('|bn| min', grab['bn'].abs().min().item())


if __name__ == '__main__' and os.environ.get('PROBE'):
    probe()

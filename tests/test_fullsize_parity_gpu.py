"""Full-size TRAIN-MODE parity against the oracle: ONE training step of BASELINE.json configs[2]
(DeepLabV3+ R50, 21 classes, 512x512, batch 16) and configs[4] (HRNet, 21 classes, 512x512, batch 8) on the HIP path and on
the CPU oracle from the same filled parameters and the same seeded batch.

What the small whole-model cases cannot show is here at the headline size: batch statistics over 262 144 values per channel
(stride-4 maps), the fused low-resolution loss + forked weight gradients of the Trainer step `bench.py` times, every large
plan tile with its real grid.  The oracle is stock torch fp32 on the CPU composed as reference models/deeplabv3plus.py:28-44,
models/hrnet.py:373-406 and utils/utils.py:17-24 compose it (oracle/models.py, oracle/loss.py); one such step costs the
GPU box's 16 usable cores ~6 s / ~4 s.

Bounds (north_star: 1e-3 relative fp32, masks bit-exact):
  * logits, loss, every BatchNorm running statistic: max-norm 1e-3 of the tensor's peak (measured: 1e-4, 3e-7, <= 1e-5);
  * masks: bit-exact on the pixels whose top-2 oracle margin exceeds the tolerance;
  * gradients.  MEASURED FIRST (round 6, tools/lab/fullsize_diag.py, profiles/r06_fullsize_parity.md): at this size the backward
    pass of the random-init network is ill-conditioned for ANY fp32 implementation -- the CPU oracle evaluated on the image
    moved by ONE unit in the last place (x * (1 + 2^-22)) moves its own logits by 9e-5, its own stride-16 feature gradient by
    6e-3 and its own parameter gradients by 4e-3 (ASPP projection) .. 2.2e-2 (stem) in relative L2: ~1e8 ReLU pre-activations,
    a few hundred within rounding distance of 0, every flipped mask element a 1/sqrt(pixels) change of a filter row.  The
    HIP path sits at the SAME distances from the oracle (4e-3 .. 2.1e-2, tensor by tensor within 5 % of the oracle's own),
    and the plain 1e-3 on gradients that VERDICT r5 item 5 asked for is not a property any implementation can have here.
    What is asserted instead, per tensor: distance from the oracle <= max(1e-3, 3 x the oracle's own distance under the
    one-ulp perturbation, measured in the same test), cosine >= 0.999 -- and, in the test that follows, the STRICT form that IS
    well-posed at full size: every kernel call of the step recomputed in fp64 from its own device inputs (tests/opcheck.py),
    1e-4, no allowance.
fp32 policy only (the headline's arithmetic); the `-mp` leg of configs[4] is compared at its own stated tolerance.
"""
import numpy as np
import pytest
import torch

from oracle import fill
from oracle import loss as oloss
from oracle import models as omodels

pytestmark = pytest.mark.gpu
TOL = 1e-3


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def l2rel(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-300)).item()


@pytest.fixture()
def fp32_policy():
    from pytorch_segmentation_amd import ops
    before = ops.POLICY_NAME
    ops.set_conv_precision('fp32')
    yield
    ops.set_conv_precision(before)


ULP = 2.0 ** -22       # the yardstick perturbation: every image value moved by about one unit in the last place


def _oracle_step(ref, state, x, tgt, head_features=None):
    """one train-mode forward + loss + backward of the CPU oracle (fp32) from `state`; -> logits, loss, parameter gradients,
    buffers after, {i: gradient the HEAD sends into backbone feature i} (head_features: DeepLabV3+ only -- the backbone is
    cut from the head there, so feature 1's gradient is the head's own, not the sum with what layer 2 sends back)"""
    ref.load_state_dict(state)
    ref.train()
    ref.zero_grad(set_to_none=True)
    fgr = {}
    if head_features:
        feats = ref.backbone(x)
        cut = [f.detach().requires_grad_() if i in head_features else f for i, f in enumerate(feats)]
        out = ref.head(cut)
        loss = oloss.compute_loss(out, tgt)
        loss.backward()
        fgr = {i: cut[i].grad.detach().clone() for i in head_features}
        torch.autograd.backward([feats[i] for i in head_features], [cut[i].grad for i in head_features])
    else:
        out = ref(x)
        loss = oloss.compute_loss(out, tgt)
        loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in ref.named_parameters()}
    bufs = {n: b.detach().clone() for n, b in ref.named_buffers()}
    return out.detach(), loss.item(), grads, bufs, fgr


def cosine(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return (a @ b / (a.norm() * b.norm() + 1e-300)).item()


def _compare_param_grads(tag, named_grads, ref_grads, ulp_grads):
    """per tensor: relative L2 from the oracle <= max(1e-3, 3 x the oracle's OWN relative-L2 move under the one-ulp image
    perturbation), cosine >= 0.999.  A tensor's own move is ONE draw of a rare-event process (a handful of flipped mask
    elements, or none: HRNet's fusion 1x1 layers moved 2e-5 in the oracle's draw and 7e-3 in the HIP path's, tensors next to
    them the other way round), so the yardstick of a tensor is the larger of its own draw and the MEDIAN draw over all
    tensors of the model.  Prints how many tensors hold the plain 1e-3 (L2 / max-norm) and the worst ratio."""
    gmax = max(v.abs().max().item() for v in ref_grads.values())
    n = n_l2 = n_mx = 0
    worst, worst_ratio, over = (0.0, 0.0, None), (0.0, None), []
    live = [name for name, _ in named_grads if ref_grads[name].abs().max().item() >= 1e-6 * gmax]
    median_own = float(np.median([l2rel(ulp_grads[name], ref_grads[name]) for name in live]))
    for name, g in named_grads:
        r = ref_grads[name]
        if r.abs().max().item() < 1e-6 * gmax:
            continue      # exactly zero in exact arithmetic (a BatchNorm bias in front of conv + BatchNorm): rounding noise on both sides
        n += 1
        e2, em, own = l2rel(g, r), rel(g, r), max(median_own, l2rel(ulp_grads[name], r))
        n_l2 += e2 < TOL
        n_mx += em < TOL
        if e2 > worst[0]:
            worst = (e2, own, name)
        if e2 > TOL and e2 / own > worst_ratio[0]:
            worst_ratio = (e2 / own, name)
        if not (e2 <= max(TOL, 3.0 * own) and cosine(g, r) >= 0.999):
            over.append((name, e2, own, cosine(g, r)))
    print('%s: %d parameter gradients; worst relative L2 from the oracle %.2e (oracle under a one-ulp image move: %.2e) at %s; '
          'worst distance / yardstick %.2f (%s; median own move %.2e); plain 1e-3 held by %d (L2) / %d (max-norm) tensors'
          % (tag, n, worst[0], worst[1], worst[2], worst_ratio[0], worst_ratio[1], median_own, n_l2, n_mx))
    assert not over, over[:8]


def _grab_reduced_grads(tr):
    """the Trainer's gradient arena x grad_scale at the moment the fused optimiser would consume it"""
    box = {}
    orig = tr.optimizer.step

    def grab(grad_scale=1.0, mp_state=None):
        box['scale'] = grad_scale
        box['grads'] = tr.arena.grads.clone()
        # (no optimiser update: the parameters stay the case's)
    tr.optimizer.step = grab
    return box, orig


def test_config2_deeplab_512_batch16_train_step_vs_oracle(fp32_policy):
    """BASELINE.json configs[2] at full size, train mode, against the CPU oracle -- through the public model API (bridge), through
    the explicit Act-level passes (for the gradients that reach the backbone's stride-4 / stride-16 features) and through the
    Trainer step that bench.py times (fused low-resolution loss, early filter transposes, forked weight gradients)."""
    from pytorch_segmentation_amd import ops, prepare
    from pytorch_segmentation_amd.models import DeepLabV3Plus
    from pytorch_segmentation_amd.nn import Env
    from pytorch_segmentation_amd.utils import Trainer, compute_loss, predict_mask
    B, S, NC = 16, 512, 21
    ref = omodels.DeepLabV3Plus(NC)
    fill.fill_module_(ref, 'cfg2t')
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    x = fill.images('cfg2t/x', (B, 3, S, S))
    tgt = fill.labels('cfg2t/t', (B, S, S), NC, block=16)
    out_ref, loss_ref, g_ref, b_ref, f_ref = _oracle_step(ref, state, x, tgt, head_features=(1, 4))
    out_u, _, g_u, _, f_u = _oracle_step(ref, state, x * (1 + ULP), tgt, head_features=(1, 4))     # the yardstick
    print('configs[2] oracle under a one-ulp image move: logits %.2e, stride-4 / stride-16 feature gradients %.2e / %.2e (rel. L2)'
          % (rel(out_u, out_ref), l2rel(f_u[1], f_ref[1]), l2rel(f_u[4], f_ref[4])))

    m = DeepLabV3Plus(NC)
    m.load_state_dict(state)
    prepare(m, 'cuda')
    m.train()
    xg, tg = x.cuda(), tgt.cuda()

    # (1) the reference's own idiom: outputs = model(inputs); loss = compute_loss(...); loss.backward()
    m._pseg_arena.zero_grad()
    out = m(xg)
    loss = compute_loss(out, tg, m)
    loss.backward()
    torch.cuda.synchronize()
    e_out = rel(out, out_ref)
    e_loss = abs(loss.item() - loss_ref) / abs(loss_ref)
    print('configs[2] train step vs oracle: logits %.2e, loss %.2e (%.6f vs %.6f)' % (e_out, e_loss, loss.item(), loss_ref))
    assert e_out < TOL and e_loss < 1e-4
    _compare_param_grads('configs[2] bridge path', [(n, p.grad) for n, p in m.named_parameters()], g_ref, g_u)
    msd = m.state_dict()
    worst_buf = max((rel(msd[n].float(), q.float()), n) for n, q in b_ref.items())
    print('configs[2] BatchNorm running statistics after the step: worst %.2e (%s)' % worst_buf)
    assert worst_buf[0] < TOL
    top2 = out_ref.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > TOL * out_ref.abs().max()
    assert safe.float().mean().item() > 0.5
    assert torch.equal(predict_mask(out).cpu()[safe], oloss.predict_mask(out_ref)[safe])

    # (2) the gradients that reach the backbone features (stride 4: [16,256,128,128]; stride 16: [16,2048,32,32])
    m.load_state_dict(state)
    env = Env(save=True, accumulate=False)
    m._pseg_arena.transpose_filters()
    env.wT_fresh = True
    out2, (s_bb, s_head) = m.model_fwd(xg, env)
    _, dl = ops.ce_fwd_bwd(out2, tg)
    dlow, dhigh = m.head_bwd(dl, s_head, env)
    torch.cuda.synchronize()
    e_low, e_high = l2rel(dlow.to_nchw(256), f_ref[1]), l2rel(dhigh.to_nchw(2048), f_ref[4])
    o_low, o_high = l2rel(f_u[1], f_ref[1]), l2rel(f_u[4], f_ref[4])
    print('configs[2] head -> feature gradients vs oracle (rel. L2): stride-4 %.2e (oracle own %.2e), stride-16 %.2e (oracle own %.2e)'
          % (e_low, o_low, e_high, o_high))
    assert e_low <= max(TOL, 3 * o_low) and e_high <= max(TOL, 3 * o_high)
    del s_bb, s_head, dlow, dhigh, out2, dl

    # (3) the step bench.py times: Trainer.train_batch (explicit passes, loss taken from the stride-4 logits, weight gradients
    # forked onto the auxiliary stream)
    m.load_state_dict(state)
    tr = Trainer(m, None, loss_fn=compute_loss, lr=1e-3, graph=False)
    box, orig = _grab_reduced_grads(tr)
    lt = tr.train_batch(xg, tg)
    torch.cuda.synchronize()
    tr.optimizer.step = orig
    assert abs(lt.item() - loss_ref) < 1e-4 * abs(loss_ref), (lt.item(), loss_ref)
    saved = tr.arena.grads.clone()
    tr.arena.grads.copy_(box['grads'] * box['scale'])
    _compare_param_grads('configs[2] Trainer step', [(n, p.grad) for n, p in m.named_parameters()], g_ref, g_u)
    tr.arena.grads.copy_(saved)
    msd = m.state_dict()
    assert max(rel(msd[n].float(), q.float()) for n, q in b_ref.items()) < TOL
    tr.close()


def test_config4_hrnet_512_batch8_train_step_vs_oracle(fp32_policy):
    """BASELINE.json configs[4] at full size, train mode, against the CPU oracle (reference models/hrnet.py:373-406): fp32 policy at
    the plain contract; then the SAME step as `train.py -mp` runs it (Trainer(mixed_precision=True): fp16 storage, loss scaling)
    against the same fp32 oracle at the half policy's stated tolerance (loss 5e-3)."""
    from pytorch_segmentation_amd import prepare
    from pytorch_segmentation_amd.models import HRNet
    from pytorch_segmentation_amd.utils import Trainer, compute_loss, predict_mask
    B, S, NC = 8, 512, 21
    ref = omodels.HRNet(NC)
    fill.fill_module_(ref, 'cfg4t')
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    x = fill.images('cfg4t/x', (B, 3, S, S))
    tgt = fill.labels('cfg4t/t', (B, S, S), NC, block=16)
    out_ref, loss_ref, g_ref, b_ref, _ = _oracle_step(ref, state, x, tgt)
    out_u, _, g_u, _, _ = _oracle_step(ref, state, x * (1 + ULP), tgt)                              # the yardstick
    print('configs[4] oracle under a one-ulp image move: logits %.2e' % rel(out_u, out_ref))

    m = HRNet(NC)
    m.load_state_dict(state)
    prepare(m, 'cuda')
    m.train()
    xg, tg = x.cuda(), tgt.cuda()
    m._pseg_arena.zero_grad()
    out = m(xg)
    loss = compute_loss(out, tg, m)
    loss.backward()
    torch.cuda.synchronize()
    e_out = rel(out, out_ref)
    e_loss = abs(loss.item() - loss_ref) / abs(loss_ref)
    print('configs[4] train step vs oracle: logits %.2e, loss %.2e (%.6f vs %.6f)' % (e_out, e_loss, loss.item(), loss_ref))
    assert e_out < TOL and e_loss < 1e-4
    _compare_param_grads('configs[4] bridge path', [(n, p.grad) for n, p in m.named_parameters()], g_ref, g_u)
    msd = m.state_dict()
    worst_buf = max((rel(msd[n].float(), q.float()), n) for n, q in b_ref.items())
    print('configs[4] BatchNorm running statistics after the step: worst %.2e (%s)' % worst_buf)
    assert worst_buf[0] < TOL
    top2 = out_ref.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > TOL * out_ref.abs().max()
    assert safe.float().mean().item() > 0.5
    assert torch.equal(predict_mask(out).cpu()[safe], oloss.predict_mask(out_ref)[safe])

    # the Trainer step (fp32): explicit passes
    m.load_state_dict(state)
    tr = Trainer(m, None, loss_fn=compute_loss, lr=1e-3, graph=False)
    box, orig = _grab_reduced_grads(tr)
    lt = tr.train_batch(xg, tg)
    torch.cuda.synchronize()
    tr.optimizer.step = orig
    assert abs(lt.item() - loss_ref) < 1e-4 * abs(loss_ref)
    saved = tr.arena.grads.clone()
    tr.arena.grads.copy_(box['grads'] * box['scale'])
    _compare_param_grads('configs[4] Trainer step', [(n, p.grad) for n, p in m.named_parameters()], g_ref, g_u)
    tr.arena.grads.copy_(saved)
    tr.close()

    # the -mp leg: same parameters, same batch, fp16 storage + loss scaling; the scaled gradient arena x grad_scale / S against
    # the fp32 oracle in relative L2 over the WHOLE arena (per-tensor fp16 noise is bounded by the half policy's own tests)
    m.load_state_dict(state)
    trh = Trainer(m, None, loss_fn=compute_loss, lr=1e-3, graph=False, mixed_precision=True)
    assert trh.env.half
    boxh, origh = _grab_reduced_grads(trh)
    lh = trh.train_batch(xg, tg)
    torch.cuda.synchronize()
    trh.optimizer.step = origh
    S_loss = trh.loss_scale_state()['scale']
    assert abs(lh.item() - loss_ref) < 5e-3 * abs(loss_ref), (lh.item(), loss_ref)
    saved = trh.arena.grads.clone()
    trh.arena.grads.copy_(boxh['grads'] * (boxh['scale'] / S_loss))
    # yardstick of this leg: the fp32 oracle on the image ROUNDED TO fp16 ONCE (the -mp path rounds every activation of ~110 layers)
    _, loss_h, g_h, _, _ = _oracle_step(ref, state, x.half().float(), tgt)
    num = den = own = dot = nrm = 0.0
    for n, p in m.named_parameters():
        a, r = p.grad.detach().double().cpu(), g_ref[n].double()
        num += (a - r).pow(2).sum().item()
        den += r.pow(2).sum().item()
        own += (g_h[n].double() - r).pow(2).sum().item()
        dot += (a * r).sum().item()
        nrm += a.pow(2).sum().item()
    e_half, e_own, cos = (num / den) ** 0.5, (own / den) ** 0.5, dot / (nrm * den) ** 0.5
    print('configs[4] -mp Trainer step vs fp32 oracle: loss %.2e, gradient arena relative L2 %.2e, cosine %.4f '
          '(the fp32 oracle on the fp16-rounded image moves its own gradient by %.2e)'
          % (abs(lh.item() - loss_ref) / abs(loss_ref), e_half, cos, e_own))
    # the gradient of this random-init network amplifies a 1e-7 perturbation to 1e-2 (above): at fp16's 5e-4 the direction
    # is what survives.  Per-call correctness of the -mp step is tests/test_half_models_gpu.py's (every call vs fp64).
    assert np.isfinite(e_half) and cos > 0.9 and e_half < max(0.5, 3 * e_own)
    trh.arena.grads.copy_(saved)
    trh.close()


@pytest.mark.parametrize('name', ['deeplabv3plus', 'hrnet'])
def test_fullsize_step_every_call_strict(fp32_policy, name):
    """The strict form of full-size parity: EVERY kernel call of one real training step of configs[2] (DeepLabV3+ 512x512 B=16:
    ~570 calls) / configs[4] (HRNet 512x512 B=8: ~1000 calls) -- the headline's own shapes, grids, plan tiles, pixel strides, concat
    slices and accumulate flags -- recomputed on the CPU in fp64 from the call's own device inputs (tests/opcheck.py) and held to
    1e-4 in max-norm, no outlier allowance.  Independent of the graph's conditioning (each call is judged on its actual inputs),
    so this is where "the train-mode step at the benchmark size equals the reference's arithmetic" is decided; the composition
    (which tensor feeds which call) is pinned by the whole-model tests at 128x128 and by the forward quantities above."""
    import time
    from opcheck import OpCheck
    from pytorch_segmentation_amd import models
    from pytorch_segmentation_amd.utils import compute_loss
    hip_cls, nc, S, B, key = {'deeplabv3plus': (models.DeepLabV3Plus, 21, 512, 16, 'cfg2t'),
                              'hrnet': (models.HRNet, 21, 512, 8, 'cfg4t')}[name]
    ref = {'deeplabv3plus': omodels.DeepLabV3Plus, 'hrnet': omodels.HRNet}[name](nc)
    fill.fill_module_(ref, key)
    m = hip_cls(nc)
    m.load_state_dict(ref.state_dict())
    del ref
    m.cuda().train()
    x = fill.images(key + '/x', (B, 3, S, S)).cuda()
    tgt = fill.labels(key + '/t', (B, S, S), nc, block=16).cuda()
    t0 = time.time()
    import os
    with OpCheck(verbose=os.environ.get('PSEG_OPCHECK_VERBOSE', '0') == '1') as oc:
        out = m(x)
        loss = compute_loss(out, tgt, m)
        loss.backward()
        torch.cuda.synchronize()
    kinds = {}
    for op, err, info in oc.calls:
        k = kinds.setdefault(op, [0, 0.0])
        k[0] += 1
        k[1] = max(k[1], err)
    print('full-size every-call check [%s %dx%d B=%d]: %d calls in %.0f s; worst per op: %s'
          % (name, S, S, B, len(oc.calls), time.time() - t0,
             ', '.join('%s x%d %.1e' % (k, v[0], v[1]) for k, v in sorted(kinds.items()))))
    assert len(oc.calls) > 400
    for need in ('conv2d_fwd', 'conv2d_dgrad', 'conv2d_wgrad', 'bn_act_fwd', 'bn_act_bwd.dy', 'bn_finalize', 'ce.dlogits'):
        assert need in kinds, need
    bad = [(op, err, info) for op, err, info in oc.calls if not err < 1e-4]
    assert not bad, bad[:8]

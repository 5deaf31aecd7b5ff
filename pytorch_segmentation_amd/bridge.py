"""Autograd bridge: a whole block / model is ONE torch.autograd.Function.

``model(x)`` returns an ordinary NCHW tensor with a grad_fn, so the reference's training idiom
(``outputs = model(inputs); loss = compute_loss(outputs, targets, model); loss.backward()``, reference
train.py:71-72 via the external Trainer) runs unchanged, while everything between the two calls is explicit
HIP launches with no autograd bookkeeping.

Parameter gradients are written by the kernels straight into the gradient arena (``param.grad`` is a view of it)
with autograd's accumulate semantics: backward ADDS to ``.grad`` unless the caller's Env says overwrite.
"""
import torch

from . import arena as _arena
from . import ops as _ops
from .nn import Env
from .ops import Act, _round4


def ensure_prepared(module, device):
    """Arena-back `module` on first use (or after it was moved)."""
    ar = getattr(module, '_pseg_arena', None)
    if ar is not None and ar.device == device:
        return ar
    for m in module.modules():
        if any(p is not None for p in m._parameters.values()) and not hasattr(m, '_raw'):
            return _arena.prepare(module, device)
        for p in m._parameters.values():
            if p is not None and p.device != device:
                return _arena.prepare(module, device)
    return ar


def _anchor(module, device):
    a = getattr(module, '_pseg_anchor', None)
    if a is None or a.device != device:
        a = torch.zeros((), device=device, requires_grad=True)
        object.__setattr__(module, '_pseg_anchor', a)
    return a


def _env_for(module, grad):
    env = getattr(module, '_pseg_env', None)
    if env is not None:
        env.save = grad
        return env
    return Env(save=grad, accumulate=True, overlap_wgrad=True)


def _cpad(c, env):
    return (c + 7) // 8 * 8 if env.half else _round4(c)


def _prep_half(module, env, transposed):
    """half-precision policy: refresh the fp16 filter copies of the arena that backs `module` (one launch)"""
    if env.half and not env.half_fresh:
        ar = getattr(module, '_pseg_arena', None)
        if ar is None:
            raise RuntimeError('half-precision pass on a module without a parameter arena')
        ar.prepare_half(transposed=transposed)


def _fix_none_grads(module):
    """Autograd semantics when the caller did ``zero_grad(set_to_none=True)``: treat missing grads as zeros."""
    ar = getattr(module, '_pseg_arena', None)
    if ar is None:
        return
    if any(s.param.grad is None for s in ar.segments):
        ar.restore_grad_views()
        ar.zero_grad()


class _BlockFn(torch.autograd.Function):
    """Single-input single-output block (Conv2d, ConvNormAct, ASPP, ...) on NCHW tensors."""

    @staticmethod
    def forward(ctx, module, x, anchor):
        env = _env_for(module, True)
        _prep_half(module, env, True)
        xa = Act.from_nchw(x, _cpad(x.shape[1], env), dtype=env.act_dtype)
        y, saved = module.block_fwd(xa, env)
        ctx.module, ctx.saved, ctx.env = module, saved, env
        ctx.need_dx = x.requires_grad
        ctx.cin = x.shape[1]
        return y.to_nchw(module.block_out_channels)

    @staticmethod
    def backward(ctx, gy):
        module = ctx.module
        _fix_none_grads(module)
        dya = Act.from_nchw(gy.contiguous(), _cpad(gy.shape[1], ctx.env), dtype=ctx.env.act_dtype)
        dx = module.block_bwd(dya, ctx.saved, ctx.env, need_dx=ctx.need_dx)
        _ops.join_aux(gy.device)
        ctx.saved = None
        return None, (dx.to_nchw(ctx.cin) if ctx.need_dx else None), None


def run_module(module, x):
    if not x.is_cuda:
        raise RuntimeError('pytorch_segmentation_amd runs on the MI355X HIP path only; got a %s tensor '
                           '(there is no CPU fallback -- the CPU restatement lives in oracle/ for tests)' % x.device)
    if x.dtype != torch.float32:
        raise TypeError('HIP path computes in fp32; got %s' % x.dtype)
    ensure_prepared(module, x.device)
    if torch.is_grad_enabled():
        return _BlockFn.apply(module, x, _anchor(module, x.device))
    env = _env_for(module, False)
    _prep_half(module, env, False)
    y, _ = module.block_fwd(Act.from_nchw(x, _cpad(x.shape[1], env), dtype=env.act_dtype), env)
    return y.to_nchw(module.block_out_channels)


class _BackboneFn(torch.autograd.Function):
    """Encoder with the reference's backbone contract (pytorch_modules.backbones.*): NCHW image in, list of NCHW
    feature maps out -- for callers that keep their own decoder in torch ops and only swap the encoder."""

    @staticmethod
    def forward(ctx, module, x, anchor):
        env = _env_for(module, True)
        _prep_half(module, env, True)
        feats, saved = module.fwd(Act.from_nchw(x, _cpad(4, env), dtype=env.act_dtype), env)
        ctx.module, ctx.saved, ctx.env = module, saved, env
        ctx.set_materialize_grads(False)
        return tuple(f.to_nchw(c) for f, c in zip(feats, module.out_channels))

    @staticmethod
    def backward(ctx, *grads):
        module = ctx.module
        _fix_none_grads(module)
        env = ctx.env
        dfeats = [Act.from_nchw(g.contiguous(), _cpad(g.shape[1], env), dtype=env.act_dtype) if g is not None else None
                  for g in grads]
        if any(d is not None for d in dfeats):
            module.bwd(dfeats, ctx.saved, ctx.env)
            _ops.join_aux(next(d for d in dfeats if d is not None).device)
        ctx.saved = None
        return None, None, None


def run_backbone(module, x):
    if not x.is_cuda:
        raise RuntimeError('pytorch_segmentation_amd runs on the MI355X HIP path only; got a %s tensor '
                           '(there is no CPU fallback -- the CPU restatement lives in oracle/ for tests)' % x.device)
    if x.dtype != torch.float32:
        raise TypeError('HIP path computes in fp32; got %s' % x.dtype)
    ensure_prepared(module, x.device)
    if torch.is_grad_enabled():
        return list(_BackboneFn.apply(module, x, _anchor(module, x.device)))
    env = _env_for(module, False)
    _prep_half(module, env, False)
    feats, _ = module.fwd(Act.from_nchw(x, _cpad(4, env), dtype=env.act_dtype), env)
    return [f.to_nchw(c) for f, c in zip(feats, module.out_channels)]


class _ModelFn(torch.autograd.Function):
    """Top-level segmentation model: NCHW image in, NCHW logits out (written NCHW by the last upsample kernel)."""

    @staticmethod
    def forward(ctx, model, x, anchor):
        env = _env_for(model, True)
        _prep_half(model, env, True)
        out, saved = model.model_fwd(x, env)
        ctx.model, ctx.saved, ctx.env = model, saved, env
        ctx.mark_non_differentiable()
        return out

    @staticmethod
    def backward(ctx, gout):
        model = ctx.model
        _fix_none_grads(model)
        ar = getattr(model, '_pseg_arena', None)
        if ar is not None and not ctx.env.half:     # (half policy: the fp16 transposed copies were made in forward)
            ar.transpose_filters()
            ctx.env.wT_fresh = True
        model.model_bwd(gout.contiguous(), ctx.saved, ctx.env)
        ctx.env.wT_fresh = False
        _ops.join_aux(gout.device)
        ctx.saved = None
        return None, None, None


def run_model(model, x):
    if not x.is_cuda:
        raise RuntimeError('pytorch_segmentation_amd runs on the MI355X HIP path only; got a %s tensor '
                           '(there is no CPU fallback -- the CPU restatement lives in oracle/ for tests)' % x.device)
    if x.dtype != torch.float32:
        raise TypeError('HIP path computes in fp32; got %s' % x.dtype)
    ensure_prepared(model, x.device)
    if torch.is_grad_enabled():
        return _ModelFn.apply(model, x, _anchor(model, x.device))
    env = _env_for(model, False)
    _prep_half(model, env, False)
    out, _ = model.model_fwd(x, env)
    return out

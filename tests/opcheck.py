"""Teacher-forced per-call checker (test infrastructure).

Inside ``with OpCheck() as oc:`` every ``pytorch_segmentation_amd.ops`` call of a real model step -- with the real
shapes, pixel strides, concat slices, accumulate flags and precision policy of that model -- is recomputed on the CPU
in fp64 *from the call's own device inputs* and compared in max-norm.  Because each call is judged on its actual
inputs, the check is independent of how ill-conditioned the whole graph is: a ReLU mask that flips between two fp32
implementations of a 50-layer network (which moves late whole-model gradients by percents in ANY fp32 implementation,
the CPU reference included -- see DESIGN.md section 4) cannot hide or fake an error here, while a 1 % systematic error
in any one mid-network data / weight gradient is a 100x violation.  Composition (which tensor feeds which call) is
what the whole-model tests check; together they pin the backward pass.

``oc.calls`` = list of (op name, max-norm relative error, description).
"""
import torch
import torch.nn.functional as F

from pytorch_segmentation_amd import ops


def nchw(a, C=None):
    """Act -> cpu fp64 NCHW"""
    v = a.view4().detach().cpu().double().permute(0, 3, 1, 2).contiguous()
    return v if C is None else v[:, :C]


def rel(got, ref):
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-30)).item()


def _vec(t):
    return t.detach().cpu().double().view(1, -1, 1, 1)


def _w_oihw(w_raw, Cout, kh, kw, Cin):
    return w_raw.detach().cpu().double().view(Cout, kh, kw, Cin).permute(0, 3, 1, 2).contiguous()


def _act(t, act):
    return F.relu(t) if act == 1 else (F.relu6(t) if act == 2 else t)


def _mask(g, zz, act):
    if act == 1:
        return g * (zz > 0)
    if act == 2:
        return g * ((zz > 0) & (zz < 6))
    return g


class OpCheck:
    def __init__(self, verbose=False):
        self.calls = []
        self.verbose = verbose
        self._orig = {}
        self._batch_co = {}     # id(co tensor) -> (eps) for coefficient sets produced by bn_finalize (batch statistics)

    def report(self, name, err, info=''):
        self.calls.append((name, err, info))
        if self.verbose:
            print('%-18s err %.2e  %s' % (name, err, info), flush=True)

    def worst(self, prefix=''):
        sel = [c for c in self.calls if c[0].startswith(prefix)]
        return max(sel, key=lambda c: c[1]) if sel else None

    # ------------------------------------------------------------------------------------------ wrappers
    def __enter__(self):
        o = self._orig
        rep = self.report

        def wrap(name, fn):
            o[name] = getattr(ops, name)
            setattr(ops, name, fn)

        def conv2d_fwd(x, w_raw, bias_raw, y, kh, kw, stride, pad, dil, accumulate=False, want_stats=False, **kx):
            prev = nchw(y) if accumulate else None
            xin = nchw(x)
            r = o['conv2d_fwd'](x, w_raw, bias_raw, y, kh, kw, stride, pad, dil, accumulate=accumulate,
                                want_stats=want_stats, **kx)
            ref = F.conv2d(xin, _w_oihw(w_raw, y.C, kh, kw, x.C),
                           bias_raw.detach().cpu().double() if bias_raw is not None else None, stride, pad, dil)
            if accumulate:
                ref = ref + prev
            rep('conv2d_fwd', rel(nchw(y), ref), 'x%s -> y%s k%d s%d p%d d%d ldx%d ldy%d'
                % ((x.B, x.C, x.H, x.W), (y.B, y.C, y.H, y.W), kh, stride, pad, dil, x.ld, y.ld))
            if want_stats and r is not None:
                st, rows, group = r
                st = st.double().cpu()
                # group g covers rows [g*group, min(M, (g+1)*group)) -- possibly none (tiny maps under a big tile)
                cnt = (y.M - group * torch.arange(rows, dtype=torch.float64)).clamp(min=0.0, max=float(group))
                K, S1, S2 = st[0], st[1], st[2]
                colsum = (S1 + K * cnt[:, None]).sum(0)
                colsq = (S2 + 2 * K * S1 + K * K * cnt[:, None]).sum(0)
                # fp16 results: the statistics are those of the values AS STORED (rounded to fp16) -- what the layer
                # normalises and what its backward pass reads -- so that is what they are checked against
                sref = nchw(y).double() if getattr(y, 'half', False) else ref
                rep('conv2d_fwd.stats', max(rel(colsum, sref.sum((0, 2, 3))), rel(colsq, (sref * sref).sum((0, 2, 3)))),
                    'rows %d' % rows)
            return r

        def conv2d_dgrad(dy, wT_raw, dx, kh, kw, stride, pad, dil, accumulate=False, **kx):
            prev = nchw(dx) if accumulate else None
            g = nchw(dy)
            o['conv2d_dgrad'](dy, wT_raw, dx, kh, kw, stride, pad, dil, accumulate=accumulate, **kx)
            Cout, Cin = dy.C, dx.C
            w = wT_raw.detach().cpu().double().view(Cin, kh * kw, Cout).permute(2, 0, 1).reshape(Cout, Cin, kh, kw)
            ref = torch.nn.grad.conv2d_input((dx.B, Cin, dx.H, dx.W), w, g, stride, pad, dil)
            if accumulate:
                ref = ref + prev
            rep('conv2d_dgrad', rel(nchw(dx), ref), 'dy%s -> dx%s k%d s%d p%d d%d acc%d'
                % ((dy.B, dy.C, dy.H, dy.W), (dx.B, dx.C, dx.H, dx.W), kh, stride, pad, dil, accumulate))

        def conv2d_dgrad_planes(dy_planes, dy, wT_planes, dx, kh, kw, stride, pad, dil, accumulate=False):
            prev = nchw(dx) if accumulate else None
            g = nchw(dy)
            # the planes handed over are the limbs of dy itself (bn_act_bwd(want_planes=True)): bit-exact
            want = ops.split_planes(dy)
            same = torch.equal(want.hi, dy_planes.hi) and torch.equal(want.lo, dy_planes.lo)
            rep('dgrad_planes.limbs', 0.0 if same else 1.0)
            o['conv2d_dgrad_planes'](dy_planes, dy, wT_planes, dx, kh, kw, stride, pad, dil, accumulate=accumulate)
            Cout, Cin = dy.C, dx.C

            def f32(p):
                return (p.to(torch.int32) << 16).view(torch.float32).cpu().double()
            w = (f32(wT_planes.hi) + f32(wT_planes.lo)).view(Cin, -1)[:, :kh * kw * Cout]
            w = w.reshape(Cin, kh * kw, Cout).permute(2, 0, 1).reshape(Cout, Cin, kh, kw)
            ref = torch.nn.grad.conv2d_input((dx.B, Cin, dx.H, dx.W), w, g, stride, pad, dil)
            if accumulate:
                ref = ref + prev
            rep('conv2d_dgrad', rel(nchw(dx), ref), 'planes dy%s -> dx%s k%d s%d p%d d%d acc%d'
                % ((dy.B, dy.C, dy.H, dy.W), (dx.B, dx.C, dx.H, dx.W), kh, stride, pad, dil, accumulate))

        def conv2d_wgrad(x, dy, dw_raw, kh, kw, stride, pad, dil, accumulate=False, **kx):
            prev = dw_raw.detach().cpu().double().clone().reshape(-1) if accumulate else None
            xin, g = nchw(x), nchw(dy)
            o['conv2d_wgrad'](x, dy, dw_raw, kh, kw, stride, pad, dil, accumulate=accumulate, **kx)
            ref = torch.nn.grad.conv2d_weight(xin, (dy.C, x.C, kh, kw), g, stride, pad, dil).permute(0, 2, 3, 1).reshape(-1)
            if accumulate:
                ref = ref + prev
            rep('conv2d_wgrad', rel(dw_raw.detach().cpu().double().reshape(-1), ref), 'x%s dy%s k%d s%d p%d d%d acc%d'
                % ((x.B, x.C, x.H, x.W), (dy.B, dy.C, dy.H, dy.W), kh, stride, pad, dil, accumulate))

        def bn_finalize(stats, count, gamma, beta, running_mean, running_var, momentum, eps):
            co = o['bn_finalize'](stats, count, gamma, beta, running_mean, running_var, momentum, eps)
            self._batch_co[id(co)] = (co, eps, gamma, beta)
            return co

        def bn_act_fwd(y, co, act, z, residual=None, want_mask=False):
            yin = nchw(y)
            rin = nchw(residual) if residual is not None else None
            mask = o['bn_act_fwd'](y, co, act, z, residual=residual, want_mask=want_mask)
            if mask is not None:      # the activation bitmask must be exactly act'(z) of the z just written
                zz = z.view4()[..., :y.C].reshape(y.M, y.C)
                on = (zz > 0) if act == 1 else ((zz > 0) & (zz < 6))
                w = (on.view(y.M, y.C // 32, 32).to(torch.int64) << torch.arange(32, device=on.device)).sum(-1)
                w = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)
                rep('bn_act_fwd.mask', float((w.reshape(-1) != mask).sum().item()), 'y%s' % ((y.B, y.C, y.H, y.W),))
                self.bn_masks = getattr(self, 'bn_masks', 0) + 1
            t = yin
            if co is not None:
                if id(co) in self._batch_co:      # batch statistics: mean / invstd / scale must be those of THIS y
                    _, eps, gamma, beta = self._batch_co[id(co)]
                    mu = yin.mean((0, 2, 3))
                    var = yin.var((0, 2, 3), unbiased=False)
                    is_ = 1.0 / (var + eps).sqrt()
                    gm = gamma.detach().cpu().double() if gamma is not None else torch.ones_like(mu)
                    e_mu = ((co[0].detach().cpu().double() - mu).abs().max() / (mu.abs().max() + var.sqrt().max() + 1e-30)).item()
                    rep('bn_finalize', max(e_mu, rel(co[1].detach().cpu().double(), is_),
                                           rel(co[2].detach().cpu().double(), gm * is_)), 'C%d M%d' % (y.C, y.M))
                t = (t - _vec(co[0])) * _vec(co[2]) + _vec(co[3])
            if rin is not None:
                t = t + rin
            rep('bn_act_fwd', rel(nchw(z), _act(t, act)), 'y%s act%d res%d' % ((y.B, y.C, y.H, y.W), act, residual is not None))
            return mask

        def bn_fwd_fused(stats, count, gamma, beta, running_mean, running_var, momentum, eps, y, act, z, residual=None):
            yin = nchw(y)
            rin = nchw(residual) if residual is not None else None
            rm0 = running_mean.detach().cpu().double().clone() if running_mean is not None else None
            rv0 = running_var.detach().cpu().double().clone() if running_var is not None else None
            co = o['bn_fwd_fused'](stats, count, gamma, beta, running_mean, running_var, momentum, eps, y, act, z,
                                   residual=residual)
            mu = yin.mean((0, 2, 3))
            var = yin.var((0, 2, 3), unbiased=False)
            is_ = 1.0 / (var + eps).sqrt()
            gm = gamma.detach().cpu().double() if gamma is not None else torch.ones_like(mu)
            bt = beta.detach().cpu().double() if beta is not None else torch.zeros_like(mu)
            e_mu = ((co[0].detach().cpu().double() - mu).abs().max() / (mu.abs().max() + var.sqrt().max() + 1e-30)).item()
            rep('bn_finalize', max(e_mu, rel(co[1].detach().cpu().double(), is_), rel(co[2].detach().cpu().double(), gm * is_)),
                'fused C%d M%d' % (y.C, y.M))
            if rm0 is not None:
                n = float(y.M)
                rep('bn_running_stats', max(rel(running_mean.detach().cpu().double(), (1 - momentum) * rm0 + momentum * mu),
                                            rel(running_var.detach().cpu().double(), (1 - momentum) * rv0 + momentum * var * n / max(n - 1, 1))))
            t = (yin - mu.view(1, -1, 1, 1)) * (gm * is_).view(1, -1, 1, 1) + bt.view(1, -1, 1, 1)
            if rin is not None:
                t = t + rin
            rep('bn_act_fwd', rel(nchw(z), _act(t, act)), 'fused y%s act%d res%d' % ((y.B, y.C, y.H, y.W), act, residual is not None))
            return co

        def bn_act_bwd(dz, z, y, co, act, dy, gamma_grad, beta_grad, accumulate=False, dres=None, res_accumulate=False,
                       frozen=False, mask=None, want_planes=False, part=None):
            g, yy = nchw(dz), nchw(y)
            zz = nchw(z) if z is not None else (yy - _vec(co[0])) * _vec(co[2]) + _vec(co[3])
            pg = gamma_grad.detach().cpu().double().clone() if gamma_grad is not None else None
            pb = beta_grad.detach().cpu().double().clone() if beta_grad is not None else None
            pres = nchw(dres) if (dres is not None and res_accumulate) else None
            o['bn_act_bwd'](dz, z, y, co, act, dy, gamma_grad, beta_grad, accumulate=accumulate, dres=dres,
                            res_accumulate=res_accumulate, frozen=frozen, mask=mask, want_planes=want_planes, part=part)
            g = _mask(g, zz, act)
            xh = (yy - _vec(co[0])) * _vec(co[1])
            M = y.M
            db = g.sum((0, 2, 3))
            dg = (g * xh).sum((0, 2, 3))
            # yardstick of a column sum: the largest sum of |terms| (a BatchNorm bias that feeds conv + BatchNorm has
            # dbeta == 0 in exact arithmetic: sum / max|sum| would compare rounding noise with rounding noise)
            db_scale = g.abs().sum((0, 2, 3)).max().item() + 1e-30
            dg_scale = (g * xh).abs().sum((0, 2, 3)).max().item() + 1e-30
            ref = _vec(co[2]) * (g if frozen else (g - db.view(1, -1, 1, 1) / M - xh * dg.view(1, -1, 1, 1) / M))
            rep('bn_act_bwd.dy', rel(nchw(dy), ref), 'y%s act%d frozen%d' % ((y.B, y.C, y.H, y.W), act, frozen))
            if gamma_grad is not None:
                rep('bn_act_bwd.dgamma', (gamma_grad.detach().cpu().double() - dg - (pg if accumulate else 0)).abs().max().item()
                    / (dg_scale + (pg.abs().max().item() if accumulate else 0.0)))
                rep('bn_act_bwd.dbeta', (beta_grad.detach().cpu().double() - db - (pb if accumulate else 0)).abs().max().item()
                    / (db_scale + (pb.abs().max().item() if accumulate else 0.0)))
            if dres is not None:
                rep('bn_act_bwd.dres', rel(nchw(dres), g + (pres if pres is not None else 0)))

        def act_bwd(dz, z, act, dy, scale=None, dres=None, res_accumulate=False):
            g = nchw(dz)
            zz = nchw(z) if z is not None else None
            pres = nchw(dres) if (dres is not None and res_accumulate) else None
            o['act_bwd'](dz, z, act, dy, scale=scale, dres=dres, res_accumulate=res_accumulate)
            g = _mask(g, zz, act) if act else g
            if dy is not None:
                rep('act_bwd.dy', rel(nchw(dy), g * _vec(scale) if scale is not None else g))
            if dres is not None:
                rep('act_bwd.dres', rel(nchw(dres), g + (pres if pres is not None else 0)))

        def bilinear_fwd(x, y, align_corners):
            xin = nchw(x)
            o['bilinear_fwd'](x, y, align_corners)
            ref = F.interpolate(xin, size=(y.H, y.W), mode='bilinear', align_corners=bool(align_corners))
            rep('bilinear_fwd', rel(nchw(y), ref), 'x%s -> %dx%d' % ((x.B, x.C, x.H, x.W), y.H, y.W))

        def bilinear_fwd_nchw(x, C, Ho, Wo, align_corners):
            xin = nchw(x, C)
            out = o['bilinear_fwd_nchw'](x, C, Ho, Wo, align_corners)
            ref = F.interpolate(xin, size=(Ho, Wo), mode='bilinear', align_corners=bool(align_corners))
            rep('bilinear_fwd_nchw', rel(out.detach().cpu().double(), ref), 'x%s -> %dx%d' % ((x.B, C, x.H, x.W), Ho, Wo))
            return out

        def _bil_grad(g, shape, align_corners):
            with torch.enable_grad():
                xin = torch.zeros(*shape, dtype=torch.float64, requires_grad=True)
                F.interpolate(xin, size=tuple(g.shape[2:]), mode='bilinear', align_corners=bool(align_corners)).backward(g)
            return xin.grad

        def bilinear_bwd(dy, dx, align_corners, accumulate=False):
            g = nchw(dy)
            prev = nchw(dx) if accumulate else 0
            o['bilinear_bwd'](dy, dx, align_corners, accumulate=accumulate)
            rep('bilinear_bwd', rel(nchw(dx), _bil_grad(g, (dx.B, dx.C, dx.H, dx.W), align_corners) + prev),
                'dy%s acc%d' % ((dy.B, dy.C, dy.H, dy.W), accumulate))

        def bilinear_bwd_nchw(dy_nchw, dx, C, align_corners, accumulate=False):
            g = dy_nchw.detach().cpu().double()
            prev = nchw(dx, C) if accumulate else 0
            o['bilinear_bwd_nchw'](dy_nchw, dx, C, align_corners, accumulate=accumulate)
            rep('bilinear_bwd_nchw', rel(nchw(dx, C), _bil_grad(g, (dx.B, C, dx.H, dx.W), align_corners) + prev),
                'dy%s' % (tuple(dy_nchw.shape),))

        def copy2d(x, y, accumulate=False):
            prev = nchw(y) if accumulate else 0
            xin = nchw(x)
            o['copy2d'](x, y, accumulate=accumulate)
            rep('copy2d', rel(nchw(y), xin + prev), 'x%s acc%d' % ((x.B, x.C, x.H, x.W), accumulate))

        def pool_sum(x, out, scale):
            xin = nchw(x)
            o['pool_sum'](x, out, scale)
            rep('pool_sum', rel(nchw(out), scale * xin.sum((2, 3), keepdim=True)), 'x%s' % ((x.B, x.C, x.H, x.W),))

        def broadcast(x, y, scale=1.0, accumulate=False):
            prev = nchw(y) if accumulate else 0
            xin = nchw(x)
            o['broadcast'](x, y, scale=scale, accumulate=accumulate)
            rep('broadcast', rel(nchw(y), scale * xin.expand(-1, -1, y.H, y.W) + prev), 'acc%d' % accumulate)

        def maxpool_fwd(x, y, k, stride, pad, want_argmax=True):
            xin = nchw(x)
            arg = o['maxpool_fwd'](x, y, k, stride, pad, want_argmax=want_argmax)
            rep('maxpool_fwd', rel(nchw(y), F.max_pool2d(xin, k, stride, pad)), 'x%s' % ((x.B, x.C, x.H, x.W),))
            self._pool_in = xin
            return arg

        def bn_act_maxpool_fwd(x, co, act, y, k, stride, pad, want_argmax=True):
            # the ResNet stem without its activated map: BatchNorm (coefficients checked against THIS x) + activation + max-pool
            xin = nchw(x)
            arg = o['bn_act_maxpool_fwd'](x, co, act, y, k, stride, pad, want_argmax=want_argmax)
            if id(co) in self._batch_co:
                _, eps, gamma, beta = self._batch_co[id(co)]
                mu = xin.mean((0, 2, 3))
                var = xin.var((0, 2, 3), unbiased=False)
                is_ = 1.0 / (var + eps).sqrt()
                gm = gamma.detach().cpu().double() if gamma is not None else torch.ones_like(mu)
                e_mu = ((co[0].detach().cpu().double() - mu).abs().max() / (mu.abs().max() + var.sqrt().max() + 1e-30)).item()
                rep('bn_finalize', max(e_mu, rel(co[1].detach().cpu().double(), is_), rel(co[2].detach().cpu().double(), gm * is_)),
                    'C%d M%d' % (x.C, x.M))
            z = _act((xin - _vec(co[0])) * _vec(co[2]) + _vec(co[3]), act)
            if x.half:      # the pooling compares the values as the separate pass would have STORED them (ties after rounding)
                z = z.half().double()
            rep('bn_act_maxpool_fwd', rel(nchw(y), F.max_pool2d(z, k, stride, pad)), 'x%s act%d' % ((x.B, x.C, x.H, x.W), act))
            self._pool_in = z
            return arg

        def maxpool_bwd(dy, arg, dx, k, stride, pad, accumulate=False):
            g = nchw(dy)
            prev = nchw(dx) if accumulate else 0
            o['maxpool_bwd'](dy, arg, dx, k, stride, pad, accumulate=accumulate)
            with torch.enable_grad():
                xin = self._pool_in.clone().requires_grad_()
                F.max_pool2d(xin, k, stride, pad).backward(g)
            rep('maxpool_bwd', rel(nchw(dx), xin.grad + prev), 'dy%s' % ((dy.B, dy.C, dy.H, dy.W),))

        def dwconv_fwd(x, w_raw, y, k, stride, pad):
            xin = nchw(x)
            o['dwconv_fwd'](x, w_raw, y, k, stride, pad)
            w = w_raw.detach().cpu().double().view(k, k, x.C).permute(2, 0, 1).unsqueeze(1)
            rep('dwconv_fwd', rel(nchw(y), F.conv2d(xin, w, None, stride, pad, 1, groups=x.C)), 'x%s s%d' % ((x.B, x.C, x.H, x.W), stride))

        def dwconv_dgrad(dy, w_raw, dx, k, stride, pad):
            g = nchw(dy)
            o['dwconv_dgrad'](dy, w_raw, dx, k, stride, pad)
            w = w_raw.detach().cpu().double().view(k, k, dx.C).permute(2, 0, 1).unsqueeze(1)
            ref = torch.nn.grad.conv2d_input((dx.B, dx.C, dx.H, dx.W), w, g, stride, pad, 1, groups=dx.C)
            rep('dwconv_dgrad', rel(nchw(dx), ref), 'dx%s s%d' % ((dx.B, dx.C, dx.H, dx.W), stride))

        def dwconv_wgrad(x, dy, dw_raw, k, stride, pad, accumulate=False):
            prev = dw_raw.detach().cpu().double().clone() if accumulate else 0
            xin, g = nchw(x), nchw(dy)
            o['dwconv_wgrad'](x, dy, dw_raw, k, stride, pad, accumulate=accumulate)
            ref = torch.nn.grad.conv2d_weight(xin, (x.C, 1, k, k), g, stride, pad, 1, groups=x.C)[:, 0].permute(1, 2, 0) + prev
            rep('dwconv_wgrad', rel(dw_raw.detach().cpu().double().view(k, k, x.C), ref), 'x%s s%d acc%d' % ((x.B, x.C, x.H, x.W), stride, accumulate))

        def col_sum(dy, out, accumulate=False, C=None):
            Cc = dy.C if C is None else C
            prev = out.detach().cpu().double().clone() if accumulate else 0
            g = nchw(dy, Cc)
            o['col_sum'](dy, out, accumulate=accumulate, C=C)
            rep('col_sum', rel(out.detach().cpu().double()[:Cc], (g.sum((0, 2, 3)) + (prev[:Cc] if accumulate else 0))), 'C%d' % Cc)

        def ce_fwd_bwd(logits, target, want_grad=True, ignore_index=-100):
            out, dl = o['ce_fwd_bwd'](logits, target, want_grad=want_grad, ignore_index=ignore_index)
            with torch.enable_grad():
                lg = logits.detach().cpu().double().requires_grad_()
                ref = F.cross_entropy(lg, target.cpu(), ignore_index=ignore_index)
                ref.backward()
            rep('ce.loss', abs(out[0].item() - ref.item()) / abs(ref.item()))
            if dl is not None:
                rep('ce.dlogits', rel(dl.detach().cpu().double(), lg.grad))
            return out, dl

        for name, fn in list(locals().items()):
            if callable(fn) and hasattr(ops, name) and name not in ('wrap',):
                wrap(name, fn)
        return self

    def __exit__(self, *exc):
        for k, v in self._orig.items():
            setattr(ops, k, v)
        self._orig.clear()
        self._batch_co.clear()

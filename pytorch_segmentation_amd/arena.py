"""Flat fp32 parameter / gradient arenas.

Every parameter of a model becomes a (possibly strided) view into ONE flat device buffer, and every ``.grad`` a
view into a second one laid out identically.  That gives:

* kernel-native layouts behind torch-shaped parameters: a conv weight is stored [Cout][kh][kw][Cin] (what the
  implicit-GEMM kernels read) while ``module.weight`` keeps its OIHW shape and its state-dict key; Cin / Cout are
  padded to multiples of 4 inside the arena (3-channel stem, 21-class classifier) and the padding stays zero;
* one fused optimiser launch over the whole model (ops.sgd_step / ops.adam_step on the flat buffers);
* gradient buckets for the data-parallel all-reduce that are plain contiguous ranges of the gradient arena
  (no flatten / unflatten copies), in forward order so that backward completes them from the top down.
"""
import torch


def _round4(n):
    return (n + 3) // 4 * 4


class Segment:
    __slots__ = ('module', 'name', 'param', 'offset', 'numel', 'raw_shape')

    def __init__(self, module, name, param, offset, numel, raw_shape):
        self.module, self.name, self.param = module, name, param
        self.offset, self.numel, self.raw_shape = offset, numel, raw_shape


def _layout(module, name, p):
    """-> (raw_shape, view_fn).  Modules may define ``_pseg_layout(name, param)`` for kernel-native layouts."""
    fn = getattr(module, '_pseg_layout', None)
    if fn is not None:
        lay = fn(name, p)
        if lay is not None:
            return lay
    n = p.numel()
    shape = tuple(p.shape)
    return (_round4(n),), (lambda raw, n=n, shape=shape: raw[:n].view(shape))


class ParamArena:
    def __init__(self, model, device):
        self.device = torch.device(device)
        segs, seen, off = [], set(), 0
        for mod in model.modules():
            for name, p in mod._parameters.items():
                if p is None or id(p) in seen:
                    continue
                seen.add(id(p))
                if p.dtype != torch.float32:
                    raise TypeError('parameter %s is %s; the arena is fp32' % (name, p.dtype))
                raw_shape, view_fn = _layout(mod, name, p)
                numel = 1
                for s in raw_shape:
                    numel *= s
                numel = _round4(numel)
                segs.append((Segment(mod, name, p, off, numel, raw_shape), view_fn))
                off += numel
        self.numel = off
        self.params = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.segments = []
        for seg, view_fn in segs:
            n = 1
            for s in seg.raw_shape:
                n *= s
            raw_p = self.params[seg.offset:seg.offset + n].view(seg.raw_shape)
            raw_g = self.grads[seg.offset:seg.offset + n].view(seg.raw_shape)
            pv, gv = view_fn(raw_p), view_fn(raw_g)
            assert tuple(pv.shape) == tuple(seg.param.shape), (seg.name, pv.shape, seg.param.shape)
            with torch.no_grad():
                pv.copy_(seg.param.data)
            seg.param.data = pv
            seg.param.grad = gv
            mod = seg.module
            if not hasattr(mod, '_raw'):
                mod._raw, mod._raw_grad = {}, {}
            mod._raw[seg.name] = raw_p
            mod._raw_grad[seg.name] = raw_g
            self.segments.append(seg)
        self._grad_views = {id(s.param): s.param.grad for s in self.segments}
        self._wT_jobs = None
        self._amax_jobs = None

    def filter_amax(self):
        """max|w| of every dense conv filter in ONE launch (the fp16-limb forward convs scale their filter operand by it);
        each conv finds its scalar in ``module._wamax_view``.  Call once per forward pass while the weights are fixed."""
        from . import _lib, ops
        if self._amax_jobs is None:
            jobs, mods, blocks = [], [], 0
            for seg in self.segments:
                mod = seg.module
                if seg.name != 'weight' or not getattr(mod, 'kernel_size', None) or getattr(mod, 'depthwise', True):
                    continue
                n = 1
                for d in seg.raw_shape:
                    n *= d
                nb = max(1, min(64, (n + 4095) // 4096))
                jobs.append([self.params.data_ptr() + 4 * seg.offset, n, blocks])
                mods.append(mod)
                blocks += nb
            self.wamax = torch.zeros(max(len(jobs), 1), dtype=torch.float32, device=self.device)
            table = torch.tensor(jobs, dtype=torch.int64, device=self.device) if jobs else None
            self._amax_jobs = (table, len(jobs), blocks)
            for i, mod in enumerate(mods):
                mod._wamax_view = self.wamax[i:i + 1]
        table, n, blocks = self._amax_jobs
        if n:
            _lib.call('pseg_amax_batch', table.data_ptr(), n, blocks, self.wamax.data_ptr(), ops._stream())

    def transpose_filters(self):
        """[Cin][taps][Cout] copies of every dense conv filter (what the data-gradient kernels read), refreshed with ONE
        launch at the start of a backward pass; each conv finds its slice in ``module._wT_view``."""
        from . import _lib, ops
        if self._wT_jobs is None:
            jobs, views, off, tiles = [], [], 0, 0
            for seg in self.segments:
                mod = seg.module
                if seg.name != 'weight' or not getattr(mod, 'kernel_size', None) or getattr(mod, 'depthwise', True):
                    continue
                co, kh, kw, ci = seg.raw_shape
                jobs.append([seg.offset, off, co, kh * kw, ci, tiles])
                views.append((mod, off, co * kh * kw * ci))
                off += co * kh * kw * ci
                tiles += kh * kw * ((co + 31) // 32) * ((ci + 31) // 32)
            self.wT = torch.empty(max(off, 1), dtype=torch.float32, device=self.device)
            base_w, base_t = self.params.data_ptr(), self.wT.data_ptr()
            table = [[base_w + 4 * j[0], base_t + 4 * j[1], j[2], j[3], j[4], j[5]] for j in jobs]
            self._wT_jobs = (torch.tensor(table, dtype=torch.int64, device=self.device) if table else None, len(table), tiles)
            for mod, o, n in views:
                mod._wT_view = self.wT[o:o + n]
        table, n, tiles = self._wT_jobs
        if n:
            _lib.call('pseg_filter_transpose_batch', table.data_ptr(), n, tiles, ops._stream())

    def prepare_half(self, transposed=True):
        """fp16 copies of every dense conv filter for the half-precision (`-mp`) policy, refreshed from the fp32 master
        weights with ONE launch: ``module._w_h_view`` ([Cout'][taps][Cin'], what the forward convs read) and -- when
        `transposed` -- ``module._wT_h_view`` ([Cin'][taps][Cout'], what the data gradients read).  Channel counts are padded
        to multiples of 8 in these copies (16-byte granules of fp16).  Call once per pass while the weights are fixed."""
        from . import _lib, ops
        if getattr(self, '_half_jobs', None) is None:
            jobs, views, off, tiles = [], [], 0, 0
            for seg in self.segments:
                mod = seg.module
                if seg.name != 'weight' or not getattr(mod, 'kernel_size', None) or getattr(mod, 'depthwise', True):
                    continue
                co, kh, kw, ci = seg.raw_shape
                cop, cip = (co + 7) // 8 * 8, (ci + 7) // 8 * 8
                n = cop * kh * kw * cip
                jobs.append([seg.offset, off, off + n, co, kh * kw, ci, cop, cip, tiles])
                views.append((mod, off, n))
                off += 2 * n
                tiles += kh * kw * ((cop + 31) // 32) * ((cip + 31) // 32)
            self.params_h = torch.zeros(max(off, 8), dtype=torch.float16, device=self.device)
            base_w, base_h = self.params.data_ptr(), self.params_h.data_ptr()
            table = [[base_w + 4 * j[0], base_h + 2 * j[1], base_h + 2 * j[2]] + j[3:] for j in jobs]
            notr = [[r[0], r[1], 0] + r[3:] for r in table]
            dev = self.device
            self._half_jobs = (torch.tensor(table, dtype=torch.int64, device=dev) if table else None,
                               torch.tensor(notr, dtype=torch.int64, device=dev) if table else None, len(table), tiles)
            for mod, o, n in views:
                mod._w_h_view = self.params_h[o:o + n]
                mod._wT_h_view = self.params_h[o + n:o + 2 * n]
        table, notr, n, tiles = self._half_jobs
        if n:
            _lib.call('pseg_filter_prepare_h', (table if transposed else notr).data_ptr(), n, tiles, ops._stream())

    def restore_grad_views(self):
        """Re-point ``.grad`` at the arena (after an external ``zero_grad(set_to_none=True)``)."""
        for s in self.segments:
            if s.param.grad is None or s.param.grad.data_ptr() != self._grad_views[id(s.param)].data_ptr():
                s.param.grad = self._grad_views[id(s.param)]

    def zero_grad(self):
        self.grads.zero_()

    def range_of(self, module):
        """[begin, end) element range of the arena that holds ``module``'s own and descendant parameters."""
        mods = set(id(m) for m in module.modules())
        offs = [(s.offset, s.offset + s.numel) for s in self.segments if id(s.module) in mods]
        if not offs:
            return None
        return min(o[0] for o in offs), max(o[1] for o in offs)


def prepare(model, device=None):
    """Move ``model`` to ``device`` and re-home its parameters into a fresh arena (idempotent per device)."""
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device())
    device = torch.device(device)
    if device.type == 'cuda' and device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    ar = getattr(model, '_pseg_arena', None)
    if ar is not None and ar.device == device and all(
            s.param.data_ptr() >= ar.params.data_ptr() and
            s.param.data_ptr() < ar.params.data_ptr() + ar.numel * 4 for s in ar.segments):
        return ar
    model.to(device)
    ar = ParamArena(model, device)
    object.__setattr__(model, '_pseg_arena', ar)
    return ar

"""Building blocks with explicit forward / backward on the HIP kernels.

Mirror of the reference's external block library (``pytorch_modules.nn.ConvNormAct`` -- contract restated from
the call sites listed in oracle/blocks.py) plus the torch.nn layers the model files use directly
(``nn.Conv2d`` for the classifiers, reference models/deeplabv3plus.py:22, models/unet.py:23).

Design: there is no autograd graph inside a model.  Every block has

    y, saved = block.fwd(x, env, ...)         # x, y: ops.Act (NHWC handles)
    dx       = block.bwd(dy, saved, env, ...)

and composite blocks call their children's fwd/bwd directly, so producers write straight into concat slices,
residual gradients merge in dgrad epilogues, and weight gradients land in the flat gradient arena the moment
their kernel finishes (which is what lets the data-parallel all-reduce start bucket by bucket).  Autograd sees a
whole model as ONE ``torch.autograd.Function`` (see bridge.py), so ``loss.backward()`` keeps working.

Parameters are ordinary ``nn.Parameter``s with the reference's names / shapes (state-dicts interchange with the
reference and the oracle); their storage lives in the arena (arena.py) in kernel-native layout.
"""
import os

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_NONE, ACT_RELU, ACT_RELU6, Act


def _round4(n):
    return (n + 3) // 4 * 4


def _round8(n):
    return (n + 7) // 8 * 8


class Env:
    """Per-step execution context handed down through fwd/bwd."""
    __slots__ = ('save', 'accumulate', 'grad_ready', 'overlap_wgrad', 'wT_fresh', 'wamax_fresh', 'policy', 'slab_pool',
                 'loss_scale', 'half_fresh')

    def __init__(self, save=True, accumulate=False, grad_ready=None, overlap_wgrad=False, policy=None):
        self.save = save              # keep what backward needs
        self.accumulate = accumulate  # parameter gradients += (micro-batch > 0 of an accumulation window)
        self.grad_ready = grad_ready  # callable(module): all parameter grads of `module` are enqueued
        # weight gradients go to the auxiliary stream (ops.fork_aux); whoever sets this joins it (ops.join_aux) before
        # anything reads the gradient arena
        self.overlap_wgrad = overlap_wgrad
        # the arena's transposed filters were refreshed for this backward pass (ParamArena.transpose_filters)
        self.wT_fresh = False
        # the arena's per-filter max|w| scalars were refreshed for this forward pass (ParamArena.filter_amax)
        self.wamax_fresh = False
        # conv arithmetic policy of THIS execution context ('fp32' | 'mixed' | 'limb' | ...); None = the process default
        # (ops.set_conv_precision / PSEG_PRECISION).  Trainer(mixed_precision=True) sets it on its own Env only.
        self.policy = policy
        # ops.SlabPool of the pass in flight: split weight gradients leave their slabs there and whoever set it folds
        # them with ONE launch at the end of backward (Trainer._fwd_loss_bwd); None = every conv reduces its own
        self.slab_pool = None
        # half-precision policy: device scalar (fp32, one element) the loss gradient is multiplied by where it enters the
        # fp16 network (dynamic loss scaling; the optimiser divides it out again); None = 1
        self.loss_scale = None
        # half-precision policy: the arena's fp16 filter copies were refreshed for this pass by whoever drives it (the bridge
        # then skips its per-block refresh)
        self.half_fresh = False

    @property
    def policy_name(self):
        return self.policy if self.policy is not None else ops.POLICY_NAME

    @property
    def fwd_prec(self):
        return ops._POLICIES[self.policy][0] if self.policy is not None else ops.FWD_PRECISION

    @property
    def bwd_prec(self):
        return ops._POLICIES[self.policy][1] if self.policy is not None else ops.BWD_PRECISION

    @property
    def half(self):
        """True under the half-precision (`-mp`) policy: fp16 activations, gradients and filter copies."""
        return self.policy_name == 'half'

    @property
    def act_dtype(self):
        return torch.float16 if self.half else torch.float32

    @property
    def track_amax(self):
        """True when the forward conv policy needs per-tensor maxima (fp16 limbs)."""
        return self.fwd_prec == ops.PREC_FP16X3


def loss_grad_in(dlr, env):
    """The gradient of the loss with respect to the (fp32) class logits enters the network: unchanged under the fp32
    policies; under the half-precision policy multiplied by the loss scale and rounded to fp16 in one pass."""
    if env.half and not dlr.half:
        return dlr.to(torch.float16, scale=env.loss_scale)
    return dlr


def _raw(module, name):
    try:
        return module._raw[name], module._raw_grad[name]
    except AttributeError:
        raise RuntimeError('%s is not arena-backed: call pytorch_segmentation_amd.prepare(model) (or run the model '
                           'once on a CUDA tensor) before using the HIP path' % type(module).__name__)


def _act_code(activate):
    if activate is None or activate is False:
        return ACT_NONE
    if activate is True or isinstance(activate, nn.ReLU):
        return ACT_RELU
    if isinstance(activate, nn.ReLU6):
        return ACT_RELU6
    raise NotImplementedError('activation %r has no HIP kernel (ReLU / ReLU6 / None)' % (activate,))


LIMB_MIN_CHANNELS = int(os.environ.get('PSEG_LIMB_MIN_CHANNELS', '0'))
# the ResNet stem's BatchNorm + ReLU + max-pool as ONE pass when nobody reads the activated stride-2 map (BatchNorm2d.fwd_pooled)
FUSE_STEM_POOL = os.environ.get('PSEG_FUSE_STEM_POOL', '1') == '1'


class Conv2d(nn.Conv2d):
    """nn.Conv2d parameter holder (same state-dict) whose arithmetic runs on the implicit-GEMM MFMA kernels
    (dense) or the direct depthwise kernels (groups == channels)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        k, s, p, d = self.kernel_size, self.stride, self.padding, self.dilation
        if k[0] != k[1] and groups != 1:
            raise NotImplementedError('depthwise kernels are square')
        if s[0] != s[1] or p[0] != p[1] or d[0] != d[1]:
            raise NotImplementedError('HIP conv supports symmetric stride / padding / dilation')
        self.depthwise = groups != 1
        if self.depthwise and not (groups == in_channels == out_channels and d[0] == 1 and k[0] <= 3):
            raise NotImplementedError('grouped conv other than depthwise k<=3 has no HIP kernel')
        self.cin_p, self.cout_p = _round4(in_channels), _round4(out_channels)
        # half-precision path: 8-channel (16-byte) granules; where that differs from the arena's padding (3-channel stem,
        # 2-class classifier) the fp16 filter copies carry their own padding and the weight gradient goes through a scratch
        self.cin_h, self.cout_h = _round8(in_channels), _round8(out_channels)
        # PSEG_LIMB_MIN_CHANNELS (default 0 = off): keep layers with fewer channels on exact fp32 under the limb policies.
        # Measured unnecessary once the weight-gradient split planner stopped starving small problems of blocks: the limb
        # kernels are at least as fast as the exact-fp32 ones on every HRNet / UNet / ResNet shape
        # (tools/bench_limb_breakeven.py), HRNet 512x512 batch 8: 19.7 ms/step fp32, 17.3 mixed.
        self.limb_pays = min(self.cin_p, self.cout_p) >= LIMB_MIN_CHANNELS

    # kernel-native storage inside the arena
    def _pseg_layout(self, name, p):
        co, ci = self.out_channels, self.in_channels
        kh, kw = self.kernel_size
        if name == 'weight':
            if self.depthwise:
                return (kh, kw, self.cout_p), (lambda raw: raw[:, :, :co].permute(2, 0, 1).unsqueeze(1))
            return (self.cout_p, kh, kw, self.cin_p), (lambda raw: raw[:co, :, :, :ci].permute(0, 3, 1, 2))
        if name == 'bias':
            return (self.cout_p,), (lambda raw: raw[:co])
        return None

    def out_hw(self, H, W):
        k, s, p, d = self.kernel_size, self.stride[0], self.padding[0], self.dilation[0]
        return ops.conv_out_size(H, k[0], s, p, d), ops.conv_out_size(W, k[1], s, p, d)

    def _half_filters(self):
        wh, wT = getattr(self, '_w_h_view', None), getattr(self, '_wT_h_view', None)
        if wh is None:
            raise RuntimeError('fp16 filter copies are missing: call ParamArena.prepare_half() before a half-precision pass')
        return wh, wT

    def _fwd_half(self, x, env, out, want_stats, out_f32):
        assert x.C == self.cin_h, 'conv expects %d (padded) input channels, got %d' % (self.cin_h, x.C)
        Ho, Wo = self.out_hw(x.H, x.W)
        y = out if out is not None else Act.empty(x.B, Ho, Wo, self.cout_h, x.device,
                                                  dtype=torch.float32 if out_f32 else torch.float16)
        assert (y.B, y.H, y.W, y.C) == (x.B, Ho, Wo, self.cout_h)
        kh, kw = self.kernel_size
        s, p, d = self.stride[0], self.padding[0], self.dilation[0]
        stats = None
        if self.depthwise:
            assert self.bias is None, 'depthwise conv with bias is not on the hot path'
            # the depthwise kernels read the fp32 filter / write the gradient straight in the arena, padded to 4 channels, while
            # fp16 activations are padded to 8: a width with C % 8 == 4 would read and write past the filter segment
            # (MobileNetV2's widths are multiples of 8; anything else is refused rather than mis-addressed)
            if self.cin_h != self.cin_p:
                raise NotImplementedError('depthwise conv with %d channels under the half policy: channel count must be a '
                                          'multiple of 8' % self.in_channels)
            ops.dwconv_fwd(x, _raw(self, 'weight')[0], y, kh, s, p)      # (fp32 filter, fp16 activations)
            if want_stats:
                stats = ops.col_stats(y)
        else:
            b = self._bias_h() if self.bias is not None else None
            stats = ops.conv2d_fwd(x, self._half_filters()[0], b, y, kh, kw, s, p, d, want_stats=want_stats)
        return y, stats, (x if env.save else None)

    def _bias_h(self):
        """fp32 bias padded to the fp16 path's output channels (zeros beyond the arena's padding)."""
        b = _raw(self, 'bias')[0]
        if self.cout_h == self.cout_p:
            return b
        pad = getattr(self, '_bias_pad', None)
        if pad is None or pad.device != b.device:
            pad = self._bias_pad = torch.zeros(self.cout_h, dtype=torch.float32, device=b.device)
        # (a kernel launch, not a tensor copy: a device-to-device memcpy node would keep a captured step off the lane executor)
        ops.copy2d(Act(b, 1, 1, 1, self.cout_p, self.cout_p), Act(pad, 1, 1, 1, self.cout_p, self.cout_h))
        return pad

    def fwd(self, x, env, out=None, want_stats=False, out_f32=False):
        """x: Act with C == padded in_channels.  Returns (y, stats|None, saved).
        out_f32 (half-precision policy only): write y as fp32 (the class logits, which the loss reads)."""
        if x.half:
            return self._fwd_half(x, env, out, want_stats, out_f32)
        assert x.C == self.cin_p, 'conv expects %d (padded) input channels, got %d' % (self.cin_p, x.C)
        w, _ = _raw(self, 'weight')
        b = _raw(self, 'bias')[0] if self.bias is not None else None
        Ho, Wo = self.out_hw(x.H, x.W)
        y = out if out is not None else Act.empty(x.B, Ho, Wo, self.cout_p, x.device)
        assert (y.B, y.H, y.W, y.C) == (x.B, Ho, Wo, self.cout_p)
        kh, kw = self.kernel_size
        s, p, d = self.stride[0], self.padding[0], self.dilation[0]
        stats = None
        if self.depthwise:
            assert b is None, 'depthwise conv with bias is not on the hot path'
            ops.dwconv_fwd(x, w, y, kh, s, p)
            if want_stats:
                stats = ops.col_stats(y)
        else:
            am = {}
            fprec = env.fwd_prec if self.limb_pays else ops.PREC_FP32
            if fprec == ops.PREC_FP16X3:
                # fp16-limb forward: both operands are scaled by an exact power of two from their max|.| bound
                wam = getattr(self, '_wamax_view', None) if env.wamax_fresh else None
                am = dict(amax_x=x.amax if x.amax is not None else ops.amax_of(x),
                          amax_w=wam if wam is not None else ops.amax_of(w))
            stats = ops.conv2d_fwd(x, w, b, y, kh, kw, s, p, d, want_stats=want_stats, precision=fprec, **am)
        return y, stats, (x if env.save else None)

    def wants_dy_planes(self, x, env):
        """Should the producer of this conv's output gradient (BatchNorm backward) also write it as bf16 limb planes?
        Yes under the limb policies for the convs whose data gradient the pre-split LDS-DMA kernel covers AND whose
        contraction is deep enough for the gain (1.15-1.35x on the kernel) to exceed the extra 4 bytes per element written:
        3x3 convs with >= 128 output channels (layer 2-4 bottleneck interiors, the ASPP branches)."""
        if not ops.DY_PLANES or self.depthwise or x is None:
            return False
        bprec = env.bwd_prec if self.limb_pays else ops.PREC_FP32
        kh, kw = self.kernel_size
        if bprec != ops.PREC_BF16X3 or kh * kw < 9 or self.cout_p % 8 != 0 or self.cout_p < 128:
            return False
        s, p, d = self.stride[0], self.padding[0], self.dilation[0]
        Ho, Wo = self.out_hw(x.H, x.W)
        return ops.dgrad_planes_ok_shape(x.B, x.H, x.W, self.cin_p, Ho, Wo, self.cout_p, kh, kw, s, p, d)

    def bwd(self, dy, saved, env, need_dx=True, dx_out=None, dx_accumulate=False, bn_prev=None, wgrad_on_main=False,
            wgrad_concurrent=None):
        """Enqueue wgrad (+bias grad) into the gradient arena and, if asked, dgrad.  Returns dx or None.
        wgrad_on_main: keep the weight gradient on the CURRENT stream although the pass forks its weight gradients (the ResNet
        stem: the last kernel of backward, meant to run beside the auxiliary stream's queue).  wgrad_concurrent: how the launch is
        PLANNED -- True: as one that runs beside another stream's kernels (one resident block per CU, half the slabs), False: as one
        that runs alone, None: concurrent exactly when it is forked.  (both are arguments now; the shared Env flag is
        not toggled around the call any more.)
        bn_prev: the saved state of the BatchNorm2d whose output (after its activation) is this conv's input, when this conv is
        that output's ONLY consumer and the layer has no residual -- dx is then exactly that layer's dz, and the data gradient
        computes its backward partial sums on the way out (ops.conv2d_dgrad(bn=...); BatchNorm2d.bwd picks them up)."""
        x = saved
        if x.half:
            return self._bwd_half(dy, x, env, need_dx, dx_out, dx_accumulate, bn_prev, wgrad_on_main)
        w, dw = _raw(self, 'weight')
        kh, kw = self.kernel_size
        s, p, d = self.stride[0], self.padding[0], self.dilation[0]
        bprec = env.bwd_prec if self.limb_pays else ops.PREC_FP32
        # (a gradient that a reducer picks up layer by layer must be complete when this call returns)
        pool = env.slab_pool if env.grad_ready is None else None

        forked = bool(env.overlap_wgrad and ops.OVERLAP_WGRAD)      # (the weight gradient runs beside the data gradients)
        if wgrad_concurrent is not None:
            forked = bool(wgrad_concurrent)
        elif wgrad_on_main:
            forked = False

        def wgrad_body():
            if self.depthwise:
                ops.dwconv_wgrad(x, dy, dw, kh, s, p, accumulate=env.accumulate)
            else:
                ops.conv2d_wgrad(x, dy, dw, kh, kw, s, p, d, accumulate=env.accumulate, precision=bprec, pool=pool,
                                 concurrent=forked)
            if self.bias is not None:
                ops.col_sum(dy, _raw(self, 'bias')[1], accumulate=env.accumulate)

        def wgrad():
            # (depthwise weight gradients too, since round 4: in the MobileNetV2 UNet they were 17 launches of the serial
            # backward chain -- 0.18 ms of a 3.6 ms replayed step)
            if env.overlap_wgrad and ops.OVERLAP_WGRAD and not wgrad_on_main:
                side = ops.fork_aux(x.device)
                with torch.cuda.stream(side):
                    wgrad_body()
                x.t.record_stream(side)     # the caching allocator must not hand these blocks out again before the
                dy.t.record_stream(side)    # auxiliary stream is done with them
            else:
                wgrad_body()
            if env.grad_ready is not None:
                env.grad_ready(self)

        # Order on the device.  Default: the weight gradient is forked to the auxiliary stream BEFORE the data gradient is
        # enqueued.  PSEG_WGRAD_AFTER_DGRAD=1 forks it after (so that it would start when the data gradient -- matrix-pipe
        # bound like itself -- has finished and run beside the HBM-bound BatchNorm backward passes of the next layer down):
        # measured no better, 51.0 -> 52.2 ms/step at the headline config, kept selectable.
        late = need_dx and ops.WGRAD_AFTER_DGRAD
        if not late:
            wgrad()
        if not need_dx:
            return None
        dx = dx_out if dx_out is not None else Act.empty(x.B, x.H, x.W, self.cin_p, x.device)
        if self.depthwise:
            if dx_accumulate:
                tmp = Act.empty(x.B, x.H, x.W, self.cin_p, x.device)
                ops.dwconv_dgrad(dy, w, tmp, kh, s, p)
                ops.copy2d(tmp, dx, accumulate=True)
            else:
                ops.dwconv_dgrad(dy, w, dx, kh, s, p)
        else:
            wT = getattr(self, '_wT_view', None) if env.wT_fresh else None
            if wT is None:
                wT = ops.filter_transpose(w, self.cout_p, kh * kw, self.cin_p)
            if dy.planes is not None and bprec == ops.PREC_BF16X3 and ops.dgrad_planes_ok(dy, dx, kh, kw, s, p, d):
                # dy arrived pre-split (BatchNorm backward wrote the limb planes): split the filter once, DMA kernel
                wp = ops.split_planes(wT.view(self.cin_p, kh * kw * self.cout_p))
                ops.conv2d_dgrad_planes(dy.planes, dy, wp, dx, kh, kw, s, p, d, accumulate=dx_accumulate)
            else:
                bn = None
                if bn_prev is not None and not dx_accumulate:
                    by, bz, bco, bact, use_batch, bmask = bn_prev
                    if bz is None and bmask is None and use_batch and by.C == dx.C and by.M == dx.M:
                        bn = (by, bco, bact)
                ops.conv2d_dgrad(dy, wT, dx, kh, kw, s, p, d, accumulate=dx_accumulate, precision=bprec, bn=bn)
        if late:
            wgrad()
        return dx

    def _bwd_half(self, dy, x, env, need_dx, dx_out, dx_accumulate, bn_prev=None, wgrad_on_main=False):
        """fp16 operands; the weight gradient lands in the fp32 gradient arena (scaled by the loss scale, which the
        optimiser divides out)."""
        assert dy.half and dy.C == self.cout_h
        w, dw = _raw(self, 'weight')
        kh, kw = self.kernel_size
        s, p, d = self.stride[0], self.padding[0], self.dilation[0]
        padded = (self.cin_h, self.cout_h) != (self.cin_p, self.cout_p)
        pool = env.slab_pool if (env.grad_ready is None and not padded) else None

        def wgrad_body():
            if self.depthwise:
                ops.dwconv_wgrad(x, dy, dw, kh, s, p, accumulate=env.accumulate)
            elif padded:
                # the fp16 path's channel padding differs from the arena's: gradient into a scratch, its arena-shaped
                # corner (first cout_p rows, first cin_p channels of every tap) copied / added into the arena
                taps = kh * kw
                tmp = torch.empty(self.cout_h * taps * self.cin_h, dtype=torch.float32, device=x.device)
                ops.conv2d_wgrad(x, dy, tmp, kh, kw, s, p, d, accumulate=False)
                ops.copy2d(Act(tmp, 1, 1, self.cout_p * taps, self.cin_p, self.cin_h),
                           Act(dw.view(-1), 1, 1, self.cout_p * taps, self.cin_p, self.cin_p), accumulate=env.accumulate)
            else:
                ops.conv2d_wgrad(x, dy, dw, kh, kw, s, p, d, accumulate=env.accumulate, pool=pool)
            if self.bias is not None:
                ops.col_sum(dy, _raw(self, 'bias')[1], accumulate=env.accumulate, C=self.cout_p)

        if env.overlap_wgrad and ops.OVERLAP_WGRAD and not wgrad_on_main:
            side = ops.fork_aux(x.device)
            with torch.cuda.stream(side):
                wgrad_body()
            x.t.record_stream(side)
            dy.t.record_stream(side)
        else:
            wgrad_body()
        if env.grad_ready is not None:
            env.grad_ready(self)
        if not need_dx:
            return None
        dx = dx_out if dx_out is not None else Act.empty(x.B, x.H, x.W, self.cin_h, x.device, dtype=torch.float16)
        if self.depthwise:
            if dx_accumulate:
                tmp = dx.like()
                ops.dwconv_dgrad(dy, w, tmp, kh, s, p)
                ops.copy2d(tmp, dx, accumulate=True)
            else:
                ops.dwconv_dgrad(dy, w, dx, kh, s, p)
        else:
            bn = None
            if bn_prev is not None and not dx_accumulate:
                by, bz, bco, bact, use_batch, bmask = bn_prev
                if bz is None and bmask is None and use_batch and by.C == dx.C and by.M == dx.M:
                    bn = (by, bco, bact)
            ops.conv2d_dgrad(dy, self._half_filters()[1], dx, kh, kw, s, p, d, accumulate=dx_accumulate, bn=bn)
        return dx

    def forward(self, x):
        from .bridge import run_module
        return run_module(self, x)

    # uniform block protocol used by the bridge
    def block_fwd(self, x, env):
        y, _, saved = self.fwd(x, env)
        return y, saved

    def block_bwd(self, dy, saved, env, need_dx=True):
        return self.bwd(dy, saved, env, need_dx=need_dx)

    @property
    def block_out_channels(self):
        return self.out_channels


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d parameter / buffer holder; normalisation, activation and residual add are one fused pass."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        # num_batches_tracked is bumped on the host and written back lazily (state_dict time): a device-side
        # `+= 1` per layer per step is 60 extra launches for a value nothing on the hot path reads
        self._nbt_pending = 0
        self.register_state_dict_pre_hook(BatchNorm2d._flush_counter)
        # ... and a loaded state REPLACES the count: batches counted since the last flush belong to the state that is being
        # overwritten (found by tests/test_fullsize_parity_gpu.py: reload + one step reported two batches)
        self.register_load_state_dict_pre_hook(BatchNorm2d._drop_counter)

    @staticmethod
    def _drop_counter(module, state_dict, prefix, *unused):
        if prefix + 'num_batches_tracked' in state_dict:
            module.__dict__['_nbt_pending'] = 0

    @staticmethod
    def _flush_counter(module, prefix, keep_vars):
        if module._nbt_pending and module.num_batches_tracked is not None:
            module.num_batches_tracked += module._nbt_pending
            module._nbt_pending = 0

    def fwd(self, y, stats, env, act=ACT_NONE, residual=None, out=None):
        """z = act(BN(y) (+ residual)).  `stats` are the column partials of y (training mode)."""
        C = self.num_features
        assert y.C == C and C % 4 == 0, 'BatchNorm2d HIP path needs C %% 4 == 0 (got %d)' % C
        g = _raw(self, 'weight')[0] if self.affine else None
        b = _raw(self, 'bias')[0] if self.affine else None
        use_batch = self.training or not self.track_running_stats
        mask = None
        if use_batch:
            if stats is None:
                stats = ops.col_stats(y)
            if y.M <= 1:
                raise ValueError('Expected more than 1 value per channel when training, got input size %s'
                                 % ((y.B, C, y.H, y.W),))
            mom = self.momentum
            rm = self.running_mean if self.track_running_stats else None
            rv = self.running_var if self.track_running_stats else None
            if self.track_running_stats and self.training:
                self.__dict__['_nbt_pending'] += 1      # (nn.Module.__setattr__ costs ~3 us per write)
                if mom is None:  # cumulative moving average
                    mom = 1.0 / float(int(self.num_batches_tracked) + self._nbt_pending)
            z = out if out is not None else y.like()
            if env.track_amax and z.amax is None:
                z.amax = ops.new_amax(z.device)
            if ops.bn_small_path(stats[1], y.M, C):
                # small tensors: finalize + normalise + activation (+ residual) in ONE launch
                co = ops.bn_fwd_fused(stats, y.M, g, b, rm if self.training else None, rv if self.training else None,
                                      mom if mom is not None else 0.0, self.eps, y, act, z, residual=residual)
            else:
                co = ops.bn_finalize(stats, y.M, g, b, rm if self.training else None, rv if self.training else None,
                                     mom if mom is not None else 0.0, self.eps)
                # a residual layer's activation mask cannot be recomputed from y: keep it as one bit per element
                mask = ops.bn_act_fwd(y, co, act, z, residual=residual,
                                      want_mask=residual is not None and env.save and ops.BN_MASK)
        else:
            co = ops.bn_eval_coeffs(g, b, self.running_mean, self.running_var, self.eps)
            z = out if out is not None else y.like()
            if env.track_amax and z.amax is None:
                z.amax = ops.new_amax(z.device)
            ops.bn_act_fwd(y, co, act, z, residual=residual)
        # without a residual the backward kernels recompute the activation mask from y: z need not be re-read
        saved = (y, z if (residual is not None or not use_batch) else None, co, act, use_batch, mask) if env.save else None
        return z, saved

    def fwd_pooled(self, y, stats, env, act, k, stride, pad):
        """maxpool(act(BN(y))) in training mode without the activated map: -> (pooled, argmax, saved) -- `saved` is what fwd() would
        have saved (backward recomputes the activation mask from y), or None when this form does not apply (eval-mode statistics,
        per-tensor maxima wanted): the caller then runs fwd() and the pooling pass."""
        use_batch = self.training or not self.track_running_stats
        if not (FUSE_STEM_POOL and use_batch and not env.track_amax and y.M > 1):
            return None
        C = self.num_features
        assert y.C == C and C % 4 == 0
        g = _raw(self, 'weight')[0] if self.affine else None
        b = _raw(self, 'bias')[0] if self.affine else None
        if stats is None:
            stats = ops.col_stats(y)
        mom = self.momentum
        rm = self.running_mean if self.track_running_stats else None
        rv = self.running_var if self.track_running_stats else None
        if self.track_running_stats and self.training:
            self.__dict__['_nbt_pending'] += 1
            if mom is None:
                mom = 1.0 / float(int(self.num_batches_tracked) + self._nbt_pending)
        co = ops.bn_finalize(stats, y.M, g, b, rm if self.training else None, rv if self.training else None,
                             mom if mom is not None else 0.0, self.eps)
        Hp, Wp = ops.conv_out_size(y.H, k, stride, pad, 1), ops.conv_out_size(y.W, k, stride, pad, 1)
        p = y.new(y.B, Hp, Wp, C)
        arg = ops.bn_act_maxpool_fwd(y, co, act, p, k, stride, pad, want_argmax=env.save)
        saved = (y, None, co, act, use_batch, None) if env.save else None
        return p, arg, saved

    def bwd(self, dz, saved, env, dy_out=None, dres=None, res_accumulate=False, want_planes=False):
        """Returns dy (gradient w.r.t. the BN input).  dres (optional Act) receives the residual-branch gradient.
        want_planes: also write dy as bf16 limb planes (dy.planes) for the consumer conv's pre-split data gradient."""
        y, z, co, act, use_batch, mask = saved
        dy = dy_out if dy_out is not None else y.like()
        dg = _raw(self, 'weight')[1] if self.affine else None
        db = _raw(self, 'bias')[1] if self.affine else None
        # eval mode (frozen running statistics): same two passes, with the statistics treated as constants --
        # dy = scale * dz * act', dgamma = sum(dz * act' * xhat), dbeta = sum(dz * act') (what autograd gives for
        # F.batch_norm(training=False))
        ops.bn_act_bwd(dz, z, y, co, act, dy, dg, db, accumulate=env.accumulate, dres=dres,
                       res_accumulate=res_accumulate, frozen=not use_batch, mask=mask, want_planes=want_planes,
                       part=dz.bnpart if use_batch else None)
        if env.grad_ready is not None:
            env.grad_ready(self)
        return dy


class ConvNormAct(nn.Module):
    """conv -> BatchNorm2d -> activation with the reference's external contract (oracle/blocks.py):
    ``ConvNormAct(cin, cout, ksize=3, stride=1, groups=1, dilation=1, activate=True)``, 'same' padding
    ``(ksize-1)//2*dilation``, bias-free conv, children named '0' / '1' / '2' like the nn.Sequential it mirrors."""

    def __init__(self, in_channels, out_channels, ksize=3, stride=1, groups=1, dilation=1, activate=True):
        super().__init__()
        pad = (ksize - 1) // 2 * dilation
        self.add_module('0', Conv2d(in_channels, out_channels, ksize, stride, pad, dilation, groups, bias=False))
        self.add_module('1', BatchNorm2d(out_channels))
        self.act = _act_code(activate)
        if self.act == ACT_RELU:
            self.add_module('2', nn.ReLU(inplace=True))
        elif self.act == ACT_RELU6:
            self.add_module('2', nn.ReLU6(inplace=True))

    @property
    def conv(self):
        return self._modules['0']

    @property
    def bn(self):
        return self._modules['1']

    def fwd(self, x, env, out=None):
        y, stats, sc = self.conv.fwd(x, env, want_stats=self.bn.training)
        z, sb = self.bn.fwd(y, stats, env, act=self.act, out=out)
        return z, (sc, sb)

    def bwd(self, dz, saved, env, need_dx=True, dx_out=None, dx_accumulate=False):
        sc, sb = saved
        dy = self.bn.bwd(dz, sb, env, want_planes=need_dx and self.conv.wants_dy_planes(sc, env))
        return self.conv.bwd(dy, sc, env, need_dx=need_dx, dx_out=dx_out, dx_accumulate=dx_accumulate)

    def forward(self, x):
        from .bridge import run_module
        return run_module(self, x)

    def block_fwd(self, x, env):
        return self.fwd(x, env)

    def block_bwd(self, dy, saved, env, need_dx=True):
        return self.bwd(dy, saved, env, need_dx=need_dx)

    @property
    def block_out_channels(self):
        return self.conv.out_channels


def initialize_weights(module):
    """Role of pytorch_modules.utils.initialize_weights at the reference's call sites
    (models/deeplabv3plus.py:24-26, models/unet.py:24-25): Kaiming-normal convs, unit BN."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)

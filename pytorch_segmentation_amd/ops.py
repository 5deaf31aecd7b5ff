"""Tensor-level wrappers over the C-ABI (include/pseg_amd.h).

PyTorch is plumbing here: it owns device memory (caching allocator) and the HIP stream.  Every function
enqueues on ``torch.cuda.current_stream()`` and never synchronises the host.  Activations travel as
:class:`Act` handles -- fp32 NHWC with an explicit pixel stride, so a channel slice of a concat buffer is
just another handle over the same storage (the reference's ``torch.cat`` calls, models/aspp.py:36,
models/deeplabv3plus.py:38, models/unet.py:34-46, cost no copy).
"""
import os

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2
PREC_FP32, PREC_BF16X3, PREC_BF16X6, PREC_FP16X3 = 0, 1, 2, 3
_PREC_NAMES = {'fp32': PREC_FP32, 'bf16x3': PREC_BF16X3, 'bf16x6': PREC_BF16X6, 'fp16x3': PREC_FP16X3}
# Conv arithmetic policy (forward convs, backward convs) used when a call does not name a precision.
#   fp32   : exact fp32 MFMA everywhere (bit-tight against the CPU oracle; the reference's own arithmetic)
#   mixed  : forward exact fp32, backward (dgrad + wgrad) split-bf16 three-product (op-level error <= 2e-4)
#   bf16x3 / bf16x6 : everything on bf16 limbs (three / six partial products)
_POLICIES = {'fp32': (PREC_FP32, PREC_FP32), 'mixed': (PREC_FP32, PREC_BF16X3),
             'bf16x3': (PREC_BF16X3, PREC_BF16X3), 'bf16x6': (PREC_BF16X6, PREC_BF16X6),
             # forward on fp16 limbs of the amax-scaled operands (~2^-22 per product), backward on bf16 limbs
             'limb': (PREC_FP16X3, PREC_BF16X3),
             # half-precision storage (`train.py -mp`): fp16 activations / gradients / filter copies, one fp16 MFMA pass with
             # fp32 accumulation, fp32 master weights and dynamic loss scaling (the precisions named here are unused: the
             # fp16 kernels have one arithmetic)
             'half': (PREC_FP32, PREC_FP32)}
# Default: 'fp32' -- the reference's arithmetic for every conv (drop-in fidelity first).  The faster reduced-product
# policies are opt-in: PSEG_PRECISION=mixed|limb, ops.set_conv_precision(...), Env(policy=...); train.py -mp selects 'half' (PSEG_MP_POLICY=limb: the fp32-storage limb policy instead).
POLICY_NAME = os.environ.get('PSEG_PRECISION', 'fp32')
FWD_PRECISION, BWD_PRECISION = _POLICIES[POLICY_NAME]


def set_conv_precision(name):
    """Select the conv arithmetic policy: 'fp32' | 'mixed' | 'bf16x3' | 'bf16x6' (see the table above)."""
    global FWD_PRECISION, BWD_PRECISION, POLICY_NAME
    FWD_PRECISION, BWD_PRECISION = _POLICIES[name]
    POLICY_NAME = name


def _prec(precision, backward=False):
    if precision is None:
        return BWD_PRECISION if backward else FWD_PRECISION
    return precision


def _round4(n):
    return (n + 3) // 4 * 4


def _round8(n):
    return (n + 7) // 8 * 8


def _stream():
    # raw handle of the current stream of the current device (0.2 us; torch.cuda.current_stream() builds a Python
    # Stream object every time, ~2 us, and this runs once per launch)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


# ---------------------------------------------------------------------------------------------- auxiliary stream
# Weight gradients have no consumer inside backward (only the optimiser / the gradient all-reduce read them), while the
# chain  BN-backward -> dgrad -> BN-backward ...  is strictly serial.  The weight-gradient kernels are bound by VALU +
# matrix work, the BatchNorm passes by HBM, the data gradients by LDS: enqueued on a second HIP stream they fill the
# same CUs side by side.  `fork_aux` orders the auxiliary stream after everything enqueued so far on the current one;
# `join_aux` makes the current stream wait for it (end of backward).
OVERLAP_WGRAD = os.environ.get('PSEG_OVERLAP_WGRAD', '1') == '1'
# limb policies: BatchNorm backward writes dy of wide 3x3 convs as bf16 limb planes too, their data gradient runs on the
# pre-split LDS-DMA kernel (nn.Conv2d.wants_dy_planes)
DY_PLANES = os.environ.get('PSEG_DY_PLANES', '1') == '1'
# residual BatchNorm layers keep their activation mask as a bitmask for backward (bn_act_fwd(want_mask=True))
BN_MASK = os.environ.get('PSEG_BN_MASK', '1') == '1'
# enqueue a conv's weight gradient after its data gradient (see nn.Conv2d.bwd)
WGRAD_AFTER_DGRAD = os.environ.get('PSEG_WGRAD_AFTER_DGRAD', '0') == '1'
_aux_streams = {}
_aux_dirty = {}


_aux_events = {}


_cur_stream_cache = {}


def _current_stream_obj(device):
    """torch.cuda.current_stream(device), memoised on the raw handle (building the Stream object costs ~3 us, and the
    backward pass asks for it once per conv)."""
    raw = _stream()
    hit = _cur_stream_cache.get(raw)
    if hit is None:
        hit = _cur_stream_cache[raw] = torch.cuda.current_stream(device)
        if len(_cur_stream_cache) > 64:
            _cur_stream_cache.clear()
    return hit


_own_stream_handles = {}       # device index -> torch stream ids of every stream this package has taken from torch's pool
OWN_STREAMS = os.environ.get('PSEG_OWN_STREAMS', '1') == '1'      # 0: torch.cuda.Stream() as it comes, torch's default capture stream


def new_stream(device):
    """A torch stream whose HIP handle differs from every stream this package already holds on the device.  torch hands out
    its 32 pool streams per device ROUND-ROBIN: in a process that has created many (a long test session; a service that builds
    Trainers over and over) a fresh ``torch.cuda.Stream()`` can BE the weight-gradient stream, the exchange stream or the
    capture stream -- a 'forked' launch then runs in line, silently (round 6: a lane-executor test saw a forked graph with one
    lane after the suite grew by two files)."""
    device = torch.device(device)
    if not OWN_STREAMS:
        return torch.cuda.Stream(device=device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    held = _own_stream_handles.setdefault(idx, set())
    s = None
    # (identity = torch's stream id, NOT the HIP handle: torch creates a pool stream's HIP stream on first access to the handle,
    # and the order in which a process's HIP streams come into being decides which hardware queue each lands on -- reading
    # `.cuda_stream` here materialised the never-used exchange stream of a one-rank Trainer ahead of the lane pool and cost the
    # replayed HRNet step 8 %: 7.19 -> 7.79 ms -mp, profiles/r06_ab_streams.txt)
    for _ in range(64):
        s = torch.cuda.Stream(device=device)
        if s.stream_id not in held:
            break
    held.add(s.stream_id)
    return s


_role_streams = {}


def role_stream(role, device=None):
    """One stream per (device, role) for the whole process, distinct from the package's other streams (new_stream): 'exchange' =
    the gradient reducer's side stream -- shared by every Trainer of the device: sharing only adds ordering, and a stream per
    Trainer would walk through torch's 32-stream pool."""
    idx = torch.cuda.current_device() if device is None or torch.device(device).index is None else torch.device(device).index
    st = _role_streams.get((idx, role))
    if st is None:
        st = _role_streams[(idx, role)] = new_stream(torch.device('cuda', idx))
    return st


def _make_aux_stream(device):
    """The auxiliary (weight-gradient) stream.  (A CU-masked stream that kept N CUs free for the main chain was measured in
    round 3 -- 44.6 -> 54.7-65.9 ms -- and is gone: profiles/EXPERIMENTS.md section 1.)"""
    return new_stream(device)


# PSEG_AUX_STREAMS: how many auxiliary streams the forked work (weight gradients) is dealt onto, round-robin.  The large
# configurations fill the chip with every weight-gradient launch and want ONE (a second would only interleave them); the
# narrow layers of the launch-bound ones (HRNet's 32 / 64-channel branches in fp32) leave it half empty, and there the
# auxiliary lane -- not the backward chain -- ends the step: two lanes overlap them.
AUX_STREAMS = max(1, min(3, int(os.environ.get('PSEG_AUX_STREAMS', '1'))))
# ... and while a step is being CAPTURED, as many as the model asks for (`capture_wgrad_lanes`, set by the Trainer around the
# capture; PSEG_AUX_STREAMS_CAPTURE forces a number).  The weight gradients are independent of one another -- only the ONE
# stream they share makes a chain of them -- and in a replayed HRNet step that chain (103 weight gradients + 104 slab
# reductions: 4.2 ms) was the longest lane of backward by a millisecond: two lanes, HRNet -mp 8.0 -> 7.5 ms, fp32 15.2 -> 14.7.
# UNet and DeepLabV3+ lose with two (3.36 -> 3.61, 14.14 -> 14.31 ms: their backward chain is the longest lane, and a second
# weight-gradient lane only takes CUs from it): they keep one.
_AUX_CAPTURE_FORCED = os.environ.get('PSEG_AUX_STREAMS_CAPTURE')
AUX_STREAMS_CAPTURE = max(1, min(3, int(_AUX_CAPTURE_FORCED))) if _AUX_CAPTURE_FORCED else 1
AUX_STREAMS_MAX = max(AUX_STREAMS, AUX_STREAMS_CAPTURE, 2)


def set_capture_wgrad_lanes(n):
    """-> the previous value.  Called by the Trainer around a capture with the model's `capture_wgrad_lanes` (default 1)."""
    global AUX_STREAMS_CAPTURE
    prev = AUX_STREAMS_CAPTURE
    if not _AUX_CAPTURE_FORCED:
        AUX_STREAMS_CAPTURE = max(1, min(AUX_STREAMS_MAX, int(n)))
    return prev


_aux_next = {}


def fork_aux(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    pool = _aux_streams.get(idx)
    if pool is None:
        pool = _aux_streams[idx] = [_make_aux_stream(device) for _ in range(AUX_STREAMS_MAX)]
        _aux_events[idx] = [torch.cuda.Event() for _ in range(AUX_STREAMS_MAX)]
        _aux_dirty[idx] = set()
        _aux_next[idx] = 0
    n = AUX_STREAMS_CAPTURE if CAPTURING > 0 else AUX_STREAMS
    k = _aux_next[idx] % n
    _aux_next[idx] = (k + 1) % n
    aux = pool[k]
    ev = _aux_events[idx][k]       # one event object per stream, re-recorded: a wait captures the record that precedes it
    ev.record(_current_stream_obj(device))
    aux.wait_event(ev)
    _aux_dirty[idx].add(k)
    return aux


def aux_streams_in_use(device):
    """The auxiliary streams that work was forked onto since the last join (possibly none)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    pool = _aux_streams.get(idx)
    return [pool[k] for k in sorted(_aux_dirty.get(idx, ()))] if pool else []


def join_aux(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    dirty = _aux_dirty.get(idx)
    if dirty:
        cur = torch.cuda.current_stream(device)
        for k in sorted(dirty):
            cur.wait_stream(_aux_streams[idx][k])
        dirty.clear()
        _aux_next[idx] = 0


# ---------------------------------------------------------------------------------------------- branch lanes
# A model with independent branches (HRNet: up to four resolution branches of 8 convs each between two fusion points, then
# one independent accumulation chain per fused output) is a few hundred launches of 5-20 us on ONE stream: every small
# kernel of a low-resolution branch waits for the one before it while 200 CUs idle.  While a step is being CAPTURED the
# branches are enqueued on streams of their own (forked behind an event of the current stream, joined before the next
# fusion); the lane executor (csrc/lanes.hip) finds them in the captured graph as parallel chains and replays them on
# parallel lanes.  Eager steps stay on one stream -- they are bound by the host's enqueue rate, not by the device -- unless
# PSEG_BRANCH_EAGER=1 (tests).  Same kernels, same arguments, same order within every chain: results are bit-identical.
# PSEG_BRANCH_STREAMS=0 switches the forks off.
# Three streams (one per non-main branch of HRNet).  The HIP runtime multiplexes the streams of a process onto four hardware
# queues (GPU_MAX_HW_QUEUES), a lane that shares its queue with the main chain blocks it (head of the line), and five busy
# queues (GPU_MAX_HW_QUEUES=8) fall off a cliff -- HRNet -mp 18 ms/step against 9.1; profiles/EXPERIMENTS.md section 0.12.
# The replay itself is capped by the model (`replay_lanes`: HRNet 5 -- forward: main chain + three branch lanes; backward: six
# chains on five lanes).  Measured (branch streams, lanes): (2, 4) 7.75, (2, 5) 7.40, (3, 4) 7.21, (3, 5) 7.11-7.17, (3, 6) 7.30 ms.
BRANCH_STREAMS = max(0, min(6, int(os.environ.get('PSEG_BRANCH_STREAMS', '3'))))
BRANCH_EAGER = os.environ.get('PSEG_BRANCH_EAGER', '0') == '1'
_branch_pool = {}
_branch_depth = 0
_branch_deferred = []


def after_branches(fn):
    """Run fn() now, or -- inside a Branches region, where "everything enqueued so far" is spread over several streams --
    when the region has been joined (the gradient-bucket markers of a captured data-parallel step)."""
    if _branch_depth > 0:
        _branch_deferred.append(fn)
    else:
        fn()


def reset_branches():
    """After an exception inside a Branches region (a failed capture): forget the open region and its deferred callbacks, so
    that the next step forks again instead of running with the forks silently off."""
    global _branch_depth
    _branch_depth = 0
    del _branch_deferred[:]


class _Lane:
    __slots__ = ('ctx',)

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


class Branches:
    """b = Branches(device, n); for i: `with b.lane(i, *inputs): ...`; b.join(*outputs).

    lane 0 is the current stream; lane i > 0 a pooled stream ordered behind everything enqueued on the current stream when
    the region opened.  `inputs`: the Acts / tensors the lane reads that were allocated elsewhere, `outputs`: what the
    code after the join reads -- both are recorded with the caching allocator for the stream that uses them (a block must
    not be handed out again on its own stream while another stream still reads it; inside a capture that defers the
    release to the end of the capture).  Off (everything on the current stream) outside a capture, inside another region,
    or for a single lane."""

    def __init__(self, device, n):
        global _branch_depth
        self.on = n > 1 and BRANCH_STREAMS > 0 and (CAPTURING > 0 or BRANCH_EAGER) and _branch_depth == 0
        if not self.on:
            return
        idx = device.index if device.index is not None else torch.cuda.current_device()
        pool = _branch_pool.get(idx)
        if pool is None:
            pool = _branch_pool[idx] = [new_stream(device) for _ in range(BRANCH_STREAMS)]
        self.pool = pool
        self.main = torch.cuda.current_stream(device)
        self.fork = torch.cuda.Event()
        self.fork.record(self.main)
        self.used = []
        _branch_depth += 1

    @staticmethod
    def _record(items, stream):
        for a in items:
            if a is None:
                continue
            t = a.t if isinstance(a, Act) else a
            t.record_stream(stream)

    def lane(self, i, *inputs):
        if not self.on or i == 0:
            return _Lane(None)
        s = self.pool[(i - 1) % len(self.pool)]
        if s not in self.used:
            s.wait_event(self.fork)
            self.used.append(s)
        self._record(inputs, s)
        return _Lane(torch.cuda.stream(s))

    def join(self, *outputs):
        global _branch_depth
        if not self.on:
            return
        for s in self.used:
            self.main.wait_stream(s)
        self._record(outputs, self.main)
        _branch_depth -= 1
        self.on = False
        if _branch_depth == 0 and _branch_deferred:
            fns = list(_branch_deferred)
            del _branch_deferred[:]
            for fn in fns:
                fn()


def _ptr(t):
    return 0 if t is None else t.data_ptr()


class Act:
    """NHWC activation [B,H,W,C] with pixel stride ``ld`` (elements); ``t`` is a 1-D tensor whose first element is
    element (0,0,0,0) and which keeps the storage alive.  fp32 everywhere except under the half-precision (`-mp`)
    policy, where activations and their gradients are fp16 (C % 8 == 0, ld % 8 == 0: 16 bytes = 8 channels)."""
    __slots__ = ('t', 'B', 'H', 'W', 'C', 'ld', 'amax', 'planes', 'bnpart')

    def __init__(self, t, B, H, W, C, ld, amax=None):
        assert t.dtype in (torch.float32, torch.float16) and t.dim() == 1
        q = 4 if t.dtype == torch.float32 else 8
        assert ld % q == 0 and ld >= C and t.data_ptr() % 16 == 0, \
            'NHWC handle must be 16-byte aligned, ld %% %d == 0' % q
        assert t.dtype == torch.float32 or C % 8 == 0, 'fp16 activations carry multiples of 8 channels'
        need = ((B * H * W - 1) * ld + C) if B * H * W > 0 else 0
        assert t.numel() >= need, 'backing tensor too small'
        self.t, self.B, self.H, self.W, self.C, self.ld = t, B, H, W, C, ld
        # optional device scalar: an upper bound of max|x| over this tensor (and every slice sharing it); the fp16-limb
        # conv kernels scale their operands by it.  Producers raise it atomically; it must start at 0.
        self.amax = amax
        # bf16 limb planes of this tensor (Planes), when its producer wrote them alongside (bn_act_bwd(want_planes=True))
        self.planes = None
        # BatchNorm-backward partial sums of this tensor seen as a layer's dz, when the data gradient that produced it
        # computed them alongside (conv2d_dgrad(bn=...)): BnPart
        self.bnpart = None

    @property
    def M(self):
        return self.B * self.H * self.W

    @property
    def ptr(self):
        return self.t.data_ptr()

    @property
    def device(self):
        return self.t.device

    @property
    def dtype(self):
        return self.t.dtype

    @property
    def half(self):
        return self.t.dtype == torch.float16

    @staticmethod
    def empty(B, H, W, C, device, zero=False, ld=None, amax=False, dtype=torch.float32):
        if ld is None:
            ld = _round4(C) if dtype == torch.float32 else _round8(C)
        n = B * H * W * ld
        t = torch.zeros(n, dtype=dtype, device=device) if zero else torch.empty(n, dtype=dtype, device=device)
        return Act(t, B, H, W, C, ld, new_amax(t.device) if amax else None)

    def new(self, B, H, W, C, zero=False, amax=False):
        """A fresh activation of this one's storage type on its device."""
        return Act.empty(B, H, W, C, self.t.device, zero=zero, amax=amax, dtype=self.t.dtype)

    def like(self, C=None, zero=False, dtype=None):
        return Act.empty(self.B, self.H, self.W, self.C if C is None else C, self.t.device, zero=zero,
                         dtype=self.t.dtype if dtype is None else dtype)

    def slice(self, c0, c1):
        """Channels [c0, c1) of this activation (no copy)."""
        q = 8 if self.half else 4
        assert 0 <= c0 < c1 <= self.ld and c0 % q == 0
        return Act(self.t[c0:], self.B, self.H, self.W, c1 - c0, self.ld, self.amax)

    def to(self, dtype, scale=None):
        """A converted copy (fp32 <-> fp16), optionally multiplied by the device scalar `scale` (the loss scale)."""
        out = self.like(dtype=dtype)
        _lib.call('pseg_convert2d', self.ptr, int(self.half), self.ld, out.ptr, int(out.half), out.ld, self.M, self.C,
                  _ptr(scale), _stream())
        return out

    def view4(self):
        """Strided torch view [B,H,W,C] in the handle's own dtype (tests / debugging / host-side glue only)."""
        return torch.as_strided(self.t, (self.B, self.H, self.W, self.C),
                                (self.H * self.W * self.ld, self.W * self.ld, self.ld, 1))

    def to_nchw(self, C=None):
        """Contiguous fp32 NCHW torch tensor with the first C channels (layout kernel, not torch.permute)."""
        C = self.C if C is None else C
        src = self.to(torch.float32) if self.half else self
        out = torch.empty(self.B, C, self.H, self.W, dtype=torch.float32, device=self.t.device)
        _lib.call('pseg_nhwc_to_nchw', src.ptr, src.ld, out.data_ptr(), self.B, C, self.H * self.W, _stream())
        return out

    @staticmethod
    def from_nchw(x, Cpad=None, dtype=torch.float32):
        """fp32 NCHW torch tensor -> NHWC handle, channels zero-padded up to Cpad (default: next multiple of 4; 8 for
        fp16 handles)."""
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
        x = x.contiguous()
        B, C, H, W = x.shape
        if dtype == torch.float16:
            Cpad = _round8(C if Cpad is None else Cpad)
        else:
            Cpad = _round4(C) if Cpad is None else Cpad
        if dtype == torch.float16 and C <= 8 and Cpad == 8:      # the image: straight into 8-channel fp16 pixels, one pass
            out = Act.empty(B, H, W, 8, x.device, dtype=torch.float16)
            _lib.call('pseg_nchw_to_nhwc_h', x.data_ptr(), out.ptr, out.ld, B, C, H * W, _stream())
            return out
        out = Act.empty(B, H, W, Cpad, x.device, ld=_round4(Cpad))
        _lib.call('pseg_nchw_to_nhwc', x.data_ptr(), out.ptr, out.ld, B, C, H * W, Cpad, _stream())
        return out.to(dtype) if dtype != torch.float32 else out


class _Workspace:
    """One growing scratch tensor per (device, stream).  Kernels on a stream run in order, so a single
    buffer per stream is race-free; torch's allocator makes growth stream-safe."""

    def __init__(self):
        self._bufs = {}
        self._retired = []

    def get(self, nbytes, device):
        key = (device.index, _stream())     # raw handle of the current stream (0.3 us; current_stream() builds objects: 5 us)
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            if buf is not None and EVER_CAPTURED:
                # a captured step (Trainer graph mode) has this buffer's address baked into its launches: a larger request
                # must not free it under the graph's feet -- retired buffers stay alive for the life of the process
                self._retired.append(buf)
            nbytes = max(int(nbytes), 1 << 20)
            buf = torch.empty((nbytes + 255) // 256 * 256, dtype=torch.uint8, device=device)
            self._bufs[key] = buf
        return buf


workspace = _Workspace()


def conv_out_size(n, k, stride, pad, dil):
    return (n + 2 * pad - dil * (k - 1) - 1) // stride + 1


# ---------------------------------------------------------------------------------------------- convolution
class AmaxPool:
    """Per-step pool of device scalars for the per-tensor max|x| bounds of the fp16-limb forward: one zero-fill per
    forward pass instead of one torch.zeros(1) launch per BatchNorm layer (~100 launches in HRNet)."""

    def __init__(self, device, n=2048):
        self.buf = torch.zeros(n, dtype=torch.float32, device=device)
        self.n, self.used = n, 0

    def reset(self):
        if self.used:
            self.buf[:self.used].zero_()
        self.used = 0

    def take(self):
        if self.used >= self.n:
            return torch.zeros(1, dtype=torch.float32, device=self.buf.device)
        v = self.buf[self.used:self.used + 1]
        self.used += 1
        return v


_amax_pool = None      # the pool of the forward pass in flight (Trainer._fwd_loss_bwd installs / removes it)


def new_amax(device):
    """A zeroed device scalar for an amax bound."""
    if _amax_pool is not None and _amax_pool.buf.device == device:
        return _amax_pool.take()
    return torch.zeros(1, dtype=torch.float32, device=device)


def amax_of(t_or_act):
    """Device scalar holding max|x| of a torch tensor or an Act (fresh computation with the generic kernel)."""
    out = new_amax(t_or_act.device)
    if isinstance(t_or_act, Act):
        a = t_or_act
        _lib.call('pseg_amax', a.ptr, a.ld, a.M, a.C, out.data_ptr(), _stream())
    else:
        t = t_or_act
        assert t.is_contiguous()
        _lib.call('pseg_amax', t.data_ptr(), t.numel(), 1, t.numel(), out.data_ptr(), _stream())
    return out


# ---- operands above 2 GiB (round 6) --------------------------------------------------------------------------------------
# The conv kernels read their GATHERED operand (x of a forward conv, dy of a data gradient, x and dy of a weight gradient)
# through ONE buffer descriptor with 32-bit byte offsets: the range check that makes padding and ragged edges free caps that
# tensor at 2 GiB (csrc/conv_common.h kMaxBytes; results are addressed with 64-bit pointers and are not capped).  On a 288 GB
# part that is per-GPU batch ~120 at 512x512 -- and `-s` / `-bs` are free-form in the reference (train.py:88-90).  A conv is
# independent per image, so an operand above the cap is handed to the library in equal BATCH CHUNKS (the smallest divisor of B
# whose chunk fits): forward convs and data gradients write disjoint image ranges of their result, weight gradients add chunk
# after chunk into the gradient (fp32, fixed chunk order: reproducible), fused BatchNorm statistics are the chunks' partial rows
# side by side (legal when a chunk is whole row groups -- checked).  What does not chunk (a single image above 2 GiB, B with
# no fitting divisor) keeps the library's "exceeds 2 GiB" error.
_CAP_BYTES = (1 << 31) - 64


def _span_bytes(a):
    return ((a.M - 1) * a.ld + a.C) * (2 if a.half else 4) if a.M > 0 else 0


def _batch_chunks(*acts):
    """-> images per chunk (== B when every operand fits its descriptor)"""
    B = acts[0].B
    for a in acts:          # (the common case in two comparisons per operand: this sits in front of every conv call)
        if ((a.B * a.H * a.W - 1) * a.ld + a.C) * (2 if a.t.dtype == torch.float16 else 4) >= _CAP_BYTES:
            break
    else:
        return B
    for d in range(2, B + 1):
        if B % d == 0 and all(_span_bytes(_sub(a, 0, B // d)) < _CAP_BYTES for a in acts):
            return B // d
    return B          # (no fitting divisor: the library reports the cap)


def _sub(a, b0, nb):
    """images [b0, b0 + nb) of an activation (no copy)"""
    return Act(a.t[b0 * a.H * a.W * a.ld:], nb, a.H, a.W, a.C, a.ld, a.amax)


def conv2d_fwd(x, w_raw, bias_raw, y, kh, kw, stride, pad, dil, accumulate=False, want_stats=False, precision=None,
               amax_x=None, amax_w=None):
    """y = conv(x, w) (+bias); an x above the 2 GiB descriptor cap runs in batch chunks (see above)."""
    nb = _batch_chunks(x)
    if nb == x.B:
        return _conv2d_fwd_one(x, w_raw, bias_raw, y, kh, kw, stride, pad, dil, accumulate, want_stats, precision, amax_x, amax_w)
    parts = []
    for b0 in range(0, x.B, nb):
        parts.append(_conv2d_fwd_one(_sub(x, b0, nb), w_raw, bias_raw, _sub(y, b0, nb), kh, kw, stride, pad, dil, accumulate,
                                     want_stats, precision, amax_x, amax_w))
    if not want_stats:
        return None
    rows, group = parts[0][1], parts[0][2]
    # row group g of chunk c is row group c * rows + g of the whole tensor only when a chunk is WHOLE groups
    if any(p[1] != rows or p[2] != group for p in parts) or rows * group != nb * y.H * y.W:
        raise _lib.PsegError('conv2d_fwd: a %d-image chunk of this %d-image tensor (operand above 2 GiB) is not whole '
                             'statistics groups (%d rows of %d for %d pixels)' % (nb, x.B, rows, group, nb * y.H * y.W))
    return torch.cat([p[0] for p in parts], dim=1).contiguous(), rows * len(parts), group


def _conv2d_fwd_one(x, w_raw, bias_raw, y, kh, kw, stride, pad, dil, accumulate=False, want_stats=False, precision=None,
                    amax_x=None, amax_w=None):
    """y = conv(x, w) (+bias).  w_raw is [Cout][kh][kw][Cin] with Cin == x.C, Cout == y.C.
    Returns (stat[3][rows][Cout], rows, group) when want_stats (fused into the epilogue when the plan allows,
    otherwise a separate column-statistics pass), else None."""
    Cout, Cin = y.C, x.C
    assert w_raw.numel() == Cout * kh * kw * Cin and w_raw.is_contiguous()
    assert y.B == x.B and y.H == conv_out_size(x.H, kh, stride, pad, dil) and y.W == conv_out_size(x.W, kw, stride, pad, dil)
    dev = x.device
    if x.half:
        # half-precision path: fp16 operands, one fp16 MFMA pass, fp32 accumulate; y fp16 (or fp32: the class logits)
        assert w_raw.dtype == torch.float16, 'fp16 activations need the fp16 filter copy (ParamArena.prepare_half)'
        geo = (x.B, y.H, y.W, Cin, Cout, kh, kw, stride, pad, dil)
        st, rows, group = None, 0, 0
        if want_stats:
            rows = _lib.query('pseg_conv2d_stat_rows_h', *geo)
            group = _lib.query('pseg_conv2d_stat_group_h', *geo)
            st = torch.empty(3, rows, Cout, dtype=torch.float32, device=dev)
        _lib.call('pseg_conv2d_fwd_h', x.ptr, x.ld, w_raw.data_ptr(), _ptr(bias_raw), y.ptr, y.ld, int(not y.half), x.B,
                  x.H, x.W, Cin, y.H, y.W, Cout, kh, kw, stride, pad, dil, int(accumulate), _ptr(st), _stream())
        return (st, rows, group) if want_stats else None
    ws_bytes = _lib.query('pseg_conv2d_fwd_workspace_bytes', x.B, y.H, y.W, Cin, Cout, kh, kw)
    ws = workspace.get(ws_bytes, dev) if ws_bytes else None
    fused = want_stats and ws_bytes == 0
    st = None
    rows = group = 0
    if fused:
        rows = _lib.query('pseg_conv2d_stat_rows', x.B, y.H, y.W, Cin, Cout, kh, kw, stride, pad, dil)
        group = _lib.query('pseg_conv2d_stat_group', x.B, y.H, y.W, Cin, Cout, kh, kw, stride, pad, dil)
        st = torch.empty(3, rows, Cout, dtype=torch.float32, device=dev)
    _lib.call('pseg_conv2d_fwd', x.ptr, x.ld, w_raw.data_ptr(), _ptr(bias_raw), y.ptr, y.ld, x.B, x.H, x.W, Cin,
              y.H, y.W, Cout, kh, kw, stride, pad, dil, int(accumulate), _fwd_prec(precision, amax_x, amax_w),
              _ptr(amax_x), _ptr(amax_w), _ptr(st), _ptr(ws), ws_bytes, _stream())
    if want_stats and not fused:
        return col_stats(y)
    return (st, rows, group) if want_stats else None


def filter_transpose(w_raw, Cout, taps, Cin):
    wT = torch.empty_like(w_raw)
    _lib.call('pseg_filter_transpose', w_raw.data_ptr(), wT.data_ptr(), Cout, taps, Cin, _stream())
    return wT


def _fwd_prec(precision, amax_a, amax_b, backward=False):
    """fp16 limbs need both per-tensor maxima; without them the call runs on the exact fp32 kernel."""
    pr = _prec(precision, backward)
    if pr == PREC_FP16X3 and (amax_a is None or amax_b is None):
        return PREC_FP32
    return pr


class BnPart:
    """[2][rows][C] partial sums (dbeta, dgamma) of a BatchNorm backward, written by the data gradient that produced dz;
    `key` names the layer they belong to (address of its saved y)."""
    __slots__ = ('part', 'rows', 'key')

    def __init__(self, part, rows, key):
        self.part, self.rows, self.key = part, rows, key


FUSE_BN_BWD = os.environ.get('PSEG_FUSE_BN_BWD', '1') == '1'
# ... under the half-precision policy: OFF by default.  Built, parity-tested (tests/test_half_gpu.py) and measured SLOWER in the
# step: DeepLabV3+ -mp 13.85-13.93 ms without, 13.96-14.04 with (32 reduction launches of ~12 us gone, the 32 data gradients that
# carry their sums instead each a few us longer -- and under -mp the data-gradient chain IS the critical path;
# profiles/EXPERIMENTS.md 5.8).  PSEG_FUSE_BN_BWD_H=1 switches it on.
FUSE_BN_BWD_H = os.environ.get('PSEG_FUSE_BN_BWD_H', '0') == '1'


_WARNED_ONCE = set()


def _warn_once(key, msg):
    if key not in _WARNED_ONCE:
        _WARNED_ONCE.add(key)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def conv2d_dgrad(dy, wT_raw, dx, kh, kw, stride, pad, dil, accumulate=False, precision=None, amax_dy=None,
                 amax_w=None, bn=None):
    """dx (+)= conv_transpose(dy, w); a dy above the 2 GiB descriptor cap runs in batch chunks (without the fused
    BatchNorm-backward sums: dx.bnpart stays None and bn_act_bwd takes its own reduction pass)."""
    nb = _batch_chunks(dy)
    if nb == dy.B:
        return _conv2d_dgrad_one(dy, wT_raw, dx, kh, kw, stride, pad, dil, accumulate, precision, amax_dy, amax_w, bn)
    for b0 in range(0, dy.B, nb):
        _conv2d_dgrad_one(_sub(dy, b0, nb), wT_raw, _sub(dx, b0, nb), kh, kw, stride, pad, dil, accumulate, precision, amax_dy,
                          amax_w, None)
    dx.bnpart = None


def _conv2d_dgrad_one(dy, wT_raw, dx, kh, kw, stride, pad, dil, accumulate=False, precision=None, amax_dy=None,
                      amax_w=None, bn=None):
    """dx (+)= conv_transpose(dy, w); wT_raw is the [Cin][kh][kw][Cout] transposed filter.
    bn = (y, co, act) of the BatchNorm + activation layer that produced the conv's input, when dx is that layer's dz and
    nothing else adds to it: the kernel then also writes the layer's backward partial sums (dx.bnpart; pseg_conv2d_dgrad_bnstat /
    _bnstat_h) where the problem runs on a kernel that can (exact fp32: the LDS-DMA tiles; fp16: every gather kernel) -- bn_act_bwd
    skips its reduction pass."""
    Cout, Cin = dy.C, dx.C
    assert wT_raw.numel() == Cout * kh * kw * Cin
    if bn is not None and FUSE_BN_BWD and not dy.half and not accumulate and _prec(precision, True) == PREC_FP32:
        rows = _lib.query('pseg_conv2d_dgrad_bnstat_rows', dx.B, dx.H, dx.W, Cin, dy.H, dy.W, Cout, kh, kw, stride, pad, dil)
        y, co, act = bn
        if rows > 0 and y.C == Cin and y.M == dx.M and not y.half:
            part = torch.empty(2, rows, Cin, dtype=torch.float32, device=dx.device)
            c0, cs = co.data_ptr(), co.shape[1] * 4
            p0 = part.data_ptr()
            try:
                _lib.call('pseg_conv2d_dgrad_bnstat', dy.ptr, dy.ld, wT_raw.data_ptr(), dx.ptr, dx.ld, dx.B, dx.H, dx.W, Cin, dy.H,
                          dy.W, Cout, kh, kw, stride, pad, dil, y.ptr, y.ld, c0, c0 + cs, c0 + 2 * cs, c0 + 3 * cs, act,
                          p0, p0 + rows * Cin * 4, rows, _stream())
                dx.bnpart = BnPart(part, rows, y.ptr)
                return
            except _lib.PsegError as e:
                # The rows query (dgrad_bnstat_plan) and the launch (run_gather) derive the kernel choice separately:
                # should they ever disagree for a shape, the library refuses BEFORE launching anything -- take the unfused path
                # (plain data gradient + bn_bwd_reduce) instead of aborting the step, and say so once.
                _warn_once('dgrad_bnstat', 'pseg_conv2d_dgrad_bnstat refused a shape its rows query accepted (%s); '
                                           'using the unfused data gradient for it' % e)
                dx.bnpart = None
    if dy.half:
        assert wT_raw.dtype == torch.float16 and dx.half
        if bn is not None and FUSE_BN_BWD_H and not accumulate:
            y, co, act = bn
            rows = _lib.query('pseg_conv2d_dgrad_bnstat_rows_h', dx.B, dx.H, dx.W, Cin, dy.H, dy.W, Cout, kh, kw, stride, pad, dil)
            if rows > 0 and y.C == Cin and y.M == dx.M and y.half:
                part = torch.empty(2, rows, Cin, dtype=torch.float32, device=dx.device)
                c0, cs = co.data_ptr(), co.shape[1] * 4
                p0 = part.data_ptr()
                _lib.call('pseg_conv2d_dgrad_bnstat_h', dy.ptr, dy.ld, wT_raw.data_ptr(), dx.ptr, dx.ld, dx.B, dx.H, dx.W, Cin,
                          dy.H, dy.W, Cout, kh, kw, stride, pad, dil, y.ptr, y.ld, c0, c0 + cs, c0 + 2 * cs, c0 + 3 * cs, act,
                          p0, p0 + rows * Cin * 4, rows, _stream())
                dx.bnpart = BnPart(part, rows, y.ptr)
                return
        _lib.call('pseg_conv2d_dgrad_h', dy.ptr, dy.ld, wT_raw.data_ptr(), dx.ptr, dx.ld, dx.B, dx.H, dx.W, Cin, dy.H, dy.W,
                  Cout, kh, kw, stride, pad, dil, int(accumulate), _stream())
        return
    # the dgrad GEMM has M = input pixels, N = Cin, K = kh*kw*Cout
    ws_bytes = _lib.query('pseg_conv2d_fwd_workspace_bytes', dx.B, dx.H, dx.W, Cout, Cin, kh, kw)
    ws = workspace.get(ws_bytes, dx.device) if ws_bytes else None
    _lib.call('pseg_conv2d_dgrad', dy.ptr, dy.ld, wT_raw.data_ptr(), dx.ptr, dx.ld, dx.B, dx.H, dx.W, Cin, dy.H,
              dy.W, Cout, kh, kw, stride, pad, dil, int(accumulate), _fwd_prec(precision, amax_dy, amax_w, True),
              _ptr(amax_dy), _ptr(amax_w), _ptr(ws), ws_bytes, _stream())


class Planes:
    """bf16 limb planes of an fp32 [M][C] tensor: hi = bf16(x), lo = bf16(x - hi), each [M][ldp] (int16 storage)."""
    __slots__ = ('hi', 'lo', 'ldp', 'M', 'C')

    def __init__(self, hi, lo, ldp, M, C):
        self.hi, self.lo, self.ldp, self.M, self.C = hi, lo, ldp, M, C


def split_planes(x):
    """Act or contiguous 2-D-like torch tensor ([rows][C] fp32) -> Planes (one HBM-bound pass)."""
    if isinstance(x, Act):
        ptr, ld, M, C, dev = x.ptr, x.ld, x.M, x.C, x.device
    else:
        assert x.is_contiguous() and x.dtype == torch.float32
        M, C = x.shape[0], x.numel() // x.shape[0]
        ptr, ld, dev = x.data_ptr(), C, x.device
    ldp = (C + 7) // 8 * 8
    hi = torch.empty(M * ldp, dtype=torch.int16, device=dev)
    lo = torch.empty(M * ldp, dtype=torch.int16, device=dev)
    _lib.call('pseg_split_planes', ptr, ld, M, C, hi.data_ptr(), lo.data_ptr(), ldp, _stream())
    return Planes(hi, lo, ldp, M, C)


def dgrad_planes_ok(dy, dx, kh, kw, stride, pad, dil):
    """True when the pre-split LDS-DMA limb kernel covers this data gradient (see pseg_conv2d_dgrad_planes_ok)."""
    return bool(_lib.query('pseg_conv2d_dgrad_planes_ok', dx.B, dx.H, dx.W, dx.C, dy.H, dy.W, dy.C, kh, kw, stride, pad, dil))


def dgrad_planes_ok_shape(B, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, dil):
    return bool(_lib.query('pseg_conv2d_dgrad_planes_ok', B, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, dil))


def conv2d_dgrad_planes(dy_planes, dy, wT_planes, dx, kh, kw, stride, pad, dil, accumulate=False):
    """dx (+)= conv_transpose(dy, w) in BF16X3 arithmetic from pre-split operands (dy: Act giving the geometry)."""
    _lib.call('pseg_conv2d_dgrad_planes', dy_planes.hi.data_ptr(), dy_planes.lo.data_ptr(), dy_planes.ldp,
              wT_planes.hi.data_ptr(), wT_planes.lo.data_ptr(), dx.ptr, dx.ld, dx.B, dx.H, dx.W, dx.C, dy.H, dy.W, dy.C,
              kh, kw, stride, pad, dil, int(accumulate), _stream())


class SlabPool:
    """Deferred reduction of split weight gradients: every split conv2d_wgrad of a backward pass leaves its slabs in a
    region owned by this pool (one per (gradient tensor, geometry), allocated once and kept), and `reduce()` folds them
    all into the gradient arena with ONE launch (pseg_slab_reduce_batch) instead of one launch-bound reduction per
    layer.  The job table lives on the device and is rebuilt only when the sequence of jobs of a pass changes (first
    step), so a pass can be captured in a hipGraph."""

    def __init__(self, device):
        self.device = device
        self.regions = {}        # key -> (slab tensor, elements per slab, splits, gradient tensor)
        self.pending = []        # keys of this pass, in call order
        # device job tables, one per job sequence ever seen, NEVER freed: a captured step has its table's address (and the
        # job count / block count of that sequence) baked into its slab_reduce_batch launch
        self.tables = {}         # tuple(keys) -> (device table, blocks)
        self.block = _lib.query('pseg_slab_reduce_block')

    def region(self, dw_raw, geom, splits):
        key = (dw_raw.data_ptr(), geom, splits)
        hit = self.regions.get(key)
        if hit is None:
            hit = self.regions[key] = (torch.empty(splits * dw_raw.numel(), dtype=torch.float32, device=self.device),
                                       dw_raw.numel(), splits, dw_raw)
        if key in self.pending:      # the same gradient twice in one pass: fold what is there first
            return None
        self.pending.append(key)
        return hit[0]

    def reduce(self, accumulate=False):
        """Fold every pending slab set into its gradient (dw = or += sum of slabs); call on the stream that joined the
        weight-gradient stream, before anything reads the gradients."""
        if not self.pending:
            return
        seq = tuple(self.pending)
        hit = self.tables.get(seq)
        if hit is None:
            if CAPTURING:
                # (the upload below is a pageable host-to-device copy: it would abort the capture.  The Trainer runs the
                # first step of every shape eagerly under the capture's own configuration, so this cannot happen there)
                raise RuntimeError('SlabPool: a new job sequence inside a stream capture; run one eager pass of this '
                                   'shape with the same Env first')
            rows, first = [], 0
            for key in self.pending:
                slabs, elems, splits, dw = self.regions[key]
                assert elems % 4 == 0
                rows.append([slabs.data_ptr(), dw.data_ptr(), elems, splits, first])
                first += (elems + self.block - 1) // self.block
            hit = self.tables[seq] = (torch.tensor(rows, dtype=torch.int64).to(self.device), first)
        table, blocks = hit
        _lib.call('pseg_slab_reduce_batch', table.data_ptr(), len(self.pending), blocks, int(accumulate), _stream())
        self.pending = []


def conv2d_wgrad(x, dy, dw_raw, kh, kw, stride, pad, dil, accumulate=False, precision=None, pool=None, concurrent=False):
    """dw (+)= x^T * dy; operands above the 2 GiB descriptor cap run in batch chunks that ADD into dw in chunk order (no slab
    pool for them: every chunk reduces its own slabs)."""
    nb = _batch_chunks(x, dy)
    if nb == x.B:
        return _conv2d_wgrad_one(x, dy, dw_raw, kh, kw, stride, pad, dil, accumulate, precision, pool, concurrent)
    for b0 in range(0, x.B, nb):
        _conv2d_wgrad_one(_sub(x, b0, nb), _sub(dy, b0, nb), dw_raw, kh, kw, stride, pad, dil, accumulate or b0 > 0, precision,
                          None, concurrent)


def _conv2d_wgrad_one(x, dy, dw_raw, kh, kw, stride, pad, dil, accumulate=False, precision=None, pool=None, concurrent=False):
    """pool (SlabPool): a split plan leaves its slabs with the pool -- `accumulate` is then the POOL's business
    (pool.reduce(accumulate)), and dw_raw is complete only after that call.
    concurrent: the launch runs beside another stream's kernels (Conv2d.bwd forks it onto the weight-gradient stream): the
    exact-fp32 plan then keeps ONE resident block per CU (half the slabs); alone it is planned for two."""
    cc = int(bool(concurrent))
    Cout, Cin = dy.C, x.C
    assert dw_raw.numel() == Cout * kh * kw * Cin and dw_raw.is_contiguous()
    if x.half:
        # fp16 operands, fp32 gradient (the master gradient arena); same split / slab protocol as the fp32 path
        assert dy.half and dw_raw.dtype == torch.float32
        if pool is not None:
            splits = _lib.query('pseg_conv2d_wgrad_splits_h', x.B, dy.H, dy.W, Cin, Cout, kh, kw)
            if splits > 1:
                slabs = pool.region(dw_raw, (x.B, x.H, x.W, dy.H, dy.W, kh, kw, stride, pad, dil, 'h'), splits)
                if slabs is not None:
                    _lib.call('pseg_conv2d_wgrad_slabs_h', x.ptr, x.ld, dy.ptr, dy.ld, slabs.data_ptr(), x.B, x.H, x.W, Cin,
                              dy.H, dy.W, Cout, kh, kw, stride, pad, dil, slabs.numel() * 4, _stream())
                    return
                pool.reduce(accumulate)
                accumulate = True
        ws_bytes = _lib.query('pseg_conv2d_wgrad_workspace_bytes_h', x.B, dy.H, dy.W, Cin, Cout, kh, kw)
        ws = workspace.get(ws_bytes, x.device) if ws_bytes else None
        _lib.call('pseg_conv2d_wgrad_h', x.ptr, x.ld, dy.ptr, dy.ld, dw_raw.data_ptr(), x.B, x.H, x.W, Cin, dy.H, dy.W,
                  Cout, kh, kw, stride, pad, dil, int(accumulate), _ptr(ws), ws_bytes, _stream())
        return
    if pool is not None:
        prec = _prec(precision, True)
        splits = _lib.query('pseg_conv2d_wgrad_splits', x.B, dy.H, dy.W, Cin, Cout, kh, kw, prec, cc)
        if splits > 1:
            slabs = pool.region(dw_raw, (x.B, x.H, x.W, dy.H, dy.W, kh, kw, stride, pad, dil, prec, cc), splits)
            if slabs is not None:
                _lib.call('pseg_conv2d_wgrad_slabs', x.ptr, x.ld, dy.ptr, dy.ld, slabs.data_ptr(), x.B, x.H, x.W, Cin,
                          dy.H, dy.W, Cout, kh, kw, stride, pad, dil, prec, cc, slabs.numel() * 4, _stream())
                return
            # a second gradient for the same filter in one pass (shared weights): fold what is parked, then add
            pool.reduce(accumulate)
            accumulate = True
    ws_bytes = _lib.query('pseg_conv2d_wgrad_workspace_bytes', x.B, dy.H, dy.W, Cin, Cout, kh, kw)
    ws = workspace.get(ws_bytes, x.device) if ws_bytes else None
    _lib.call('pseg_conv2d_wgrad', x.ptr, x.ld, dy.ptr, dy.ld, dw_raw.data_ptr(), x.B, x.H, x.W, Cin, dy.H, dy.W,
              Cout, kh, kw, stride, pad, dil, int(accumulate), _prec(precision, True), cc, _ptr(ws), ws_bytes, _stream())


def _h(name, act):
    """entry point for the activation's storage type: the fp16 instantiation carries the suffix _h"""
    return name + '_h' if act.t.dtype == torch.float16 else name


def dwconv_fwd(x, w_raw, y, k, stride, pad):
    _lib.call(_h('pseg_dwconv_fwd', x), x.ptr, x.ld, w_raw.data_ptr(), y.ptr, y.ld, x.B, x.H, x.W, x.C, y.H, y.W, k, stride,
              pad, _stream())


def dwconv_dgrad(dy, w_raw, dx, k, stride, pad):
    _lib.call(_h('pseg_dwconv_dgrad', dy), dy.ptr, dy.ld, w_raw.data_ptr(), dx.ptr, dx.ld, dx.B, dx.H, dx.W, dx.C, dy.H, dy.W,
              k, stride, pad, _stream())


def dwconv_wgrad(x, dy, dw_raw, k, stride, pad, accumulate=False):
    ws_bytes = _lib.query('pseg_dwconv_wgrad_workspace_bytes', x.B, dy.H, dy.W, x.C, k)
    ws = workspace.get(ws_bytes, x.device)
    _lib.call(_h('pseg_dwconv_wgrad', x), x.ptr, x.ld, dy.ptr, dy.ld, dw_raw.data_ptr(), x.B, x.H, x.W, x.C, dy.H, dy.W, k,
              stride, pad, int(accumulate), ws.data_ptr(), ws_bytes, _stream())


# ---------------------------------------------------------------------------------------------- batch norm
def col_stats(y):
    """-> (stat[3][rows][C] = pivot / shifted sum / shifted sum of squares per row group, rows, group)."""
    rows = _lib.query('pseg_col_stats_rows', y.M, y.C)
    st = torch.empty(3, rows, y.C, dtype=torch.float32, device=y.device)
    _lib.call(_h('pseg_col_stats', y), y.ptr, y.ld, y.M, y.C, st.data_ptr(), _stream())
    return st, rows, _lib.query('pseg_col_stats_group', y.M, y.C)


def bn_finalize(stats, count, gamma, beta, running_mean, running_var, momentum, eps):
    """-> coeff tensor [4][C]: mean, invstd, scale = gamma*invstd, beta (and updates the running statistics)."""
    st, rows, group = stats
    C = st.shape[-1]
    co = torch.empty(4, C, dtype=torch.float32, device=st.device)
    ws_bytes = _lib.query('pseg_bn_finalize_workspace_bytes', rows, C)
    ws = workspace.get(ws_bytes, st.device) if ws_bytes else None
    base, step = co.data_ptr(), C * 4      # (row pointers by arithmetic: indexing a tensor costs ~2 us a time)
    _lib.call('pseg_bn_finalize', st.data_ptr(), rows, group, count, C, _ptr(gamma), _ptr(beta),
              _ptr(running_mean), _ptr(running_var), float(momentum), float(eps), base, base + step, base + 2 * step,
              base + 3 * step, _ptr(ws), ws_bytes, _stream())
    return co


BN_SMALL_IN_GRAPH = os.environ.get('PSEG_BN_SMALL_GRAPH', '0') == '1'
CAPTURING = 0     # > 0 while a Trainer step is being captured for replay (utils/trainer.py)
EVER_CAPTURED = False   # a captured step exists (or existed): device buffers whose addresses it baked in are never freed


class no_gc_capture:
    """``with no_gc_capture(graph): ...`` = ``torch.cuda.graph(graph, capture_error_mode='thread_local')`` with the garbage
    collector held off while the capture is open.  A collection that happens to run between two captured launches finalizes
    whatever cycles earlier steps left behind (trainers, graphs, streams, events); a destructor that makes a synchronising
    HIP call inside a capture aborts the process (seen in rounds 2 and 3: a `Stream` / `Event` finalizer; torch.cuda.graph
    itself collects once BEFORE the capture begins, not during).  Every capture of the package and its tests goes through
    here."""

    def __init__(self, graph, **kw):
        kw.setdefault('capture_error_mode', 'thread_local')
        self._ctx = torch.cuda.graph(graph, **kw)
        # torch's default capture stream is one of its 32 pool streams too: should it BE one of the streams this package forks onto
        # (new_stream), the capture gets a stream of its own.  Only then -- a different capture stream moves the hardware queues the
        # replay's lanes land on: HRNet replayed 14.0 -> 14.7 ms fp32, 7.2 -> 7.75 ms -mp with an own capture stream always
        # (profiles/r06_ab_streams.txt).
        cap = getattr(self._ctx, 'capture_stream', None)
        if OWN_STREAMS and 'stream' not in kw and cap is not None:
            idx = cap.device.index if cap.device.index is not None else torch.cuda.current_device()
            if cap.stream_id in _own_stream_handles.get(idx, ()):
                kw['stream'] = new_stream(cap.device)
                self._ctx = torch.cuda.graph(graph, **kw)
        self._gc_was_on = False

    def __enter__(self):
        import gc
        self._gc_was_on = gc.isenabled()
        gc.disable()
        try:
            return self._ctx.__enter__()
        except BaseException:
            if self._gc_was_on:
                gc.enable()
            raise

    def __exit__(self, *exc):
        import gc
        try:
            return self._ctx.__exit__(*exc)
        finally:
            if self._gc_was_on:
                gc.enable()


def bn_small_path(rows, M, C):
    """True when the library recommends the fused finalize + apply launches (small tensors, launch-bound regime).
    Not while a step is being captured: a replayed step pays no host time per launch, and on the device the two plain
    launches are faster than the fused one (every block of which re-derives the coefficients): UNet 5.2 -> 5.0 ms."""
    return (CAPTURING == 0 or BN_SMALL_IN_GRAPH) and bool(_lib.query('pseg_bn_small_path', rows, M, C))


def bn_fwd_fused(stats, count, gamma, beta, running_mean, running_var, momentum, eps, y, act, z, residual=None):
    """bn_finalize + bn_act_fwd in ONE launch (small tensors): -> coeff tensor [4][C]; z is written."""
    st, rows, group = stats
    C = st.shape[-1]
    co = torch.empty(4, C, dtype=torch.float32, device=st.device)
    base, step = co.data_ptr(), C * 4
    args = (st.data_ptr(), rows, group, count, C, _ptr(gamma), _ptr(beta), _ptr(running_mean),
            _ptr(running_var), float(momentum), float(eps), base, base + step, base + 2 * step, base + 3 * step,
            y.ptr, y.ld, residual.ptr if residual is not None else 0, residual.ld if residual is not None else 0, act,
            z.ptr, z.ld, y.M)
    if y.half:
        _lib.call('pseg_bn_fwd_fused_h', *args, _stream())
    else:
        _lib.call('pseg_bn_fwd_fused', *args, _ptr(z.amax), _stream())
    return co


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps):
    C = running_mean.numel()
    co = torch.empty(4, C, dtype=torch.float32, device=running_mean.device)
    _lib.call('pseg_bn_eval_coeffs', _ptr(gamma), _ptr(beta), running_mean.data_ptr(), running_var.data_ptr(),
              float(eps), C, co[0].data_ptr(), co[1].data_ptr(), co[2].data_ptr(), co[3].data_ptr(), _stream())
    return co


def track_amax():
    """True when the forward conv policy needs per-tensor maxima (fp16 limbs)."""
    return FWD_PRECISION == PREC_FP16X3


def raise_amax(dst, src):
    """dst.amax = max(dst.amax, src.amax) on the device (bound propagation through max-preserving ops)."""
    if dst.amax is not None and src.amax is not None and dst.amax is not src.amax:
        torch.maximum(dst.amax, src.amax, out=dst.amax)


def bn_act_fwd(y, co, act, z, residual=None, want_mask=False):
    """z = act((y - mean)*scale + beta (+ residual)); co None -> plain activation / residual add.
    When z carries an amax scalar the kernel raises it to max|z|.
    want_mask (C % 32 == 0, an activation): also returns the activation bitmask [M][C/32] (int32) for bn_act_bwd."""
    assert z.M == y.M and z.C == y.C
    mu = sc = sh = 0
    if co is not None:
        mu, step = co.data_ptr(), co.shape[1] * 4
        sc, sh = mu + 2 * step, mu + 3 * step
    mask = torch.empty(y.M * (y.C // 32), dtype=torch.int32, device=y.device) \
        if (want_mask and act != ACT_NONE and y.C % 32 == 0) else None
    args = (y.ptr, y.ld, mu, sc, sh, residual.ptr if residual is not None else 0,
            residual.ld if residual is not None else 0, act, z.ptr, z.ld, y.M, y.C)
    if y.half:
        _lib.call('pseg_bn_act_fwd_h', *args, _ptr(mask), _stream())
    else:
        _lib.call('pseg_bn_act_fwd', *args, _ptr(z.amax), _ptr(mask), _stream())
    return mask


def bn_act_bwd(dz, z, y, co, act, dy, gamma_grad, beta_grad, accumulate=False, dres=None, res_accumulate=False,
               frozen=False, mask=None, want_planes=False, part=None):
    """Backward through act(BN(y) (+res)).  Writes dy, (+)= dgamma/dbeta, optional dres.
    z=None (allowed when the forward had no residual): the activation mask is recomputed from y.
    mask (bn_act_fwd(want_mask=True)): the large-tensor passes read this bitmask instead of z.
    want_planes (C % 8 == 0, dense dy, large-tensor path): the apply pass also writes dy as bf16 limb planes -> dy.planes.
    frozen: eval-mode BatchNorm (co from bn_eval_coeffs) -- the statistics are constants, dy = scale * dz * act'."""
    C, M, dev = y.C, y.M, y.device
    fused = part is not None and part.key == y.ptr and z is None and mask is None and tuple(part.part.shape[::2]) == (2, C)
    if fused:       # the data gradient that produced dz left the partial sums behind (conv2d_dgrad(bn=...)): no reduction pass
        rows, part = part.rows, part.part
    else:
        rows = _lib.query('pseg_col_stats_rows', M, C)
        part = torch.empty(2, rows, C, dtype=torch.float32, device=dev)
    zp, zld = (z.ptr, z.ld) if z is not None else (0, 0)
    # row pointers by arithmetic (indexing a tensor costs ~2 us a time, and this function runs once per layer and step)
    c0, cs = co.data_ptr(), co.shape[1] * 4
    c1, c2, c3 = c0 + cs, c0 + 2 * cs, c0 + 3 * cs
    p0 = part.data_ptr()
    p1 = p0 + rows * C * 4
    dzp, dzl, yp, yl, dyp, dyl = dz.ptr, dz.ld, y.ptr, y.ld, dy.ptr, dy.ld
    drp, drl = (dres.ptr, dres.ld) if dres is not None else (0, 0)
    st = _stream()
    if not fused:
        _lib.call(_h('pseg_bn_act_bwd_reduce', y), dzp, dzl, zp, zld, yp, yl, c0, c1, c2, c3, act, M, C, p0, p1, _ptr(mask), st)
    if bn_small_path(rows, M, C):      # finalize folded into the apply pass: one launch fewer
        _lib.call(_h('pseg_bn_bwd_fused', y), p0, p1, rows, M, C, _ptr(gamma_grad), _ptr(beta_grad), int(accumulate),
                  int(frozen), dzp, dzl, zp, zld, yp, yl, c0, c1, c2, c3, act, dyp, dyl, drp, drl, int(res_accumulate), M,
                  st)
        return
    cc = torch.empty(2, C, dtype=torch.float32, device=dev)
    k0 = cc.data_ptr()
    k1 = k0 + C * 4
    _lib.call('pseg_bn_bwd_finalize', p0, p1, rows, M, C, _ptr(gamma_grad), _ptr(beta_grad), int(accumulate),
              int(frozen), k0, k1, st)
    if y.half:
        _lib.call('pseg_bn_act_bwd_apply_h', dzp, dzl, zp, zld, yp, yl, c0, c1, c2, c3, k0, k1, act, dyp, dyl, drp, drl,
                  int(res_accumulate), M, C, _ptr(mask), st)
        return
    hi = lo = None
    if want_planes and C % 8 == 0 and dy.ld == C:
        hi = torch.empty(M * C, dtype=torch.int16, device=dev)
        lo = torch.empty(M * C, dtype=torch.int16, device=dev)
        dy.planes = Planes(hi, lo, C, M, C)
    _lib.call('pseg_bn_act_bwd_apply', dzp, dzl, zp, zld, yp, yl, c0, c1, c2, c3, k0, k1, act, dyp, dyl, drp, drl,
              int(res_accumulate), M, C, _ptr(mask), _ptr(hi), _ptr(lo), C, st)


def act_bwd(dz, z, act, dy, scale=None, dres=None, res_accumulate=False):
    """dy = scale * dz * act'(z) (eval-mode BN / plain activation backward); optional dres = dz * act'(z)."""
    _lib.call(_h('pseg_act_bwd', dz), dz.ptr, dz.ld, z.ptr if z is not None else 0, z.ld if z is not None else 0, _ptr(scale),
              act, dy.ptr if dy is not None else 0, dy.ld if dy is not None else 0,
              dres.ptr if dres is not None else 0, dres.ld if dres is not None else 0, int(res_accumulate), dz.M, dz.C,
              _stream())


def col_sum(dy, out, accumulate=False, C=None):
    C = dy.C if C is None else C
    rows = _lib.query('pseg_col_stats_rows', dy.M, C)
    nbytes = rows * C * 4
    ws = workspace.get(nbytes, dy.device)
    _lib.call(_h('pseg_col_sum', dy), dy.ptr, dy.ld, dy.M, C, out.data_ptr(), int(accumulate), ws.data_ptr(), nbytes, _stream())


def copy2d(x, y, accumulate=False):
    assert x.M == y.M and x.C == y.C and x.dtype == y.dtype
    _lib.call(_h('pseg_copy2d', x), x.ptr, x.ld, y.ptr, y.ld, x.M, x.C, int(accumulate), _stream())


# ---------------------------------------------------------------------------------------------- pool / resize
def pool_sum(x, out, scale):
    """out[b,0,0,c] = scale * sum over pixels (out is an Act with H = W = 1)."""
    assert out.B == x.B and out.H == 1 and out.W == 1 and out.C == x.C
    _lib.call(_h('pseg_pool_sum', x), x.ptr, x.ld, x.B, x.H * x.W, x.C, float(scale), out.ptr, out.ld, _stream())


def broadcast(x, y, scale=1.0, accumulate=False):
    """y[b,h,w,c] (+)= scale * x[b,0,0,c]."""
    assert x.H == 1 and x.W == 1 and x.B == y.B and x.C == y.C
    _lib.call(_h('pseg_broadcast', x), x.ptr, x.ld, y.B, y.H * y.W, y.C, float(scale), y.ptr, y.ld, int(accumulate), _stream())


def bilinear_fwd(x, y, align_corners):
    assert x.B == y.B and x.C == y.C and x.dtype == y.dtype
    if x.half:
        _lib.call('pseg_bilinear_fwd_h', x.ptr, x.ld, x.B, x.H, x.W, x.C, y.ptr, y.ld, y.H, y.W, int(align_corners), _stream())
        return
    _lib.call('pseg_bilinear_fwd', x.ptr, x.ld, x.B, x.H, x.W, x.C, y.ptr, y.ld, y.H, y.W, int(align_corners), 0, _stream())


def bilinear_fwd_nchw(x, C, Ho, Wo, align_corners):
    """NHWC handle -> contiguous fp32 NCHW torch tensor [B,C,Ho,Wo] (first C channels)."""
    if x.half:
        x = x.to(torch.float32)
    out = torch.empty(x.B, C, Ho, Wo, dtype=torch.float32, device=x.device)
    _lib.call('pseg_bilinear_fwd', x.ptr, x.ld, x.B, x.H, x.W, C, out.data_ptr(), 0, Ho, Wo, int(align_corners), 1, _stream())
    return out


def bilinear_bwd(dy, dx, align_corners, accumulate=False):
    assert dx.B == dy.B and dx.C == dy.C and dx.dtype == dy.dtype
    if dy.half:
        _lib.call('pseg_bilinear_bwd_h', dy.ptr, dy.ld, dx.B, dx.H, dx.W, dx.C, dx.ptr, dx.ld, dy.H, dy.W, int(align_corners),
                  int(accumulate), _stream())
        return
    _lib.call('pseg_bilinear_bwd', dy.ptr, dy.ld, dx.B, dx.H, dx.W, dx.C, dx.ptr, dx.ld, dy.H, dy.W, int(align_corners), 0,
              int(accumulate), 0, 0, _stream())


def bilinear_bwd_nchw(dy_nchw, dx, C, align_corners, accumulate=False):
    """Gradient w.r.t. the NHWC source from a contiguous NCHW gradient [B,C,Ho,Wo]."""
    assert dy_nchw.is_contiguous() and dy_nchw.shape[0] == dx.B and dy_nchw.shape[1] == C
    Ho, Wo = dy_nchw.shape[2], dy_nchw.shape[3]
    nbytes = _lib.query('pseg_bilinear_bwd_workspace_bytes', dx.B, dx.H, dx.W, C, Ho, Wo, 1)
    ws = workspace.get(nbytes, dx.device)
    _lib.call('pseg_bilinear_bwd', dy_nchw.data_ptr(), 0, dx.B, dx.H, dx.W, C, dx.ptr, dx.ld, Ho, Wo, int(align_corners), 1,
              int(accumulate), ws.data_ptr(), nbytes, _stream())


def maxpool_fwd(x, y, k, stride, pad, want_argmax=True):
    arg = torch.empty(y.M * y.C, dtype=torch.uint8, device=x.device) if want_argmax else None
    _lib.call(_h('pseg_maxpool_fwd', x), x.ptr, x.ld, x.B, x.H, x.W, x.C, y.ptr, y.ld, _ptr(arg), y.H, y.W, k, stride, pad, _stream())
    return arg


def bn_act_maxpool_fwd(x, co, act, y, k, stride, pad, want_argmax=True):
    """y = maxpool(act(BN(x))) with BN's coefficient tensor `co` ([4][C]: mean, invstd, scale, shift) -- the activated map never exists
    (pseg_bn_act_maxpool_fwd; same values and argmax as bn_act_fwd + maxpool_fwd)."""
    arg = torch.empty(y.M * y.C, dtype=torch.uint8, device=x.device) if want_argmax else None
    c0, cs = co.data_ptr(), co.shape[1] * 4
    _lib.call(_h('pseg_bn_act_maxpool_fwd', x), x.ptr, x.ld, c0, c0 + 2 * cs, c0 + 3 * cs, act, x.B, x.H, x.W, x.C, y.ptr, y.ld,
              _ptr(arg), y.H, y.W, k, stride, pad, _stream())
    return arg


def maxpool_bwd(dy, arg, dx, k, stride, pad, accumulate=False):
    _lib.call(_h('pseg_maxpool_bwd', dy), dy.ptr, dy.ld, arg.data_ptr(), dx.B, dx.H, dx.W, dx.C, dx.ptr, dx.ld, dy.H, dy.W, k,
              stride, pad, int(accumulate), _stream())


# ---------------------------------------------------------------------------------------------- loss / masks
def ce_fwd_bwd(logits, target, want_grad=True, ignore_index=-100):
    """logits: contiguous NCHW fp32 cuda, target: NHW int64.
    -> (loss_out[3] = [mean loss, n_valid, n_out_of_range_targets], dlogits|None)."""
    assert logits.is_cuda and logits.dtype == torch.float32 and logits.is_contiguous() and logits.dim() == 4
    assert target.dtype == torch.int64 and target.is_contiguous() and target.shape == (logits.shape[0],) + logits.shape[2:]
    B, C, H, W = logits.shape
    dl = torch.empty_like(logits) if want_grad else None
    out = torch.empty(3, dtype=torch.float32, device=logits.device)
    nbytes = _lib.query('pseg_ce_workspace_bytes', B * H * W)
    ws = workspace.get(nbytes, logits.device)
    _lib.call('pseg_ce_fwd_bwd', logits.data_ptr(), target.data_ptr(), B, C, H * W, ignore_index, _ptr(dl),
              out.data_ptr(), ws.data_ptr(), nbytes, _stream())
    return out, dl


def ce_upsampled_ok_shape(h, w, C, H, W, align_corners):
    return bool(_lib.query('pseg_ce_upsampled_ok', h, w, C, H, W, int(align_corners)))


def ce_upsampled_ok(lr, C, H, W, align_corners):
    return ce_upsampled_ok_shape(lr.H, lr.W, C, H, W, align_corners)


def ce_upsampled_fwd_bwd(lr, C, target, align_corners, want_grad=True, ignore_index=-100):
    """CrossEntropy(interpolate(logits, target.shape[1:], 'bilinear', align_corners), target) straight from the LOW-resolution
    NHWC logits `lr` (first C channels): -> (loss_out[3], dlr | None) with dlr an Act shaped like lr (gradient with respect
    to the low-resolution logits; padded channels zero).  The full-resolution logits never exist."""
    assert target.dtype == torch.int64 and target.is_contiguous() and target.dim() == 3 and target.shape[0] == lr.B
    assert not lr.half, 'the loss reads fp32 logits (the classifier conv writes them in fp32 under the half policy)'
    H, W = int(target.shape[1]), int(target.shape[2])
    dlr = Act.empty(lr.B, lr.H, lr.W, lr.C, lr.device) if want_grad else None
    out = torch.empty(3, dtype=torch.float32, device=lr.device)
    nbytes = _lib.query('pseg_ce_upsampled_workspace_bytes', lr.B, lr.H, lr.W)
    ws = workspace.get(nbytes, lr.device)
    _lib.call('pseg_ce_upsampled_fwd_bwd', lr.ptr, lr.ld, lr.B, lr.H, lr.W, C, target.data_ptr(), H, W, int(align_corners),
              ignore_index, dlr.ptr if dlr is not None else 0, dlr.ld if dlr is not None else 0, out.data_ptr(),
              ws.data_ptr(), nbytes, _stream())
    return out, dlr


def scale_inplace(x, gscale):
    """x *= gscale (0-dim / 1-element device tensor); a no-op on the device when gscale == 1."""
    assert x.is_contiguous() and gscale.numel() == 1 and gscale.dtype == torch.float32
    _lib.call('pseg_scale_inplace', x.data_ptr(), x.numel(), gscale.data_ptr(), _stream())


def argmax(logits):
    assert logits.is_cuda and logits.dtype == torch.float32 and logits.is_contiguous() and logits.dim() == 4
    B, C, H, W = logits.shape
    mask = torch.empty(B, H, W, dtype=torch.int64, device=logits.device)
    _lib.call('pseg_argmax', logits.data_ptr(), B, C, H * W, mask.data_ptr(), _stream())
    return mask


def confusion(pred, target, counters):
    """counters: int64 [3][C] (tp, fn, fp), accumulated in place."""
    assert pred.dtype == torch.int64 and target.dtype == torch.int64 and counters.dtype == torch.int64
    assert pred.is_contiguous() and target.is_contiguous() and counters.is_contiguous() and counters.shape[0] == 3
    _lib.call('pseg_confusion', pred.data_ptr(), target.data_ptr(), pred.numel(), counters.shape[1],
              counters.data_ptr(), _stream())


# ---------------------------------------------------------------------------------------------- optimiser
def sgd_step(param, grad, mbuf, lr, momentum, weight_decay, nesterov, grad_scale, first_step):
    _lib.call('pseg_sgd_step', param.data_ptr(), grad.data_ptr(), _ptr(mbuf), param.numel(), float(lr), float(momentum),
              float(weight_decay), int(nesterov), float(grad_scale), int(first_step), _stream())


def adam_step(param, grad, m, v, lr, beta1, beta2, eps, weight_decay, decoupled, grad_scale, step):
    _lib.call('pseg_adam_step', param.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), param.numel(), float(lr),
              float(beta1), float(beta2), float(eps), float(weight_decay), int(decoupled), float(grad_scale), int(step),
              _stream())


def fill(x, value):
    _lib.call('pseg_fill', x.data_ptr(), x.numel(), float(value), _stream())

"""Trainer / Fetcher: the runtime the reference imports from pytorch_modules.utils (train.py:14,39,55,61-81).

Contract kept (from the call sites): ``Trainer(model, fetcher, loss_fn=, workdir=, accumulate=, adam=, lr=, weights=,
resume=, mixed_precision=)``; attributes ``epoch``, ``metrics``, ``model``; methods ``step()`` (one epoch over the
fetcher) and ``save(best)``; checkpoint files hold ``{'model': state_dict, ...}`` (test.py:103-104).
``Fetcher(loader, post_fetch_fn)`` iterates ``(inputs, targets)`` on the device and exposes ``.loader``.

What is different underneath: parameters and gradients live in flat arenas, the optimiser is one fused HIP launch,
gradient accumulation is an accumulate flag on the weight-gradient kernels (no zeroing pass), and data-parallel
gradients are exchanged bucket-by-bucket on a side stream while backward runs (utils/dist.py).

Launch-bound configurations (UNet at 256x256 batch 8 enqueues ~900 kernels in 10 ms of host time for ~5 ms of GPU
work) can run the forward + loss + backward of a micro-step as ONE captured hipGraph: ``Trainer(..., graph=True)`` or
``PSEG_GRAPH=1``.  The step body is explicit kernel launches with no host synchronisation, so it captures as is; the
first batch of a given shape runs eagerly, the second is captured, later ones copy the batch into the graph's static
input buffers and replay.  The optimiser launch stays outside the graph (its learning rate is a by-value argument).
The optimiser hyper-parameters of the external Trainer are not observable from the reference tree; torch.optim
defaults are used (SGD: momentum 0.9, no weight decay; Adam: betas (0.9, 0.999)) and can be overridden.
"""
import os

import torch
import torch.distributed as dist

from .. import ops
from ..arena import prepare
from ..nn import BatchNorm2d, Conv2d, Env
from .dist import GradReducer, broadcast_buffers
from .loss import compute_loss as _default_loss


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError('pytorch_segmentation_amd needs a HIP device (no CPU fallback)')
    return torch.device('cuda', torch.cuda.current_device())


class Fetcher:
    def __init__(self, loader, post_fetch_fn=None, device=None):
        self.loader = loader
        self.post_fetch_fn = post_fetch_fn
        self.device = device

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        dev = self.device or _device()
        for batch in self.loader:
            batch = tuple(t.to(dev, non_blocking=True) if torch.is_tensor(t) else t for t in batch)
            if self.post_fetch_fn is not None:
                batch = self.post_fetch_fn(batch)
            yield batch


class FlatOptimizer:
    """SGD(momentum) / Adam over the flat parameter arena: one kernel launch per step."""

    def __init__(self, arena, adam=False, lr=1e-3, momentum=0.9, weight_decay=0.0, nesterov=False,
                 betas=(0.9, 0.999), eps=1e-8):
        self.arena, self.adam, self.lr = arena, adam, lr
        self.momentum, self.weight_decay, self.nesterov = momentum, weight_decay, nesterov
        self.betas, self.eps = betas, eps
        self.steps = 0
        # dynamic loss scaling of the half-precision policy (apex / torch.cuda.amp defaults)
        self.growth_factor, self.backoff_factor, self.growth_interval = 2.0, 0.5, 2000
        n, dev = arena.numel, arena.device
        if adam:
            self.m = torch.zeros(n, dtype=torch.float32, device=dev)
            self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        else:
            self.m = torch.zeros(n, dtype=torch.float32, device=dev) if momentum else None
            self.v = None

    def step(self, grad_scale=1.0, mp_state=None):
        """mp_state (half-precision policy): the device-resident loss-scale state.  The step is then: flag inf / nan
        gradients, update with grad_scale / loss_scale unless flagged, adjust the loss scale -- three launches, no host
        synchronisation (whether the step was applied is only known on the device: state[4] / state[5] count)."""
        a = self.arena
        self.steps += 1
        if mp_state is not None:
            from .. import _lib
            st, sp = ops._stream(), mp_state.data_ptr()
            _lib.call('pseg_mp_check', a.grads.data_ptr(), a.numel, sp, st)
            if self.adam:
                _lib.call('pseg_adam_step_mp', a.params.data_ptr(), a.grads.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                          a.numel, float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                          float(self.weight_decay), 0, float(grad_scale), sp, st)
            else:
                _lib.call('pseg_sgd_step_mp', a.params.data_ptr(), a.grads.data_ptr(), ops._ptr(self.m), a.numel,
                          float(self.lr), float(self.momentum), float(self.weight_decay), int(self.nesterov),
                          float(grad_scale), sp, st)
            _lib.call('pseg_mp_update', sp, float(self.growth_factor), float(self.backoff_factor), int(self.growth_interval),
                      1.0, float(2 ** 24), st)
            return
        if self.adam:
            ops.adam_step(a.params, a.grads, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps,
                          self.weight_decay, False, grad_scale, self.steps)
        else:
            ops.sgd_step(a.params, a.grads, self.m, self.lr, self.momentum, self.weight_decay, self.nesterov,
                         grad_scale, self.steps == 1)

    def state_dict(self):
        return {'adam': self.adam, 'steps': self.steps, 'm': self.m, 'v': self.v, 'lr': self.lr}

    def load_state_dict(self, sd):
        self.steps = sd['steps']
        if sd.get('m') is not None and self.m is not None:
            self.m.copy_(sd['m'])
        if sd.get('v') is not None and self.v is not None:
            self.v.copy_(sd['v'])


FUSE_CE_UPSAMPLE = os.environ.get('PSEG_FUSE_CE_UPSAMPLE', '1') == '1'
# hipGraphLaunch of a captured step is NOT a path of this package (DESIGN.md section 5 "Fault records": two host faults inside
# the runtime's launch of forked graphs, round 4): a captured step is replayed by the lane executor or not at all.  (Round 6: the
# PSEG_DEBUG_HIPGRAPHLAUNCH switch that kept the old engine reachable is gone.)
# AUTO mode replays a shape whose eager step spends at least this fraction of its device span enqueueing launches.  Measured
# ratios: DeepLabV3+ fp32 0.18 (eager for good), DeepLabV3+ -mp 0.52 (replay 14.23 ms against 14.41 eager), UNet / HRNet
# 0.68-1.0; the opt-in limb policies of DeepLabV3+ 0.24-0.30.  The line sits inside the gap between 0.30 and 0.52 (round 4 had
# it at 0.5, on top of DeepLabV3+ -mp), a factor 1.3 from either side: no workload of the reference straddles it.
AUTO_REPLAY_RATIO = float(os.environ.get('PSEG_AUTO_REPLAY_RATIO', '0.4'))
_GRAVEYARD = []     # (lane-executor handle, its CUDAGraph) of collected _StepGraph objects, see _StepGraph.__del__


class GraphRefused(RuntimeError):
    """The lane executor cannot express a captured step (memcpy nodes, `extra`-style kernel nodes): the step runs eagerly."""


def _drain_graveyard():
    """Destroy the lane executors of collected step graphs (called outside any capture).  Order: the executor first -- its
    nodes point into the hipGraph's argument blocks --, then the graph and with it the capture's private memory pool;
    pseg_lanes_destroy drains the device itself before it releases an event."""
    if _GRAVEYARD and ops.CAPTURING == 0:
        from .. import _lib
        kept = []
        while _GRAVEYARD:
            handle, graph = _GRAVEYARD.pop()
            try:
                _lib.call('pseg_lanes_destroy', handle)
            except _lib.PsegError:
                kept.append((handle, graph))      # (nothing was released: try again at the next step)
                continue
            del graph
        _GRAVEYARD.extend(kept)


class _StepGraph:
    """One captured forward + loss + backward for a fixed batch shape.

    Replay is the lane executor's (csrc/lanes.hip, `lanes` > 0): the captured graph is walked once and re-issued as plain
    launches on up to `lanes` streams from a C loop -- the weight gradients keep their own stream, the host pays ~2 us per
    kernel instead of ~15 us of Python, and no hipGraphExec is ever instantiated (its launch costs 12-24 ms of host time
    for a forked graph, and it is the call both host faults of round 4 died in).  lanes == 1: everything on the caller's
    stream, still without hipGraphExec.  A graph the executor cannot express raises GraphRefused: the Trainer then runs that
    shape eagerly."""

    def __init__(self, trainer, inputs, targets, lanes=1):
        lanes = max(1, lanes)
        bad = [m for m in trainer.model.modules() if isinstance(m, BatchNorm2d) and m.training and
               m.track_running_stats and m.momentum is None]
        if bad:
            raise NotImplementedError('graph mode bakes the BatchNorm momentum into the captured launch; '
                                      'momentum=None (cumulative average) changes it every step')
        self.x = inputs.detach().clone().contiguous()
        self.t = targets.detach().to(torch.int64).clone().contiguous()
        # host-side effects of one step that a replay does not repeat: the lazy num_batches_tracked counters
        self.bns = [m for m in trainer.model.modules() if isinstance(m, BatchNorm2d) and m.training and
                    m.track_running_stats]
        torch.cuda.synchronize()
        self.lanes = 0
        if lanes > 1:
            # the lane executor's stream pool (csrc/lanes.hip): created and touched BEFORE the capture's own streams (capture
            # stream, branch streams), so that the lanes of the replay do not land on the compute stream's hardware queue
            from .. import _lib
            _lib.call('pseg_lanes_reserve', min(int(lanes), 5))
        self.graph = torch.cuda.CUDAGraph(keep_graph=True) if lanes > 0 else torch.cuda.CUDAGraph()
        # Data-parallel runs: the captured step carries no collective, but it MARKS where each gradient bucket is complete
        # (a one-word memset on the compute stream, and on the weight-gradient stream when that is in use: words 2k, 2k+1);
        # the lane executor records events there and run() hangs the bucket's all-reduce behind them.
        self.reducer = trainer.reducer if (trainer.reducer.enabled and lanes > 0) else None
        self.marked = {}
        self.marks = None
        hook = None
        if self.reducer is not None:
            from .. import _lib
            stride = 1 + ops.AUX_STREAMS_MAX  # marker words per bucket: the compute stream, then each auxiliary stream
            self.marks = torch.zeros(stride * len(self.reducer.buckets), dtype=torch.int32, device=self.x.device)
            base, dev = self.marks.data_ptr(), self.x.device

            def mark(k):
                words = [stride * k]
                _lib.call('pseg_mark', base + 4 * words[0], ops._stream())
                for j, aux in enumerate(ops.aux_streams_in_use(dev)):
                    words.append(stride * k + 1 + j)
                    _lib.call('pseg_mark', base + 4 * words[-1], aux.cuda_stream)
                self.marked[k] = tuple(words)

            def complete(k):
                # (inside a region of parallel branch lanes -- ops.Branches -- the bucket's gradients are spread over
                # several streams: the marker is set where they have been joined)
                ops.after_branches(lambda: mark(k))
            hook = self.reducer.capture_hook(complete)
        ops.EVER_CAPTURED = True      # (workspaces / job tables a captured launch points at are never freed from here on)
        ops.CAPTURING += 1
        prev_lanes = ops.set_capture_wgrad_lanes(getattr(trainer.model, 'capture_wgrad_lanes', 1))
        saved_hook, trainer.env.grad_ready = trainer.env.grad_ready, hook
        try:
            with ops.no_gc_capture(self.graph):      # (no garbage collection while the capture is open: see the helper)
                self.loss_out = trainer._fwd_loss_bwd(self.x, self.t)
        except BaseException:
            ops.reset_branches()
            raise
        finally:
            ops.CAPTURING -= 1
            ops.set_capture_wgrad_lanes(prev_lanes)
            trainer.env.grad_ready = saved_hook
        for m in self.bns:                      # the capture pass ran the host code once but no kernel
            m.__dict__['_nbt_pending'] -= 1
        if lanes > 0:
            import ctypes
            from .. import _lib
            h = ctypes.c_int64(0)
            try:
                _lib.call('pseg_lanes_build', self.graph.raw_cuda_graph(), int(lanes), ctypes.byref(h))
                info = [ctypes.c_int(0) for _ in range(4)]
                _lib.call('pseg_lanes_info', h.value, *[ctypes.byref(i) for i in info])
                self.lanes = h.value
                self.lane_info = dict(zip(('nodes', 'launches', 'lanes', 'events'), (i.value for i in info)))
                if self.marked:
                    bound = ctypes.c_int(0)
                    _lib.call('pseg_lanes_bind_markers', h.value, self.marks.data_ptr(), self.marks.numel(),
                              ctypes.byref(bound))
                    want = sum(len(v) for v in self.marked.values())
                    if bound.value != want:
                        raise RuntimeError('captured step holds %d bucket markers, %d were set' % (bound.value, want))
                    self.lane_info['markers'] = bound.value
            except _lib.PsegError as e:
                if h.value:
                    _GRAVEYARD.append((h.value, self.graph))
                self.lanes = 0
                raise GraphRefused(str(e)) from e

    def __del__(self):
        # The executor owns streams and events, and the capture's memory pool may still be in use by them: both are released
        # later, from a point where synchronising is legal (a garbage collection can run this in the middle of ANOTHER
        # step's stream capture, where any synchronising call aborts the process).
        if getattr(self, 'lanes', 0):
            _GRAVEYARD.append((self.lanes, self.graph))
            self.lanes = 0

    def run(self, inputs, targets, exchange=False):
        """exchange: this micro-step ends an accumulation window of a data-parallel run -- enqueue the bucket all-reduces
        behind the replay's markers (what is not marked is left to reducer.finish())."""
        self.x.copy_(inputs, non_blocking=True)
        self.t.copy_(targets, non_blocking=True)
        if self.lanes:
            from .. import _lib
            _lib.call('pseg_lanes_launch', self.lanes, ops._stream())
            if exchange and self.reducer is not None and self.marked:
                def wait(k, side):
                    for word in self.marked[k]:
                        _lib.call('pseg_lanes_wait_marker', self.lanes, word, side.cuda_stream)
                self.reducer.launch_behind(self.marked, wait)
        else:
            raise RuntimeError('captured step without a lane executor')
        for m in self.bns:
            m.__dict__['_nbt_pending'] += 1
        return self.loss_out[0].clone()


class Trainer:
    def __init__(self, model, fetcher, loss_fn=None, workdir='weights', accumulate=1, adam=False, lr=1e-3,
                 weights='', resume=False, mixed_precision=False, momentum=0.9, weight_decay=0.0,
                 bucket_bytes=32 << 20, device=None, graph=None, max_graphs=8):
        # mixed_precision: the reference's -mp flag asks apex for fp16 compute with fp32 master weights and loss scaling
        # (train.py:70,102-105,138; README.md:12).  That is the `half` policy here: fp16 activations / gradients / filter
        # copies in HBM, every conv one fp16 MFMA pass with fp32 accumulation, BatchNorm statistics, the loss and the
        # master weights / optimiser state in fp32, dynamic loss scaling with the overflow check and the skipped step on
        # the device.  (The three-product fp32-storage policy of rounds 1-2 stays reachable as PSEG_PRECISION=limb or
        # PSEG_MP_POLICY=limb.)  The policy is scoped to this Trainer's execution context (its Env, below).
        self.device = device or _device()
        self.model = model
        self.fetcher = fetcher
        self.loss_fn = loss_fn or _default_loss
        self.workdir = workdir
        self.accumulate = max(1, int(accumulate))
        self.epoch = 0
        self.metrics = 0
        if weights:
            sd = torch.load(weights, map_location='cpu')
            model.load_state_dict(sd['model'] if 'model' in sd else sd)
        self.arena = prepare(model, self.device)
        self.optimizer = FlatOptimizer(self.arena, adam=adam, lr=lr, momentum=momentum, weight_decay=weight_decay)
        owners = [(s.module, s.offset, s.numel) for s in self.arena.segments]
        self.reducer = GradReducer(self.arena.grads, owners, bucket_bytes=bucket_bytes)
        mp_policy = os.environ.get('PSEG_MP_POLICY', 'half')
        self.env = Env(save=True, accumulate=False, grad_ready=self.reducer.grad_ready if self.reducer.enabled else None,
                       overlap_wgrad=True, policy=mp_policy if mixed_precision else None)
        self.mp_state = None
        self._ensure_mp_state()
        self.reducer.extra_stream = lambda: ops.aux_streams_in_use(self.device)
        object.__setattr__(model, '_pseg_env', self.env)
        self._micro = 0
        self._amax_pool = None
        # one launch for every split weight gradient's slab reduction (ops.SlabPool) -- pays when backward runs on ONE
        # stream (HRNet 512x512 B=8: 22.6 -> 20.3 ms; the captured-graph mode of the launch-bound configurations); with the
        # weight gradients on the auxiliary stream the per-layer reductions are off the critical path already and one
        # big reduction after the join is 0.3-0.5 ms slower (DeepLabV3+ 47.1 -> 47.5 ms).  PSEG_DEFER_SLABS=1 / 0 forces.
        defer = os.environ.get('PSEG_DEFER_SLABS', 'auto')
        self._slab_pool = ops.SlabPool(self.device) if (defer == '1' or (defer == 'auto' and not ops.OVERLAP_WGRAD)) \
            else None
        # graph: True / False, or None = AUTO (PSEG_GRAPH=1 / 0 force it for a default-constructed Trainer): a micro-step is
        # captured and replayed when it is launch-bound -- see _auto_graph
        env_graph = os.environ.get('PSEG_GRAPH', 'auto')
        self.graph = bool(graph) if graph is not None else (True if env_graph == '1' else (False if env_graph == '0' else 'auto'))
        self._auto = {}       # AUTO: shape key -> {'n': steps seen, 'use': None (undecided) | True | False, ...}
        # streams of the lane executor that replays a captured step (1: everything on the compute stream)
        self.graph_lanes = max(1, int(os.environ.get('PSEG_GRAPH_LANES', '6')))
        # The lane executor's stream pool (csrc/lanes.hip) is created when the first step is captured: a Trainer that never
        # replays (DeepLabV3+: eager) creates no stream it does not use, and its weight-gradient / exchange / RCCL streams keep
        # the hardware queues they always had.  A model that is known to live on the replay (HRNet: `replay_lanes`) gets the pool
        # HERE, before this process's other streams exist -- the pool then owns the queues next to the compute stream, worth 4 %
        # of its step (7.8 -> 7.5 ms -mp).  PSEG_LANES_RESERVE=early / capture forces either.
        want_early = os.environ.get('PSEG_LANES_RESERVE', 'early' if getattr(model, 'replay_lanes', 0) > 1 else 'capture')
        if self.device.type == 'cuda' and self.graph_lanes > 1 and want_early == 'early' and self.graph is not False:
            from .. import _lib
            with torch.cuda.device(self.device):
                _lib.call('pseg_lanes_reserve', min(self.graph_lanes, max(2, getattr(model, 'replay_lanes', 5))))
        self.max_graphs = max_graphs
        self._graphs = {}     # key -> _StepGraph | None (None: seen once, run eagerly)
        self._graph_refused = {}   # key -> why the lane executor refused the captured step (that shape stays eager)
        self._first_sight = False
        if resume:
            path = os.path.join(workdir, 'last.pt')
            if os.path.exists(path):
                self.load(path)
        self.sync_initial_state()

    def close(self):
        """Release what the Trainer holds outside torch's allocator: the library's RCCL communicator, if one was made."""
        red = getattr(self, 'reducer', None)
        if red is not None:
            red.close()

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 -- interpreter shutdown: the library / process group may be gone already
            pass

    def _ensure_mp_state(self):
        """The device-resident loss-scale state of the half-precision policy (created when the policy is first seen: the
        policy of a Trainer's Env may be switched after construction, as bench.py does)."""
        if self.env.half and self.mp_state is None:
            from .. import _lib
            self.mp_state = torch.zeros(8, dtype=torch.float32, device=self.device)
            _lib.call('pseg_mp_state_init', self.mp_state.data_ptr(), float(os.environ.get('PSEG_LOSS_SCALE', 2.0 ** 16)),
                      ops._stream())
            self._seed_applied_steps()
        self.env.loss_scale = self.mp_state[0:1] if (self.env.half and self.mp_state is not None) else None

    def _seed_applied_steps(self):
        """Under the half policy the optimiser's first-step flag (SGD: momentum buffer := gradient) and Adam's bias
        correction come from the device counter of APPLIED steps, mp_state[4].  A state created after optimiser steps were
        already taken -- the policy switched to `half` mid-run, or an fp32 checkpoint resumed with -mp -- starts that counter
        at the optimiser's own count, so warm moments are not treated as a first step."""
        if self.mp_state is not None and self.optimizer.steps > 0:
            self.mp_state[4] = float(self.optimizer.steps)

    def _bridge_half(self):
        """Half policy through the autograd bridge for a model WITHOUT explicit model_fwd / model_bwd (a container of
        ConvNormAct / backbone blocks glued by torch ops): every block must run under this Trainer's Env -- fp16 storage --
        and find its fp16 filter copies in the Trainer's arena; the loss-scaled gradient enters through the scaled loss
        (train_batch).  A nested full model scales internally (nn.loss_grad_in) and would be scaled twice: refused."""
        if not getattr(self, '_bridge_half_ready', False):
            for m in self.model.modules():
                if m is not self.model and hasattr(m, 'model_bwd'):
                    raise NotImplementedError('mixed_precision=True with a segmentation model nested inside a container: '
                                              'pass the model itself to the Trainer')
                object.__setattr__(m, '_pseg_env', self.env)
                if getattr(m, '_pseg_arena', None) is None:
                    object.__setattr__(m, '_pseg_arena', self.arena)
            self._bridge_half_ready = True
        self.arena.prepare_half()        # one refresh per micro-step; the blocks' own refresh is switched off below
        self.env.half_fresh = True

    def sync_initial_state(self):
        """Data-parallel replicas must start from ONE model: rank 0's parameters (one flat arena), BatchNorm running
        statistics / counters and optimiser moments are broadcast to every rank -- what DistributedDataParallel does
        at construction inside the reference's external Trainer (train.py:61,112-117).  Without it every rank would
        keep its own random initialisation and the summed gradients would belong to no model at all."""
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        dist.broadcast(self.arena.params, 0)
        if self.mp_state is not None:
            dist.broadcast(self.mp_state, 0)
        for t in (self.optimizer.m, self.optimizer.v):
            if t is not None:
                dist.broadcast(t, 0)
        steps = torch.tensor([self.optimizer.steps, self.epoch], dtype=torch.int64, device=self.device)
        dist.broadcast(steps, 0)
        self.optimizer.steps, self.epoch = int(steps[0]), int(steps[1])
        broadcast_buffers(self.model, 0)

    # ---- one optimisation micro-step; the optimiser fires every `accumulate` micro-batches (train.py:65)
    def train_batch(self, inputs, targets):
        if _GRAVEYARD:
            _drain_graveyard()
        self._ensure_mp_state()
        first = self._micro == 0
        last = self._micro == self.accumulate - 1
        self.env.accumulate = not first
        # only the last micro-batch of a window exchanges gradients
        self.env.grad_ready = self.reducer.grad_ready if (self.reducer.enabled and last) else None
        loss = None
        if self._explicit(inputs, targets):
            # stock loss at the logits' own resolution: forward, loss and backward as explicit launches on this thread
            # (no autograd graph, no hop to the autograd worker), optionally replayed from a captured hipGraph
            if self.graph is True or (self.graph == 'auto' and self._auto_graph(inputs, targets)):
                # The captured micro-step carries NO collective; with the reducer on it carries one MARKER per gradient bucket
                # (_StepGraph), and the replay hangs the bucket all-reduces behind them on the side stream: they overlap the
                # replayed backward like the eager per-bucket callbacks do.
                # The first step of a shape runs eagerly but under the CAPTURE's configuration (no per-bucket callbacks, the
                # slab pool active): whatever is built lazily on first use -- slab job tables, workspaces, plans -- exists
                # before the capture, which cannot upload or reallocate anything.
                # (with markers to come, that eager step gets a no-op callback: like the capture's hook it keeps the weight
                # gradients off the deferred slab pool, whose single reduction would come after the markers)
                ready = self.env.grad_ready
                self.env.grad_ready = (lambda module: None) if (self.reducer.enabled and self.graph_lanes > 0) else None
                try:
                    loss = self._graph_step(inputs, targets, exchange=self.reducer.enabled and last)
                    if loss is None and self._first_sight:
                        loss = self._fwd_loss_bwd(inputs, targets.to(torch.int64).contiguous())[0]
                finally:
                    self.env.grad_ready = ready
            if loss is None and getattr(self, '_auto_loss', None) is not None:
                loss, self._auto_loss = self._auto_loss, None      # (AUTO: the judged step already ran, eagerly)
            if loss is None:
                loss = self._fwd_loss_bwd(inputs, targets.to(torch.int64).contiguous())[0]
        if loss is None:
            # the reference's own idiom through the autograd bridge (custom loss_fn, eval-mode BatchNorm, a model without
            # model_fwd).  Half policy: a model with model_bwd multiplies the loss scale in itself (nn.loss_grad_in); any
            # other module tree gets it through the loss -- the optimiser divides it out either way
            blockwise = not hasattr(self.model, 'model_bwd')
            scale_loss = self.env.half and blockwise
            if blockwise:
                # a container of blocks: every block is its own autograd node with autograd's ACCUMULATE semantics (a module
                # may be used twice in one forward), so the window starts from a zeroed gradient arena -- the explicit path
                # instead overwrites on the first micro-batch and never zeroes
                if first:
                    self.arena.zero_grad()
                self.env.accumulate = True
            if scale_loss:
                self._bridge_half()
            try:
                outputs = self.model(inputs)
                loss = self.loss_fn(outputs, targets, self.model)
                (loss * self.mp_state[0] if scale_loss else loss).backward()
            finally:
                self.env.half_fresh = False
        self._micro += 1
        if last:
            self.reducer.finish()
            self.optimizer.step(grad_scale=self.reducer.grad_scale / self.accumulate, mp_state=self.mp_state if self.env.half else None)
            self._micro = 0
        return loss

    # ---- hipGraph-captured micro-step
    def _explicit(self, inputs, targets):
        """The plain case: the stock loss at the logits' own resolution, a model with explicit model_fwd / model_bwd,
        training mode."""
        return (self.loss_fn is _default_loss and hasattr(self.model, 'model_fwd') and self.model.training and
                inputs.is_cuda and inputs.dtype == torch.float32 and targets.is_cuda and
                tuple(targets.shape) == (inputs.shape[0],) + tuple(inputs.shape[2:]))

    def _fwd_loss_bwd(self, x, t):
        """forward + cross-entropy + backward as explicit launches (what model(x) / compute_loss / loss.backward() do
        through the autograd bridge, minus the bridge)."""
        self.env.save = True      # (an evaluation pass through the bridge in between switches it off on the shared Env)
        with torch.no_grad():
            if self.env.track_amax:          # fp16-limb forward: every filter's max|w| in one launch,
                self.arena.filter_amax()     # activation bounds from one per-step pool of zeroed scalars
                self.env.wamax_fresh = True
                if self._amax_pool is None:
                    self._amax_pool = ops.AmaxPool(self.device)
                self._amax_pool.reset()
                ops._amax_pool = self._amax_pool
            # A model whose last op is a bilinear up-sampling of the class logits (DeepLabV3+) hands out the low-resolution
            # logits and the loss + its gradient are taken from them directly (pseg_ce_upsampled_fwd_bwd): the
            # full-resolution logits and their gradient (2 x 352 MB at the benchmark shape) never exist
            # the [Cin][taps][Cout] filter copies the data gradients read depend on the weights only: refreshed on the second
            # stream beside the forward pass (one bandwidth-bound launch, 0.09 ms for DeepLabV3+) instead of between the
            # loss and the first backward kernel; joined before the backward pass starts
            half = self.env.half
            if half:
                # fp16 filter copies (forward) and their transposes (data gradients) from the fp32 master weights: one launch
                self.arena.prepare_half()
            early_wT = ops.OVERLAP_WGRAD and self.env.overlap_wgrad and not half
            if early_wT:
                with torch.cuda.stream(ops.fork_aux(x.device)):
                    self.arena.transpose_filters()
            lr_spec = getattr(self.model, 'lowres_loss', None) if FUSE_CE_UPSAMPLE else None
            lowres = lr_spec is not None and x.shape[2] % lr_spec[0] == 0 and x.shape[3] % lr_spec[0] == 0 and \
                ops.ce_upsampled_ok_shape(x.shape[2] // lr_spec[0], x.shape[3] // lr_spec[0], self.model.num_classes,
                                          x.shape[2], x.shape[3], lr_spec[1])
            try:
                if lowres:
                    out, saved = self.model.model_fwd(x, self.env, lowres=True)
                else:
                    out, saved = self.model.model_fwd(x, self.env)
            finally:
                ops._amax_pool = None
            self.env.wamax_fresh = False
            if lowres:
                loss_out, dl = ops.ce_upsampled_fwd_bwd(out, self.model.num_classes, t, lr_spec[1], want_grad=True)
            else:
                loss_out, dl = ops.ce_fwd_bwd(out, t, want_grad=True)
            if early_wT:
                ops.join_aux(x.device)
            elif not half:
                self.arena.transpose_filters()
            self.env.wT_fresh = not half
            # split weight gradients park their slabs in the pool; one launch folds them all after the join
            self.env.slab_pool = self._slab_pool
            try:
                if lowres:
                    self.model.model_bwd(dl, saved, self.env, lowres=True)
                else:
                    self.model.model_bwd(dl, saved, self.env)
            finally:
                self.env.slab_pool = None
            self.env.wT_fresh = False
            ops.join_aux(x.device)
            if self._slab_pool is not None:
                self._slab_pool.reduce(accumulate=self.env.accumulate)
        return loss_out

    def _auto_graph(self, inputs, targets):
        """AUTO mode (Trainer(graph=None), the default): should this micro-step go through capture + replay?
        The reference's loop just runs (train.py:59,71-72); here a drop-in user gets the replayed step whenever it pays, without
        an environment variable.  Per (shape, accumulate flag, policy) key: the first step runs eagerly (plans, workspaces and
        allocator pools come into being), the second runs eagerly between two events from an EMPTY queue (one host
        synchronisation per shape, ever) and is judged: host enqueue time >= AUTO_REPLAY_RATIO (0.4) of the device span means
        the device spends a good part of the step waiting for launches (HRNet / UNet: ~1000 / ~560 launches of 5-10 us; measured
        ratios 0.68-1.0; DeepLabV3+ -mp 0.52: replayed 14.23 ms, eager 14.41) and from the third step on the shape is replayed by
        the lane executor; otherwise (DeepLabV3+ fp32 at 512x512: 8 ms of enqueue under 45 ms of kernels, ratio 0.18) it stays
        eager for good.  Shapes beyond `max_graphs` captured ones (train.py --multi-scale) stay eager silently."""
        import time
        key = (tuple(inputs.shape), self.env.accumulate, self.arena.params.data_ptr(), self.env.policy_name)
        st = self._auto.setdefault(key, {'n': 0, 'use': None})
        if st['use'] is not None:
            return st['use']
        st['n'] += 1
        if st['n'] == 1:
            return False
        if sum(1 for v in self._auto.values() if v['use']) >= self.max_graphs:
            st['use'] = False
            return False
        # measured eager step (runs here, its loss is handed back through _auto_loss)
        torch.cuda.current_stream(self.device).synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        self._auto_loss = self._fwd_loss_bwd(inputs, targets.to(torch.int64).contiguous())[0]
        e1.record()
        host_ms = (time.perf_counter() - t0) * 1e3
        e1.synchronize()
        dev_ms = e0.elapsed_time(e1)
        st['host_ms'], st['dev_ms'] = host_ms, dev_ms
        # A per-rank verdict: ranks of a data-parallel run may disagree (a --multi-scale run gives every rank its own shape
        # sequence) because the exchange does not depend on it -- eager steps, replayed steps and reducer.finish() all issue the
        # bucket collectives in bucket-index order (utils/dist.py).  (Round 4 all-reduced the verdict here: a collective behind
        # a per-rank, data-dependent condition -- removed.)
        st['use'] = bool(host_ms >= AUTO_REPLAY_RATIO * dev_ms)
        if os.environ.get('PSEG_GRAPH_VERBOSE', '0') == '1':
            print('[pseg] auto graph %s: host enqueue %.2f ms, device span %.2f ms -> %s'
                  % (key[0], host_ms, dev_ms, 'replay' if st['use'] else 'eager'), flush=True)
        return False

    def step_mode(self, shape=None):
        """'replayed' / 'eager': how micro-steps of input `shape` (default: any shape seen) run now under the Env's current
        policy -- for run records (bench.py) and tests."""
        keys = [k for k in self._graphs if (shape is None or tuple(k[0]) == tuple(shape)) and k[3] == self.env.policy_name]
        for k in reversed(keys):
            sg = self._graphs.get(k)
            if sg is not None and sg.lanes:
                return 'replayed'
        return 'eager'

    def graph_decisions(self):
        """AUTO mode: {input shape: {'use': replayed?, 'host_ms', 'dev_ms'}} of the shapes judged so far (logging / tests)."""
        return {k[0]: {kk: vv for kk, vv in v.items() if kk != 'n'} for k, v in self._auto.items() if v['use'] is not None}

    def _graph_step(self, inputs, targets, exchange=False):
        key = (tuple(inputs.shape), self.env.accumulate, self.arena.params.data_ptr(), self.env.policy_name)
        sg = self._graphs.get(key, False)
        self._first_sight = False
        if sg is False:                         # first sight of this shape: eager (also warms the allocator)
            if len(self._graphs) >= self.max_graphs:
                return None
            self._graphs[key] = None
            self._first_sight = True
            return None
        if key in self._graph_refused:         # the executor refused this shape's graph once: eager for good
            self._first_sight = True
            return None
        if sg is None:
            # (a model may cap the lanes of its replay: HRNet's six chains run best on five -- `replay_lanes`)
            cap = getattr(self.model, 'replay_lanes', 0)
            lanes = min(self.graph_lanes, cap) if (cap > 0 and self.graph_lanes > 0) else self.graph_lanes
            try:
                sg = self._graphs[key] = _StepGraph(self, inputs, targets, lanes=lanes)
            except GraphRefused as e:
                # The capture pass enqueued nothing (its launches became nodes of a graph that is dropped here) and its host-side
                # effects were undone by _StepGraph: this micro-step -- and every later one of the shape -- runs eagerly.  There is
                # no second replay engine to fall back to.
                import warnings
                warnings.warn('captured step refused by the lane executor (%s): this shape runs eagerly' % e)
                self._graph_refused[key] = str(e)
                self._first_sight = True
                return None
        return sg.run(inputs, targets, exchange=exchange)

    def step(self):
        """One epoch (reference train.py:71-72)."""
        self.model.train()
        total, n = None, 0
        for inputs, targets in self.fetcher:
            loss = self.train_batch(inputs, targets).detach()
            total = loss if total is None else total + loss
            n += 1
        self.epoch += 1
        return (total / max(n, 1)).item() if total is not None else float('nan')

    def state(self):
        sd = {k: v.detach().contiguous().clone() for k, v in self.model.state_dict().items()}
        st = {'model': sd, 'epoch': self.epoch, 'metrics': self.metrics, 'optimizer': self.optimizer.state_dict()}
        if self.mp_state is not None:
            st['loss_scaler'] = self.mp_state.detach().cpu().clone()
            # the optimiser's host-side count includes steps the device skipped (overflow); what a resumed fp32 run needs is
            # the number of steps that were APPLIED
            st['optimizer'] = dict(st['optimizer'], steps=int(st['loss_scaler'][4].item()))
        return st

    def loss_scale_state(self):
        """(half-precision policy) host copy of the loss-scale state: dict(scale, steps_applied, steps_skipped).
        Synchronises the host -- for logging and tests, not for the step."""
        if self.mp_state is None:
            return None
        s = self.mp_state.detach().cpu().tolist()
        return {'scale': s[0], 'good_steps': int(s[2]), 'steps_applied': int(s[4]), 'steps_skipped': int(s[5])}

    def sync_buffers(self):
        """Rank 0's BatchNorm running statistics to every rank (DistributedDataParallel's broadcast_buffers semantics, see
        utils/dist.py::broadcast_buffers): called before a checkpoint is written; test() does the same before it evaluates."""
        broadcast_buffers(self.model, 0)

    def save(self, best=False):
        self.sync_buffers()             # (collective: every rank takes part, rank 0 writes)
        if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
            return
        os.makedirs(self.workdir, exist_ok=True)
        st = self.state()
        torch.save(st, os.path.join(self.workdir, 'last.pt'))
        if best:
            torch.save(st, os.path.join(self.workdir, 'best.pt'))

    def load(self, path):
        st = torch.load(path, map_location='cpu')
        self.model.load_state_dict(st['model'])
        self.epoch = st.get('epoch', 0)
        self.metrics = st.get('metrics', 0)
        if 'optimizer' in st:
            self.optimizer.load_state_dict(st['optimizer'])
        if self.mp_state is not None:
            if 'loss_scaler' in st:
                self.mp_state.copy_(st['loss_scaler'])
            else:                   # an fp32 checkpoint resumed with -mp: warm moments, no scaler state
                self._seed_applied_steps()

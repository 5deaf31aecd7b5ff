"""Data-parallel gradient exchange: bucketed all-reduce of the flat gradient arena, overlapped with backward.

The reference trains with `torch.distributed.launch` + DistributedDataParallel inside its external Trainer
(README.md:42-44, train.py:112-117).  Re-done here for one process per GPU over RCCL/xGMI:

* buckets are contiguous ranges of the gradient arena (arena.py) -- no flatten/unflatten copies;
* each block reports `grad_ready(module)` the moment its weight-gradient kernels are enqueued; when every
  parameter-owning module of a bucket has reported, an event is recorded on the compute stream, the side stream
  waits on it and issues ONE sum-all-reduce for the bucket while backward keeps running on the compute stream;
* the collectives of a step are ALWAYS issued in bucket-index order, whatever order backward completes the buckets in and
  whichever way the step runs (eager with per-bucket callbacks, eager without, replayed from a captured graph, or a mix of
  those across the ranks): a completed bucket waits for the ones before it, as DistributedDataParallel's reducer does.  The
  ranks therefore never have to agree on HOW they run a step (Trainer AUTO mode decides per rank);
* the 1/world_size of the mean is folded into the fused optimiser kernel (grad_scale), not a separate pass;
* bucket size defaults to 32 MiB: the 8-GPU xGMI mesh is point-to-point (7 links/GPU), so few large
  collectives beat many small ones (DeepLabV3+ R50: 156.6 MB of gradients -> 7 buckets: cuts fall on
  parameter-segment boundaries, so the large layer-4 / ASPP filters end buckets early).

The reducer only touches torch tensors and torch.distributed, so the same code runs on CPU tensors over gloo
(tests/test_dist_cpu.py) and on HIP tensors over RCCL.  PSEG_NATIVE_ALLREDUCE=1: the collective itself goes through the
library's own RCCL binding (pseg_allreduce_bucket, csrc/comm.hip) on a communicator created from a unique id that rank 0
hands out through torch.distributed -- one C call per bucket on the side stream, no Work objects.
"""
import ctypes
import os

import torch
import torch.distributed as dist


class _Enqueued:
    """What a natively issued collective returns: ordering is the side stream's, there is nothing to wait for on the host."""

    def wait(self):
        return True


class NativeComm:
    """An RCCL communicator of the library (include/pseg_amd.h: pseg_comm_*) over the ranks of a torch.distributed
    process group: rank 0 draws the unique id, the group's own broadcast distributes it."""

    def __init__(self, device, group=None):
        from .. import _lib
        self._lib = _lib
        if not _lib.load().pseg_comm_available():
            raise RuntimeError('librccl.so could not be bound by libpseg_amd.so (PSEG_RCCL_PATH)')
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        buf = (ctypes.c_char * 128)()
        if rank == 0:
            _lib.call('pseg_comm_unique_id', ctypes.addressof(buf))
        on_dev = dist.get_backend(group) == 'nccl'
        t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        t = t.to(device) if on_dev else t
        dist.broadcast(t, dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ident = bytes(t.cpu().tolist())
        h = ctypes.c_int64(0)
        with torch.cuda.device(device):
            _lib.call('pseg_comm_init', ident, world, rank, ctypes.byref(h))
        self.handle = h.value

    def all_reduce(self, view, stream):
        self._lib.call('pseg_allreduce_bucket', self.handle, view.data_ptr(), view.numel(), stream.cuda_stream)
        return _Enqueued()

    def reduce_scatter(self, view, per_rank, rank, stream):
        self._lib.call('pseg_reduce_scatter_bucket', self.handle, view.data_ptr(), per_rank, rank, stream.cuda_stream)

    def all_gather(self, view, per_rank, rank, stream):
        self._lib.call('pseg_all_gather_bucket', self.handle, view.data_ptr(), per_rank, rank, stream.cuda_stream)

    def version(self):
        v = ctypes.c_int(0)
        self._lib.call('pseg_comm_version', ctypes.byref(v))
        return v.value

    def close(self):
        if self.handle:
            self._lib.call('pseg_comm_destroy', self.handle)
            self.handle = 0


class _Works:
    """Several outstanding collectives of one bucket behind one .wait()."""

    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()
        return True


class Bucket:
    __slots__ = ('begin', 'end', 'pending', 'total', 'work', 'modules', 'ev', 'ev2')

    def __init__(self, begin, end):
        self.begin, self.end = begin, end
        self.pending = self.total = 0
        self.work = None
        self.modules = set()
        self.ev = self.ev2 = None     # per-bucket events, created once and re-recorded every step


class GradReducer:
    def __init__(self, flat_grads, segments, bucket_bytes=32 << 20, process_group=None, tail_bytes=4 << 20):
        """flat_grads: 1-D fp32 tensor (the gradient arena).  segments: iterable of (module, offset, numel) in
        arena order.  Buckets are cut in REVERSE arena order (backward finishes the top of the network first)."""
        self.flat = flat_grads
        self.group = process_group
        import os
        inited = dist.is_available() and dist.is_initialized()
        # PSEG_FORCE_REDUCER=1 runs the full bucket / side-stream / collective path even with one rank (testing)
        self.enabled = inited and (dist.get_world_size(process_group) > 1 or os.environ.get('PSEG_FORCE_REDUCER') == '1')
        self.world = dist.get_world_size(process_group) if self.enabled else 1
        self.rank = dist.get_rank(process_group) if self.enabled else 0
        # How a bucket is summed over the ranks.  'allreduce' (default): one all-reduce, algorithm left to RCCL.  'rs_ag': the
        # same sum as a reduce-scatter + all-gather pair (+ an all-reduce of the < world remainder elements): on the fully
        # connected xGMI node of eight GPUs a direct exchange drives all seven links of a GPU where a ring all-reduce is bound
        # by one (~0.26 ms against ~1.8 ms for the 156.6 MB of DeepLabV3+).  Nobody has been able to
        # measure the two against each other yet (one-GPU boxes): PSEG_EXCHANGE=allreduce|rs_ag is the switch, bench.py
        # records which one ran.  Two ranks give bit-identical sums either way (tests/test_dist_cpu.py).
        self.exchange = os.environ.get('PSEG_EXCHANGE', 'allreduce')
        if self.exchange not in ('allreduce', 'rs_ag'):
            raise ValueError('PSEG_EXCHANGE must be allreduce or rs_ag, got %r' % self.exchange)
        # measurement aid (bench.py exposed_comm_ms): the whole reducer machinery runs -- events, side stream, joins -- but no
        # collective is issued.  Gradients are then NOT summed: never set outside a timing run.
        self.skip_collectives = False
        segs = sorted(((off, off + n, mod) for mod, off, n in segments), key=lambda s: s[0])
        self.buckets = []
        limit = max(1, bucket_bytes // 4)
        cur = None
        # A module whose parameters are all frozen (requires_grad=False) never reports grad_ready: it is not waited for -- with
        # the strict bucket-index order one such module in a low-index bucket would hold EVERY later collective back until
        # finish() and the exchange would silently lose its overlap.  Its gradient range is still reduced (zeros).
        def frozen(mod):
            ps = [p for p in getattr(mod, '_parameters', {}).values() if p is not None]
            return bool(ps) and not any(p.requires_grad for p in ps)
        self._frozen = {id(mod) for _, _, mod in segs if frozen(mod)}
        for b, e, mod in reversed(segs):
            if cur is None or (cur.end - b) > limit and cur.modules:
                cur = Bucket(b, e)
                self.buckets.append(cur)
            cur.begin = min(cur.begin, b)
            cur.modules.add(id(mod))
        # The LAST bucket (the earliest layers) completes when backward ends, so its all-reduce is the one nothing
        # overlaps: keep it small -- cut the tail of that bucket off at a module boundary (<= tail_bytes).
        tail = max(1, tail_bytes // 4)
        if self.buckets and (self.buckets[-1].end - self.buckets[-1].begin) > 2 * tail:
            last = self.buckets[-1]
            inside = [(b, e, mod) for b, e, mod in segs if b >= last.begin and e <= last.end]
            cut = None
            for b, e, mod in inside:          # ascending addresses = backward's finishing order reversed
                if e - last.begin <= tail:
                    cut = e
            if cut is not None and last.begin < cut < last.end:
                head, small = Bucket(cut, last.end), Bucket(last.begin, cut)
                for b, e, mod in inside:
                    (small if e <= cut else head).modules.add(id(mod))
                if head.modules and small.modules:
                    self.buckets[-1:] = [head, small]
        self._by_module = {}
        for bk in self.buckets:
            bk.modules -= self._frozen
            bk.total = len(bk.modules)
            for mid in bk.modules:
                self._by_module.setdefault(mid, []).append(bk)
        self._reported = 0          # grad_ready calls of the step in flight
        self._warned_stall = False
        if flat_grads.is_cuda:
            from .. import ops as _ops
            self._side = _ops.role_stream('exchange', flat_grads.device)      # (one per device, distinct from the package's other streams)
        else:
            self._side = None
        self.extra_stream = None   # callable -> the further streams gradients are being produced on (a list, possibly empty)
        self.native = None
        if self.enabled and flat_grads.is_cuda and flat_grads.dtype == torch.float32 and \
                os.environ.get('PSEG_NATIVE_ALLREDUCE', '0') == '1':
            self.native = NativeComm(flat_grads.device, process_group)
        self.reset()

    def _all_reduce(self, view):
        """one bucket's sum over the ranks on the CURRENT stream (the side stream); -> something with .wait()"""
        if self.skip_collectives:
            return _Enqueued()
        if self.exchange == 'rs_ag' and self.world > 1 and view.numel() >= self.world:
            return self._rs_ag(view)
        if self.native is not None:
            return self.native.all_reduce(view, torch.cuda.current_stream(view.device))
        return dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _rs_ag(self, view):
        """reduce-scatter + all-gather of one bucket, in place: rank r owns slice r of the first world * per elements"""
        n = view.numel()
        per = n // self.world
        main = per * self.world
        if self.native is not None:
            st = torch.cuda.current_stream(view.device)
            self.native.reduce_scatter(view, per, self.rank, st)
            self.native.all_gather(view, per, self.rank, st)
            if main < n:
                self.native.all_reduce(view[main:], st)
            return _Enqueued()
        mine = view[self.rank * per:(self.rank + 1) * per]
        # (wait() of an RCCL work only orders the current stream behind it; over gloo -- CPU tests -- it blocks the host)
        dist.reduce_scatter_tensor(mine, view[:main], op=dist.ReduceOp.SUM, group=self.group, async_op=True).wait()
        works = [dist.all_gather_into_tensor(view[:main], mine, group=self.group, async_op=True)]
        if main < n:
            works.append(dist.all_reduce(view[main:], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return _Works(works)

    def describe(self):
        """what a run record should say about the exchange (bench.py)"""
        d = {'buckets': len(self.buckets), 'bucket_bytes': [(b.end - b.begin) * 4 for b in self.buckets],
             'bytes': int(sum((b.end - b.begin) * 4 for b in self.buckets)), 'mode': self.exchange,
             'native': self.native is not None, 'world': self.world, 'enabled': bool(self.enabled)}
        if self.native is not None:
            try:
                d['rccl_version'] = self.native.version()
            except Exception as e:      # noqa: BLE001 -- a record, not a requirement
                d['rccl_version'] = 'unavailable: %s' % e
        return d

    def close(self):
        """release the library's RCCL communicator (PSEG_NATIVE_ALLREDUCE=1); the reducer falls back to torch.distributed"""
        if self.native is not None:
            native, self.native = self.native, None
            native.close()

    def reset(self):
        for bk in self.buckets:
            bk.pending = bk.total
            bk.work = None
        self._reported = 0
        self._next = 0        # index of the first bucket whose collective has not been issued this step

    # called from backward (possibly the autograd worker thread) for every parameter-owning module
    def grad_ready(self, module):
        if not self.enabled:
            return
        self._reported += 1
        for bk in self._by_module.get(id(module), ()):
            bk.pending -= 1
        # strictly in index order: bucket k goes out when it AND every bucket before it is complete (the arena is laid out
        # in forward order and the buckets are cut from its end, so this is backward's own order but for small inversions
        # -- DeepLabV3+'s low-level projection sits before the ASPP in the arena and completes after it)
        while self._next < len(self.buckets) and self.buckets[self._next].pending <= 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def _launch(self, bk):
        view = self.flat[bk.begin:bk.end]
        if self._side is not None:
            if bk.ev is None:
                bk.ev, bk.ev2 = torch.cuda.Event(), []
            bk.ev.record(torch.cuda.current_stream(self.flat.device))
            self._side.wait_event(bk.ev)
            extra = self.extra_stream() if self.extra_stream is not None else ()
            for j, st in enumerate(extra or ()):      # weight gradients enqueued on the auxiliary stream(s)
                if j >= len(bk.ev2):
                    bk.ev2.append(torch.cuda.Event())
                bk.ev2[j].record(st)
                self._side.wait_event(bk.ev2[j])
            with torch.cuda.stream(self._side):
                bk.work = self._all_reduce(view)
        else:
            bk.work = self._all_reduce(view)

    def capture_hook(self, on_complete):
        """For a step that is CAPTURED once and replayed (Trainer graph mode): a `grad_ready` callback with bucket counters of
        its own that calls on_complete(bucket index) at the point of the enqueue order where the bucket's last gradient
        kernel has been issued -- the capture puts a marker there (csrc/lanes.hip) instead of launching a collective."""
        pending = [bk.total for bk in self.buckets]
        index = {id(bk): k for k, bk in enumerate(self.buckets)}

        def hook(module):
            for bk in self._by_module.get(id(module), ()):
                k = index[id(bk)]
                pending[k] -= 1
                if pending[k] == 0:
                    on_complete(k)
        return hook

    def launch_behind(self, which, wait_fn):
        """Replay of a captured step: bucket k's all-reduce (k in `which`) goes onto the side
        stream behind wait_fn(k, side_stream) -- the marker events of the replay -- instead of behind events recorded now,
        so it overlaps the rest of the replayed backward.  Buckets not in `which` are left to finish()."""
        if not self.enabled:
            return
        # index order, and nothing past the first bucket that carries no marker (finish() takes over from there): the order
        # of the collectives is the eager step's
        while self._next < len(self.buckets) and self._next in which:
            k, bk = self._next, self.buckets[self._next]
            view = self.flat[bk.begin:bk.end]
            if self._side is not None:
                wait_fn(k, self._side)
                with torch.cuda.stream(self._side):
                    bk.work = self._all_reduce(view)
            else:
                bk.work = self._all_reduce(view)
            self._next += 1

    def finish(self):
        """Block the compute stream until every bucket is reduced (call before the optimiser step).
        Buckets whose modules never reported (unused parameters) are reduced here."""
        if not self.enabled:
            return
        left = len(self.buckets) - self._next
        if left > 1 and self._reported and not self._warned_stall:
            # a step that DID report its gradients layer by layer and still left more than the tail bucket to finish(): some
            # module of bucket %d never reported (an unused parameter, a path not taken this step) and the strict bucket-index
            # order held every later collective back -- the sums are right, the overlap with backward is lost
            self._warned_stall = True
            import warnings
            warnings.warn('GradReducer: %d of %d gradient buckets were still waiting when backward ended (bucket %d never '
                          'completed: a parameter that received no gradient this step?); their all-reduces ran after backward '
                          'instead of beside it' % (left, len(self.buckets), self._next), RuntimeWarning, stacklevel=2)
        for bk in self.buckets[self._next:]:
            self._launch(bk)
        self._next = len(self.buckets)
        for bk in self.buckets:
            bk.work.wait()
        if self._side is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self._side)
        self.reset()

    @property
    def grad_scale(self):
        return 1.0 / self.world


def broadcast_buffers(model, src=0, group=None):
    """Rank `src`'s module buffers (BatchNorm running_mean / running_var / num_batches_tracked) to every rank, in two
    flat broadcasts (floating point, integer).  torch's DistributedDataParallel -- what the reference's external Trainer
    wraps the model in for `torch.distributed.launch` runs (README.md:42-44, train.py:112-117) -- does this before every
    forward (broadcast_buffers=True), so all replicas evaluate and checkpoint rank 0's statistics.  Here the replicas keep
    their own running statistics while training (they are never read in train mode) and are brought into line with rank 0
    where they ARE read: before an evaluation pass (test.py:15-58: the per-class counters of all ranks are summed, so they
    must come from ONE model) and before a checkpoint is written.  No-op without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return
    from ..nn import BatchNorm2d
    for m in model.modules():             # write lazily counted num_batches_tracked back before it is broadcast
        if isinstance(m, BatchNorm2d):
            BatchNorm2d._flush_counter(m, '', False)
    bufs = [b for b in model.buffers() if b is not None]
    for sel in (lambda b: b.is_floating_point(), lambda b: not b.is_floating_point()):
        part = [b for b in bufs if sel(b)]
        if not part:
            continue
        flat = torch.cat([b.detach().reshape(-1) for b in part])
        dist.broadcast(flat, src, group=group)
        off = 0
        with torch.no_grad():
            for b in part:
                n = b.numel()
                b.copy_(flat[off:off + n].view_as(b))
                off += n


def all_reduce_counters(counters, group=None):
    """Sum the per-class tp/fn/fp counters over ranks (reference test.py:51-58: three [num_classes] all-reduces;
    here one [3][num_classes] int64 all-reduce)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(counters, op=dist.ReduceOp.SUM, group=group)
    return counters

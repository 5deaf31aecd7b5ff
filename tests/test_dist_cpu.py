"""The N>1 path on CPU: two gloo ranks drive the gradient reducer (bucketing, readiness tracking, overlap-safe
finish, unused-parameter fallback) and the eval-counter all-reduce.  Same code runs over RCCL on HIP tensors."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pytorch_segmentation_amd.utils.dist import GradReducer, all_reduce_counters


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Mod:  # stand-in for a parameter-owning block
    pass


def _layout():
    """8 'modules' with (offset, numel) like an arena: sizes chosen so a 1 KiB bucket limit cuts several buckets."""
    sizes = [64, 128, 32, 256, 64, 16, 300, 100]
    mods, off, segs = [], 0, []
    for n in sizes:
        m = _Mod()
        mods.append(m)
        n4 = (n + 3) // 4 * 4
        segs.append((m, off, n4))
        off += n4
    return mods, segs, off


def _worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        mods, segs, total = _layout()
        flat = torch.arange(total, dtype=torch.float32) * (rank + 1)          # rank-dependent "gradients"
        expect = torch.arange(total, dtype=torch.float32) * sum(r + 1 for r in range(world))
        red = GradReducer(flat, segs, bucket_bytes=1024)
        assert red.enabled and red.world == world and abs(red.grad_scale - 1.0 / world) < 1e-12
        # buckets tile the arena exactly, in reverse order, none above the limit unless a single segment is
        covered = sorted((b.begin, b.end) for b in red.buckets)
        assert covered[0][0] == 0 and covered[-1][1] == total
        assert all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
        assert len(red.buckets) >= 3
        assert red.buckets[0].end == total                                      # first bucket = top of the network
        # step 1: every module reports, in backward (reverse) order
        for m in reversed(mods):
            red.grad_ready(m)
        red.finish()
        assert torch.equal(flat, expect)
        # step 2: a module never reports (unused parameter) -> finish() still reduces its bucket; state resets
        flat.copy_(torch.arange(total, dtype=torch.float32) * (rank + 1))
        for m in reversed(mods[1:]):
            red.grad_ready(m)
        red.finish()
        assert torch.equal(flat, expect)
        # mean = sum * grad_scale matches a single-process average
        avg = flat * red.grad_scale
        assert torch.allclose(avg, torch.arange(total, dtype=torch.float32) * (sum(r + 1 for r in range(world)) / world))
        cnt = torch.tensor([[1, 2, 3], [4, 5, 6], [7, 8, 9]], dtype=torch.int64) * (rank + 1)
        all_reduce_counters(cnt)
        assert torch.equal(cnt, torch.tensor([[1, 2, 3], [4, 5, 6], [7, 8, 9]], dtype=torch.int64) * 3)
        results[rank] = 'ok'
    except Exception as e:  # surface the failure in the parent
        results[rank] = 'fail: %r' % (e,)
    finally:
        dist.destroy_process_group()


def test_grad_reducer_two_ranks_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, port, results), nprocs=world, join=True)
    assert dict(results) == {0: 'ok', 1: 'ok'}, dict(results)


def _exchange_worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        mods, segs, total = _layout()
        g = torch.Generator().manual_seed(100 + rank)
        grads = torch.randn(total, generator=g)
        out = {}
        for mode in ('allreduce', 'rs_ag'):
            os.environ['PSEG_EXCHANGE'] = mode
            flat = grads.clone()
            red = GradReducer(flat, segs, bucket_bytes=1024)
            assert red.exchange == mode and red.describe()['mode'] == mode and red.describe()['buckets'] == len(red.buckets)
            # bucket sizes that are NOT multiples of the world size exercise the remainder all-reduce
            assert any(((b.end - b.begin) % world) != 0 for b in red.buckets) or world == 2
            for m in reversed(mods):
                red.grad_ready(m)
            red.finish()
            out[mode] = flat
        both = [torch.zeros(total) for _ in range(world)]
        dist.all_gather(both, grads)
        assert torch.equal(out['allreduce'], both[0] + both[1])
        assert torch.equal(out['rs_ag'], out['allreduce'])          # two ranks: the same two numbers are added either way
        # collectives skipped (bench.py exposed_comm_ms): the machinery runs, the gradients stay local
        flat = grads.clone()
        red = GradReducer(flat, segs, bucket_bytes=1024)
        red.skip_collectives = True
        for m in reversed(mods):
            red.grad_ready(m)
        red.finish()
        assert torch.equal(flat, grads)
        results[rank] = 'ok'
    except Exception as e:
        results[rank] = 'fail: %r' % (e,)
    finally:
        os.environ.pop('PSEG_EXCHANGE', None)
        dist.destroy_process_group()


def test_exchange_switch_two_ranks_gloo():
    """PSEG_EXCHANGE=allreduce|rs_ag: one all-reduce per bucket, or the same sum as reduce-scatter + all-gather (+ the
    remainder) -- identical gradients on two ranks, bucket by bucket, through the production reducer."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_exchange_worker, args=(world, port, results), nprocs=world, join=True)
    assert dict(results) == {0: 'ok', 1: 'ok'}, dict(results)


def _order_worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        mods, segs, total = _layout()
        g = torch.Generator().manual_seed(7 + rank)
        issued = []
        for step, how in enumerate((('shuffled', 'reverse'), ('replay', 'finish-only'), ('reverse', 'replay-partial'))):
            grads = torch.randn(total, generator=g)
            flat = grads.clone()
            red = GradReducer(flat, segs, bucket_bytes=1024)
            order = []
            plain = red._all_reduce
            red._all_reduce = lambda view, plain=plain, order=order: (order.append(view.numel()), plain(view))[1]
            mine = how[rank]
            nb = len(red.buckets)
            if mine == 'reverse':                      # backward's order
                for m in reversed(mods):
                    red.grad_ready(m)
            elif mine == 'shuffled':                   # a completion order with inversions (another shape, another lane layout)
                for i in (5, 7, 6, 1, 0, 3, 2, 4):
                    red.grad_ready(mods[i])
            elif mine == 'replay':                     # a replayed step: every bucket marked, handed over after the replay
                red.launch_behind({k: () for k in reversed(range(nb))}, lambda k, side: None)
            elif mine == 'replay-partial':             # markers for buckets 0 and 2 only: 0 goes out, 1.. wait for finish()
                red.launch_behind({0: (), 2: ()}, lambda k, side: None)
                assert red._next == 1
            red.finish()                               # 'finish-only': nothing reported at all
            assert order == [b.end - b.begin for b in red.buckets], (mine, order)     # bucket-index order, every time
            both = [torch.zeros(total) for _ in range(world)]
            dist.all_gather(both, grads)
            assert torch.equal(flat, both[0] + both[1]), mine
            issued.append(order)
        results[rank] = 'ok'
    except Exception as e:
        results[rank] = 'fail: %r' % (e,)
    finally:
        dist.destroy_process_group()


def test_ranks_may_run_a_step_differently_two_ranks_gloo():
    """the Trainer's AUTO mode decides per rank whether a shape is replayed, and under --multi-scale the
    ranks see different shape sequences -- so in one and the same step one rank may run eagerly with per-bucket callbacks
    (in whatever order its backward completes the buckets), another replay a captured step (launch_behind), a third leave
    everything to finish().  No collective may hang on such a per-rank condition: every path issues the bucket collectives
    in bucket-index order, and the sums are right whatever the mix."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_order_worker, args=(world, port, results), nprocs=world, join=True)
    assert dict(results) == {0: 'ok', 1: 'ok'}, dict(results)


def test_reducer_is_inert_without_process_group():
    mods, segs, total = _layout()
    flat = torch.ones(total)
    red = GradReducer(flat, segs, bucket_bytes=1024)
    assert not red.enabled and red.grad_scale == 1.0
    for m in mods:
        red.grad_ready(m)
    red.finish()
    assert torch.equal(flat, torch.ones(total))


def test_arena_layout_and_buckets_on_real_model():
    """Arena views keep torch shapes / state-dict keys; conv weights are stored [Cout][kh][kw][Cin] padded to 4."""
    from pytorch_segmentation_amd import prepare
    from pytorch_segmentation_amd.models import DeepLabV3Plus
    from oracle import models as omodels
    ref = omodels.DeepLabV3Plus(21)
    m = DeepLabV3Plus(21)
    m.load_state_dict(ref.state_dict())
    ar = prepare(m, 'cpu')
    assert set(m.state_dict()) == set(ref.state_dict())
    for k, v in ref.state_dict().items():
        assert torch.equal(m.state_dict()[k], v), k
    w = m.cls_conv
    assert tuple(w._raw['weight'].shape) == (24, 3, 3, 384) and w._raw['weight'][21:].abs().max() == 0
    assert torch.equal(w._raw['weight'][:21].permute(0, 3, 1, 2), ref.cls_conv.weight)
    stem = m.backbone.conv1
    assert tuple(stem._raw['weight'].shape) == (64, 7, 7, 4) and stem._raw['weight'][..., 3].abs().max() == 0
    # every parameter and its gradient are views into the two flat buffers
    lo, hi = ar.params.data_ptr(), ar.params.data_ptr() + ar.numel * 4
    assert all(lo <= p.data_ptr() < hi for p in m.parameters())
    assert all(p.grad is not None and ar.grads.data_ptr() <= p.grad.data_ptr() < ar.grads.data_ptr() + ar.numel * 4
               for p in m.parameters())
    red = GradReducer(ar.grads, [(s.module, s.offset, s.numel) for s in ar.segments], bucket_bytes=32 << 20)
    sizes = [(b.end - b.begin) * 4 for b in red.buckets]
    assert sum(sizes) == ar.numel * 4 and 4 <= len(red.buckets) <= 8, sizes



def _buffers_worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import torch.nn as nn
        from pytorch_segmentation_amd.nn import BatchNorm2d
        from pytorch_segmentation_amd.utils import broadcast_buffers
        torch.manual_seed(rank)
        m = nn.Sequential(BatchNorm2d(8), nn.Conv2d(8, 4, 1), BatchNorm2d(4))
        with torch.no_grad():
            for b in m.buffers():
                if b.is_floating_point():
                    b.copy_(torch.randn_like(b) + rank)
        m[0].__dict__['_nbt_pending'] = 5 + rank          # lazily counted batches: flushed before the broadcast
        m[2].num_batches_tracked += 3 * (rank + 1)
        mine = {k: v.clone() for k, v in m.state_dict().items() if 'running' in k}
        broadcast_buffers(m, 0)
        sd = m.state_dict()
        gathered = [None] * world
        dist.all_gather_object(gathered, {k: v.clone() for k, v in sd.items() if 'running' in k or 'num_batches' in k})
        for other in gathered[1:]:
            for k in gathered[0]:
                assert torch.equal(other[k], gathered[0][k]), k
        if rank == 0:
            for k, v in mine.items():
                assert torch.equal(sd[k], v), k                # rank 0's values are the ones kept
        assert int(sd['0.num_batches_tracked']) == 5 and int(sd['2.num_batches_tracked']) == 3
        results[rank] = 'ok'
    except Exception as e:
        results[rank] = 'fail: %r' % (e,)
    finally:
        dist.destroy_process_group()


def test_broadcast_buffers_two_ranks_gloo():
    """DistributedDataParallel's broadcast_buffers semantics (utils/dist.py::broadcast_buffers): BatchNorm running
    statistics AND counters (incl. the lazily counted num_batches_tracked) of every rank become rank 0's."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_buffers_worker, args=(world, port, results), nprocs=world, join=True)
    assert dict(results) == {0: 'ok', 1: 'ok'}, dict(results)


# ---- world sizes 3 / 4 / 8 on the REAL DeepLabV3+ gradient arena ------------------------------------
# BASELINE configs[3] is 8 ranks; no 8-GPU node has ever run this code, so everything about the exchange that does not need
# RCCL itself -- bucket cuts, in-place reduce-scatter / all-gather offsets, the < world remainder all-reduce, the tail bucket,
# the issue order under mixed step modes -- runs here with 3, 4 and 8 gloo ranks over the arena layout of the benchmark model.
# Reference: train.py:33-35,112-117 (DistributedSampler + init_process_group), test.py:51-58 (counter all-reduce).

_DL_LAYOUT = None


def _deeplab_layout():
    """[(module index, offset, numel)] of the DeepLabV3+ (21 classes) arena -- built once in the parent, shipped to the ranks
    (a rank only needs module identity, offsets and sizes; 39.2 M floats)."""
    global _DL_LAYOUT
    if _DL_LAYOUT is None:
        from pytorch_segmentation_amd import prepare
        from pytorch_segmentation_amd.models import DeepLabV3Plus
        m = DeepLabV3Plus(21)
        ar = prepare(m, 'cpu')
        index = {}
        _DL_LAYOUT = ([(index.setdefault(id(s.module), len(index)), s.offset, s.numel) for s in ar.segments], ar.numel)
    return _DL_LAYOUT


def _int_grads(total, rank, step):
    """integer-valued fp32 'gradients' (|v| < 2^11): every partial sum over <= 8 ranks is exact in fp32, so the reduced arena
    must equal the single-process sum BIT FOR BIT whatever order a ring / tree / direct exchange adds them in."""
    g = torch.Generator().manual_seed(1000 * step + rank)
    return torch.randint(-2047, 2048, (total,), generator=g, dtype=torch.int32).to(torch.float32)


def _real_arena_worker(rank, world, port, layout, total, results):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        nmod = 1 + max(i for i, _, _ in layout)
        mods = [_Mod() for _ in range(nmod)]
        segs = [(mods[i], off, n) for i, off, n in layout]
        # backward's reporting order: modules in reverse arena order
        order_bwd = sorted(range(nmod), key=lambda i: -max(off for j, off, _ in layout if j == i))
        expect = {}
        step = 0
        for mode in ('allreduce', 'rs_ag'):
            os.environ['PSEG_EXCHANGE'] = mode
            flat = torch.empty(total, dtype=torch.float32)
            red = GradReducer(flat, segs, bucket_bytes=32 << 20)
            sizes = [b.end - b.begin for b in red.buckets]
            assert red.world == world and red.rank == rank and sum(sizes) == total and 4 <= len(sizes) <= 8, sizes
            assert sizes[-1] * 4 <= (4 << 20), sizes                 # the tail bucket (nothing overlaps it) was cut small
            covered = sorted((b.begin, b.end) for b in red.buckets)
            assert covered[0][0] == 0 and covered[-1][1] == total
            assert all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
            if world == 3:                                           # arena segments are multiples of 4: world 3 has remainders
                assert any(s % world for s in sizes), sizes
            issued = []
            plain = red._all_reduce
            red._all_reduce = lambda view, plain=plain, issued=issued: (issued.append(view.numel()), plain(view))[1]
            # how each rank runs the step: eager in backward order, eager with inversions, replayed (all buckets marked),
            # replayed with a gap in the markers, nothing reported (finish() only)
            hows = ('reverse', 'shuffled', 'replay', 'replay-partial', 'finish-only')
            for trial in range(2):
                step += 1
                flat.copy_(_int_grads(total, rank, step))
                del issued[:]
                mine = hows[(rank + trial) % len(hows)] if trial else 'reverse'
                nb = len(red.buckets)
                if mine == 'reverse':
                    for i in order_bwd:
                        red.grad_ready(mods[i])
                elif mine == 'shuffled':
                    g = torch.Generator().manual_seed(rank)
                    for i in torch.randperm(nmod, generator=g).tolist():
                        red.grad_ready(mods[i])
                elif mine == 'replay':
                    red.launch_behind({k: () for k in range(nb)}, lambda k, side: None)
                    assert red._next == nb
                elif mine == 'replay-partial':
                    red.launch_behind({0: (), 2: ()}, lambda k, side: None)
                    assert red._next == 1
                red.finish()
                assert issued == sizes, (mine, issued, sizes)          # bucket-index order on every rank, every mode
                want = expect.get(step)
                if want is None:
                    want = torch.zeros(total, dtype=torch.float32)
                    for r in range(world):
                        want += _int_grads(total, r, step)
                    expect[step] = want
                assert torch.equal(flat, want), (mode, mine, float((flat - want).abs().max()))
        results[rank] = 'ok'
    except Exception as e:
        import traceback
        results[rank] = 'fail: %r\n%s' % (e, traceback.format_exc())
    finally:
        os.environ.pop('PSEG_EXCHANGE', None)
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [3, 4, 8])
def test_grad_reducer_real_arena_many_ranks_gloo(world):
    """3, 4 and 8 gloo ranks over the DeepLabV3+ arena (39.2 M floats, the production 32 MiB buckets + 4 MiB tail) under
    both exchanges; in the second trial of each the ranks run the step five different ways at once.  Reduced arena == the
    single-process sum bit for bit (integer-valued gradients: exact in any order), collectives in bucket-index order."""
    layout, total = _deeplab_layout()
    port = _free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_real_arena_worker, args=(world, port, layout, total, results), nprocs=world, join=True)
    assert dict(results) == {r: 'ok' for r in range(world)}, dict(results)


def _remainder_worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        # segment sizes that leave every remainder 1 .. world-1 somewhere, plus buckets SMALLER than the world size
        # (rs_ag must fall back to the plain all-reduce there: per-rank slice would be empty)
        sizes = [world * 37 + r for r in range(1, world)] + [3, 1, world - 1, world, world + 1, 5 * world + 2]
        mods, segs, off = [], [], 0
        for n in sizes:
            m = _Mod()
            mods.append(m)
            segs.append((m, off, n))
            off += n
        total = off
        for mode in ('allreduce', 'rs_ag'):
            os.environ['PSEG_EXCHANGE'] = mode
            for bucket_bytes in (4, 4 * (world + 3), 4 * 64 * world):   # one segment per bucket ... everything in two
                flat = _int_grads(total, rank, 7)
                red = GradReducer(flat, segs, bucket_bytes=bucket_bytes, tail_bytes=4 * 2)
                bs = [b.end - b.begin for b in red.buckets]
                assert sum(bs) == total
                if bucket_bytes == 4:
                    assert sorted(bs) == sorted(sizes) and any(0 < s < world for s in bs)
                    assert {s % world for s in bs} >= set(range(world)), bs          # every remainder class is present
                for m in reversed(mods):
                    red.grad_ready(m)
                red.finish()
                want = torch.zeros(total)
                for r in range(world):
                    want += _int_grads(total, r, 7)
                assert torch.equal(flat, want), (mode, bucket_bytes, bs)
        cnt = torch.arange(3 * 21, dtype=torch.int64).view(3, 21) * (rank + 1)      # test.py:51-58 at 21 classes
        all_reduce_counters(cnt)
        assert torch.equal(cnt, torch.arange(3 * 21, dtype=torch.int64).view(3, 21) * (world * (world + 1) // 2))
        results[rank] = 'ok'
    except Exception as e:
        import traceback
        results[rank] = 'fail: %r\n%s' % (e, traceback.format_exc())
    finally:
        os.environ.pop('PSEG_EXCHANGE', None)
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [3, 4, 8])
def test_exchange_remainders_many_ranks_gloo(world):
    """numel % world != 0, explicitly: every remainder 1 .. world-1, buckets shorter than the world size, one segment per bucket
    up to everything in two buckets -- reduce-scatter slices, all-gather offsets and the remainder all-reduce of
    utils/dist.py::_rs_ag against the plain all-reduce and against the exact sum; plus the evaluation counters."""
    port = _free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_remainder_worker, args=(world, port, results), nprocs=world, join=True)
    assert dict(results) == {r: 'ok' for r in range(world)}, dict(results)


def test_frozen_modules_are_not_waited_for_and_a_stall_warns(monkeypatch):
    """with the strict bucket-index order a module that never reports (all parameters frozen, or simply unused this
    step) in a LOW-index bucket holds every later collective back until finish().  Frozen modules are taken out of the wait
    set at construction; a step that reported layer by layer and still left more than the tail bucket to finish() warns once."""
    import warnings
    import torch.nn as nn
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', str(_free_port()))
    monkeypatch.setenv('PSEG_FORCE_REDUCER', '1')
    dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        mods = [nn.Linear(16, 16, bias=False) for _ in range(6)]            # 256 floats each: one bucket per module at 1 KiB
        for p in mods[5].parameters():                                      # the TOP module (bucket 0) is frozen
            p.requires_grad_(False)
        segs = [(m, 256 * i, 256) for i, m in enumerate(mods)]
        flat = torch.ones(256 * 6)
        red = GradReducer(flat, segs, bucket_bytes=1024, tail_bytes=4)
        assert red.enabled and len(red.buckets) == 6 and red.buckets[0].total == 0
        issued = []
        plain = red._all_reduce
        red._all_reduce = lambda view, plain=plain: (issued.append(view.numel()), plain(view))[1]
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            red.grad_ready(mods[4])            # bucket 1 completes: buckets 0 (frozen: nothing to wait for) and 1 go out NOW
            assert len(issued) == 2
            for m in reversed(mods[:4]):
                red.grad_ready(m)
            assert len(issued) == 6
            red.finish()
        assert not [x for x in w if 'GradReducer' in str(x.message)]
        # an unused parameter in bucket 1: everything behind it waits for finish() -- and says so, once
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            for _ in range(2):
                del issued[:]
                for m in reversed(mods[:4]):
                    red.grad_ready(m)
                assert len(issued) == 1            # only the frozen bucket 0 could go
                red.finish()
                assert len(issued) == 6
        msgs = [str(x.message) for x in w if 'GradReducer' in str(x.message)]
        assert len(msgs) == 1 and 'bucket 1 never' in msgs[0], msgs
    finally:
        dist.destroy_process_group()

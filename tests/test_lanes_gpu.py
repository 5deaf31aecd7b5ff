"""Lane executor (csrc/lanes.hip) at the C ABI: a hipGraph captured by torch -- kernels on two forked streams and a memset --
replayed as plain launches must produce what eager execution produces, also after the inputs change, for max_lanes 1
(everything on the caller's stream) and 4.  A graph holding a device-to-device copy node must be REFUSED (hipGraphMemcpyNode
parameters of a captured hipMemcpyAsync cannot be read back), never mis-replayed -- and never handed to hipGraphLaunch."""
import ctypes

import pytest
import torch

from pytorch_segmentation_amd import ops as ops_mod

pytestmark = pytest.mark.gpu


def _distinct_streams(dev, n):
    """n + 1 torch streams with pairwise different HIP handles: [capture stream, side streams ...].  torch hands out its 32 pool
    streams per device round-robin, so in a long test session a fresh `torch.cuda.Stream()` can BE the stream torch.cuda.graph
    captures on -- a 'forked' step then has no fork and one lane (round 6: seen once the suite grew by two files)."""
    got = []
    for _ in range(80):
        s = torch.cuda.Stream(device=dev)
        if all(s.cuda_stream != g.cuda_stream for g in got):
            got.append(s)
            if len(got) == n + 1:
                return got
    raise RuntimeError('torch stream pool exhausted')


def _step(x, out, side, with_copy=False):
    """a small forked computation writing `out` (static buffers, graph-capturable)."""
    cur = torch.cuda.current_stream()
    a = x * 2.0
    ev = torch.cuda.Event()
    ev.record(cur)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        b = torch.sin(x) + 1.0          # second lane
        b2 = b * b
    z = torch.empty_like(x)
    z.zero_()                           # memset node (or a fill kernel)
    z.add_(a)
    if with_copy:
        w = torch.empty_like(x)
        w.copy_(a)                      # device-to-device copy node
    else:
        w = a + 0.0
    ev2 = torch.cuda.Event()
    ev2.record(side)
    cur.wait_event(ev2)
    torch.add(z + w, b2, out=out)
    b.record_stream(cur)
    b2.record_stream(cur)


@pytest.mark.parametrize('max_lanes', [1, 4])
def test_lane_executor_replays_a_forked_graph(max_lanes):
    from pytorch_segmentation_amd import _lib
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    x = torch.randn(1 << 16, device=dev)
    out = torch.zeros_like(x)
    cap, side = _distinct_streams(dev, 1)
    _step(x, out, side)                 # warm the allocator / lazy init outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with ops_mod.no_gc_capture(g, stream=cap):
        _step(x, out, side)
    h = ctypes.c_int64(0)
    _lib.call('pseg_lanes_build', g.raw_cuda_graph(), max_lanes, ctypes.byref(h))
    info = [ctypes.c_int(0) for _ in range(4)]
    _lib.call('pseg_lanes_info', h.value, *[ctypes.byref(i) for i in info])
    nodes, launches, lanes, events = (i.value for i in info)
    assert launches >= 6 and nodes >= launches
    assert lanes == (1 if max_lanes == 1 else 2) and (events > 0) == (max_lanes > 1)
    for trial in range(3):
        x.copy_(torch.randn(1 << 16, device=dev))
        out.fill_(float('nan'))
        _lib.call('pseg_lanes_launch', h.value, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        a = x * 2.0
        b = torch.sin(x) + 1.0
        ref = (a + (a + 0.0)) + b * b
        assert torch.equal(out, ref), trial
    _lib.call('pseg_lanes_destroy', h.value)


def test_lane_executor_refuses_memcpy_nodes():
    from pytorch_segmentation_amd import _lib
    dev = torch.device('cuda', 0)
    x = torch.randn(1 << 12, device=dev)
    out = torch.zeros_like(x)
    cap, side = _distinct_streams(dev, 1)
    _step(x, out, side, with_copy=True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with ops_mod.no_gc_capture(g, stream=cap):
        _step(x, out, side, with_copy=True)
    h = ctypes.c_int64(0)
    with pytest.raises(_lib.PsegError, match='memcpy node'):
        _lib.call('pseg_lanes_build', g.raw_cuda_graph(), 4, ctypes.byref(h))
    assert h.value == 0
    # A refused graph is NOT replayed through hipGraphLaunch (round 4: this very call -- a hipGraphLaunch of this forked
    # graph -- died inside the runtime, profiles/r04_segfault_full_suite_run6.log): the caller runs the
    # step eagerly.  tests/test_models_gpu.py::test_trainer_runs_a_refused_step_eagerly covers the Trainer's side of that.
    del g
    x.copy_(torch.randn(1 << 12, device=dev))
    _step(x, out, side, with_copy=True)
    torch.cuda.synchronize()
    a = x * 2.0
    assert torch.equal(out, a + a + (torch.sin(x) + 1.0) ** 2)


def test_lane_executor_markers_order_outside_work():
    """pseg_mark / pseg_lanes_bind_markers / pseg_lanes_wait_marker: a marker set in the middle of a captured chain (on both
    forked streams) is an event of the REPLAY; a consumer on a third stream that waits for it sees everything enqueued before
    the marker -- and the consumer is not part of the graph (this is how the gradient exchange hangs behind a replayed
    backward pass).  Also: unknown marker ids are refused, markers bind once."""
    from pytorch_segmentation_amd import _lib
    dev = torch.device('cuda', 0)
    n = 1 << 22
    x = torch.randn(n, device=dev)
    early, late = torch.zeros_like(x), torch.zeros_like(x)
    side_early = torch.zeros_like(x)
    marks = torch.zeros(4, dtype=torch.int32, device=dev)
    cap, side, consumer = _distinct_streams(dev, 2)

    def step():
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            torch.mul(x, 3.0, out=side_early)
            _lib.call('pseg_mark', marks.data_ptr() + 4 * 3, side.cuda_stream)          # marker 3: the side stream's part
        torch.mul(x, 2.0, out=early)
        _lib.call('pseg_mark', marks.data_ptr() + 4 * 2, cur.cuda_stream)               # marker 2: after `early`
        t = early
        for _ in range(24):                                                              # a long tail behind the marker
            t = torch.sin(t)
        torch.add(t, 0.0, out=late)
        ev2 = torch.cuda.Event()
        ev2.record(side)
        cur.wait_event(ev2)

    step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with ops_mod.no_gc_capture(g, stream=cap):
        step()
    h = ctypes.c_int64(0)
    _lib.call('pseg_lanes_build', g.raw_cuda_graph(), 4, ctypes.byref(h))
    bound = ctypes.c_int(0)
    _lib.call('pseg_lanes_bind_markers', h.value, marks.data_ptr(), 4, ctypes.byref(bound))
    assert bound.value == 2
    with pytest.raises(_lib.PsegError):
        _lib.call('pseg_lanes_bind_markers', h.value, marks.data_ptr(), 4, ctypes.byref(bound))     # once per executor
    for trial in range(3):
        x.copy_(torch.randn(n, device=dev))
        early.fill_(float('nan')), late.fill_(float('nan')), side_early.fill_(float('nan'))
        torch.cuda.synchronize()
        _lib.call('pseg_lanes_launch', h.value, torch.cuda.current_stream().cuda_stream)
        _lib.call('pseg_lanes_wait_marker', h.value, 2, consumer.cuda_stream)
        _lib.call('pseg_lanes_wait_marker', h.value, 3, consumer.cuda_stream)
        with torch.cuda.stream(consumer):
            seen = early + side_early          # reads what lies BEFORE the markers, while the tail is still running
        with pytest.raises(_lib.PsegError, match='no marker'):
            _lib.call('pseg_lanes_wait_marker', h.value, 1, consumer.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(seen, x * 2.0 + x * 3.0), trial
        assert torch.isfinite(late).all()
    _lib.call('pseg_lanes_destroy', h.value)


def test_lane_executor_rejects_bad_arguments():
    from pytorch_segmentation_amd import _lib
    h = ctypes.c_int64(0)
    with pytest.raises(_lib.PsegError):
        _lib.call('pseg_lanes_build', 0, 2, ctypes.byref(h))
    with pytest.raises(_lib.PsegError):
        _lib.call('pseg_lanes_launch', 0, 0)
    _lib.call('pseg_lanes_destroy', 0)


def test_lane_executor_reuses_joined_lanes_and_borrows_streams():
    """A step that forks two side chains and joins them again, six times over (HRNet's modules: ops.Branches).  The executor
    must put the side chains of EVERY round on the same two extra lanes -- a lane whose tail has been joined is idle -- not
    open lanes until they run out and queue the rest behind the main chain; max_lanes = 3 is enough.  Two executors of the same
    graph share the process-wide lane streams (pseg_lanes_reserve); results equal eager execution."""
    from pytorch_segmentation_amd import _lib
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    x = torch.randn(1 << 14, device=dev)
    out = torch.zeros_like(x)
    cap, *sides = _distinct_streams(dev, 2)

    def step():
        cur = torch.cuda.current_stream()
        acc = x * 1.0
        keep = []
        for r in range(6):
            ev = torch.cuda.Event()
            ev.record(cur)
            parts = []
            for k, s in enumerate(sides):
                s.wait_event(ev)
                with torch.cuda.stream(s):
                    p = torch.sin(acc * float(k + 1)) * 0.5
                    p = p + float(r)
                parts.append(p)
            m = acc * 0.5                       # the main chain works meanwhile
            for p, s in zip(parts, sides):
                cur.wait_stream(s)
                p.record_stream(cur)
                m = m + p
            acc.record_stream(sides[0])
            acc.record_stream(sides[1])
            keep.append(acc)
            acc = m
        torch.add(acc, 0.0, out=out)          # (a kernel, not a device-to-device copy node)
        return keep

    def reference():
        acc = x * 1.0
        for r in range(6):
            m = acc * 0.5
            for k in range(2):
                m = m + (torch.sin(acc * float(k + 1)) * 0.5 + float(r))
            acc = m
        return acc

    step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with ops_mod.no_gc_capture(g, stream=cap):
        step()
    _lib.call('pseg_lanes_reserve', 3)
    with pytest.raises(_lib.PsegError, match='lanes'):
        _lib.call('pseg_lanes_reserve', 99)
    handles = []
    for _ in range(2):
        h = ctypes.c_int64(0)
        _lib.call('pseg_lanes_build', g.raw_cuda_graph(), 3, ctypes.byref(h))
        info = [ctypes.c_int(0) for _ in range(4)]
        _lib.call('pseg_lanes_info', h.value, *[ctypes.byref(i) for i in info])
        nodes, launches, lanes, events = (i.value for i in info)
        assert lanes == 3 and events >= 6 * 3, (lanes, events)
        handles.append(h.value)
    for trial in range(4):
        x.copy_(torch.randn(1 << 14, device=dev))
        out.fill_(float('nan'))
        _lib.call('pseg_lanes_launch', handles[trial % 2], torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(out, reference()), trial
    for hv in handles:
        _lib.call('pseg_lanes_destroy', hv)


def test_branches_fork_join_and_deferred_callbacks(monkeypatch):
    """ops.Branches by itself (eager forks switched on): lane i > 0 runs on a stream of its own, ordered behind everything the
    current stream held when the region opened; join() makes the current stream wait for every lane; callbacks registered with
    after_branches inside the region run at the join, outside it at once; a region opened inside a region stays on one stream."""
    monkeypatch.setattr(ops_mod, 'BRANCH_EAGER', True)
    dev = torch.device('cuda', 0)
    main = torch.cuda.current_stream(dev)
    x = torch.randn(1 << 22, device=dev)
    for _ in range(4):
        x = torch.sin(x) * 1.0001 + 0.1            # a long producer on the main stream: the lanes must wait for it
    ran = []
    ops_mod.after_branches(lambda: ran.append('now'))
    assert ran == ['now']
    br = ops_mod.Branches(dev, 3)
    assert br.on
    outs, streams = [None] * 3, []
    for i in range(3):
        with br.lane(i, x):
            streams.append(torch.cuda.current_stream(dev).cuda_stream)
            y = x
            for _ in range(6):
                y = torch.cos(y) + float(i)
            outs[i] = y
            ops_mod.after_branches(lambda i=i: ran.append(i))
            inner = ops_mod.Branches(dev, 2)       # nested: off
            assert not inner.on
            inner.join()
    assert streams[0] == main.cuda_stream and len(set(streams)) == 3
    assert ran == ['now']                          # deferred while the region is open
    br.join(*outs)
    assert ran == ['now', 0, 1, 2]
    total = outs[0] + outs[1] + outs[2]            # on the main stream, right behind the join
    ref = x
    refs = []
    for i in range(3):
        y = ref
        for _ in range(6):
            y = torch.cos(y) + float(i)
        refs.append(y)
    torch.cuda.synchronize()
    assert torch.equal(total, refs[0] + refs[1] + refs[2])
    # off outside a capture when the eager switch is off
    monkeypatch.setattr(ops_mod, 'BRANCH_EAGER', False)
    off = ops_mod.Branches(dev, 3)
    assert not off.on
    with off.lane(1, x):
        assert torch.cuda.current_stream(dev).cuda_stream == main.cuda_stream
    off.join()

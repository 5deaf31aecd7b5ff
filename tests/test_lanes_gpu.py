"""Lane executor (csrc/lanes.hip) at the C ABI: a hipGraph captured by torch -- kernels on two forked streams and a memset --
replayed as plain launches must produce what eager execution produces, also after the inputs change, for max_lanes 1
(everything on the caller's stream) and 4.  A graph holding a device-to-device copy node must be REFUSED (hipGraphMemcpyNode
parameters of a captured hipMemcpyAsync cannot be read back), never mis-replayed."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


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
    side = torch.cuda.Stream(device=dev)
    _step(x, out, side)                 # warm the allocator / lazy init outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
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
    side = torch.cuda.Stream(device=dev)
    _step(x, out, side, with_copy=True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        _step(x, out, side, with_copy=True)
    h = ctypes.c_int64(0)
    with pytest.raises(_lib.PsegError, match='memcpy node'):
        _lib.call('pseg_lanes_build', g.raw_cuda_graph(), 4, ctypes.byref(h))
    assert h.value == 0
    g.instantiate()                     # the fallback the Trainer takes: hipGraphLaunch
    x.copy_(torch.randn(1 << 12, device=dev))
    g.replay()
    torch.cuda.synchronize()
    a = x * 2.0
    assert torch.equal(out, a + a + (torch.sin(x) + 1.0) ** 2)


def test_lane_executor_rejects_bad_arguments():
    from pytorch_segmentation_amd import _lib
    h = ctypes.c_int64(0)
    with pytest.raises(_lib.PsegError):
        _lib.call('pseg_lanes_build', 0, 2, ctypes.byref(h))
    with pytest.raises(_lib.PsegError):
        _lib.call('pseg_lanes_launch', 0, 0)
    _lib.call('pseg_lanes_destroy', 0)

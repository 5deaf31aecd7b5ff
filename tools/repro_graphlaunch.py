"""Does a lane executor that ran on BORROWED torch streams break a later hipGraphLaunch (torch CUDAGraph.replay) of a forked
graph?  usage: python tools/repro_graphlaunch.py [borrow=1] [destroy=1] [same_side=1]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from pytorch_segmentation_amd import _lib, ops as ops_mod  # noqa: E402
from test_lanes_gpu import _step  # noqa: E402

opts = dict(a.split('=') for a in sys.argv[1:])
borrow = opts.get('borrow', '1') == '1'
destroy = opts.get('destroy', '1') == '1'
same_side = opts.get('same_side', '1') == '1'
dev = torch.device('cuda', 0)
x = torch.randn(1 << 14, device=dev)
out = torch.zeros_like(x)
side = torch.cuda.Stream(device=dev)
_step(x, out, side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(keep_graph=True)
with ops_mod.no_gc_capture(g):
    _step(x, out, side)
h = ctypes.c_int64(0)
_lib.call('pseg_lanes_build', g.raw_cuda_graph(), 4, ctypes.byref(h))
if borrow:
    used = ctypes.c_int(0)
    _lib.call('pseg_lanes_use_streams', h.value, (ctypes.c_int64 * 1)(side.cuda_stream), 1, ctypes.byref(used))
    print('borrowed', used.value)
for _ in range(3):
    _lib.call('pseg_lanes_launch', h.value, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
if destroy:
    _lib.call('pseg_lanes_destroy', h.value)
print('executor done')
side2 = side if same_side else torch.cuda.Stream(device=dev)
x2 = torch.randn(1 << 12, device=dev)
out2 = torch.zeros_like(x2)
_step(x2, out2, side2, with_copy=True)
torch.cuda.synchronize()
g2 = torch.cuda.CUDAGraph(keep_graph=True)
with ops_mod.no_gc_capture(g2):
    _step(x2, out2, side2, with_copy=True)
g2.instantiate()
torch.cuda.synchronize()
for _ in range(3):
    g2.replay()
torch.cuda.synchronize()
a = x2 * 2.0
print('replay ok', torch.equal(out2, a + a + (torch.sin(x2) + 1.0) ** 2))

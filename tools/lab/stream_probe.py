import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ss = [torch.cuda.Stream() for _ in range(40)]
hs = [s.cuda_stream for s in ss]
print('40 fresh torch streams: %d distinct handles; first repeats at index %s' % (len(set(hs)), next((i for i, h in enumerate(hs) if h in hs[:i]), None)))
print([hex(h) for h in hs[:8]])
g = torch.cuda.graph(torch.cuda.CUDAGraph())
print('default capture stream', hex(g.capture_stream.cuda_stream), 'index in list', hs.index(g.capture_stream.cuda_stream) if g.capture_stream.cuda_stream in hs else None)
from pytorch_segmentation_amd import ops
from pytorch_segmentation_amd import models as zoo
from pytorch_segmentation_amd.utils import Trainer, compute_loss
import bench
m = zoo.HRNet(21)
tr = Trainer(m, None, loss_fn=compute_loss, lr=1e-3, device=torch.device('cuda', 0))
m.train()
x, t = bench.synthetic_batch(8, 512, 21, torch.device('cuda', 0), 1)
for _ in range(6):
    tr.train_batch(x, t)
torch.cuda.synchronize()
print('OWN_STREAMS', ops.OWN_STREAMS, 'held', {k: [hex(h) for h in v] for k, v in ops._own_stream_handles.items()})
print('aux', [hex(s.cuda_stream) for p in ops._aux_streams.values() for s in p], 'branch', [hex(s.cuda_stream) for p in ops._branch_pool.values() for s in p] if hasattr(ops, '_branch_pool') else None)
print('exchange', hex(tr.reducer._side.cuda_stream) if tr.reducer._side is not None else None)
print('capture', hex(torch.cuda.graph.default_capture_stream.cuda_stream) if torch.cuda.graph.default_capture_stream is not None else None)
print('step mode', tr.step_mode())

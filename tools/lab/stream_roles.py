import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_segmentation_amd import ops
from pytorch_segmentation_amd import models as zoo
from pytorch_segmentation_amd.utils import Trainer, compute_loss
import bench
dev = torch.device('cuda', 0)
m = zoo.DeepLabV3Plus(21)
tr = Trainer(m, None, loss_fn=compute_loss, lr=1e-3, device=dev)
m.train()
x, t = bench.synthetic_batch(4, 256, 21, dev, 1)
seen = []
orig_w, orig_d, orig_f = ops.conv2d_wgrad, ops.conv2d_dgrad, ops.conv2d_fwd
def tag(name, fn):
    def w(*a, **k):
        cs = torch.cuda.current_stream()
        seen.append((name, cs.stream_id, hex(cs.cuda_stream)))
        return fn(*a, **k)
    return w
ops.conv2d_wgrad, ops.conv2d_dgrad, ops.conv2d_fwd = tag('wgrad', orig_w), tag('dgrad', orig_d), tag('fwd', orig_f)
for _ in range(2):
    tr.train_batch(x, t)
torch.cuda.synchronize()
from collections import Counter
print(Counter(seen).most_common())
print('null stream', torch.cuda.default_stream().stream_id, hex(torch.cuda.default_stream().cuda_stream), 'current', torch.cuda.current_stream().stream_id)
print('aux pool', [(s.stream_id, hex(s.cuda_stream)) for p in ops._aux_streams.values() for s in p])

import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, '/root/repo')
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = '29577'; os.environ['PSEG_FORCE_REDUCER'] = '1'
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
import bench
from pytorch_segmentation_amd.models import DeepLabV3Plus
from pytorch_segmentation_amd.utils import Trainer, compute_loss
m = DeepLabV3Plus(21)
tr = Trainer(m, None, loss_fn=compute_loss, accumulate=1, lr=1e-3)
assert tr.reducer.enabled
m.train()
x, t = bench.synthetic_batch(16, 512, 21, torch.device('cuda', 0), 1)
for _ in range(3): tr.train_batch(x, t)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): tr.train_batch(x, t)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print('forced reducer (1-rank RCCL, %d buckets): %.2f ms/step  %.1f img/s' % (len(tr.reducer.buckets), dt * 1e3, 16 / dt))
dist.destroy_process_group()

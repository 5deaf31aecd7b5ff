import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import models as om, loss as ol
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
try:
    print('cgroup cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as e:
    print('no cgroup cpu.max', e)
m = om.DeepLabV3Plus(21).train()
x = torch.randn(2, 3, 256, 256); t = torch.randint(0, 21, (2, 256, 256))
for n in (16, 32, 64, 256):
    torch.set_num_threads(n)
    def step():
        m.zero_grad(); ol.compute_loss(m(x), t).backward()
    step()
    t0 = time.perf_counter(); step(); dt = time.perf_counter() - t0
    print('threads %3d: %.2f s/step (B=2, 256x256)' % (n, dt), flush=True)

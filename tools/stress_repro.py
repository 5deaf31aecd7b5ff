"""Race screen for the LDS-DMA kernels: N full-size training steps from the same state must give bit-identical losses,
logits and gradient arenas (every reduction is fixed-order; a missed wait / early read would show as a rare mismatch)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pytorch_segmentation_amd import ops
from pytorch_segmentation_amd.models import DeepLabV3Plus, HRNet
from pytorch_segmentation_amd.utils import Trainer, compute_loss
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for name, cls, B, S in (('deeplabv3plus', DeepLabV3Plus, 16, 512), ('hrnet', HRNet, 8, 512)):
    for pol in ('fp32', 'mixed'):
        torch.manual_seed(0)
        m = cls(21)
        tr = Trainer(m, None, loss_fn=compute_loss, lr=0.0)      # lr 0: the state never changes
        tr.env.policy = pol
        m.train()
        x, t = bench.synthetic_batch(B, S, 21, 'cuda', 7)
        ref = None
        bad = 0
        for i in range(n):
            loss = tr.train_batch(x, t)
            torch.cuda.synchronize()
            cur = (loss.item(), tr.arena.grads.clone())
            if ref is None:
                ref = cur
            elif cur[0] != ref[0] or not torch.equal(cur[1], ref[1]):
                bad += 1
        print('%s %s: %d steps, %d mismatching' % (name, pol, n, bad), flush=True)
        assert bad == 0

# Second screen: the state DOES change (lr > 0) and nothing synchronises between steps -- two runs from the same seed must
# end in bit-identical parameters (the filter transposes run on the second stream beside the forward pass; a step that read
# them before they were refreshed, or refreshed them before the previous optimiser step had landed, would diverge).
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for pol, graph in (('fp32', False), ('mixed', False), ('limb', False), ('fp32', True), ('limb', True)):
    finals = []
    for run in range(2):
        torch.manual_seed(0)
        m = DeepLabV3Plus(21)
        tr = Trainer(m, None, loss_fn=compute_loss, lr=1e-2, graph=graph)
        tr.env.policy = pol
        m.train()
        x, t = bench.synthetic_batch(16, 512, 21, 'cuda', 7)
        losses = [tr.train_batch(x, t) for _ in range(steps)]
        torch.cuda.synchronize()
        finals.append((torch.stack([l.reshape(()) for l in losses]).cpu(), tr.arena.params.clone()))
        del tr, m
    same = torch.equal(finals[0][0], finals[1][0]) and torch.equal(finals[0][1], finals[1][1])
    print('train %s graph=%s: %d steps twice, identical: %s (loss %.4f -> %.4f)' % (
        pol, graph, steps, same, finals[0][0][0].item(), finals[0][0][-1].item()), flush=True)
    assert same

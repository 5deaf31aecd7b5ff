"""The data-parallel path on the real device: one RCCL rank with the reducer forced on, so bucket events, the side
stream, the NCCL(=RCCL) all-reduce calls and the join before the fused optimiser all execute on the MI355X.
(Multi-rank semantics are covered on CPU/gloo by tests/test_dist_cpu.py; the 8-GPU run is the driver's.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_trainer_step_through_rccl_reducer(monkeypatch):
    from oracle import fill
    from pytorch_segmentation_amd.models import DeepLabV3Plus
    from pytorch_segmentation_amd.utils import Trainer, compute_loss
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', str(_free_port()))
    monkeypatch.setenv('PSEG_FORCE_REDUCER', '1')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        x = fill.images('distgpu/x', (4, 3, 64, 64)).cuda()
        t = fill.labels('distgpu/t', (4, 64, 64), 21, block=8).cuda()

        def run(force):
            os.environ['PSEG_FORCE_REDUCER'] = '1' if force else '0'
            torch.manual_seed(0)
            m = DeepLabV3Plus(21)
            fill.fill_module_(m, 'distgpu')
            tr = Trainer(m, None, loss_fn=compute_loss, accumulate=2, lr=1e-2, bucket_bytes=8 << 20)
            assert tr.reducer.enabled == force
            m.train()
            losses = [tr.train_batch(x, t).item() for _ in range(4)]   # two optimiser steps of two micro-batches
            torch.cuda.synchronize()
            return losses, tr.arena.params.clone(), len(tr.reducer.buckets)

        l1, p1, nb = run(True)
        l0, p0, _ = run(False)
        assert nb >= 10                       # 157 MB of gradients in 8 MiB buckets
        assert l1 == l0                       # one rank: the all-reduce is the identity -> bit-identical training
        assert torch.equal(p1, p0)
        assert l1[2] < l1[0]                  # and it actually trains
    finally:
        dist.destroy_process_group()


def test_bench_two_rank_rehearsal():
    """bench.py under the driver's N>1 launch line (torch.distributed.run, two ranks) -- both ranks on the one GPU of the
    test box, gradients over gloo (PSEG_BENCH_REHEARSAL=1).  Not a measurement: it proves that the launch contract, the
    barriers, the max-over-ranks timing and the metered extra step (a full training step with its bucketed all-reduce,
    which EVERY rank has to take part in) run to completion and that rank 0 prints exactly one JSON line."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PSEG_BENCH_REHEARSAL='1')
    env.pop('PSEG_FORCE_REDUCER', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(_free_port()), os.path.join(repo, 'bench.py'), '--gpus', '2', '--steps', '2',
           '--warmup', '1', '--batch', '2', '--size', '128']
    r = subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['value'] > 0
    assert out['config']['global_batch'] == 4 and out['scaling'] == 'weak'
    assert out['roofline'] is not None and out['cpu_baseline'] is None

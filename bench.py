#!/usr/bin/env python3
"""Headline benchmark: images/sec of one DeepLabV3+ (ResNet-50, 21 classes) training step at 512x512.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = forward + per-pixel cross-entropy + backward + (N>1: bucketed RCCL gradient all-reduce overlapped with
backward) + fused SGD update, on a synthetic batch that is already resident in HBM (weak scaling: 16 images per GPU).
Prints ONE JSON line (rank 0).

`value` / `ms_per_step` / `dtype` are the STRICT fp32 policy: every conv -- forward, data gradient, weight gradient --
on PSEG_PREC_FP32 (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate), the reference's own arithmetic.
`other_policies` carries the same K timed steps, measured in the same process right after, under the opt-in policies:
`half` (= train.py -mp: fp16 storage, one fp16 MFMA pass, fp32 master weights, dynamic loss scaling -- with its own
roofline objects: the dominant conv class against the 2.5 PF dense fp16 peak and the BatchNorm passes against HBM), and
the reduced-product fp32-storage policies (`mixed`: backward convs on split-bf16 limbs; `limb`: forward on fp16 limbs too).
`other_configs` (N = 1): the launch-bound single-GPU configurations of BASELINE.json -- configs[4] HRNet 512x512 batch 8 and
configs[1] UNet 256x256 batch 8 -- under fp32 and `-mp`, each with a default-constructed Trainer (which replays such steps by
itself: `replayed`, `lane_executor`); parity-test cases first, bench lines second: they are NOT the headline.
Extra objects:
  roofline     -- the implicit-GEMM conv kernel class with the most device time (fwd / dgrad / wgrad) against the
                  fp32-MFMA peak.  achieved = algorithmic (in-bounds taps) conv FLOPs of a step / time inside those
                  kernels, measured with HIP events on the launch stream in a separate instrumented step (the timed
                  region itself carries no instrumentation).  traffic = L2<->fabric bytes per launch of that class from the
                  rocprofv3 PMC passes tracked under profiles/ (tools/pmc_step.sh; the counters cannot be read from
                  inside the timed process) -- the file is named in roofline.traffic_source.
  cpu_baseline -- the CPU oracle (same graph, stock torch fp32 ops = what the reference runs) timed on this host.
"""
import argparse
import json
import os

# multi-process GPU work on this ROCm host: only dmabuf IPC is supported (RCCL / tensor sharing fail on the legacy mode)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TF = 157.3  # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md, chip-level parameters)
FP16_MFMA_PEAK_TF = 2500.0  # dense fp16 / bf16 matrix peak (same table; the 5 PF headline figure includes 2:1 sparsity)
HBM_PEAK_GBS = 8000.0       # HBM3E peak (spec; ~6.3 TB/s achievable by a streaming copy)
DTYPES = {'fp32': 'f32',
          'mixed': 'f32 (forward convs exact fp32 MFMA; backward convs split-bf16 3-product MFMA, fp32 accumulate)',
          'bf16x3': 'bf16x3 (fp32 operands split into two bf16 limbs, 3 partial products, fp32 accumulate)',
          'bf16x6': 'bf16x6 (three bf16 limbs, 6 partial products, fp32 accumulate)',
          'limb': 'f32 via limbs (forward convs: 2 fp16 limbs of the amax-scaled fp32 operands, 3 partial products; '
                  'backward convs: 2 bf16 limbs, 3 partial products; fp32 accumulate)',
          'half': 'f16 storage (train.py -mp: fp16 activations / gradients / filter copies, one fp16 MFMA pass with fp32 '
                  'accumulate, fp32 BatchNorm statistics / loss / master weights, dynamic loss scaling)'}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=16, help='images per GPU (BASELINE.json configs[2]: 16)')
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--classes', type=int, default=21)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=2)
    ap.add_argument('--precision', choices=['fp32', 'mixed', 'limb', 'half', 'bf16x3', 'bf16x6'], default='fp32',
                    help='conv arithmetic policy of the headline value (default: fp32 = every conv on exact fp32 MFMA)')
    ap.add_argument('--configs', default='hrnet,unet',
                    help='N=1: after the headline, time the other single-GPU BASELINE.json configurations (configs[4] HRNet 512x512 '
                         'B=8, configs[1] UNet 256x256 B=8; fp32 and -mp) with a default-constructed Trainer -> "other_configs".  '
                         'Skipped when --also is empty (profiling runs) or this is ""')
    ap.add_argument('--only-config', default='', help=argparse.SUPPRESS)     # (child process of --configs: one configuration, JSON out)
    ap.add_argument('--also', default='half,mixed,limb',
                    help='comma-separated policies measured after the headline in the same process (N=1 only; "" = none)')
    return ap.parse_args()


def pmc_traffic(policy, op):
    """(bytes per launch, source file) of kernel class `op` under `policy` from the newest tracked PMC summary
    (profiles/r*_pmc_traffic.json, written by tools/pmc_step.py from `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` passes
    over this very script; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  (None, None) if absent."""
    import glob
    # the file named by profiles/CURRENT (one line: the end-of-round summary; written by tools/refresh_profiles.sh) first, then
    # the rest newest-first by modification time -- never by name order ('r03a_' sorts after 'r03_')
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')), key=os.path.getmtime)
    try:
        cur = os.path.join(ROOT, 'profiles', open(os.path.join(ROOT, 'profiles', 'CURRENT')).read().split()[0])
        if os.path.exists(cur):
            files = [f for f in files if os.path.abspath(f) != os.path.abspath(cur)] + [cur]
    except (OSError, IndexError):
        pass
    for f in reversed(files):
        try:
            d = json.load(open(f))
            e = d['policies'][policy][op]
            global _TRAFFIC_SHA
            _TRAFFIC_SHA = d.get('kernels_sha16')
            return (e['fetch_bytes_per_step'] + e['write_bytes_per_step']) / e['launches_per_step'], os.path.relpath(f, ROOT)
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            continue
    return None, None


_TRAFFIC_SHA = None


def kernels_sha16():
    """sha256 over the library's HIP sources, as tools/pmc_step.py stamps its counter summaries: the line says whether the
    tracked `traffic` figure was taken on THESE kernels (traffic_stale false) or on earlier ones (true / 'unknown')"""
    import hashlib
    here = os.path.join(ROOT, 'pytorch_segmentation_amd', 'csrc')
    h = hashlib.sha256()
    for f in sorted(os.listdir(here)):
        if f.endswith('.hip') or f.endswith('.h'):
            h.update(f.encode())
            h.update(open(os.path.join(here, f), 'rb').read())
    return h.hexdigest()[:16]


def synthetic_batch(batch, size, classes, device, seed):
    """uint8-uniform images through the reference's normalisation (utils/datasets.py:199-205), uniform labels."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    img = torch.randint(0, 256, (batch, 3, size, size), generator=g, dtype=torch.uint8).float()
    mean = torch.tensor([123.675, 116.28, 103.53]).view(1, 3, 1, 1)
    std = torch.tensor([58.395, 57.12, 57.375]).view(1, 3, 1, 1)
    tgt = torch.randint(0, classes, (batch, size, size), generator=g, dtype=torch.int64)
    return ((img - mean) / std).to(device), tgt.to(device)


class ConvMeter:
    """Wraps the three conv entry points with HIP-event pairs on the launch stream and counts their FLOPs."""

    def __init__(self, ops, model=None):
        self.ops = ops
        self.records = []
        self._orig = {}
        # algorithmic FLOPs are counted on the layers' LOGICAL channel counts: the kernels see channels padded to 4 (fp32) or
        # 8 (fp16 storage) -- the 3-channel stem, the 21-class classifier -- and that padding is not work the model asked for
        self._logical = {}
        if model is not None:
            for m in model.modules():
                if isinstance(m, torch.nn.Conv2d) and m.groups == 1:
                    for q in (4, 8):
                        key = ((m.in_channels + q - 1) // q * q, (m.out_channels + q - 1) // q * q) + tuple(m.kernel_size)
                        self._logical.setdefault(key, (m.in_channels, m.out_channels))

    def _channels(self, cin_p, cout_p, kh, kw):
        return self._logical.get((cin_p, cout_p, kh, kw), (cin_p, cout_p))

    @staticmethod
    def _inbounds_fraction(H, Ho, k, stride, pad, dil):
        if k == 1:
            return 1.0
        cnt = 0
        for o in range(Ho):
            for r in range(k):
                i = o * stride - pad + r * dil
                cnt += 0 <= i < H
        return cnt / float(Ho * k)

    def __enter__(self):
        ops = self.ops

        def timed(name, flops_fn):
            orig = getattr(ops, name)
            self._orig[name] = orig

            def wrapper(*a, **kw):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = orig(*a, **kw)
                e1.record()
                self.records.append((name, e0, e1) + flops_fn(*a, **kw))
                return r
            setattr(ops, name, wrapper)

        def f_fwd(x, w, b, y, kh, kw, s, p, d, **_):
            dense = 2.0 * y.M * y.C * x.C * kh * kw             # what the kernel executes (padded channels, every tap)
            cin, cout = self._channels(x.C, y.C, kh, kw)
            fr = self._inbounds_fraction(x.H, y.H, kh, s, p, d) * self._inbounds_fraction(x.W, y.W, kw, s, p, d)
            return dense, 2.0 * y.M * cout * cin * kh * kw * fr

        def f_dgrad(dy, wT, dx, kh, kw, s, p, d, **_):
            dense = 2.0 * dx.M * dx.C * dy.C * kh * kw          # what the kernel executes (gather form)
            cin, cout = self._channels(dx.C, dy.C, kh, kw)
            fr = self._inbounds_fraction(dx.H, dy.H, kh, s, p, d) * self._inbounds_fraction(dx.W, dy.W, kw, s, p, d)
            return dense, 2.0 * dy.M * cout * cin * kh * kw * fr     # algorithmic = the forward conv's in-bounds MACs

        def f_wgrad(x, dy, dw, kh, kw, s, p, d, **_):
            dense = 2.0 * dy.M * dy.C * x.C * kh * kw
            cin, cout = self._channels(x.C, dy.C, kh, kw)
            fr = self._inbounds_fraction(x.H, dy.H, kh, s, p, d) * self._inbounds_fraction(x.W, dy.W, kw, s, p, d)
            return dense, 2.0 * dy.M * cout * cin * kh * kw * fr

        def esz(a):
            return 2.0 if a.half else 4.0

        # BatchNorm passes (HBM-bound): algorithmic bytes = every activation-sized tensor read or written once per pass
        def b_fwd(y, co, act, z, residual=None, want_mask=False):
            n = y.M * y.C * esz(y)
            return n * (2 + (residual is not None)), n * (2 + (residual is not None))

        def b_bwd(dz, z, y, co, act, dy, gg, bg, accumulate=False, dres=None, res_accumulate=False, frozen=False, mask=None,
                  want_planes=False, part=None):
            n = y.M * y.C * esz(y)
            # (part: the data gradient that produced dz already took the reduction's sums -- no reduce pass, two reads fewer)
            reads = (0 if part is not None else 2) + 2 + (2 if (z is not None and mask is None) else 0) + \
                (1 if (dres is not None and res_accumulate) else 0)
            writes = 1 + (dres is not None)
            return n * (reads + writes), n * (reads + writes)

        timed('conv2d_fwd', f_fwd)
        timed('conv2d_dgrad', f_dgrad)
        timed('conv2d_wgrad', f_wgrad)
        timed('bn_act_fwd', b_fwd)
        timed('bn_act_bwd', b_bwd)
        return self

    def __exit__(self, *exc):
        for k, v in self._orig.items():
            setattr(self.ops, k, v)

    def summary(self):
        torch.cuda.synchronize()
        ms = sum(e0.elapsed_time(e1) for _, e0, e1, _, _ in self.records)
        dense = sum(r[3] for r in self.records)
        useful = sum(r[4] for r in self.records)
        per = {}
        for name, e0, e1, dn, us in self.records:
            t = per.setdefault(name, [0.0, 0.0, 0])
            t[0] += e0.elapsed_time(e1)
            t[1] += dn
            t[2] += 1
        return ms, dense, useful, len(self.records), per


def usable_cores():
    """Cores this process may actually use: min(affinity mask, cgroup CPU quota).  (On the GPU box the quota is 16 of
    256 hardware threads; running 256 torch threads against it is ~1000x slower than 16.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(args):
    """The CPU oracle (stock torch fp32 ops composed as the reference composes them) on this host's cores."""
    from oracle import loss as oloss
    from oracle import models as omodels
    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = omodels.DeepLabV3Plus(args.classes).train()
    opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9)
    x, t = synthetic_batch(args.cpu_batch, args.size, args.classes, 'cpu', 1)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = oloss.compute_loss(model(x), t)
        loss.backward()
        opt.step()

    step()  # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:  # a bounded sample: ~12 s of CPU work, at most 40 steps
        step()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= 12.0 or n >= 40:
            break
    return {'value': args.cpu_batch * n / dt, 'unit': 'images/sec', 'cores': cores, 'cpu': cpu_model(), 'kind': 'port',
            'sample': '%d timed fwd+loss+bwd+SGD steps (after 1 warm-up) of the torch-CPU oracle, batch %d, %dx%d, '
                      'torch.set_num_threads(%d) = usable cores (affinity / cgroup quota)' % (n, args.cpu_batch, args.size, args.size, cores)}


def _decisions(tr):
    """AUTO mode's verdicts of a Trainer, JSON-able: [{shape, use, host_ms, dev_ms}] (host enqueue time against device span of
    the judged eager step; Trainer._auto_graph)"""
    return [dict(shape=list(k), **{kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()})
            for k, v in tr.graph_decisions().items()]


def timed_config(name, steps, device):
    """The launch-bound BASELINE.json configurations as `python train.py [-mp]` runs them: a default-constructed Trainer
    (AUTO graph mode: it replays these steps by itself from the fourth step on), K timed steps after eight warm-up steps."""
    from pytorch_segmentation_amd import models as zoo
    from pytorch_segmentation_amd.utils import Trainer, compute_loss
    name, _, only = name.partition(':')
    cls, B, S, nc, what = {'hrnet': (zoo.HRNet, 8, 512, 21, 'BASELINE.json configs[4]: HRNet 512x512, batch 8, 21 classes'),
                           'unet': (zoo.UNet, 8, 256, 2, 'BASELINE.json configs[1]: UNet 256x256, batch 8, 2 classes')}[name]
    xb, tb = synthetic_batch(B, S, nc, device, 4321)
    res = {'workload': what + '; fwd + cross-entropy + bwd + SGD(momentum) step'}
    for key, mp in (('fp32', False), ('half', True)):
        if only and key != only:
            continue
        torch.manual_seed(0)
        m = cls(nc)
        tr = Trainer(m, fetcher=None, loss_fn=compute_loss, accumulate=1, adam=False, lr=1e-3, device=device, mixed_precision=mp)
        m.train()
        for _ in range(8):
            tr.train_batch(xb, tb)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            ls = tr.train_batch(xb, tb)
        torch.cuda.synchronize()
        d = time.perf_counter() - t1
        lanes = [sg.lane_info for sg in tr._graphs.values() if sg is not None and getattr(sg, 'lanes', 0)]
        res[key] = {'value': B * steps / d, 'unit': 'images/sec', 'ms_per_step': d / steps * 1e3, 'steps': steps,
                    'dtype': DTYPES[key], 'loss': ls.item(), 'step_mode': tr.step_mode(), 'auto_decisions': _decisions(tr),
                    'replayed': bool(lanes), 'lane_executor': lanes[0] if lanes else None}
        tr.close()
        del tr, m
    return res



def main():
    args = parse()
    if args.only_config:
        torch.cuda.set_device(0)
        print(json.dumps(timed_config(args.only_config, args.steps, torch.device('cuda', 0))), flush=True)
        return
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback for the measured path)')
    # rehearsal on a one-GPU box (control flow of the N>1 path only, never a measurement): PSEG_BENCH_REHEARSAL=1 puts
    # every rank on cuda:0 and exchanges gradients over gloo
    rehearsal = os.environ.get('PSEG_BENCH_REHEARSAL', '0') == '1'
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='gloo' if rehearsal else 'nccl', init_method='env://', world_size=world, rank=rank)
    if args.gpus != world and rank == 0:
        print('warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE' % (args.gpus, world), file=sys.stderr)

    from pytorch_segmentation_amd import ops
    from pytorch_segmentation_amd.models import DeepLabV3Plus
    from pytorch_segmentation_amd.utils import Trainer, compute_loss

    torch.manual_seed(0)
    model = DeepLabV3Plus(args.classes)  # random init (no network for checkpoints)
    trainer = Trainer(model, fetcher=None, loss_fn=compute_loss, accumulate=1, adam=False, lr=1e-3, device=device)
    trainer.env.policy = args.precision      # the arithmetic policy lives in the trainer's execution context
    model.train()
    x, t = synthetic_batch(args.batch, args.size, args.classes, device, 1234 + rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.train_batch(x, t)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.train_batch(x, t)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    loss_val = loss.item()
    headline_mode = trainer.step_mode()

    def timed_policy(name):
        """K timed steps under another conv policy, same process, same model / batch (N=1 only)."""
        trainer.env.policy = name
        for _ in range(max(5, args.warmup)):      # (a policy switch re-plans and re-allocates: two more steps than the headline's)
            trainer.train_batch(x, t)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            ls = trainer.train_batch(x, t)
        torch.cuda.synchronize()
        d = time.perf_counter() - t1
        trainer.env.policy = args.precision
        mode = trainer.step_mode()          # (read before the policy goes back: the verdict is per policy)
        return {'value': args.batch * args.steps / d, 'unit': 'images/sec', 'ms_per_step': d / args.steps * 1e3,
                'steps': args.steps, 'dtype': DTYPES[name], 'loss': ls.item(), 'step_mode': mode}

    # ---- N > 1: make the run explain itself.  Nobody can rehearse the 8-GPU case (the builder's boxes have one GPU and RCCL
    # refuses two ranks on one device), so the line records who took part and what the exchange cost:
    #   ranks_seen / distinct_devices / devices -- every rank's device name, PCI address and uuid, all-gathered
    #   exchange        -- bucket count / sizes, all-reduce vs reduce-scatter + all-gather (PSEG_EXCHANGE), native RCCL binding
    #                      or torch.distributed, RCCL version
    #   step_nocomm_ms / exposed_comm_ms -- the same K steps timed again on every rank with the collectives SKIPPED (events,
    #                      side stream and joins still run; gradients stay local): step - step_nocomm is the part of the
    #                      exchange that backward did not hide
    multi = None
    if world > 1:
        props = torch.cuda.get_device_properties(device)
        mine = {'rank': rank, 'local_rank': local_rank, 'name': props.name,
                'pci': '%04x:%02x:%02x' % (getattr(props, 'pci_domain_id', 0), getattr(props, 'pci_bus_id', 0),
                                           getattr(props, 'pci_device_id', 0)),
                'uuid': str(getattr(props, 'uuid', '')), 'host': os.uname().nodename}
        seen = [None] * world
        dist.all_gather_object(seen, mine)
        red = trainer.reducer
        red.skip_collectives = True
        for _ in range(2):
            trainer.train_batch(x, t)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            trainer.train_batch(x, t)
        barrier()
        dt_nc = time.perf_counter() - t1
        red.skip_collectives = False
        tt = torch.tensor([dt_nc], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_nc = tt.item()
        # the replicas diverged while nothing was exchanged: bring them back to rank 0's model (the line is printed after)
        trainer.sync_initial_state()
        try:
            rccl = '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:      # noqa: BLE001 -- a record, not a requirement
            rccl = 'unavailable: %s' % e
        multi = {'ranks_seen': len([s for s in seen if s is not None]),
                 'distinct_devices': len({(s['host'], s['pci'], s['uuid']) for s in seen if s is not None}),
                 'devices': seen, 'backend': dist.get_backend(), 'rccl_version': rccl,
                 'exchange': red.describe(),
                 'step_nocomm_ms': dt_nc / args.steps * 1e3,
                 'exposed_comm_ms': (dt - dt_nc) / args.steps * 1e3,
                 'note': 'no scaling curve had been measured by the builder when this was written (one-GPU boxes): this run '
                         'IS the first N>1 execution over RCCL'}

    others = None
    if world == 1 and rank == 0 and args.also:
        others = {n: timed_policy(n) for n in args.also.split(',') if n and n != args.precision}

    other_configs = None
    if world == 1 and rank == 0 and args.also and args.configs:
        # each in a process of its own: the Trainer of a replayed configuration lays out its streams (and with them the
        # hardware queues its lanes land on) as `python train.py` would, not behind the streams this process already has
        import subprocess
        other_configs = {}
        for n in [c for c in args.configs.split(',') if c]:
            for pol in ('fp32', 'half'):
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--only-config', n + ':' + pol, '--steps',
                                        str(args.steps)], capture_output=True, text=True, timeout=300)
                    other_configs.setdefault(n, {}).update(json.loads(r.stdout.strip().splitlines()[-1]))
                except (IndexError, ValueError):
                    other_configs.setdefault(n, {})[pol] = {'error': (r.stderr or r.stdout)[-400:]}
                except (OSError, subprocess.SubprocessError) as e:      # (a box that refuses child processes: the headline stands)
                    other_configs.setdefault(n, {})[pol] = {'error': repr(e)[-400:]}

    def measure_roofline(policy):
        """One extra, instrumented training step under `policy` (kernel durations are taken one launch at a time: the weight
        gradients, which the timed steps run on a second stream beside the BatchNorm / data-gradient chain, stay on the
        launch stream for it).  -> (conv roofline object, BatchNorm / HBM roofline object)"""
        trainer.env.policy = policy
        overlap, ops.OVERLAP_WGRAD = ops.OVERLAP_WGRAD, False
        graph_mode, trainer.graph = trainer.graph, False     # (the metered step is eager: a replayed step calls no Python op)
        trainer.train_batch(x, t)                      # (re-plan / re-allocate for this stream layout)
        with ConvMeter(ops, model) as meter:
            trainer.train_batch(x, t)
            meter.summary()
        ops.OVERLAP_WGRAD = overlap
        trainer.graph = graph_mode
        half = trainer.env.half
        fwd_prec, bwd_prec = trainer.env.fwd_prec, trainer.env.bwd_prec
        trainer.env.policy = args.precision
        kinds = {}
        for name, e0, e1, dn, us in meter.records:
            k = kinds.setdefault(name, {'ms': 0.0, 'dense': 0.0, 'useful': 0.0, 'launches': 0})
            k['ms'] += e0.elapsed_time(e1)
            k['dense'] += dn
            k['useful'] += us
            k['launches'] += 1
        bn = {n: kinds.pop(n) for n in ('bn_act_fwd', 'bn_act_bwd') if n in kinds}
        peaks = {0: FP32_MFMA_PEAK_TF, 1: FP16_MFMA_PEAK_TF / 3.0, 2: FP16_MFMA_PEAK_TF / 6.0, 3: FP16_MFMA_PEAK_TF / 3.0}
        pname = {0: 'exact fp32 MFMA (v_mfma_f32_32x32x2_f32), peak 157.3 TF',
                 1: 'split-bf16 3-product MFMA, peak 2500/3 TF fp32-equivalent',
                 2: 'split-bf16 6-product MFMA, peak 2500/6 TF fp32-equivalent',
                 3: 'split-fp16 3-product MFMA on amax-scaled operands, peak 2500/3 TF fp32-equivalent'}

        def entry(name, k):
            prec = fwd_prec if name == 'conv2d_fwd' else bwd_prec
            peak = FP16_MFMA_PEAK_TF if half else peaks[prec]
            arith = 'fp16 operands, one v_mfma_f32_32x32x16_f16 pass, fp32 accumulate; dense peak 2500 TF' if half else pname[prec]
            ach = k['useful'] / (k['ms'] * 1e-3) / 1e12
            tr_bytes, tr_src = pmc_traffic(policy, name)
            stale = 'unknown' if (tr_src is None or _TRAFFIC_SHA is None) else (_TRAFFIC_SHA != kernels_sha16())
            return {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
                    'traffic': tr_bytes, 'traffic_unit': 'bytes per launch (L2<->fabric, FETCH_SIZE x2 + WRITE_SIZE)',
                    'traffic_source': tr_src, 'traffic_stale': stale, 'achieved_executed': k['dense'] / (k['ms'] * 1e-3) / 1e12, 'ms_per_step': k['ms'],
                    'launches_per_step': k['launches'], 'avg_launch_us': 1e3 * k['ms'] / k['launches'],
                    'algorithmic_gflop_per_step': k['useful'] / 1e9, 'arithmetic': arith}

        dom = max(kinds, key=lambda n: kinds[n]['ms'])
        if half:
            kernel_names = {'conv2d_fwd': 'pseg::gather_h_kernel (implicit-GEMM conv forward, fp16)',
                            'conv2d_dgrad': 'pseg::gather_h_kernel (data gradient, fp16)',
                            'conv2d_wgrad': 'pseg::wgrad_h_kernel + slab_reduce (weight gradient, fp16 operands, fp32 result)'}
        else:
            kernel_names = {'conv2d_fwd': 'pseg::gather_f32_dma_kernel / gather_f32_pw_kernel / gather_f32_halo_kernel / gather_conv_kernel (implicit-GEMM conv forward)',
                            'conv2d_dgrad': 'pseg::gather_f32_dma_kernel / gather_f32_pw_kernel / gather_f32_halo_kernel / gather_limb_dma_kernel / gather_conv_kernel (data gradient)',
                            'conv2d_wgrad': 'pseg::wgrad_f32_dma_kernel / wgrad_limb_kernel / wgrad_kernel + slab_reduce (weight gradient)'}
        roof = entry(dom, kinds[dom])
        roof['kernel'] = kernel_names[dom] + ' -- the kernel class with the most device time per step'
        roof['other_conv_kernels'] = {kernel_names[n]: entry(n, k) for n, k in kinds.items() if n != dom}
        tot_ms = sum(k['ms'] for k in kinds.values())
        roof['all_conv_kernels'] = {'ms_per_step': tot_ms,
                                    'algorithmic_tflops': sum(k['useful'] for k in kinds.values()) / (tot_ms * 1e-3) / 1e12,
                                    'executed_tflops': sum(k['dense'] for k in kinds.values()) / (tot_ms * 1e-3) / 1e12}
        hbm = None
        if bn:
            ms = sum(k['ms'] for k in bn.values())
            byt = sum(k['useful'] for k in bn.values())
            ach = byt / (ms * 1e-3) / 1e9
            hbm = {'bound': 'hbm', 'kernel': 'pseg::bn_act_fwd_kernel / bn_bwd_reduce_kernel / bn_act_bwd_apply_kernel (+ finalize '
                                             'launches inside the timed span): the BatchNorm passes of one step',
                   'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                   'algorithmic_gb_per_step': byt / 1e9, 'ms_per_step': ms,
                   'launches_per_step': sum(k['launches'] for k in bn.values()), 'traffic': None,
                   'note': 'algorithmic bytes = every activation-sized operand of a pass once (fp16: 2 bytes per element); '
                           'HIP events around each call, so the small finalize launches and inter-launch gaps are inside'}
        return roof, hbm

    roof = roof_hbm = None
    if not args.no_roofline and rank != 0:
        # the metered extra steps are full training steps with their gradient all-reduce: every rank takes part
        overlap, ops.OVERLAP_WGRAD = ops.OVERLAP_WGRAD, False
        graph_mode, trainer.graph = trainer.graph, False
        trainer.train_batch(x, t)
        trainer.train_batch(x, t)
        ops.OVERLAP_WGRAD = overlap
        trainer.graph = graph_mode
    if not args.no_roofline and rank == 0:
        roof, roof_hbm = measure_roofline(args.precision)
        if roof_hbm is not None:
            roof['batchnorm_passes'] = roof_hbm
        # the WHOLE step against the same peak: the algorithmic conv FLOPs of one step (in-bounds taps,
        # logical channels; all three classes) over the headline's own wall time per step -- BatchNorm, loss, resize,
        # optimiser and every gap included in the denominator, nothing but conv MACs in the numerator
        gflop = sum(e['algorithmic_gflop_per_step'] for e in [roof] + list(roof['other_conv_kernels'].values()))
        step_ms = dt / args.steps * 1e3
        peak = roof['peak']
        roof['whole_step'] = {'useful_gflop_per_step': gflop, 'ms_per_step': step_ms, 'achieved': gflop / step_ms,
                              'unit': 'TFLOP/s', 'peak': peak, 'frac': gflop / step_ms / peak,
                              'note': 'algorithmic conv GFLOP of one training step (fwd + dgrad + wgrad, in-bounds taps) / the timed '
                                      "headline step's wall time on this rank's GPU; everything that is not a conv is in the time only"}
        if others and 'half' in others:
            r_h, hbm_h = measure_roofline('half')
            others['half']['roofline'] = r_h
            others['half']['roofline_hbm'] = hbm_h
    if world > 1:
        barrier()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    if rank == 0:
        out = {
            'metric': 'images/sec training step, DeepLabV3+ R50 512x512 21cl',
            'value': world * args.batch * args.steps / dt,
            'unit': 'images/sec',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': DTYPES[args.precision],
            'data': 'synthetic (uint8-uniform images normalised as the reference does, uniform labels), random-init weights',
            'config': {'workload': 'DeepLabV3+ ResNet-50 OS16, %d classes, %dx%d, batch %d per GPU (BASELINE.json configs[2]); '
                                   'fwd + cross-entropy + bwd + SGD(momentum) step' % (args.classes, args.size, args.size, args.batch),
                       'global_batch': world * args.batch, 'parallelism': 'dp%d' % world, 'loss': loss_val,
                       'conv_precision_policy': args.precision, 'step_mode': headline_mode},
            'other_policies': others,
            'other_configs': other_configs,
            'roofline': roof,
            'cpu_baseline': cpu,
        }
        if multi is not None:
            out['multi_gpu'] = multi
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

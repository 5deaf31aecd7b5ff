"""Fold the rocprofv3 --pmc passes of tools/pmc_step.sh into per-kernel-class tables.
usage: python tools/pmc_step.py <tag> <steps in trace> <policy>...
Writes gpurun_out/<tag>_pmc_traffic.json (bench.py's roofline.traffic source once copied to profiles/) and .md.

Classes: conv2d_fwd / conv2d_dgrad (both `gather_conv_kernel` / `gather_f32_dma_kernel`: told apart by position -- before / after the step's
cross-entropy kernel), conv2d_wgrad (`wgrad_kernel`, `wgrad_limb_kernel` + their `slab_reduce_kernel`s are listed
separately), batchnorm passes, everything else.
FETCH_SIZE / WRITE_SIZE are in KB of 64-B requests at the L2's fabric side (Infinity-Cache hits included); on gfx950
FETCH_SIZE reports half of the bytes of wide (16 B/lane) streaming reads (MI355X_MICROARCH.md, HBM section) and is
doubled here; WRITE_SIZE is exact for 16-B-per-lane stores.
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): the fraction of SIMD-cycles in which the
matrix pipe executes, per kernel class (sum over its dispatches).
Half-precision policy: the conv kernels stage their tiles with 16-byte-per-lane LDS-DMA (calibrated like the fp32 ones); the
BatchNorm / resize passes move 8 bytes per lane, a width the guide marks as uncalibrated for FETCH_SIZE -- their byte counts
are listed as measured (x2) and are upper bounds of the truth."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def klass(name, phase):
    if ('gather_conv_kernel' in name or 'gather_f32_dma_kernel' in name or 'gather_limb_dma_kernel' in name or
            'gather_h_kernel' in name or 'gather_hp_kernel' in name or 'gather_f32_pw_kernel' in name or
            'gather_f32_halo_kernel' in name):
        return 'conv2d_fwd' if phase == 'fwd' else 'conv2d_dgrad'
    if 'wgrad_limb_kernel' in name or 'wgrad_kernel' in name or 'wgrad_f32_dma_kernel' in name or 'wgrad_h_kernel' in name:
        return 'conv2d_wgrad'
    if 'slab_reduce' in name:
        return 'slab_reduce'
    if 'bn_act_fwd' in name:
        return 'bn_act_fwd'
    if 'bn_bwd_reduce' in name:
        return 'bn_bwd_reduce'
    if 'bn_act_bwd_apply' in name:
        return 'bn_act_bwd_apply'
    if 'bn_finalize' in name or 'bn_bwd_finalize' in name or 'stat_merge' in name:
        return 'bn_finalize'
    return 'other'


def load(d):
    rows = []
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if files:      # one trace per directory: the newest (earlier runs may have left theirs behind)
        rows = list(csv.DictReader(open(max(files, key=os.path.getmtime))))
    # one row per (dispatch, counter)
    disp = {}
    for r in rows:
        k = int(r['Dispatch_Id'])
        e = disp.setdefault(k, {'name': r['Kernel_Name'], 'c': {}})
        e['c'][r['Counter_Name']] = e['c'].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    out, phase = [], 'fwd'
    for k in sorted(disp):
        n = disp[k]['name']
        out.append((klass(n, phase), n, disp[k]['c']))
        if 'ce_fused_kernel' in n or 'ce_generic_kernel' in n or 'ce_up_fused_kernel' in n or 'ce_up_combine_kernel' in n:
            phase = 'bwd'
        elif 'sgd_kernel' in n or 'adam_kernel' in n or 'sgd_mp_kernel' in n or 'adam_mp_kernel' in n:
            phase = 'fwd'
    return out


def kernels_sha16():
    """sha256 over the library's HIP sources (sorted by name): bench.py compares it with the tree it runs from and says
    `traffic_stale` when the kernels changed after the counter passes"""
    import hashlib
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pytorch_segmentation_amd', 'csrc')
    h = hashlib.sha256()
    for f in sorted(os.listdir(here)):
        if f.endswith('.hip') or f.endswith('.h'):
            h.update(f.encode())
            h.update(open(os.path.join(here, f), 'rb').read())
    return h.hexdigest()[:16]


def main():
    tag, steps, pols = sys.argv[1], float(sys.argv[2]), sys.argv[3:]
    res = {'kernels_sha16': kernels_sha16(),
           'source': 'rocprofv3 --pmc {FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES} '
                     '--kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --precision P --also ""'
                     ' (tools/pmc_step.sh); per training step = totals / %g; FETCH_SIZE doubled (gfx950), KB -> bytes' % steps,
           'policies': {}}
    md = ['# %s -- hardware counters per kernel class and training step (1x MI355X, DeepLabV3+ R50, B=16, 512x512)' % tag, '',
          res['source'], '']
    for pol in pols:
        acc = defaultdict(lambda: defaultdict(float))
        for c in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_VALU_MFMA_BUSY_CYCLES'):
            for kl, name, cnt in load('gpurun_out/pmc_%s_%s_%s' % (tag, pol, c)):
                a = acc[kl]
                if c == 'FETCH_SIZE':
                    a['fetch'] += 2.0 * cnt.get('FETCH_SIZE', 0.0) * 1024.0
                    a['launches'] += 1
                elif c == 'WRITE_SIZE':
                    a['write'] += cnt.get('WRITE_SIZE', 0.0) * 1024.0
                else:
                    a['mfma'] += cnt.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
                    a['gui'] += cnt.get('GRBM_GUI_ACTIVE', 0.0)
                    a['sqbusy'] += cnt.get('SQ_BUSY_CYCLES', 0.0)
        pol_out = {}
        md += ['## policy `%s`' % pol, '', '| kernel class | launches/step | fetch GB/step | write GB/step | bytes/launch (MB) | MFMA busy |', '|---|---|---|---|---|---|']
        for kl in ('conv2d_fwd', 'conv2d_dgrad', 'conv2d_wgrad', 'slab_reduce', 'bn_act_fwd', 'bn_bwd_reduce', 'bn_act_bwd_apply', 'bn_finalize', 'other'):
            a = acc.get(kl)
            if not a:
                continue
            busy = a['mfma'] / (1024.0 * a['gui'] / 8.0) if a['gui'] else None
            e = {'launches_per_step': a['launches'] / steps, 'fetch_bytes_per_step': a['fetch'] / steps,
                 'write_bytes_per_step': a['write'] / steps, 'mfma_busy': busy}
            pol_out[kl] = e
            md.append('| %s | %.1f | %.2f | %.2f | %.1f | %s |' % (kl, e['launches_per_step'], e['fetch_bytes_per_step'] / 1e9,
                      e['write_bytes_per_step'] / 1e9, (e['fetch_bytes_per_step'] + e['write_bytes_per_step']) / max(e['launches_per_step'], 1) / 1e6,
                      '%.2f' % busy if busy is not None else '-'))
        tot_f = sum(e['fetch_bytes_per_step'] for e in pol_out.values())
        tot_w = sum(e['write_bytes_per_step'] for e in pol_out.values())
        md += ['', 'all kernels: %.1f GB fetched + %.1f GB written per step' % (tot_f / 1e9, tot_w / 1e9), '']
        res['policies'][pol] = pol_out
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(res, open('gpurun_out/%s_pmc_traffic.json' % tag, 'w'), indent=1)
    open('gpurun_out/%s_pmc_traffic.md' % tag, 'w').write('\n'.join(md) + '\n')
    print('\n'.join(md))


if __name__ == '__main__':
    main()

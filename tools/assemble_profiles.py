"""Copy the end-of-round measurement set from gpurun_out/ (scratch) into profiles/ (tracked) and write the summary.
usage: python tools/assemble_profiles.py r04     (after tools/refresh_a.sh r04 and tools/refresh_b.sh r04 on the GPU box)"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(ROOT, 'gpurun_out'), os.path.join(ROOT, 'profiles')


def main():
    tag = sys.argv[1]
    rnd = tag.lstrip('r').lstrip('0') or '0'
    for n in ['%s_bench_n1.json', '%s_pmc_traffic.json', '%s_pmc_traffic.md', '%s_small_configs.txt'] + \
            ['%%s_kernel_stats_%s.csv' % k for k in ('fp32', 'fp32_1s', 'half', 'half_1s', 'mixed_1s')]:
        shutil.copy(os.path.join(O, n % tag), os.path.join(P, n % tag))
    for n in ('%s_layers_fp32.md',):
        if os.path.exists(os.path.join(O, n % tag)):
            shutil.copy(os.path.join(O, n % tag), os.path.join(P, n % tag))
    open(os.path.join(P, 'CURRENT'), 'w').write('%s_pmc_traffic.json\n' % tag)
    lines = [l for l in open(os.path.join(O, '%s_forced.log' % tag)).read().split('\n') if 'ms/step' in l]
    open(os.path.join(P, '%s_forced_reducer.md' % tag), 'w').write(
        '# Round %s -- the gradient-exchange path on a one-GPU box: `python tools/bench_forced_reducer.py`\n\n'
        'Same protocol as `profiles/r03_forced_reducer.md` (one MI355X, ONE RCCL rank, the bucketed side-stream reducer forced on with\n'
        '`PSEG_FORCE_REDUCER=1` against the same step with the reducer off; every line a fresh process, best of three blocks of ten steps).\n'
        'It prices the reducer\'s launches and synchronisation, NOT communication: no scaling curve has been measured.  "native" = the\n'
        'collective through the library\'s own RCCL binding (`PSEG_NATIVE_ALLREDUCE=1`; a 1-rank RCCL all-reduce is a copy kernel).\n\n```\n'
        % rnd + '\n'.join(lines) + '\n```\n')
    d = json.load(open(os.path.join(P, '%s_bench_n1.json' % tag)))
    h = d['other_policies']['half']
    tab = lambda name: open(os.path.join(O, '%s_table_%s.md' % (tag, name))).read()
    oc, oh = list(d['roofline']['other_conv_kernels'].values()), list(h['roofline']['other_conv_kernels'].values())
    md = ['# Round %s (end of round, %s) -- `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline '
          '--no-roofline --precision P --also ""`\n' % (rnd, tag),
          '1x MI355X, DeepLabV3+ R50, 21 classes, 512x512, batch 16 (`tools/refresh_a.sh %s`, `tools/refresh_b.sh %s`, `tools/assemble_profiles.py %s`).\n'
          'Bench line: `profiles/%s_bench_n1.json` (fp32 headline %.1f img/s = %.2f ms/step; `half` = train.py -mp: %.1f img/s = %.2f ms/step; mixed %.1f;\n'
          'limb %.1f img/s; boxes of this pool differ by ~1-2 %% in clock).\n'
          'Counters: `profiles/%s_pmc_traffic.md`.  Launch-bound configurations (eager / replayed / AUTO): `profiles/%s_small_configs.txt`.\n'
          'Gradient-exchange path on one rank: `profiles/%s_forced_reducer.md`.  What the round measured on the way (tile variants, per-wave counters,\n'
          'ablation, persistent kernel A/B, GEMM ceiling): `profiles/EXPERIMENTS.md` (this round: section 6).  Per-layer conv table:\n'
          '`profiles/%s_layers_fp32.md`; two-stream timelines: `profiles/%s_timeline_fp32.txt`, `_half.txt`.\n'
          % (tag, tag, tag, tag, d['value'], d['ms_per_step'], h['value'], h['ms_per_step'], d['other_policies']['mixed']['value'],
             d['other_policies']['limb']['value'], tag, tag, tag, tag, tag),
          'Roofline objects of the bench line (one-stream metered step, HIP events per call): fp32 weight gradient %.3f of 157.3 TF (%.2f ms), forward %.3f,\n'
          'data gradient %.3f; `half`: weight gradient + slab reduce %.3f of 2.5 PF (%.2f ms: ONE block per CU -- slower alone, faster beside the data gradients\n'
          'of the two-stream step), forward %.3f (%.2f ms), data gradient %.3f (%.2f ms); BatchNorm passes %.2f of 8 TB/s.  Algorithmic work per class and step:\n'
          '%.1f GFLOP (logical channel counts).\n'
          % (d['roofline']['frac'], d['roofline']['ms_per_step'], oc[0]['frac'], oc[1]['frac'], h['roofline']['frac'], h['roofline']['ms_per_step'],
             oh[0]['frac'], oh[0]['ms_per_step'], oh[1]['frac'], oh[1]['ms_per_step'], h['roofline_hbm']['frac'],
             d['roofline']['algorithmic_gflop_per_step']),
          '## A. half policy (-mp), one stream (`PSEG_OVERLAP_WGRAD=0`): undisturbed per-kernel durations\n\n' + tab('half_1s'),
          '## B. half policy, default run (weight gradients on the auxiliary stream)\n\n' + tab('half'),
          '## C. fp32 policy (headline), one stream\n\n' + tab('fp32_1s'),
          '## D. fp32 policy, default run\n\n' + tab('fp32'),
          '## E. mixed policy, one stream\n\n' + tab('mixed_1s')]
    open(os.path.join(P, '%s_summary.md' % tag), 'w').write('\n'.join(md))
    print('profiles/%s_* written; CURRENT -> %s_pmc_traffic.json' % (tag, tag))


if __name__ == '__main__':
    main()

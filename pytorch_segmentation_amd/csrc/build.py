"""Build libpseg_amd.so (gfx950 only) with hipcc, in-tree.

    python -m pytorch_segmentation_amd.csrc.build [--force]

The shared library lands next to the package (pytorch_segmentation_amd/libpseg_amd.so) so it travels to the
GPU box with the source snapshot.  hipcc cross-compiles without a GPU present.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB = os.path.join(PKG, 'libpseg_amd.so')
SOURCES = ['conv_mfma.hip', 'conv_half.hip', 'norm_act.hip', 'pool_resize.hip', 'loss.hip', 'optim.hip', 'dwconv.hip', 'lanes.hip', 'comm.hip', 'diag.hip']
HEADERS = ['common.h', 'conv_common.h', 'half_io.h', os.path.join('..', '..', 'include', 'pseg_amd.h')]
ARCH = 'gfx950'


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (set HIPCC or install ROCm)')


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=True, variant=None):
    """variant='noprio': the same library without wave priorities (-DPSEG_NO_PRIO=1) as libpseg_amd_noprio.so, for A/B
    measurements on one box (PSEG_LIB_PATH selects the file to load); always rebuilt."""
    if variant is not None:
        force = True
    if variant is None and not force and up_to_date():
        return LIB
    hipcc = _hipcc()
    objs = []
    objdir = os.path.join(HERE, 'build' if variant is None else 'build_' + variant)
    lib = LIB if variant is None else LIB.replace('.so', '_%s.so' % variant)
    os.makedirs(objdir, exist_ok=True)
    procs = []
    flavour = os.environ.get('PSEG_BUILD_TRACE', '0')
    stamp = os.path.join(objdir, '.flavour')
    same_flavour = os.path.exists(stamp) and open(stamp).read() == flavour
    hdr_time = max(os.path.getmtime(os.path.join(HERE, h)) for h in HEADERS + [os.path.basename(__file__)])
    for s in SOURCES:
        o = os.path.join(objdir, s.replace('.hip', '.o'))
        objs.append(o)
        # per-object incremental: an object newer than its source and every header is kept
        if not force and same_flavour and os.path.exists(o) and \
                os.path.getmtime(o) >= max(hdr_time, os.path.getmtime(os.path.join(HERE, s))):
            continue
        cmd = [hipcc, '--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-c', os.path.join(HERE, s), '-o', o]
        if os.environ.get('PSEG_BUILD_TRACE', '0') == '1':     # debug build: tools/conv_phases.py
            cmd.insert(-4, '-DPSEG_CONV_TRACE=1')
        if variant == 'noprio':
            cmd.insert(-4, '-DPSEG_NO_PRIO=1')
        if variant == 'trace' and '-DPSEG_CONV_TRACE=1' not in cmd:   # per-block phase timestamps (tools/conv_phases.py)
            cmd.insert(-4, '-DPSEG_CONV_TRACE=1')
        if variant == 'lab':          # measured-and-rejected kernel variants, ablation switches (conv_half.hip: PSEG_LAB)
            cmd.insert(-4, '-DPSEG_LAB=1')
        if variant == 'trbuiltin':    # A/B: ds_read_b64_tr_b16 through the builtin (hipcc then drains the LDS-DMA ring before it)
            cmd.insert(-4, '-DPSEG_TR_BUILTIN=1')
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (s, out.decode(errors='replace')))
    open(stamp, 'w').write(flavour)
    cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', lib] + objs + ['-ldl']
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == '__main__':
    print(build(force='--force' in sys.argv,
                variant=next((v for v in ('noprio', 'trbuiltin', 'lab', 'trace') if '--' + v in sys.argv), None)))

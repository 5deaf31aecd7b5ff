"""Register / LDS / occupancy table of the kernels of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py conv_half.hip [name filter]   (or a saved remark file ending in .txt)"""
import os
import re
import subprocess
import sys

HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pytorch_segmentation_amd', 'csrc')


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    if src.endswith('.txt'):
        text = open(src).read()
    else:
        r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c',
                            os.path.join(HERE, src), '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'],
                           capture_output=True, text=True)
        text = r.stderr
    blocks = re.split(r'remark: Function Name: ', text)[1:]
    names = [b.split()[0] for b in blocks]
    dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.strip().split('\n')
    for b, d in zip(blocks, dem):
        d = re.sub(r'\(.*$', '', d).replace('void pseg::', '')
        if flt and flt not in d:
            continue
        g = lambda k: re.search(k + r': (\d+)', b).group(1)
        print('%-60s VGPR %3s AGPR %3s SGPR %3s spillV %s spillS %s scratch %s occ %s LDS %s' % (
            d, g('VGPRs'), g('AGPRs'), g('TotalSGPRs'), g('VGPRs Spill'), g('SGPRs Spill'), g(r'ScratchSize \[bytes/lane\]'),
            g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))


if __name__ == '__main__':
    main()

"""CPU-only evidence for tests/test_half_models_gpu.py::test_eval_forward_half_vs_fp32_oracle: how well conditioned is the
EVAL-mode (frozen-statistics) forward of the random-init DeepLabV3+ the parity cases use?

For each residual-branch gain g (the last BatchNorm weight of every ResNet-50 bottleneck scaled by g) it prints
  * the fp32 oracle's own relative-L2 response to a 1e-3 relative scaling of the input image, and
  * the distance between the fp32 oracle and the same oracle with fp16 rounding emulated on every conv / ReLU output and
    on the filters (what the half policy stores in fp16).
Round-4 numbers (this container, 8 cores): g = 1 (the fill as it is) 0.28 / 0.16 at 128x128 B=4 and 1.0 / 0.42 at 256x256
B=2 -- an ill-conditioned map, in eval mode too; g = 0.25: 1.7e-2 / 9e-3.
usage: python tools/half_eval_conditioning.py [size] [batch]"""
import copy
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fill, margins  # noqa: E402
from oracle import models as omodels  # noqa: E402


def run(S, B, gain):
    ref = omodels.DeepLabV3Plus(21)
    key = 'heval_deeplabv3plus_%d' % S
    fill.fill_module_(ref, key)
    with torch.no_grad():
        for name, mod in ref.named_modules():
            if name.endswith('bn3'):
                mod.weight.mul_(gain)
    x = fill.images(key + '/x', (B, 3, S, S))
    margins.freeze_stats(ref, x)
    with torch.no_grad():
        out_ref = ref(x)
        moved = ref(x * (1 + 1e-3))
    em = copy.deepcopy(ref)
    for mod in em.modules():
        if isinstance(mod, nn.Conv2d):
            with torch.no_grad():
                mod.weight.copy_(mod.weight.half().float())
        if isinstance(mod, (nn.Conv2d, nn.ReLU, nn.ReLU6)):
            mod.register_forward_hook(lambda m_, i_, out: out.half().float())
    with torch.no_grad():
        out = em(x.half().float())

    def l2(a, b):
        return ((a - b).double().norm() / b.double().norm()).item()

    print('%dx%d B=%d residual gain %.2f: fp32 oracle under a 1e-3 input scaling l2 %.3e | fp16-emulating oracle vs fp32 oracle '
          'l2 %.3e max-norm %.3e' % (S, S, B, gain, l2(moved, out_ref), l2(out, out_ref),
                                     ((out - out_ref).abs().max() / out_ref.abs().max()).item()))


if __name__ == '__main__':
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    for g in (1.0, 0.5, 0.25, 0.1):
        run(S, B, g)

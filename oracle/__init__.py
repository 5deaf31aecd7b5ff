"""CPU oracle for the segmentation hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This package restates, with stock torch CPU fp32 ops, the arithmetic the
reference (WoodsGao/pytorch_segmentation) runs on its hot path.  The reference
has no kernels of its own: every op is a stock ``torch`` op composed by
``models/aspp.py``, ``models/deeplabv3plus.py``, ``models/unet.py`` and
``utils/utils.py``; the block definitions (``ConvNormAct``, backbones,
``initialize_weights``) live in the un-vendored third-party package
``pytorch-modules>=0.3.0`` (reference ``requirements.txt:5``) and are restated
here from their call sites.

Pinning: ``oracle/gen_golden.py`` imports the reference's own ``models/*.py``
(under an in-memory ``pytorch_modules`` stand-in that resolves to the
restatements of this package) and commits inputs + outputs as fixtures under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this package against
them.  The reference ships no golden vectors of its own.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package.  The product package
(``pytorch_segmentation_amd``) never does: it fails loudly when its HIP
library is missing.
"""

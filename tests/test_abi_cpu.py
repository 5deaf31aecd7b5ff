"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads without a GPU, and exports exactly
the prototypes include/pseg_amd.h declares.  No compute calls here (no GPU in the build container)."""
import ctypes
import os
import subprocess

import pytest

from pytorch_segmentation_amd import _lib
from pytorch_segmentation_amd.csrc import build as csrc_build


@pytest.fixture(scope='module')
def lib():
    csrc_build.build(verbose=False)
    return _lib.load()


def test_header_prototypes_all_exported(lib):
    protos = _lib.parse_header()
    assert len(protos) >= 41
    for name in protos:
        assert hasattr(lib, name), name
    out = subprocess.check_output(['nm', '-D', '--defined-only', _lib.LIB_PATH]).decode()
    exported = {l.split()[-1] for l in out.splitlines() if ' T pseg_' in l}
    assert exported == set(protos), (exported ^ set(protos))


def test_abi_version_and_error_string(lib):
    assert lib.pseg_abi_version() == 1
    # argument validation happens on the host before any launch: usable without a GPU
    rc = lib.pseg_conv2d_fwd(None, 4, None, None, None, 4, 1, 4, 4, 4, 4, 4, 4, 1, 1, 1, 0, 1, 0, 0, None, None, None, None, 0, None)
    assert rc == -1 and b'null' in lib.pseg_last_error()
    with pytest.raises(_lib.PsegError):
        _lib.call('pseg_fill', None, 0, 0.0, None)


def test_plan_queries_are_consistent(lib):
    # ASPP dilated conv at C3: 16384 pixels -> 128 row tiles; its wgrad is split over pixels
    # 16384 pixels, 128x64 tiles (2 wave rows of 64 pixels each) -> 256 statistics groups of 64 rows
    assert _lib.query('pseg_conv2d_stat_rows', 16, 32, 32, 2048, 256, 1, 1, 1, 0, 1) == 256
    assert _lib.query('pseg_conv2d_stat_group', 16, 32, 32, 2048, 256, 1, 1, 1, 0, 1) == 64
    assert _lib.query('pseg_conv2d_fwd_workspace_bytes', 16, 32, 32, 2048, 256, 3, 3) == 0
    wb = _lib.query('pseg_conv2d_wgrad_workspace_bytes', 16, 32, 32, 2048, 256, 3, 3)
    assert wb > 0 and wb % (256 * 9 * 2048 * 4) == 0
    # UNet decoder first conv at C2: only 512 pixels -> split-K forward
    assert _lib.query('pseg_conv2d_fwd_workspace_bytes', 8, 8, 8, 1280, 256, 3, 3) > 0
    assert _lib.query('pseg_col_stats_rows', 1000, 64) * _lib.query('pseg_col_stats_group', 1000, 64) >= 1000
    assert _lib.query('pseg_bn_finalize_workspace_bytes', 16384, 64) > 0 and _lib.query('pseg_bn_finalize_workspace_bytes', 128, 64) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.PsegError, match='no CPU or eager fallback'):
        _lib.load()


def test_reference_import_lines_resolve():
    """The import lines of the reference's scripts and model files (train.py:13-16, test.py:8-9, models/*.py:5-12,
    utils/utils.py:7) resolve, from the repo root, to the MI355X-native implementations."""
    import pytorch_segmentation_amd as pkg
    from models import DeepLabV3Plus, HRNet, UNet
    from pytorch_modules.backbones import mobilenet_v2, resnet50
    from pytorch_modules.backbones.mobilenet import InvertedResidual
    from pytorch_modules.nn import ConvNormAct, FocalBCELoss, SeparableConvNormAct
    from pytorch_modules.utils import IMG_EXT, Fetcher, Trainer, device, initialize_weights
    from utils.utils import compute_loss
    assert DeepLabV3Plus is pkg.models.DeepLabV3Plus and UNet is pkg.models.UNet and HRNet is pkg.models.HRNet
    assert ConvNormAct is pkg.nn.ConvNormAct and Trainer is pkg.utils.Trainer and Fetcher is pkg.utils.Fetcher
    assert compute_loss is pkg.utils.compute_loss and initialize_weights is pkg.nn.initialize_weights
    assert callable(resnet50) and callable(mobilenet_v2) and InvertedResidual is not None
    assert '.jpg' in IMG_EXT and device is not None
    FocalBCELoss()                      # instantiated at import time by the reference (utils/utils.py:14), never called
    import pytest
    with pytest.raises(NotImplementedError):
        SeparableConvNormAct(8, 8)

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
    assert lib.pseg_abi_version() == _lib.abi_version_of_header() >= 2
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


def _null_args(argtypes):
    return [None if t is ctypes.c_void_p else (0.0 if t in (ctypes.c_float, ctypes.c_double) else 0) for t in argtypes]


def test_every_entry_point_rejects_null_arguments(lib):
    """Error behaviour of the boundary: every entry point that takes pointers validates them on the host, BEFORE any
    launch -- all-null / all-zero arguments give PSEG_ERR_ARG and a message naming the function, never a crash, and
    (this container has no GPU) no device is needed to be told so."""
    checked = 0
    for name, (restype, argtypes, _) in sorted(_lib.prototypes().items()):
        if restype is not ctypes.c_int or ctypes.c_void_p not in argtypes:
            continue                                   # integer-only size / plan queries
        if name == 'pseg_debug_conv_trace':            # null = "tracing off", the one call where null is a value
            assert getattr(lib, name)(None) == 0
            continue
        rc = getattr(lib, name)(*_null_args(argtypes))
        msg = lib.pseg_last_error()
        assert rc == -1 and msg, name
        stem = name[len('pseg_'):].split('_')[0].encode()
        assert stem[:4] in msg or b'ce' in msg, (name, msg)
        checked += 1
    assert checked >= 40


@pytest.mark.skipif(__import__('torch').cuda.is_available(), reason='uses made-up device addresses: host-side checks only')
def test_argument_validation_messages(lib):
    """The specific contracts the header states (alignment, channel padding, mask / limb-plane preconditions, label and
    size ranges) are enforced with a message that says which one failed.  The pointers are made-up addresses: every case
    must be refused by the host-side checks, so nothing is ever dereferenced."""
    A, MIS = 0x7f0000000000, 0x7f0000000004            # 16-byte aligned / misaligned fake device addresses

    def refused(name, *args, match):
        with pytest.raises(_lib.PsegError, match=match):
            _lib.call(name, *args)

    # conv forward: x misaligned; Cin not a multiple of 4; ldx not a multiple of 4
    conv = lambda x, ldx, cin: ('pseg_conv2d_fwd', x, ldx, A, None, A, 64, 1, 8, 8, cin, 8, 8, 64, 1, 1, 1, 0, 1, 0, 0,
                                None, None, None, None, 0, None)
    refused(*conv(MIS, 64, 64), match='16-byte aligned')
    refused(*conv(A, 64, 62), match='multiples of 4')
    refused(*conv(A, 62, 64), match='multiples of 4')
    # fp16-limb forward needs the operands' max|x|
    refused('pseg_conv2d_fwd', A, 64, A, None, A, 64, 1, 8, 8, 64, 8, 8, 64, 1, 1, 1, 0, 1, 0, 3, None, None, None, None, 0, None,
            match='amax')
    # BatchNorm: activation bitmask needs C % 32 == 0; limb planes need a dense dy with C % 8 == 0; C % 4 == 0 always
    refused('pseg_bn_act_fwd', A, 48, A, A, A, None, 0, 1, A, 48, 64, 48, None, A, None, match='C %% 32|C % 32')
    bwd = lambda C, ldp, hi, lo: ('pseg_bn_act_bwd_apply', A, C, None, 0, A, C, A, A, A, A, A, A, 1, A, C, None, 0, 0, 64, C,
                                  None, hi, lo, ldp, None)
    refused(*bwd(36, 36, A, A), match='limb planes')
    refused(*bwd(64, 72, A, A), match='limb planes')
    refused(*bwd(64, 64, A, None), match='come together')
    refused('pseg_col_stats', A, 6, 64, 6, A, None, match='C %% 4|C % 4')
    # loss: class count the fused kernels are built for; up-sampled form: logits row stride holds the classes
    refused('pseg_ce_upsampled_fwd_bwd', A, 4, 1, 8, 8, 21, A, 32, 32, 1, -100, A, 24, A, A, 1 << 20, None, match='ld')
    refused('pseg_ce_upsampled_fwd_bwd', A, 40, 1, 8, 8, 40, A, 32, 32, 1, -100, A, 40, A, A, 1 << 20, None, match='not covered')
    # optimiser: momentum without a buffer; misaligned arena
    refused('pseg_sgd_step', A, A, None, 1024, 0.1, 0.9, 0.0, 0, 1.0, 1, None, match='momentum needs a buffer')
    refused('pseg_sgd_step', MIS, A, A, 1024, 0.1, 0.9, 0.0, 0, 1.0, 1, None, match='alignment')
    # weight-gradient slabs: a plan that does not split has no slab form
    refused('pseg_conv2d_wgrad_slabs', A, 64, A, 64, A, 1, 4, 4, 64, 4, 4, 64, 1, 1, 1, 0, 1, 0, 1, 1 << 20, None,
            match='does not split')


def test_library_path_override(lib, tmp_path):
    """PSEG_LIB_PATH loads another build of the same library (A/B measurements: e.g. the -DPSEG_NO_PRIO=1 variant)."""
    import shutil
    import sys
    alt = tmp_path / 'libpseg_amd_alt.so'
    shutil.copy(_lib.LIB_PATH, alt)
    code = ('from pytorch_segmentation_amd import _lib; lib = _lib.load(); '
            'assert _lib.LIB_PATH.endswith("libpseg_amd_alt.so") and lib.pseg_abi_version() == _lib.abi_version_of_header(); print("ok")')
    out = subprocess.check_output([sys.executable, '-c', code], env=dict(os.environ, PSEG_LIB_PATH=str(alt)),
                                  cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.decode().strip().endswith('ok')


def test_fault_handler_prints_native_frames():
    """csrc/diag.hip: with PSEG_SEGV_BACKTRACE=1 a host fault writes the NATIVE frames of the faulting thread before Python's
    faulthandler (installed first, as pytest does) prints its own -- round 4's two faults inside hipGraphLaunch left Python
    frames only.  The process still dies of the signal.  Without the variable nothing is installed."""
    import signal
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import faulthandler, ctypes, sys; faulthandler.enable(); '
            'from pytorch_segmentation_amd import _lib; lib = _lib.load(); '
            'print("enabled", lib.pseg_fault_backtrace_enabled(), flush=True); '
            'ctypes.string_at(8)')
    p = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, PSEG_SEGV_BACKTRACE='1'), cwd=repo,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    err = p.stderr.decode(errors='replace')
    assert p.returncode == -signal.SIGSEGV, (p.returncode, err[-2000:])
    assert b'enabled 1' in p.stdout
    assert '[pseg] fatal signal 11 (SIGSEGV), native frames of the faulting thread:' in err
    native = err.split('native frames of the faulting thread:')[1].split('[pseg] end of native frames')[0]
    assert 'libpseg_amd.so' in native and ('libc.so' in native or 'libffi' in native or '_ctypes' in native), native
    assert 'Fatal Python error: Segmentation fault' in err.split('[pseg] end of native frames')[1]    # chained, and after
    p = subprocess.run([sys.executable, '-c', code.replace('ctypes.string_at(8)', 'pass')],
                       env={k: v for k, v in os.environ.items() if k != 'PSEG_SEGV_BACKTRACE'}, cwd=repo,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and b'enabled 0' in p.stdout

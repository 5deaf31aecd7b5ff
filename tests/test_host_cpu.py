"""Host-side logic that needs no GPU: pretrained-encoder handling, loader policy of train.py."""
import os
import warnings

import pytest
import torch


def test_pretrained_encoder_is_loaded_or_loudly_random(tmp_path, monkeypatch):
    """The reference builds its encoders with pretrained=True (models/deeplabv3plus.py:17-19, models/unet.py:16-17).
    Offline: a torchvision-format state-dict named by PSEG_PRETRAINED_* is loaded; otherwise a RuntimeWarning says the
    encoder is random-init (never silently)."""
    from oracle import backbones as ob
    from oracle import fill
    from pytorch_segmentation_amd.models import DeepLabV3Plus, UNet
    monkeypatch.delenv('PSEG_PRETRAINED_RESNET50', raising=False)
    monkeypatch.delenv('PSEG_PRETRAINED_MOBILENET_V2', raising=False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        DeepLabV3Plus(21)
        UNet(2)
    msgs = [str(x.message) for x in w if issubclass(x.category, RuntimeWarning)]
    assert any('resnet50' in m and 'RANDOM-INIT' in m for m in msgs) and any('mobilenet_v2' in m for m in msgs)
    for arch, var, ctor, model in (('resnet50', 'PSEG_PRETRAINED_RESNET50',
                                    lambda: ob.resnet50(replace_stride_with_dilation=[False, False, True]), lambda: DeepLabV3Plus(21)),
                                   ('mobilenet_v2', 'PSEG_PRETRAINED_MOBILENET_V2', ob.mobilenet_v2, lambda: UNet(2))):
        ref = ctor()
        fill.fill_module_(ref, 'pretrained/' + arch)
        sd = dict(ref.state_dict())
        sd['fc.weight'] = torch.zeros(10, 10)        # torchvision checkpoints carry the classifier too
        path = str(tmp_path / (arch + '.pth'))
        torch.save(sd, path)
        monkeypatch.setenv(var, path)
        with warnings.catch_warnings():
            warnings.simplefilter('error')
            m = model()
        got = m.backbone.state_dict()
        for k, v in ref.state_dict().items():
            assert torch.equal(got[k], v), k
        torch.save({k: v for k, v in list(ref.state_dict().items())[:5]}, path)
        with pytest.raises(RuntimeError):
            model()
        monkeypatch.delenv(var)


def test_validation_loader_keeps_the_short_last_batch():
    """train.py: drop_last only for the training loader; evaluation scores every image."""
    import train as train_mod
    ds = torch.utils.data.TensorDataset(torch.arange(7))
    assert len(train_mod._loader(ds, 4, 0, train=True)) == 1
    val = train_mod._loader(ds, 4, 0, train=False)
    assert len(val) == 2 and sorted(int(v) for b in val for v in b[0]) == list(range(7))
    assert len(train_mod._loader(torch.utils.data.TensorDataset(torch.arange(3)), 32, 0, train=False)) == 1


def test_wide_buffer_stores_carry_no_sgpr_offset():
    """Source-level guard for a gfx950 hazard (csrc/half_io.h): a buffer store of more than 64 bits with an SGPR soffset may be
    followed by a VALU write of its data registers without the wait state the hardware needs -- the compiler's hazard model
    exempts exactly that form.  Every 96 / 128-bit raw buffer store of the library therefore passes soffset = 0 (the offset
    lives in the VGPR), which `tools/scan_store_hazard.py` confirms on the generated assembly."""
    import glob
    import os
    import re
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pytorch_segmentation_amd', 'csrc')
    calls = 0
    for path in sorted(glob.glob(os.path.join(here, '*.hip')) + glob.glob(os.path.join(here, '*.h'))):
        src = open(path).read()
        for m in re.finditer(r'__builtin_amdgcn_raw(?:_ptr)?_buffer_store_b(96|128)\s*\(', src):
            # the argument list up to the matching parenthesis
            depth, i = 1, m.end()
            while depth:
                depth += {'(': 1, ')': -1}.get(src[i], 0)
                i += 1
            args = src[m.end():i - 1]
            parts, depth, cur = [], 0, ''
            for ch in args:
                if ch == ',' and depth == 0:
                    parts.append(cur.strip())
                    cur = ''
                else:
                    depth += {'(': 1, ')': -1}.get(ch, 0)
                    cur += ch
            parts.append(cur.strip())
            calls += 1
            assert len(parts) == 5 and parts[3] == '0', '%s: %d-bit buffer store with soffset %r' % (
                os.path.basename(path), int(m.group(1)), parts[3] if len(parts) > 3 else args)
    assert calls >= 2


def test_loaded_state_replaces_the_lazy_batch_counter():
    """BatchNorm2d counts training batches on the host and writes num_batches_tracked back when a state-dict is taken
    (nn.py); a state that is LOADED replaces the count -- batches counted since the last flush belong to the overwritten
    state (round 6: --resume / --weights after training steps reported them on top of the checkpoint's count)."""
    import torch
    from pytorch_segmentation_amd.nn import BatchNorm2d
    m = BatchNorm2d(8)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd['num_batches_tracked'] = torch.tensor(7)
    m.__dict__['_nbt_pending'] = 3
    m.load_state_dict(sd)
    assert m._nbt_pending == 0 and int(m.state_dict()['num_batches_tracked']) == 7
    m.__dict__['_nbt_pending'] = 2
    assert int(m.state_dict()['num_batches_tracked']) == 9


def test_ignored_loader_flags_say_so_once(tmp_path):
    """--rect and `augments` are flags of the reference's loader (train.py:90-101, utils/datasets.py:161-194,26-125) that the minimal
    COCO reader accepts and does not implement (out of scope: SURVEY.md section 2, #9): each says so ONCE per process."""
    import json
    from pytorch_segmentation_amd.utils import datasets as ds
    path = tmp_path / 'train.json'
    path.write_text(json.dumps({'categories': [{'name': 'a'}], 'images': [], 'annotations': []}))
    ds._WARNED.clear()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        ds.CocoDataset(str(path), rect=True, augments=[object()])
        ds.CocoDataset(str(path), rect=True, augments=[object()])
        ds.CocoDataset(str(path))
    msgs = [str(x.message) for x in w if issubclass(x.category, RuntimeWarning)]
    assert len([m for m in msgs if m.startswith('--rect')]) == 1 and len([m for m in msgs if m.startswith('augments')]) == 1, msgs
    assert all('IGNORED' in m for m in msgs)

"""BASELINE.json configs[0] plumbing: UNet, 2 classes, 128x128, batch 2, synthetic COCO-format data through the
train.py / test.py entry points (reference train.py:19-81, test.py:15-73), plus the first batch's loss against the
CPU oracle on the very same tensors."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_and_test_entry_points(tmp_path, monkeypatch):
    from oracle import loss as oloss
    from oracle import models as omodels
    from pytorch_segmentation_amd.utils.datasets import CocoInstance, make_synthetic_coco
    from pytorch_segmentation_amd.models import UNet
    from pytorch_segmentation_amd.utils import compute_loss
    root = make_synthetic_coco(str(tmp_path / 'data'), n_train=6, n_val=4, n_classes=1)
    ds = CocoInstance(os.path.join(root, 'train.json'), img_size=[128, 128])
    assert ds.classes == ['background', 'class0'] and len(ds) == 6
    img, seg = ds[0]
    assert img.dtype == torch.uint8 and tuple(img.shape) == (3, 128, 128) and tuple(seg.shape) == (128, 128)
    assert set(seg.unique().tolist()) <= {0, 1} and seg.max() == 1
    # first batch: HIP vs oracle, same weights
    imgs = torch.stack([ds[0][0], ds[1][0]])
    segs = torch.stack([ds[0][1], ds[1][1]])
    x, t = ds.post_fetch_fn((imgs, segs))
    torch.manual_seed(0)
    ref = omodels.UNet(2).train()
    m = UNet(2)
    m.load_state_dict(ref.state_dict())
    m.cuda().train()
    l_ref = oloss.compute_loss(ref(x), t)
    l_hip = compute_loss(m(x.cuda()), t.cuda(), m)
    assert abs(l_hip.item() - l_ref.item()) < 1e-3 * abs(l_ref.item())
    # the CLI surface: two epochs, gradient accumulation, evaluation, checkpoints
    monkeypatch.chdir(tmp_path)
    import train as train_mod
    trainer, loss1 = train_mod.train(root, epochs=2, img_size=[128, 128], batch_size=2, accumulate=2, lr=1e-2,
                                     num_workers=0, notest=False, nosave=False, model_name='unet')
    assert trainer.epoch == 2 and loss1 == loss1  # finite
    assert os.path.exists(tmp_path / 'weights' / 'last.pt')
    ck = torch.load(tmp_path / 'weights' / 'last.pt', map_location='cpu')
    assert set(ck['model']) == set(ref.state_dict()) and ck['epoch'] == 2
    # checkpoint loads into the oracle's (= the reference's) module and reproduces the eval forward
    ref.load_state_dict(ck['model'])
    ref.eval()
    trainer.model.eval()
    with torch.no_grad():
        a = trainer.model(x.cuda()).cpu()
        b = ref(x)
    assert ((a - b).abs().max() / b.abs().max()).item() < 1e-3


def test_validation_smaller_than_batch_is_still_scored(tmp_path, monkeypatch):
    """len(val) < batch size: the reference evaluates every validation image (train.py:45-53); a drop_last validation
    loader would see zero batches, test() would return 0.0 and `best.pt` would never be written."""
    from pytorch_segmentation_amd.utils.datasets import make_synthetic_coco
    root = make_synthetic_coco(str(tmp_path / 'data'), n_train=4, n_val=3, n_classes=1)
    monkeypatch.chdir(tmp_path)
    import train as train_mod
    trainer, _ = train_mod.train(root, epochs=1, img_size=[64, 64], batch_size=4, accumulate=1, lr=1e-2, num_workers=0,
                                 notest=False, nosave=False, model_name='unet')
    assert trainer.metrics > 0.0
    assert os.path.exists(tmp_path / 'weights' / 'best.pt')


def test_default_trainer_replays_launch_bound_steps(tmp_path, monkeypatch):
    """`python train.py data --model hrnet -mp` with NO environment variable: the Trainer's AUTO mode (graph=None) judges the
    second step of a shape -- host enqueue time against the device span -- and a launch-bound step (HRNet: ~1000 launches of
    5-10 us) is captured and replayed by the lane executor from then on; the reference's loop (train.py:59,71-72) needs no
    switch and neither does this one.  Asserted: the shape was judged launch-bound, a captured step with a lane executor
    exists, the run trained (finite, decreasing loss over two epochs) and the checkpoint is intact."""
    from pytorch_segmentation_amd.utils.datasets import make_synthetic_coco
    monkeypatch.delenv('PSEG_GRAPH', raising=False)
    root = make_synthetic_coco(str(tmp_path / 'data'), n_train=24, n_val=2, n_classes=1)
    monkeypatch.chdir(tmp_path)
    import train as train_mod
    trainer, loss = train_mod.train(root, epochs=2, img_size=[128, 128], batch_size=2, accumulate=1, lr=1e-2, num_workers=0,
                                    mixed_precision=True, notest=True, nosave=False, model_name='hrnet')
    assert trainer.graph == 'auto' and trainer.env.half
    dec = trainer.graph_decisions()
    print('auto graph decisions:', dec)
    assert dec and all(d['use'] for d in dec.values()), dec
    sgs = [sg for sg in trainer._graphs.values() if sg is not None]
    assert sgs and all(getattr(sg, 'lanes', 0) for sg in sgs)
    assert loss == loss and torch.isfinite(trainer.arena.params).all()
    st = trainer.loss_scale_state()
    assert st['steps_applied'] + st['steps_skipped'] == 24 and st['steps_applied'] >= 20
    assert os.path.exists(tmp_path / 'weights' / 'last.pt')

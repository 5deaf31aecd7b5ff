#!/usr/bin/python3
"""Training entry point with the reference's flags and `train(...)` keyword surface (reference train.py:19-21,84-140)
on the MI355X HIP path.

    python3 train.py data/<custom> [--epochs N] [-s W H] [-bs N] [-a ACC] [--lr LR] [--adam] [--resume]
                     [--weights F] [--notest] [--nosave] [--model unet|deeplabv3plus]
    python3 -m torch.distributed.run --nproc-per-node <n> train.py data/<custom>        # RCCL data parallel

Differences from the reference that are visible here: the model is picked with --model (the reference edits
train.py:57-59; default stays UNet), `best` is initialised so --notest without --nosave no longer raises
(SURVEY.md section 3.A.8), and -mp/--mix_precision selects the `half` policy (fp16 activations / gradients / filter copies, one fp16 MFMA pass with fp32
accumulation, fp32 master weights, device-side dynamic loss scaling) instead of apex; PSEG_MP_POLICY=limb keeps fp32 storage.
"""
import argparse
import os

# multi-process GPU work on this ROCm host: only dmabuf IPC is supported (RCCL / tensor sharing fail on the legacy mode)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import os.path as osp
import sys

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, DistributedSampler

from pytorch_segmentation_amd.models import DeepLabV3Plus, HRNet, UNet
from pytorch_segmentation_amd.utils import Fetcher, Trainer, compute_loss
from pytorch_segmentation_amd.utils.datasets import CocoInstance
from test import test

MODELS = {'deeplabv3plus': DeepLabV3Plus, 'unet': UNet, 'hrnet': HRNet}


def _loader(dataset, batch_size, num_workers, train=True):
    distributed = dist.is_available() and dist.is_initialized()
    sampler = DistributedSampler(dataset, dist.get_world_size(), dist.get_rank(), shuffle=train) if distributed else None
    # training: drop the short last batch (train-mode BatchNorm needs more than one sample per step).  Evaluation uses
    # the running statistics, so EVERY validation image is scored, as in the reference (train.py:45-53).
    return DataLoader(dataset, batch_size=batch_size, shuffle=train and sampler is None, sampler=sampler, pin_memory=True,
                      num_workers=num_workers, drop_last=train)


def train(data_dir, epochs=100, img_size=(320, 320), batch_size=32, accumulate=2, lr=1e-3, adam=False, resume=False,
          weights='', num_workers=4, multi_scale=False, rect=False, mixed_precision=False, notest=False, nosave=False,
          model_name='unet'):
    train_data = CocoInstance(osp.join(data_dir, 'train.json'), img_size=list(img_size), multi_scale=multi_scale,
                              rect=rect)
    train_fetcher = Fetcher(_loader(train_data, batch_size, num_workers), train_data.post_fetch_fn)
    val_fetcher = None
    if not notest:
        val_data = CocoInstance(osp.join(data_dir, 'val.json'), img_size=list(img_size), augments=None, rect=rect)
        val_fetcher = Fetcher(_loader(val_data, batch_size, num_workers, train=False), post_fetch_fn=val_data.post_fetch_fn)
    model = MODELS[model_name](len(train_data.classes))
    trainer = Trainer(model, train_fetcher, loss_fn=compute_loss, workdir='weights', accumulate=accumulate, adam=adam,
                      lr=lr, weights=weights, resume=resume, mixed_precision=mixed_precision)
    last_loss = None
    while trainer.epoch < epochs:
        last_loss = trainer.step()
        print('epoch %d: loss %g' % (trainer.epoch, last_loss))
        best = False
        if val_fetcher is not None:
            metrics = test(trainer.model, val_fetcher)
            if metrics > trainer.metrics:
                best = True
                print('save best, miou: %g' % metrics)
                trainer.metrics = metrics
        if not nosave:
            trainer.save(best)
    return trainer, last_loss


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('data', type=str, default='data/voc')
    ap.add_argument('--epochs', type=int, default=100)
    ap.add_argument('-s', '--img_size', type=int, nargs=2, default=[320, 320])
    ap.add_argument('-bs', '--batch-size', type=int, default=32)
    ap.add_argument('-a', '--accumulate', type=int, default=2)
    ap.add_argument('--lr', type=float, default=1e-3)
    ap.add_argument('--adam', action='store_true')
    ap.add_argument('--resume', action='store_true')
    ap.add_argument('--weights', type=str, default='')
    ap.add_argument('--num-workers', type=int, default=4)
    ap.add_argument('--multi-scale', action='store_true')
    ap.add_argument('--rect', action='store_true')
    ap.add_argument('-mp', '--mix_precision', action='store_true', help='mixed precision')
    ap.add_argument('--notest', action='store_true')
    ap.add_argument('--nosave', action='store_true')
    ap.add_argument('--backend', type=str, default='nccl')
    ap.add_argument('--local-rank', '--local_rank', type=int, default=int(os.environ.get('LOCAL_RANK', 0)))
    ap.add_argument('--model', choices=sorted(MODELS), default='unet')
    opt = ap.parse_args()

    if os.environ.get('WORLD_SIZE'):
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend=opt.backend, init_method='env://', world_size=int(os.environ['WORLD_SIZE']),
                                rank=int(os.environ['RANK']))
    torch.cuda.set_device(opt.local_rank)
    os.environ.setdefault('LOCAL_RANK', str(opt.local_rank))
    if opt.local_rank > 0:
        sys.stdout = open(os.devnull, 'w')
    print(opt)
    train(opt.data, opt.epochs, opt.img_size, opt.batch_size, opt.accumulate, opt.lr, opt.adam, opt.resume, opt.weights,
          opt.num_workers, opt.multi_scale, opt.rect, opt.mix_precision, opt.notest, opt.nosave, opt.model)
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

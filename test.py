#!/usr/bin/python3
"""Evaluation entry point with the reference's flags (reference test.py:76-105) on the MI355X HIP path.

    python3 test.py data/<custom>/val.json --weights weights/best.pt [-s W H] [-bs N] [--model deeplabv3plus|unet]

`test(model, fetcher)` keeps the reference's contract: evaluates in eval mode without gradients, prints per-class
(or the five worst classes') targets / precision / recall / IoU / F1 and returns the mean IoU.  The per-class
tp/fn/fp bookkeeping runs as one device kernel per batch instead of 3 host synchronisations per class, and the
distributed reduction is a single [3, num_classes] all-reduce.
"""
import argparse

import torch
from torch.utils.data import DataLoader

from pytorch_segmentation_amd.models import DeepLabV3Plus, HRNet, UNet
from pytorch_segmentation_amd.utils import (Fetcher, all_reduce_counters, broadcast_buffers, compute_loss, compute_metrics,
                                            predict_mask, update_class_counts)
from pytorch_segmentation_amd.utils.datasets import CocoDataset

MODELS = {'deeplabv3plus': DeepLabV3Plus, 'unet': UNet, 'hrnet': HRNet}


@torch.no_grad()
def test(model, fetcher):
    model.eval()
    # distributed evaluation sums per-class counters over the ranks (reference test.py:51-58): they must all come from ONE
    # model, so every replica takes rank 0's BatchNorm running statistics first (DistributedDataParallel's broadcast_buffers)
    broadcast_buffers(model, 0)
    classes = fetcher.loader.dataset.classes
    nc = len(classes)
    counters = None
    loss_sum, batches = None, 0
    for inputs, targets in fetcher:
        outputs = model(inputs)
        loss = compute_loss(outputs, targets, model)
        loss_sum = loss if loss_sum is None else loss_sum + loss
        batches += 1
        if counters is None:
            counters = torch.zeros(3, nc, dtype=torch.int64, device=outputs.device)
        update_class_counts(counters, predict_mask(outputs), targets)
    if counters is None:
        return 0.0
    all_reduce_counters(counters)
    tp, fn, fp = (c.float() for c in counters.cpu())
    T, P, R, miou, F1 = compute_metrics(tp, fn, fp)
    print('loss: %8g, mAP: %8g, F1: %8g, miou: %8g' % (loss_sum.item() / batches, P.mean(), F1.mean(), miou.mean()))
    row = 'cls: %8s, targets: %8d, pre: %8g, rec: %8g, iou: %8g, F1: %8g'
    if nc < 10:
        order = range(nc)
    else:
        print('top error 5')
        order = torch.argsort(miou)[:5].tolist()
    for c in order:
        print(row % (classes[c], T[c], P[c], R[c], miou[c], F1[c]))
    return miou.mean().item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('val', type=str)
    ap.add_argument('--weights', type=str, default='')
    ap.add_argument('--rect', action='store_true')
    ap.add_argument('-s', '--img_size', type=int, nargs=2, default=[320, 320])
    ap.add_argument('-bs', '--batch-size', type=int, default=32)
    ap.add_argument('--num-workers', type=int, default=4)
    ap.add_argument('--model', choices=sorted(MODELS), default='deeplabv3plus')
    opt = ap.parse_args()
    data = CocoDataset(opt.val, img_size=opt.img_size, augments=None, rect=opt.rect)
    loader = DataLoader(data, batch_size=opt.batch_size, pin_memory=True, num_workers=opt.num_workers)
    fetcher = Fetcher(loader, post_fetch_fn=data.post_fetch_fn)
    model = MODELS[opt.model](len(data.classes))
    if opt.weights:
        model.load_state_dict(torch.load(opt.weights, map_location='cpu')['model'])
    print('metrics: %8g' % test(model.cuda(), fetcher))


if __name__ == '__main__':
    main()

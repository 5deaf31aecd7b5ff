"""Minimal COCO-format segmentation dataset (no cv2 / imgaug): the data-format side of the hot path.

Same on-disk format and tensor contract as the reference's ``CocoDataset`` / ``CocoInstance``
(utils/datasets.py:260-391): ``<dir>/{train,val}.json`` with ``categories[{name}]``,
``images[{id,file_name,width,height}]``, ``annotations[{image_id,category_id,segmentation:[[x,y,...]]}]``; label of a
polygon = ``category_id + 1`` (0 is background, utils/datasets.py:301); ``classes = ['background', *names]``;
``__getitem__`` -> (uint8 CHW RGB image resized to ``img_size=[w,h]``, uint8 HW mask); ``post_fetch_fn`` applies the
reference's mean/std normalisation (utils/datasets.py:199-205) and turns masks into int64.

Out of scope: the imgaug augmentation pipeline, the random instance crop of CocoInstance and
the --rect letterboxing; images are decoded with PIL and resized directly.
"""
import json
import os.path as osp
import random
import warnings

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image, ImageDraw

MEAN = (123.675, 116.28, 103.53)
STD = (58.395, 57.12, 57.375)


_WARNED = set()


def _warn_ignored(flag, why):
    """a flag of the reference's CLI that this minimal loader accepts and does NOT implement: say so once per process"""
    if flag not in _WARNED:
        _WARNED.add(flag)
        warnings.warn('%s is accepted for command-line compatibility with the reference and IGNORED: %s' % (flag, why),
                      RuntimeWarning, stacklevel=3)


class CocoDataset(torch.utils.data.Dataset):
    def __init__(self, path, img_size=224, augments=None, multi_scale=False, rect=False):
        if rect:
            _warn_ignored('--rect', 'the letterboxing of reference utils/datasets.py:161-194 is part of the cv2 loader, which is out '
                                    'of scope here (SURVEY.md section 2, #9); images are resized straight to -s W H')
        if augments:
            _warn_ignored('augments', 'the imgaug pipeline of reference utils/datasets.py:26-125 is out of scope here (SURVEY.md '
                                      'section 2, #9); samples are decoded and resized only')
        if isinstance(img_size, int):
            img_size = [img_size, img_size]
        self.img_size = list(img_size)          # [w, h] as the reference's -s flag
        self.multi_scale = multi_scale
        self.rect = rect
        with open(path, 'r') as f:
            self.coco = json.load(f)
        self.img_root = osp.dirname(path)
        self.classes = ['background'] + [c['name'] for c in self.coco['categories']]
        by_id = {}
        for info in self.coco['images']:
            by_id[info['id']] = (osp.join(self.img_root, info['file_name']), [])
        for ann in self.coco['annotations']:
            if ann['image_id'] in by_id:
                by_id[ann['image_id']][1].append(ann)
        self.data = sorted(by_id.values(), key=lambda d: d[0])

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        path, anns = self.data[idx]
        img = Image.open(path).convert('RGB')
        seg = Image.new('L', img.size, 0)
        draw = ImageDraw.Draw(seg)
        for ann in anns:
            for poly in ann['segmentation']:
                pts = [(float(poly[i]), float(poly[i + 1])) for i in range(0, len(poly) - 1, 2)]
                if len(pts) >= 3:
                    draw.polygon(pts, fill=int(ann['category_id']) + 1)
        w, h = self.img_size
        img = img.resize((w, h), Image.BILINEAR)
        seg = seg.resize((w, h), Image.NEAREST)
        img = torch.from_numpy(np.array(img, dtype=np.uint8).transpose(2, 0, 1).copy())
        seg = torch.from_numpy(np.array(seg, dtype=np.uint8))
        return img, seg

    def post_fetch_fn(self, batch):
        imgs, segs = batch
        imgs = imgs.float()
        imgs = (imgs - torch.tensor(MEAN, device=imgs.device).view(1, 3, 1, 1)) / \
            torch.tensor(STD, device=imgs.device).view(1, 3, 1, 1)
        if self.multi_scale:
            h, w = imgs.size(2), imgs.size(3)
            scale = random.uniform(0.7, 1.5)
            h, w = max(32, int(h * scale / 32) * 32), max(32, int(w * scale / 32) * 32)
            imgs = F.interpolate(imgs, (h, w))
        return imgs.contiguous(), segs.long()


class CocoInstance(CocoDataset):
    """Same files, same tensors.  (The reference's variant additionally crops around a random instance with imgaug.)"""

    def __init__(self, path, img_size=224, augments=None, multi_scale=False, rect=False):
        super().__init__(path, img_size, augments, multi_scale, rect)
        self.data = [d for d in self.data if len(d[1]) > 0]


def make_synthetic_coco(root, n_train=8, n_val=4, size=(160, 128), n_classes=1, seed=0):
    """Write a tiny COCO-format dataset (PNG images + train.json / val.json) for plumbing tests."""
    import os
    rng = random.Random(seed)
    os.makedirs(osp.join(root, 'images'), exist_ok=True)
    w, h = size
    cats = [{'id': i, 'name': 'class%d' % i} for i in range(n_classes)]

    def split(name, n, first_id):
        images, anns = [], []
        for j in range(n):
            iid = first_id + j
            arr = np.zeros((h, w, 3), dtype=np.uint8)
            arr[..., 0] = (np.arange(w)[None, :] * 255 // w).astype(np.uint8)
            arr[..., 1] = (np.arange(h)[:, None] * 255 // h).astype(np.uint8)
            arr[..., 2] = rng.randrange(256)
            img = Image.fromarray(arr)
            draw = ImageDraw.Draw(img)
            for _ in range(2):
                c = rng.randrange(n_classes)
                x0, y0 = rng.randrange(5, w // 2), rng.randrange(5, h // 2)
                bw, bh = rng.randrange(50, w // 2 - 5), rng.randrange(50, h // 2 - 5)
                poly = [x0, y0, x0 + bw, y0, x0 + bw, y0 + bh, x0, y0 + bh]
                draw.polygon([(poly[i], poly[i + 1]) for i in range(0, 8, 2)],
                             fill=(255 - 60 * c, 40 + 90 * c, rng.randrange(256)))
                anns.append({'id': len(anns), 'image_id': iid, 'category_id': c, 'segmentation': [poly]})
            fn = osp.join('images', '%s_%03d.png' % (name, j))
            img.save(osp.join(root, fn))
            images.append({'id': iid, 'file_name': fn, 'width': w, 'height': h})
        with open(osp.join(root, name + '.json'), 'w') as f:
            json.dump({'categories': cats, 'images': images, 'annotations': anns}, f)

    split('train', n_train, 0)
    split('val', n_val, 1000)
    return root

"""`from pytorch_modules.utils import Fetcher, Trainer, initialize_weights, IMG_EXT, device` (reference train.py:14,
test.py:9, inference.py:12, models/*.py, utils/datasets.py:15)."""
import torch

from pytorch_segmentation_amd.nn import initialize_weights  # noqa: F401
from pytorch_segmentation_amd.utils import Fetcher, Trainer  # noqa: F401

IMG_EXT = ['.jpg', '.jpeg', '.png', '.tif', '.bmp']
device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')


def fuse(*args, **kwargs):
    raise NotImplementedError('conv+BN fusing for export (reference export2caffe.py:9) is outside the training hot path')

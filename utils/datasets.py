"""`from utils.datasets import CocoDataset, CocoInstance` -- the reference's import path (train.py:15, test.py:10)."""
from pytorch_segmentation_amd.utils.datasets import CocoDataset, CocoInstance  # noqa: F401

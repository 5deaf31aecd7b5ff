"""`from utils.utils import compute_loss, compute_metrics` -- the reference's import path (train.py:16, test.py:11)."""
from pytorch_segmentation_amd.utils.loss import compute_loss, compute_metrics, predict_mask, update_class_counts  # noqa: F401

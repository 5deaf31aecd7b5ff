from .dist import GradReducer, all_reduce_counters, broadcast_buffers
from .loss import compute_loss, compute_metrics, predict_mask, update_class_counts
from .trainer import Fetcher, FlatOptimizer, Trainer

__all__ = ['compute_loss', 'compute_metrics', 'predict_mask', 'update_class_counts', 'GradReducer',
           'all_reduce_counters', 'broadcast_buffers', 'Fetcher', 'FlatOptimizer', 'Trainer']

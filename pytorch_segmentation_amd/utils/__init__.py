from .loss import compute_loss, compute_metrics, predict_mask, update_class_counts

__all__ = ['compute_loss', 'compute_metrics', 'predict_mask', 'update_class_counts']

"""Evaluation Dice on the device (models/segmentation.py:188-194 + lib/evalMetrics.py:17-21,58-68,184-217).

The reference copies 629 MB of logits to the host and makes 31 numpy passes per volume; here one kernel does the
first-max argmax and exact integer overlap counts, so the per-class Dice 2|P&T|/(|P|+|T|) is bit-equal to the
reference's scipy value whenever the argmax is equal (NaN when a class is absent from both, like scipy).
"""
import numpy as np
import torch

from .. import ops


def eval_dice_counts(logits, truth):
    """(counts[N][C][3] int64 on device, pred uint8)."""
    return ops.argmax_dice_counts(logits, truth)


def dice_from_counts(counts):
    """counts[..., 3] = (|P|, |T|, |P&T|) -> float64 numpy Dice, NaN where |P|+|T| == 0."""
    c = counts.detach().cpu().numpy().astype(np.float64)
    with np.errstate(invalid='ignore', divide='ignore'):
        return 2.0 * c[..., 2] / (c[..., 0] + c[..., 1])


def metricEval(eval_metric, logits, truth, num_labels=None):
    """Per-class Dice for classes 1..C-1 of every volume in the batch: ndarray [N][C-1]."""
    if eval_metric != 'dice':
        raise NotImplementedError("only 'dice' is on the accelerated eval path")
    counts, _ = eval_dice_counts(logits, truth)
    return dice_from_counts(counts)[:, 1:]


def get_multiclass_dice(pred, truth, n_class, eps=1e-11):
    """lib/evalMetrics.py:184-217 for index masks: scores B x (n_class-1), from integer counts."""
    onehot_logits = ops.one_hot(pred.reshape(pred.shape[0], 1, *pred.shape[1:]).long(), n_class)
    counts, _ = ops.argmax_dice_counts(onehot_logits, truth)
    c = counts.to(torch.float32)
    return (2. * c[:, 1:, 2]) / ((c[:, 1:, 0] + c[:, 1:, 1]) + eps)

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


def _as_device_labels(a):
    """numpy / torch label map (bool masks included) -> one flattened uint8 / int64 sample on the current device."""
    t = a if torch.is_tensor(a) else torch.as_tensor(np.ascontiguousarray(a))
    if t.dtype == torch.bool:
        t = t.to(torch.uint8)
    elif t.dtype.is_floating_point:
        t = t.to(torch.int64)
    if not t.is_cuda:
        t = t.to(torch.device('cuda', torch.cuda.current_device()))
    return t.reshape(1, -1)


def metricEval(eval_metric, output, gt, num_labels=None):
    """lib/evalMetrics.py:17-100.  Two call forms:
      * the accelerated evaluation loop (models/segmentation.py:188-194 collapsed into one kernel): metricEval('dice', logits N x C x D x H x W,
        truth N x D x H x W) -> per-class Dice of classes 1..C-1 for every volume, ndarray [N][C-1];
      * the reference's own signature on LABEL MAPS (numpy or torch, bool masks included), `output` and `gt` squeezed and flattened as one
        volume: 'iou' = mean over the num_labels labels of |P & T| / |P | T| (0 for a label absent from gt; lib/evalMetrics.py:36-57);
        'dice' (num_labels == 2) = 2 |P & T| / (|P| + |T|) of the non-zero voxels (1 - scipy dice, :59-68; NaN when both are empty);
        'recall' = TP / |T|, 'precision' = TP / |P| of label 1 (:73-100; ZeroDivisionError when the denominator is empty, as there).
      All four come from the three integer counts per label of ONE pass over the two maps (da_label_overlap_counts)."""
    if eval_metric == 'dice' and torch.is_tensor(output) and output.dtype.is_floating_point and output.dim() == 5:
        counts, _ = eval_dice_counts(output, gt)
        return dice_from_counts(counts)[:, 1:]
    if eval_metric not in ('iou', 'dice', 'recall', 'precision'):
        raise ValueError('Invalid evaluation metric value: %r' % (eval_metric,))
    if num_labels is None:
        raise ValueError('num_labels is required for label-map metrics')
    if eval_metric != 'iou' and num_labels != 2:
        raise NotImplementedError('%s evaluation score is only implemented for 2 labels' % eval_metric)
    p, t = _as_device_labels(output), _as_device_labels(gt)
    if p.shape != t.shape:
        raise AssertionError('pred shape %s gt shape %s' % (tuple(p.shape), tuple(t.shape)))
    c = ops.label_overlap_counts(p, t, num_labels)[0].cpu().numpy().astype(np.float64)      # [num_labels][3] = (|P|, |T|, |P & T|)
    if eval_metric == 'iou':
        union = c[:, 0] + c[:, 1] - c[:, 2]
        present = c[:, 1] != 0
        per = np.zeros(num_labels)
        per[present] = c[present, 2] / union[present]
        return float(per.sum() / float(num_labels))
    n_p, n_t, tp = c[1]
    if eval_metric == 'dice':
        with np.errstate(invalid='ignore', divide='ignore'):
            return float(np.float64(2.0 * tp) / np.float64(n_p + n_t))
    if eval_metric == 'recall':
        return tp / n_t if n_t != 0 else (_ for _ in ()).throw(ZeroDivisionError('float division by zero'))
    return tp / n_p if n_p != 0 else (_ for _ in ()).throw(ZeroDivisionError('float division by zero'))


def _n_class_of(*masks):
    """`max(torch.unique(a).max(), torch.unique(b).max()) + 1` of the reference (lib/evalMetrics.py:198, lib/loss.py:367)."""
    return int(max(int(m.max().item()) for m in masks)) + 1


def get_multiclass_dice(pred, truth, n_class=None, eps=1e-11):
    """lib/evalMetrics.py:184-217: per-class Dice of two index masks, background (class 0) dropped: scores B x (n_class-1)
    float32 = 2|P&T| / (|P| + |T| + eps).  The reference builds both one-hot tensors and sums them in fp32 (exact below
    2^24 voxels); here the counts are exact integers from one pass over the two label maps.
    `truth` may also be a one-hot B x C x D x M x N mask (lib/evalMetrics.py:207-208)."""
    assert pred.shape[0] == truth.shape[0]
    assert pred.shape[-3:] == truth.shape[-3:]
    if pred.dim() + 1 == truth.dim():                 # one-hot truth -> index mask (exact for 0/1 masks)
        oh = truth
        if n_class is None:
            n_class = oh.shape[1]
        if not bool(((oh == 0) | (oh == 1)).all()) or not bool((oh.sum(1) == 1).all()):
            raise NotImplementedError('soft (non one-hot) truth is outside the accelerated eval path')
        truth = oh.argmax(1)
    if n_class is None:
        n_class = _n_class_of(truth, pred)
    c = ops.label_overlap_counts(pred, truth, n_class).to(torch.float32)
    return (2. * c[:, 1:, 2]) / ((c[:, 1:, 0] + c[:, 1:, 1]) + eps)


def cal_metric_from_counts(n_pred, n_gt, n_both, eps=1e-11):
    """lib/evalMetrics.py:151-181 cal_metric from the three counts of one (sample, label): -1 everywhere when the label is
    absent from the ground truth."""
    res = {'iou': -1, 'dice': -1, 'recall': -1, 'precision': -1}
    if n_gt != 0:
        tp = float(n_both); fn = float(n_gt - n_both); fp = float(n_pred - n_both)
        union = float(n_pred + n_gt - n_both)
        res['iou'] = tp / (union + eps)
        res['recall'] = tp / (tp + fn + eps)
        res['precision'] = tp / (tp + fp + eps)
        res['dice'] = 2 * tp / (2 * tp + fn + fp + eps)
    return res


def get_multi_metric(pred, gt, eval_label_list=None, rm_bg=False):
    """lib/evalMetrics.py:103-148: iou / dice / recall / precision per (sample, label) plus the label- and batch-averages
    that skip the -1 entries; same dictionary layout (float64 numpy arrays).  pred / gt: device label maps B x ...
    (the reference takes numpy arrays and makes one host pass per label and sample over Python sets)."""
    pred_t = pred if torch.is_tensor(pred) else torch.as_tensor(np.asarray(pred))
    gt_t = gt if torch.is_tensor(gt) else torch.as_tensor(np.asarray(gt))
    dev = pred_t.device if pred_t.is_cuda else (gt_t.device if gt_t.is_cuda else torch.device('cuda', torch.cuda.current_device()))
    pred_t, gt_t = pred_t.to(dev), gt_t.to(dev)
    if int(min(int(pred_t.min().item()), int(gt_t.min().item()))) < 0:
        raise ValueError('labels must be non-negative')
    n_class = _n_class_of(gt_t, pred_t)
    counts = ops.label_overlap_counts(pred_t, gt_t, n_class).cpu().numpy()          # B x n_class x 3
    label_list = np.nonzero(counts[:, :, 1].sum(0))[0].tolist()                       # == np.unique(gt).tolist()
    if rm_bg:
        label_list = label_list[1:]
    if eval_label_list is not None:
        for label in eval_label_list:
            assert label in label_list, "label {} is not in label_list".format(label)
        label_list = eval_label_list
    num_label, num_batch = len(label_list), counts.shape[0]
    metrics = ['iou', 'dice', 'recall', 'precision']
    multi_metric_res = {m: np.zeros([num_batch, num_label]) for m in metrics}
    label_avg_res = {m: np.zeros([num_batch, 1]) for m in metrics}
    batch_avg_res = {m: np.zeros([1, num_label]) for m in metrics}
    for l, lab in enumerate(label_list):
        for b in range(num_batch):
            r = cal_metric_from_counts(int(counts[b, lab, 0]), int(counts[b, lab, 1]), int(counts[b, lab, 2]))
            for m in metrics:
                multi_metric_res[m][b][l] = r[m]
    with np.errstate(invalid='ignore'), __import__('warnings').catch_warnings():
        __import__('warnings').simplefilter('ignore', category=RuntimeWarning)
        for m in metrics:
            for s_ in range(num_batch):
                row = multi_metric_res[m][s_]
                label_avg_res[m][s_] = float(np.mean(row[np.where(row != -1)]))
            for l in range(num_label):
                col = multi_metric_res[m][:, l]
                batch_avg_res[m][:, l] = float(np.mean(col[np.where(col != -1)]))
    return {'multi_metric_res': multi_metric_res, 'label_avg_res': label_avg_res, 'batch_avg_res': batch_avg_res,
            'label_list': label_list}

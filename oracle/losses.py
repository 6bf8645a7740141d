"""Oracle: restatement of the reference's hot-path losses and eval metric.

Test infrastructure only (see oracle/__init__.py).  Cites are into /root/reference.
"""
import numpy as np
import torch
import torch.nn.functional as F


def mask_to_one_hot(mask, n_classes):
    """lib/transforms.py:675-689: B x 1 x ... index mask -> B x C x ... float one-hot (zeros + scatter_)."""
    shape = list(mask.shape)
    shape[1] = n_classes
    one_hot = torch.zeros(shape, dtype=torch.float32, device=mask.device)
    one_hot.scatter_(1, mask.long(), 1)
    return one_hot


def dice_loss(source, target, n_class, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6):
    """DiceLossMultiClass.forward, lib/loss.py:410-476."""
    assert source.shape[0] == target.shape[0]
    shape = list(source.shape)
    if softmax:
        source = F.softmax(source, dim=1)                                   # :426-427
    src = source.reshape(shape[0], shape[1], -1)
    if target.dim() == len(shape) - 1:                                      # :433-434 index target
        tgt = mask_to_one_hot(target.reshape(shape[0], 1, -1), n_class).to(src.dtype)
    elif target.shape[1] == shape[1]:                                       # :435-436 soft target
        tgt = target.reshape(shape[0], shape[1], -1)
    else:
        raise ValueError("Incorrect size of target tensor")                 # :437-440
    if no_bg:                                                               # :444-446
        src, tgt = src[:, 1:, :], tgt[:, 1:, :]
    sv, tv = src.sum(2), tgt.sum(2)                                         # :449-450
    if weight_type == 'Simple':                                             # :452-454
        w = (tv.float() ** (1. / 3.) + eps).reciprocal()
    elif weight_type == 'Volume':                                           # :458-463
        w = (tv + eps).float().reciprocal()
        tmp = torch.where(torch.isinf(w), torch.ones_like(w), w)
        mx = tmp.max(dim=1, keepdim=True)[0]
        w = torch.where(torch.isinf(w), torch.ones_like(w) * mx, w)
    elif weight_type == 'Uniform':                                          # :464-465
        w = torch.ones(shape[0], shape[1] - int(no_bg))
    else:
        raise ValueError("Class weighting type {} does not exists!".format(weight_type))
    w = (w / w.max()).to(src.dtype)                                         # :468
    inter = (src * tgt).sum(2)                                              # :472
    scores = (2. * inter + eps) / ((sv + tv) + 2 * eps)                     # :473-474
    return 1 - (w * scores).sum() / w.sum()                                 # :476


def ncc_loss(inp, tgt):
    """NormalizedCrossCorrelationLoss.forward, lib/loss.py:493-501 (no eps, not squared)."""
    x = inp.reshape(inp.shape[0], -1)
    y = tgt.reshape(tgt.shape[0], -1)
    xm = x - x.mean(1, keepdim=True)
    ym = y - y.mean(1, keepdim=True)
    ncc = (xm * ym).mean(1) / (torch.sqrt((xm ** 2).mean(1)) * torch.sqrt((ym ** 2).mean(1)))
    return 1 - ncc.mean()


def bending_energy_loss(disp, spacing=(1., 1., 1.), normalize=True, norm='L2'):
    """BendingEnergyLoss.forward, lib/loss.py:687-730 (norm != 'L2' skips the weighting block :721-727: plain means of the
    absolute differences).

    Keeps the reference's quirks: mixed differences are not divided by 4; the per-axis
    weight vector spatial_dims=(D,H,W)/min is broadcast over the CHANNEL axis (:694-696,
    :722-727), i.e. channel c is weighted by dim c of (D,H,W)."""
    sp = torch.tensor(spacing, dtype=torch.float32)
    if normalize:
        sp = sp / sp.min()
    sp = sp.to(disp.dtype)
    dims = torch.tensor(disp.shape[2:], dtype=torch.float32)
    if normalize:
        dims = dims / dims.min()
    dims = dims.to(disp.dtype)
    u = disp
    B, C = u.shape[0], u.shape[1]
    c = u[:, :, 1:-1, 1:-1, 1:-1]
    ddx = (u[:, :, 2:, 1:-1, 1:-1] + u[:, :, :-2, 1:-1, 1:-1] - 2 * c).abs().reshape(B, C, -1)
    ddy = (u[:, :, 1:-1, 2:, 1:-1] + u[:, :, 1:-1, :-2, 1:-1] - 2 * c).abs().reshape(B, C, -1)
    ddz = (u[:, :, 1:-1, 1:-1, 2:] + u[:, :, 1:-1, 1:-1, :-2] - 2 * c).abs().reshape(B, C, -1)
    dxdy = (u[:, :, 2:, 2:, 1:-1] + u[:, :, :-2, :-2, 1:-1] - u[:, :, 2:, :-2, 1:-1] - u[:, :, :-2, 2:, 1:-1]).abs().reshape(B, C, -1)
    dydz = (u[:, :, 1:-1, 2:, 2:] + u[:, :, 1:-1, :-2, :-2] - u[:, :, 1:-1, 2:, :-2] - u[:, :, 1:-1, :-2, 2:]).abs().reshape(B, C, -1)
    dxdz = (u[:, :, 2:, 1:-1, 2:] + u[:, :, :-2, 1:-1, :-2] - u[:, :, 2:, 1:-1, :-2] - u[:, :, :-2, 1:-1, 2:]).abs().reshape(B, C, -1)
    if norm == 'L2':                                                        # :721-727
        ddx = (ddx ** 2).mean(2) * (dims * sp / (sp[0] ** 2)) ** 2
        ddy = (ddy ** 2).mean(2) * (dims * sp / (sp[1] ** 2)) ** 2
        ddz = (ddz ** 2).mean(2) * (dims * sp / (sp[2] ** 2)) ** 2
        dxdy = (dxdy ** 2).mean(2) * (dims * sp / (sp[0] * sp[1])) ** 2
        dydz = (dydz ** 2).mean(2) * (dims * sp / (sp[1] * sp[2])) ** 2
        dxdz = (dxdz ** 2).mean(2) * (dims * sp / (sp[2] * sp[0])) ** 2
    return (ddx.mean() + ddy.mean() + ddz.mean() + 2 * dxdy.mean() + 2 * dydz.mean() + 2 * dxdz.mean()) / 9.0


def eval_dice_per_class(logits, truth, n_classes):
    """models/segmentation.py:188-194 + lib/evalMetrics.py:58-68: argmax (first max index,
    torch.max(pred,1)[1]) then for c in 1..n_classes-1:
    1 - scipy.spatial.distance.dice(pred==c, truth==c) = 2|P&T| / (|P|+|T|) (NaN when both empty).
    Returns (dice[n_classes-1] float64 ndarray, argmax LongTensor)."""
    pred = torch.max(logits, 1)[1]
    p = pred.reshape(-1).numpy()
    t = truth.reshape(-1).numpy()
    out = np.zeros(n_classes - 1, dtype=np.float64)
    for c in range(1, n_classes):
        pc, tc = (p == c), (t == c)
        ntt = np.count_nonzero(pc & tc)
        ntf = np.count_nonzero(pc & ~tc)
        nft = np.count_nonzero(~pc & tc)
        with np.errstate(invalid='ignore', divide='ignore'):
            out[c - 1] = 1.0 - np.float64(ntf + nft) / np.float64(2 * ntt + ntf + nft)
    return out, pred


def multiclass_dice(pred, truth, n_class, eps=1e-11):
    """lib/evalMetrics.py:184-217 get_multiclass_dice for index masks (B x D x H x W)."""
    B = truth.shape[0]
    p1 = mask_to_one_hot(pred.reshape(B, 1, -1), n_class)[:, 1:, :]
    t1 = mask_to_one_hot(truth.reshape(B, 1, -1), n_class)[:, 1:, :]
    inter = (p1 * t1).sum(2)
    return (2. * inter) / ((p1.sum(2) + t1.sum(2)) + eps)


def dice_loss_on_label(source, target, n_class=None, eps=10e-6, weight_type='Uniform'):
    """lib/loss.py:348-391 DiceLossOnLabel.forward: masks B x 1 x D x M x N -> one-hot (background dropped), per (B, class)
    scores = 2 I w / (w (S + T) + eps), loss = 1 - mean(scores)."""
    assert source.shape == target.shape
    if n_class is None:                                                         # :366-367
        n_class = int(max(torch.unique(target).max(), torch.unique(source).max()).long().item()) + 1
    ms = list(target.shape)
    s1 = mask_to_one_hot(source.reshape(ms[0], ms[1], -1), n_class)[:, 1:, :]   # :370-375
    t1 = mask_to_one_hot(target.reshape(ms[0], ms[1], -1), n_class)[:, 1:, :]
    sv, tv = s1.sum(2), t1.sum(2)                                               # :378-379
    if weight_type == 'Simple':                                                 # :381-383
        w = tv.float().reciprocal()
        w = torch.where(torch.isinf(w), torch.ones_like(w), w)
    else:                                                                       # :384-385
        w = torch.ones(ms[0], ms[1])
    inter = s1 * t1
    scores = (2. * inter.sum(2).float() * w) / (w * (sv.float() + tv.float()) + eps)   # :387-389
    return 1 - scores.mean()


def cal_metric(label_pred, label_gt):
    """lib/evalMetrics.py:151-181 on two flat 0/1 arrays (counts instead of Python sets, same arithmetic)."""
    eps = 1e-11
    res = {'iou': -1, 'dice': -1, 'recall': -1, 'precision': -1}
    n_gt = int(np.count_nonzero(label_gt == 1)); n_pred = int(np.count_nonzero(label_pred == 1))
    n_both = int(np.count_nonzero((label_gt == 1) & (label_pred == 1)))
    union = n_gt + n_pred - n_both
    tp = float(n_both); fn = float(n_gt - n_both); fp = float(n_pred - n_both)
    if n_gt != 0:
        res['iou'] = tp / (float(union) + eps)
        res['recall'] = tp / (tp + fn + eps)
        res['precision'] = tp / (tp + fp + eps)
        res['dice'] = 2 * tp / (2 * tp + fn + fp + eps)
    return res


def multi_metric(pred, gt, eval_label_list=None, rm_bg=False):
    """lib/evalMetrics.py:103-148 get_multi_metric on numpy label maps B x ..."""
    label_list = np.unique(gt).tolist()
    if rm_bg:
        label_list = label_list[1:]
    if eval_label_list is not None:
        for label in eval_label_list:
            assert label in label_list
        label_list = eval_label_list
    nl, nb = len(label_list), pred.shape[0]
    metrics = ['iou', 'dice', 'recall', 'precision']
    mm = {m: np.zeros([nb, nl]) for m in metrics}
    la = {m: np.zeros([nb, 1]) for m in metrics}
    ba = {m: np.zeros([1, nl]) for m in metrics}
    for l in range(nl):
        lp = (pred == label_list[l]).astype(np.int32); lg = (gt == label_list[l]).astype(np.int32)
        for b in range(nb):
            r = cal_metric(lp[b].reshape(-1), lg[b].reshape(-1))
            for m in metrics:
                mm[m][b][l] = r[m]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', category=RuntimeWarning)
        for m in metrics:
            for s_ in range(nb):
                la[m][s_] = float(np.mean(mm[m][s_][np.where(mm[m][s_] != -1)]))
            for l in range(nl):
                ba[m][:, l] = float(np.mean(mm[m][:, l][np.where(mm[m][:, l] != -1)]))
    return {'multi_metric_res': mm, 'label_avg_res': la, 'batch_avg_res': ba, 'label_list': label_list}


def lncc_loss(I, J, filter_size=9, eps=1e-6):
    """lib/loss.py:599-617 VoxelMorphLNCC.forward with its all-ones filter (same op order)."""
    n = float(filter_size ** 3)
    filt = torch.ones(1, 1, filter_size, filter_size, filter_size, dtype=I.dtype)
    Is = F.conv3d(I, filt, padding=0); Js = F.conv3d(J, filt, padding=0)
    I2s = F.conv3d(I ** 2, filt, padding=0); J2s = F.conv3d(J ** 2, filt, padding=0); IJs = F.conv3d(I * J, filt, padding=0)
    Im, Jm = Is / n, Js / n
    cross = IJs - Im * Js - Jm * Is + Im * Jm * n
    Iv = I2s - 2 * Im * Is + Im ** 2 * n
    Jv = J2s - 2 * Jm * Js + Jm ** 2 * n
    cc = (cross ** 2) / (Iv * Jv + eps)
    return 1 - cc.mean()


def gradient_loss(u, norm='L2', spacing=(1, 1, 1), normalize=True):
    """lib/loss.py:640-671 gradientLoss.forward: note `+` for the H and W differences (:661,:663) and the 3-vector
    weights broadcast over the channel axis."""
    sp = torch.tensor(spacing).to(u.dtype)
    if normalize:
        sp = sp / sp.min()
    dims = torch.tensor(u.shape[2:]).to(u.dtype)
    if normalize:
        dims = dims / dims.min()
    N, C = u.shape[0], u.shape[1]
    dx = torch.abs(u[:, :, 2:, :, :] - u[:, :, :-2, :, :]).reshape(N, C, -1)
    dy = torch.abs(u[:, :, :, 2:, :] + u[:, :, :, :-2, :]).reshape(N, C, -1)
    dz = torch.abs(u[:, :, :, :, 2:] + u[:, :, :, :, :-2]).reshape(N, C, -1)
    if norm == 'L2':
        dx = (dx ** 2).mean(2) * (dims * sp / sp[0]) ** 2
        dy = (dy ** 2).mean(2) * (dims * sp / sp[1]) ** 2
        dz = (dz ** 2).mean(2) * (dims * sp / sp[2]) ** 2
    return (dx.mean() + dy.mean() + dz.mean()) / 3.0


def lncc_multiscale_loss(I, J):
    """lib/loss.py:512-586 LNCCLoss.forward (+ __stepup :516-540), filters on the CPU instead of .cuda()."""
    img_sz = list(I.shape[2:])
    ms = min(img_sz)
    if ms > 128:
        scale, weight, dil = [int(ms / 16), int(ms / 8), int(ms / 4)], [0.1, 0.3, 0.6], [2, 2, 2]
    elif ms > 64:
        scale, weight, dil = [int(ms / 4), int(ms / 2)], [0.3, 0.7], [2, 2]
    else:
        scale, weight, dil = [int(ms / 2)], [1.0], [1]
    total = 0.
    for k, w, d in zip(scale, weight, dil):
        step = max(int((k + 1) / 4), 1)
        filt = torch.ones([1, 1, k, k, k], dtype=I.dtype)
        conv = lambda t: F.conv3d(t, filt, padding=0, dilation=d, stride=step).view(I.shape[0], -1)
        Is, Js, I2s, J2s, IJs = conv(I), conv(J), conv(I ** 2), conv(J ** 2), conv(I * J)
        numel = float(k ** 3)
        Im, Jm = Is / numel, Js / numel
        cross = IJs - Jm * Is - Im * Js + Jm * Im * numel
        Iv = I2s - 2 * Im * Is + Im ** 2 * numel
        Jv = J2s - 2 * Jm * Js + Jm ** 2 * numel
        lncc = cross * cross / (Iv * Jv + 1e-5)
        total = total + (1 - lncc.mean()) * w
    return total


def cross_entropy_loss(logits, target, ignore_index=-100, reduction='mean'):
    """Registry 'cross_entropy' = torch.nn.CrossEntropyLoss (lib/loss.py:749), restated from its definition: -log_softmax(x)[t]
    over the voxels whose target is not `ignore_index`, mean over those voxels or sum."""
    C = logits.shape[1]
    lp = F.log_softmax(logits, 1).movedim(1, -1).reshape(-1, C)
    t = target.reshape(-1).long()
    keep = t != ignore_index
    picked = -lp[keep].gather(1, t[keep].unsqueeze(1)).squeeze(1)
    return picked.sum() if reduction == 'sum' else picked.sum() / keep.sum()


def focal_loss(inputs, targets, class_num, alpha=None, gamma=2, size_average=True, soft_max=True):
    """FocalLoss.forward, lib/loss.py:181-213.  Quirks kept: log_p is the log-softmax of `inputs` even when soft_max=False (:200), and
    probs = F.nll_loss(P, t) = -P[t] (:201), so the modulating factor (1 - probs)^gamma is (1 + P[t])^gamma."""
    C = inputs.shape[1]
    x = inputs.movedim(1, -1).reshape(-1, C)                                # :187-189
    t = targets.reshape(-1).long()
    P = F.softmax(x, dim=1) if soft_max else x                              # :193-196
    a = (torch.ones(class_num, 1) if alpha is None else alpha).to(x.dtype)[t].reshape(-1)   # :198
    log_p = F.log_softmax(x, 1).gather(1, t.unsqueeze(1)).squeeze(1)        # :200  (= -cross_entropy)
    probs = -P.gather(1, t.unsqueeze(1)).squeeze(1)                         # :201  (= nll_loss)
    batch_loss = -a * torch.pow(1 - probs, gamma) * log_p                   # :203
    return batch_loss.mean() if size_average else batch_loss.sum()          # :205-208


def soft_cross_entropy_loss(pred, target, softmax=True):
    """SoftCrossEntropy.forward with a class-probability target, lib/loss.py:151-154."""
    if softmax:
        return torch.mean(torch.sum(-target * F.log_softmax(pred, 1), 1))
    return torch.mean(torch.sum(-target * torch.log(pred.clamp(min=1e-8)), 1))


def seg_mask_to_one_hot(mask, n_classes):
    """transforms.SegMaskToOneHot.one_mask_to_one_hot, lib/transforms.py:663-673: D x M x N mask -> C x D x M x N float."""
    one_hot = torch.zeros([n_classes] + list(mask.shape), dtype=torch.float32)
    one_hot.scatter_(0, mask.unsqueeze(0).long(), 1)
    return one_hot

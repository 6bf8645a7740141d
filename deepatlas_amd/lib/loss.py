"""Loss registry: host-side mirror of lib/loss.py:739-761 with the hot-path losses as fused HIP kernels.

Hot path: 'dice' -> DiceLossMultiClass, 'ncc' -> NormalizedCrossCorrelationLoss,
'bendingEnergy' -> BendingEnergyLoss; SURVEY.md §8f f2: 'lncc' -> VoxelMorphLNCC, 'gradient' -> gradientLoss (reglosses.hip).
'mse' / 'L2' are one-line compositions; 'focal' / 'cross_entropy' / 'soft_cross_entropy' share one voxelwise HIP kernel pair (xent.hip).
"""
import os

import torch
import torch.nn as nn

from .. import ops


class DiceLossMultiClass(nn.Module):
    """Dice loss between a probability map / logits and a mask (lib/loss.py:397-476)."""

    def __init__(self, n_class=None, weight_type='Simple', no_bg=False, softmax=False, eps=1e-7):
        super(DiceLossMultiClass, self).__init__()
        self.weight_type = weight_type
        self.n_class = n_class
        self.eps = eps
        self.no_bg = no_bg
        self.softmax = softmax

    def forward(self, source, target):
        """source: B x C x D x M x N logits (softmax=True) or probabilities; target: B x D x M x N index mask,
        or B x C x D x M x N class probabilities (lib/loss.py:410-416)."""
        assert source.shape[0] == target.shape[0]
        assert source.shape[-3:] == target.squeeze().shape[-3:]
        if isinstance(source, ops.LazyLogits):
            # the network's output convolution has not been run (model.lazy_head): head + softmax + Dice in one kernel pair when this
            # is the softmax-Dice of logits against an index mask (models/segmentation.py:141-157), real logits otherwise
            if (self.softmax and len(target.shape) == len(source.shape) - 1 and self.n_class == source.shape[1]
                    and self.weight_type in ('Simple', 'Volume', 'Uniform')):
                x = source.x
                if isinstance(x, ops.LazyAct):
                    return ops.HeadDiceFn.apply(x.raw, source.weight, source.bias, target, self.weight_type, self.no_bg, self.eps,
                                                (x.scale, x.shift, x.slope))
                return ops.HeadDiceFn.apply(x, source.weight, source.bias, target, self.weight_type, self.no_bg, self.eps)
            source = source.materialize()
        if self.n_class is None:
            self.n_class = max(torch.unique(target).max(), torch.unique(source).max()).long().item() + 1
        shape = list(source.shape)
        if self.weight_type not in ('Simple', 'Volume', 'Uniform'):
            raise ValueError("Class weighting type {} does not exists!".format(self.weight_type))
        if len(target.shape) == len(shape) - 1:
            if shape[1] != self.n_class:
                raise ValueError("source has {} channels but n_class is {}".format(shape[1], self.n_class))
            return ops.DiceFn.apply(source, target, None, self.weight_type, self.no_bg, self.softmax, self.eps)
        elif target.shape[1] == shape[1]:
            return ops.DiceFn.apply(source, None, target, self.weight_type, self.no_bg, self.softmax, self.eps)
        raise ValueError("Incorrect size of target tensor: {}, should be {} or []".format(target.shape, shape,
                                                                                         shape[:1] + [1, ] + shape[2:]))


class DiceLossOnLabel(nn.Module):
    """lib/loss.py:348-391: Dice loss between two segmentation MASKS B x 1 x D x M x N (no gradient: labels), background
    dropped.  scores = 2|S&T| w / (w (|S| + |T|) + eps), loss = 1 - mean.  One pass over the two label maps
    (`da_label_overlap_counts`) instead of two materialised one-hot tensors."""

    def __init__(self, n_class=None, eps=10e-6):
        super(DiceLossOnLabel, self).__init__()
        self.n_class = n_class
        self.eps = eps

    def forward(self, source, target, weight_type='Uniform', average=True):
        assert source.shape == target.shape
        if self.n_class is None:
            self.n_class = int(max(int(target.max().item()), int(source.max().item()))) + 1
        B = target.shape[0]
        c = ops.label_overlap_counts(source.reshape(B, -1), target.reshape(B, -1), self.n_class)[:, 1:, :].to(torch.float32)
        source_volume, target_volume, inter = c[..., 0], c[..., 1], c[..., 2]
        if weight_type == 'Simple':
            weights = target_volume.reciprocal()
            weights = torch.where(torch.isinf(weights), torch.ones_like(weights), weights)
        elif weight_type == 'Uniform':
            weights = torch.ones(B, target.shape[1], device=c.device)
        else:
            raise ValueError("Class weighting type {} does not exists!".format(weight_type))
        scores = (2. * inter * weights) / (weights * (source_volume + target_volume) + self.eps)
        return 1 - scores.mean()


class NormalizedCrossCorrelationLoss(nn.Module):
    """1 - NCC (lib/loss.py:485-501)."""

    def __init__(self):
        super(NormalizedCrossCorrelationLoss, self).__init__()

    def forward(self, input, target):
        return ops.NCCFn.apply(input, target)


class BendingEnergyLoss(nn.Module):
    """Bending energy of a 3D displacement field (lib/loss.py:674-730).  norm='L2': weighted squared differences (:721-727); any
    other value skips that block in the reference, leaving the plain mean of the absolute differences (:729) -- same here."""

    def __init__(self, norm='L2', spacing=(1, 1, 1), normalize=True):
        super(BendingEnergyLoss, self).__init__()
        self.norm = norm
        self.spacing = torch.tensor(spacing).float()
        self.normalize = normalize
        if self.normalize:
            self.spacing /= self.spacing.min()

    def forward(self, input):
        return ops.BendingFn.apply(input, tuple(float(s) for s in self.spacing), self.normalize, 2 if self.norm == 'L2' else 1)


class VoxelMorphLNCC(nn.Module):
    """lib/loss.py:589-617 (registry name 'lncc'): local normalised cross-correlation over filter_size^3 windows.
    `filter` is kept as the reference's all-ones nn.Parameter (state_dict key 'filter'); the kernel is the separable box
    filter an all-ones window is, so a filter that is no longer all ones is refused instead of silently ignored."""

    def __init__(self, filter_size=9, eps=1e-6):
        super(VoxelMorphLNCC, self).__init__()
        self.filter_size = filter_size
        self.win_numel = self.filter_size ** 3
        self.filter = nn.Parameter(torch.ones(1, 1, filter_size, filter_size, filter_size), requires_grad=False)
        self.eps = eps

    def forward(self, I, J):
        key = (self.filter._version, self.filter.data_ptr())        # the check reads the device: once per state of the parameter, not per call
        if getattr(self, '_ones_checked', None) != key:
            if not bool((self.filter == 1).all()):
                raise NotImplementedError('VoxelMorphLNCC.filter must stay all ones on the accelerated path')
            self._ones_checked = key
        return ops.LNCCFn.apply(I, J, self.filter_size, self.eps)


class LNCCLoss(nn.Module):
    """lib/loss.py:512-586 (not in the registry): multi-scale LNCC.  Scales, weights, dilations and strides follow `__stepup`
    (:516-540): min(img) > 128 -> windows ms/16, ms/8, ms/4 (weights .1/.3/.6, dilation 2); > 64 -> ms/4, ms/2 (.3/.7,
    dilation 2); else ms/2 (1.0, dilation 1); stride max(int((k + 1) / 4), 1); eps 1e-5.  Each scale is one LNCCFn."""

    def initialize(self, kernel_sz=[9, 9, 9], voxel_weights=None):
        pass

    def _stepup(self, img_sz, use_multi_scale=True):
        max_scale = min(img_sz)
        if not use_multi_scale:
            raise NotImplementedError('the reference leaves self.scale undefined for use_multi_scale=False (loss.py:535-537)')
        if max_scale > 128:
            self.scale = [int(max_scale / 16), int(max_scale / 8), int(max_scale / 4)]
            self.scale_weight = [0.1, 0.3, 0.6]
            self.dilation = [2, 2, 2]
        elif max_scale > 64:
            self.scale = [int(max_scale / 4), int(max_scale / 2)]
            self.scale_weight = [0.3, 0.7]
            self.dilation = [2, 2]
        else:
            self.scale = [int(max_scale / 2)]
            self.scale_weight = [1.0]
            self.dilation = [1]
        self.num_scale = len(self.scale)
        self.kernel_sz = [[scale for _ in range(3)] for scale in self.scale]
        self.step = [[max(int((ksz + 1) / 4), 1) for ksz in self.kernel_sz[scale_id]] for scale_id in range(self.num_scale)]

    def forward(self, input, target, inst_weights=None, train=None):
        self._stepup(img_sz=list(input.shape[2:]))
        lncc_total = 0.
        for scale_id in range(self.num_scale):
            lncc = ops.LNCCFn.apply(input, target, self.scale[scale_id], 1e-5, self.dilation[scale_id], self.step[scale_id][0])
            lncc_total = lncc_total + lncc * self.scale_weight[scale_id]
        return lncc_total


class gradientLoss(nn.Module):
    """lib/loss.py:625-671 (registry name 'gradient'): first-difference regulariser of a N x 3 x D x H x W field, with the
    reference's `+` along H and W (loss.py:661,663) kept."""

    def __init__(self, norm='L2', spacing=(1, 1, 1), normalize=True):
        super(gradientLoss, self).__init__()
        self.norm = norm
        self.spacing = torch.tensor(spacing).float()
        self.normalize = normalize
        if self.normalize:
            self.spacing /= self.spacing.min()

    def forward(self, input):
        # the launcher normalises `spacing` itself (idempotent for an already normalised vector) and the volume dims
        return ops.GradLossFn.apply(input, tuple(self.spacing.tolist()), self.normalize, 2 if self.norm == 'L2' else 1)


class MSELoss(nn.Module):
    def forward(self, input, target):
        return ((input - target) ** 2).mean()


class L2Loss(nn.Module):
    def forward(self, input):
        return (input ** 2).mean()


class SoftCrossEntropy(nn.Module):
    """lib/loss.py:100-154 (registry 'soft_cross_entropy'): cross entropy against a class-probability target B x C x D x M x N.
    The reference's index-target branch multiplies the un-flattened B x D x M x N mask (`target`, not the one-hot `target_flat`, :151-153)
    into the B x C x D x M x N log-probabilities.  At the reference's batch size 1 that right-aligns and broadcasts the label VALUE over
    the class axis -- mean_v sum_c -label[v] log p[c, v] -- which is reproduced here (as a probability target whose every class entry is
    the label value); for B > 1 the reference's multiplication does not broadcast and the call raises, here as there."""

    def __init__(self, n_class=None, weight_type='Simple', no_bg=False, softmax=False):
        super(SoftCrossEntropy, self).__init__()
        self.weight_type = weight_type
        self.n_class = n_class
        self.no_bg = no_bg
        self.softmax = softmax

    def forward(self, pred, target):
        shape = list(pred.shape)
        if len(target.shape) == len(shape) - 1:
            # (B == C > 1 is the one other shape the reference's right-aligned multiplication accepts -- it then pairs target[b] with CLASS b
            # of every sample, which no caller can mean: rejected here on purpose like every other B > 1)
            if shape[0] != 1:
                raise RuntimeError("SoftCrossEntropy: an index target does not broadcast against the predictions in the reference for B > 1 "
                                   "(lib/loss.py:151-153 use `target`, not the one-hot `target_flat`); pass class probabilities B x C x ...")
            if os.environ.get('DA_CHECK_LABELS') == '1' and self.n_class is not None:       # the reference one-hots the labels first (mask_to_one_hot raises on out-of-range)
                lo, hi = int(target.min().item()), int(target.max().item())
                if lo < 0 or hi >= self.n_class:
                    raise RuntimeError('SoftCrossEntropy: label out of range [0, %d): min %d max %d' % (self.n_class, lo, hi))
            target = target.to(pred.dtype).unsqueeze(1).expand(shape).contiguous()        # the reference's broadcast at B = 1 (see the class docstring)
        if target.shape[1] != shape[1]:
            raise ValueError("Incorrect size of target tensor: {}, should be {} or []".format(target.shape, shape,
                                                                                             shape[:1] + [1, ] + shape[2:]))
        return ops.XentFn.apply(pred, None, target, None, 2, self.softmax, 0.0, -100, 0)


class FocalLoss(nn.Module):
    """lib/loss.py:157-213 (registry 'focal'): -alpha[t] (1 - probs)^gamma log_p with log_p = -F.cross_entropy(inputs, t) and
    probs = F.nll_loss(P, t) = -P[t] -- so the modulating factor is (1 + P[t])^gamma, as in the reference."""

    def __init__(self, class_num, alpha=None, gamma=2, size_average=True, soft_max=True):
        super(FocalLoss, self).__init__()
        self.alpha = torch.ones(class_num, 1) if alpha is None else alpha
        self.gamma = gamma
        self.class_num = class_num
        self.size_average = size_average
        self.soft_max = soft_max

    def forward(self, inputs, targets):
        return ops.XentFn.apply(inputs, targets, None, self.alpha, 1, self.soft_max, float(self.gamma), -100, 0 if self.size_average else 1)


class CrossEntropyLoss(nn.Module):
    """torch.nn.CrossEntropyLoss as registered under 'cross_entropy' (lib/loss.py:749): index targets B x D x M x N, `ignore_index`,
    reduction 'mean' | 'sum'.  Class weights / label smoothing are not part of the reference's use and raise."""

    def __init__(self, weight=None, size_average=None, ignore_index=-100, reduce=None, reduction='mean', label_smoothing=0.0):
        super(CrossEntropyLoss, self).__init__()
        if weight is not None or label_smoothing != 0.0 or reduction not in ('mean', 'sum'):
            raise NotImplementedError("HIP cross entropy: weight=None, label_smoothing=0, reduction 'mean' or 'sum'")
        self.ignore_index = ignore_index
        self.reduction = reduction

    def forward(self, input, target):
        return ops.XentFn.apply(input, target, None, None, 0, True, 0.0, self.ignore_index, 0 if self.reduction == 'mean' else 1)


loss_dict = {
    'ncc': NormalizedCrossCorrelationLoss,
    'lncc': VoxelMorphLNCC,
    'mse': MSELoss,
    'gradient': gradientLoss,
    'bendingEnergy': BendingEnergyLoss,
    'dice': DiceLossMultiClass,
    'L2': L2Loss,
    'focal': FocalLoss,
    'cross_entropy': CrossEntropyLoss,
    'soft_cross_entropy': SoftCrossEntropy,
}


def get_loss_function(loss_name):
    if loss_name in get_available_losses():
        return loss_dict[loss_name]
    else:
        raise KeyError("Network {} is not avaiable!\n Choose from: {}".format(loss_name, get_available_losses()))


def get_available_losses():
    return loss_dict.keys()

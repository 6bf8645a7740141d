"""Building blocks of the networks: host-side mirrors of the reference's blocks whose forward is HIP.

Parameter holders are the stock torch modules (nn.Conv3d / nn.ConvTranspose3d / nn.BatchNorm3d) so that
state_dict keys, shapes, default initialisation and `weights_init` behave exactly as in the reference
(lib/network_factory/unets.py:24-67, modules.py:28-86); their own forward() is never used -- the block
forward launches the HIP kernels through deepatlas_amd.ops.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from ... import ops

available_activations = {'ReLU': nn.ReLU, 'LeakyReLU': nn.LeakyReLU}


def get_activation_function(act):
    """unets.py:9-19 / modules.py:15-25 (unknown names return None, as in the reference)."""
    if act in available_activations:
        return available_activations[act]
    return None


def _slope_of(act_cls):
    if act_cls is None:
        return -1.0
    if act_cls is nn.ReLU:
        return 0.0
    if act_cls is nn.LeakyReLU:
        return 0.01                      # nn.LeakyReLU() default negative_slope
    raise NotImplementedError('activation %r has no HIP kernel' % (act_cls,))


class SegBlock(nn.Sequential):
    """unets.convBlock (unets.py:24-39): children 'conv' [, 'BN'], 'nonlinear' -- same state_dict keys.
    forward(x, skip=None) convolves concat(x, skip) without materialising it (unets.py:275)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True, batchnorm=False, act='ReLU'):
        act_F = get_activation_function(act)
        mods = OrderedDict()
        mods['conv'] = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=bias)
        if batchnorm:
            mods['BN'] = nn.BatchNorm3d(out_channels)
        mods['nonlinear'] = act_F()
        super().__init__(mods)
        if kernel_size != 3 or padding != 1 or stride not in (1, 2):
            raise NotImplementedError('HIP conv path covers kernel 3, padding 1, stride 1|2')
        self.stride = stride
        self.slope = _slope_of(act_F)
        self.batchnorm = batchnorm

    supports_lazy = True      # forward accepts ops.LazyAct inputs and `lazy_out` (deferred BatchNorm + activation, see ops.LazyAct)

    def forward(self, x, skip=None, lazy_out=False):
        conv = self.conv
        if self.batchnorm and self.stride == 1:
            bn = self.BN
            if self.training and bn.track_running_stats:
                ops.bump_batches_tracked(bn)
            # conv + BN + act as one autograd node (BN backward also yields the conv bias gradient); LazyAct inputs are consumed raw,
            # their BatchNorm + activation is applied by the convolution's input staging
            pro1 = (x.scale, x.shift, x.slope) if isinstance(x, ops.LazyAct) else None
            pro2 = (skip.scale, skip.shift, skip.slope) if isinstance(skip, ops.LazyAct) else None
            x1 = x.raw if pro1 is not None else x
            x2 = skip.raw if pro2 is not None else skip
            if pro1 is None and pro2 is None and not lazy_out:
                return ops.ConvBNActFn.apply(x1, x2, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                             self.training, bn.momentum, bn.eps, self.slope)
            out = ops.ConvBNActFn.apply(x1, x2, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                        self.training, bn.momentum, bn.eps, self.slope, False, pro1, pro2, lazy_out)
            return ops.LazyAct(out[0], out[1], out[2], self.slope) if lazy_out else out
        x, skip = ops.materialize(x), ops.materialize(skip)
        if self.batchnorm:
            bn = self.BN
            if self.training and bn.track_running_stats:
                ops.bump_batches_tracked(bn)
            y = ops.Conv3dK3Fn.apply(x, skip, conv.weight, conv.bias, self.stride, -1.0)
            return ops.BNActFn.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                     self.training, bn.momentum, bn.eps, self.slope)
        return ops.Conv3dK3Fn.apply(x, skip, conv.weight, conv.bias, self.stride, self.slope)


class SegUpBlock(nn.Sequential):
    """unets.deconvBlock (unets.py:42-58): 'deconv' [, 'BN'], 'nonlinear'; kernel 2, stride 2."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0,
                 bias=True, batchnorm=False, act='ReLU'):
        act_F = get_activation_function(act)
        mods = OrderedDict()
        mods['deconv'] = nn.ConvTranspose3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                            output_padding=output_padding, bias=bias)
        if batchnorm:
            mods['BN'] = nn.BatchNorm3d(out_channels)
        mods['nonlinear'] = act_F()
        super().__init__(mods)
        if kernel_size != 2 or stride != 2 or padding != 0 or output_padding != 0:
            raise NotImplementedError('HIP transposed-conv path covers kernel 2, stride 2')
        self.slope = _slope_of(act_F)
        self.batchnorm = batchnorm

    supports_lazy = True

    def forward(self, x, lazy_out=False):
        x = ops.materialize(x)
        if self.batchnorm:
            bn = self.BN
            if self.training and bn.track_running_stats:
                ops.bump_batches_tracked(bn)
            if lazy_out:
                out = ops.DeconvBNActFn.apply(x, self.deconv.weight, self.deconv.bias, bn.weight, bn.bias, bn.running_mean,
                                              bn.running_var, self.training, bn.momentum, bn.eps, self.slope, True)
                return ops.LazyAct(out[0], out[1], out[2], self.slope)
            return ops.DeconvBNActFn.apply(x, self.deconv.weight, self.deconv.bias, bn.weight, bn.bias, bn.running_mean,
                                           bn.running_var, self.training, bn.momentum, bn.eps, self.slope)
        y = ops.DeconvK2S2Fn.apply(x, self.deconv.weight, self.deconv.bias)
        return ops.ActFn.apply(y, self.slope)


class UNetEncBlock(nn.Sequential):
    """`UNet.encoder` (unets.py:113-124): nn.Sequential(Conv3d(k3,p1) [, BatchNorm3d], ReLU) with positional children, so the
    state_dict keys are '<name>.0.weight', '<name>.1.running_mean', ... as in the reference."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True, batchnorm=False):
        if kernel_size != 3 or padding != 1 or stride != 1:
            raise NotImplementedError('HIP conv path covers kernel 3, padding 1')
        mods = [nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=bias)]
        if batchnorm:
            mods.append(nn.BatchNorm3d(out_channels))
        mods.append(nn.ReLU())
        super().__init__(*mods)
        self.batchnorm = batchnorm

    def forward(self, x, skip=None):
        conv = self[0]
        if self.batchnorm:
            bn = self[1]
            if self.training and bn.track_running_stats:
                ops.bump_batches_tracked(bn)
            return ops.ConvBNActFn.apply(x, skip, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                         self.training, bn.momentum, bn.eps, 0.0)
        return ops.Conv3dK3Fn.apply(x, skip, conv.weight, conv.bias, 1, 0.0)


class UNetDecBlock(nn.Sequential):
    """`UNet.decoder` (unets.py:126-139): nn.Sequential(ConvTranspose3d [, BatchNorm3d], ReLU).  Two shapes occur
    (unets.py:88-96): kernel 2 / stride 2 (the up-sampler, pointwise MFMA GEMM) and kernel 3 / stride 1 / padding 1, which is a
    3x3x3 convolution with flipped taps and runs on the conv kernels (two-pointer concat input included)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0, bias=True, batchnorm=False):
        mods = [nn.ConvTranspose3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                   output_padding=output_padding, bias=bias)]
        if batchnorm:
            mods.append(nn.BatchNorm3d(out_channels))
        mods.append(nn.ReLU())
        super().__init__(*mods)
        self.batchnorm = batchnorm
        if (kernel_size, stride, padding, output_padding) == (2, 2, 0, 0):
            self.kind = 'up'
        elif (kernel_size, stride, padding, output_padding) == (3, 1, 1, 0):
            self.kind = 'conv'
        else:
            raise NotImplementedError('HIP transposed-conv path covers (k2, s2) and (k3, s1, p1)')

    def forward(self, x, skip=None):
        dc = self[0]
        bn = self[1] if self.batchnorm else None
        if bn is not None and self.training and bn.track_running_stats:
            ops.bump_batches_tracked(bn)
        if self.kind == 'up':
            if skip is not None:
                raise ValueError('the k2/s2 up-sampler takes a single input')
            if bn is not None:
                return ops.DeconvBNActFn.apply(x, dc.weight, dc.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                               self.training, bn.momentum, bn.eps, 0.0)
            return ops.ActFn.apply(ops.DeconvK2S2Fn.apply(x, dc.weight, dc.bias), 0.0)
        if bn is not None:
            return ops.ConvBNActFn.apply(x, skip, dc.weight, dc.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                         self.training, bn.momentum, bn.eps, 0.0, True)
        return ops.Conv3dK3Fn.apply(x, skip, dc.weight, dc.bias, 1, 0.0, True)


class HeadConv(nn.Conv3d):
    """nn.Conv3d(C, n_classes, 1) output layer (unets.py:249-250) -- keys '<idx>.weight', '<idx>.bias'."""

    supports_lazy = True      # accepts an ops.LazyAct input (deferred BatchNorm + activation of the last decoder block)

    lazy = False              # set through the network's `lazy_head` attribute: return an ops.LazyLogits instead of running the conv

    def forward(self, x):
        if (self.lazy and self.training and ops.FUSE_HEAD_DICE and torch.is_grad_enabled()
                and ops.head_dice_supported(self.weight.shape[1], self.weight.shape[0])):
            return ops.LazyLogits(x, self.weight, self.bias)
        if isinstance(x, ops.LazyAct):
            return ops.Conv1x1Fn.apply(x.raw, self.weight, self.bias, (x.scale, x.shift, x.slope))
        return ops.Conv1x1Fn.apply(x, self.weight, self.bias)


class DownConv(nn.Conv3d):
    """nn.Conv3d(C, C', kernel_size=2, stride=2, padding=0): the generator's maxpool=False down-sampler (unets.py:231-233);
    keys 'down_samplers.<i>.weight' / '.bias'."""

    def forward(self, x):
        return ops.ConvK2S2Fn.apply(x, self.weight, self.bias)


class UpsampleTrilinear2(nn.Upsample):
    """nn.Upsample(scale_factor=2, mode="trilinear"): the generator's upsample=True up-sampler (unets.py:236), no parameters."""

    def forward(self, x):
        return ops.UpsampleTrilinear2Fn.apply(x)


class MaxPool2(nn.MaxPool3d):
    """nn.MaxPool3d(2) (unets.py:230)."""

    def forward(self, x):
        return ops.MaxPool2Fn.apply(x)


class convBlock(nn.Module):
    """modules.convBlock (modules.py:28-62), the registration net's block: conv -> optional BN -> act class
    instantiated per call -> optional `x += x`.  Child names 'conv' / 'bn' as in the reference."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1,
                 bias=False, batchnorm=False, act=nn.ReLU, residual=False):
        super(convBlock, self).__init__()
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=bias)
        self.bn = nn.BatchNorm3d(out_channels) if batchnorm else None
        self.nonlinear = get_activation_function(act) if type(act) is str else act
        self.residual = residual
        if kernel_size != 3 or padding != 1 or stride not in (1, 2):
            raise NotImplementedError('HIP conv path covers kernel 3, padding 1, stride 1|2')
        self.stride = stride

    def up2_ok(self, x, skip, size):
        """Can this block consume `F.interpolate(cat(x, skip), size=size)` (voxel_morph.py:72-80) with the up-sampling folded into its
        convolution?  Exact x2 on every axis, no BatchNorm / residual, channel counts the folded kernels take, split matrix mode."""
        if self.bn is not None or self.residual or self.stride != 1:
            return False
        if tuple(int(v) for v in size) != tuple(2 * int(v) for v in x.shape[2:]) or (skip is not None and skip.shape[2:] != x.shape[2:]):
            return False
        return ops.upconv_supported(x.shape[1], skip.shape[1] if skip is not None else 0, self.conv.weight.shape[0])

    def forward(self, x, skip=None, fork=False, up2=False):
        """fork=True (no BatchNorm, no residual): returns the block's output twice -- for an output with two consumers (the
        registration net's skip connections) the two gradients are then summed inside the activation-backward pass.
        up2=True (only after up2_ok): x / skip are the COARSE tensors; the block computes conv(nearest-upsample x2 (cat(x, skip)))."""
        slope = _slope_of(self.nonlinear)
        if up2:
            return ops.Conv3dK3Fn.apply(x, skip, self.conv.weight, self.conv.bias, 1, slope, False, False, True)
        if fork and self.bn is None and not self.residual:
            return ops.Conv3dK3Fn.apply(x, skip, self.conv.weight, self.conv.bias, self.stride, slope, False, True)
        if fork:
            y = self.forward(x, skip)
            return y, y
        if self.bn is not None:
            y = ops.Conv3dK3Fn.apply(x, skip, self.conv.weight, self.conv.bias, self.stride, -1.0)
            bn = self.bn
            if self.training and bn.track_running_stats:
                ops.bump_batches_tracked(bn)
            y = ops.BNActFn.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, self.training, bn.momentum, bn.eps, slope)
        else:
            y = ops.Conv3dK3Fn.apply(x, skip, self.conv.weight, self.conv.bias, self.stride, slope)
        if self.residual:
            y = y + y                         # modules.py:59-60 `x += x`
        return y


class deconvBlock(nn.Module):
    """modules.deconvBlock (modules.py:65-86); only the k=2, s=2 form has a HIP kernel."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0,
                 bias=False, batchnorm=False, residual=False, act=nn.ReLU):
        super(deconvBlock, self).__init__()
        self.deconv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                         output_padding=output_padding, bias=bias)
        self.bn = nn.BatchNorm3d(out_channels) if batchnorm else None
        self.nonlinear = get_activation_function(act) if type(act) is str else act
        self.residual = residual
        if kernel_size != 2 or stride != 2 or padding != 0 or output_padding != 0:
            raise NotImplementedError('HIP transposed-conv path covers kernel 2, stride 2')

    def forward(self, input):
        slope = _slope_of(self.nonlinear)
        y = ops.DeconvK2S2Fn.apply(input, self.deconv.weight, self.deconv.bias)
        if self.bn is not None:
            bn = self.bn
            if self.training and bn.track_running_stats:
                ops.bump_batches_tracked(bn)
            y = ops.BNActFn.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, self.training, bn.momentum, bn.eps, slope)
        else:
            y = ops.ActFn.apply(y, slope)
        if self.residual:
            y = y + input
        return y

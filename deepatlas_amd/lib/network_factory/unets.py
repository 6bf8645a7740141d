"""Segmentation nets: host-side mirror of lib/network_factory/unets.py with HIP forwards/backwards.

`UNet_generator(encoders, decoders, act, upsample, maxpool, res)` returns a class with the reference's
constructor `(in_channel, n_classes, bias=False, BN=False)`, `forward(x) -> logits N x n_classes x D x H x W`,
`weights_init()` and identical state_dict keys.  All generator options are implemented: maxpool=False
(strided k2/s2 conv down-sampler), upsample=True (trilinear x2), res=True (residual adds), besides the 'UNet_light'
configuration (lib/network_factory/__init__.py:12-15); the fixed `UNet` (unets.py:70-179) is implemented in full.
"""
import torch
import torch.nn as nn

from ... import ops

from .modules import (SegBlock as convBlock, SegUpBlock as deconvBlock, HeadConv, MaxPool2, DownConv, UpsampleTrilinear2,
                      UNetEncBlock, UNetDecBlock, get_activation_function)


def init_conv_weights(m):
    """unets.py:61-67."""
    classname = m.__class__.__name__
    if classname.find('Conv') != -1:
        if not m.weight is None:
            ops.init_into(m.weight, nn.init.xavier_normal_)
        if not m.bias is None:
            m.bias.data.zero_()


def UNet_generator(encoders, decoders, act='ReLU', upsample=False, maxpool=True, res=False):
    """unets.py:182-280."""

    class UNetTemplate(nn.Module):
        def __init__(self, in_channel, n_classes, bias=False, BN=False):
            super(UNetTemplate, self).__init__()
            self.register_forward_hook(ops.flush_batches_tracked)      # the deferred num_batches_tracked increments of this pass: one launch
            self.in_channel = in_channel
            self.n_classes = n_classes
            self.levels = len(encoders)
            self.encoders = nn.ModuleList()
            self.decoders = nn.ModuleList()
            self.down_samplers = nn.ModuleList()
            self.up_samplers = nn.ModuleList()
            self.maxpool = maxpool
            self.upsample = upsample
            self.res = res

            for i, enc in enumerate(encoders):
                if i == 0:
                    enc = (in_channel,) + enc
                blocks = [convBlock(enc[k], enc[k + 1], bias=bias, batchnorm=BN, act=act) for k in range(len(enc) - 1)]
                self.encoders.append(nn.Sequential(*blocks))
                if i < len(encoders) - 1:
                    self.down_samplers.append(MaxPool2(2) if self.maxpool
                                              else DownConv(enc[-1], encoders[i + 1][0], kernel_size=2, stride=2, padding=0, bias=bias))

            for i, dec in enumerate(decoders):
                if self.upsample:
                    self.up_samplers.append(UpsampleTrilinear2(scale_factor=2, mode="trilinear"))
                else:
                    self.up_samplers.append(deconvBlock(encoders[-1][-1] if i == 0 else decoders[i - 1][-1], dec[0],
                                                        kernel_size=2, stride=2, bias=bias, batchnorm=BN, act=act))
                dec = (encoders[-(i + 2)][-1] + dec[0],) + dec[1:]
                # reference quirk kept: the block count comes from the leaked encoder loop variable (unets.py:247)
                blocks = [convBlock(dec[k], dec[k + 1], kernel_size=3, stride=1, padding=1, bias=bias, batchnorm=BN, act=act)
                          for k in range(len(enc) - 1)]
                if i == len(decoders) - 1:
                    blocks.append(HeadConv(dec[-1], n_classes, kernel_size=1, stride=1, padding=0, bias=bias))
                self.decoders.add_module('decBlock{}'.format(i), nn.Sequential(*blocks))
            ops.tag_conv_layouts(self)          # FlatAdam keeps tagged conv weights tap-major (the kernels' layout); see ops.tag_conv_layouts

        def weights_init(self):
            self.apply(init_conv_weights)
            ops.bump_weights_epoch()          # `.data` writes do not move torch's version counters: drop the cached weight layouts

        @property
        def lazy_head(self):
            """True: in training mode forward() returns an ops.LazyLogits (the output convolution is evaluated inside the fused
            head + softmax + Dice kernels by DiceLossMultiClass, or on .materialize()); default False: a logits tensor, always."""
            return any(m.lazy for m in self.modules() if isinstance(m, HeadConv))

        @lazy_head.setter
        def lazy_head(self, flag):
            for m in self.modules():
                if isinstance(m, HeadConv):
                    m.lazy = bool(flag)

        def forward(self, x):
            """unets.py:259-278.  The skip concat (up-sampled first, skip second) is a two-pointer conv input."""
            # Deferred BatchNorm + activation (ops.LazyAct): inside conv -> conv chains the activated tensor is never written; the next
            # convolution applies it while staging its input.  Block outputs that feed anything else are materialised.
            def lazy_ok(blk, nxt):
                return (ops.LAZY_BN and nxt is not None and getattr(blk, 'supports_lazy', False) and getattr(blk, 'batchnorm', False)
                        and getattr(nxt, 'supports_lazy', False) and getattr(blk, 'stride', 1) == 1
                        and ((isinstance(nxt, convBlock) and nxt.batchnorm and nxt.stride == 1) or isinstance(nxt, HeadConv)))
            temp = []
            for i, enc in enumerate(self.encoders):
                y = x
                blks = list(enc)
                # the level's last block may also stay lazy when its only consumer is the max-pool: the pool pass applies BatchNorm +
                # activation itself and writes the activated tensor as the skip connection
                into_pool = (ops.LAZY_BN and self.maxpool and not self.res and i < self.levels - 1 and getattr(blks[-1], 'supports_lazy', False)
                             and getattr(blks[-1], 'batchnorm', False) and getattr(blks[-1], 'stride', 1) == 1 and isinstance(blks[-1], convBlock))
                for k, blk in enumerate(blks):
                    nxt = blks[k + 1] if k + 1 < len(blks) else None
                    y = blk(y, lazy_out=True) if (lazy_ok(blk, nxt) or (nxt is None and into_pool)) else blk(y)
                x = (y + x) if self.res else y            # res=True: `enc(x) + x` (unets.py:264; broadcasts a 1-channel input)
                if i < self.levels - 1:
                    if self.maxpool:          # skip tensor + pooled tensor from one node (gradients summed in the pool backward)
                        if isinstance(x, ops.LazyAct):
                            skip, x = ops.MaxPool2SkipFn.apply(x.raw, x.scale, x.shift, x.slope)
                        else:
                            skip, x = ops.MaxPool2SkipFn.apply(x)
                        temp.append(skip)
                    else:
                        temp.append(x)
                        x = self.down_samplers[i](x)
            for j, dec in enumerate(self.decoders):
                blocks = list(dec)
                up = self.up_samplers[j]
                x = up(x, lazy_out=True) if (ops.LAZY_BN_UPSAMPLER and not self.res and lazy_ok(up, blocks[0])) else up(x)
                skip = temp.pop()
                nxt = blocks[1] if len(blocks) > 1 else None
                y = blocks[0](x, skip, lazy_out=True) if lazy_ok(blocks[0], nxt) else blocks[0](x, skip)
                for k in range(1, len(blocks)):
                    nxt = blocks[k + 1] if k + 1 < len(blocks) else None
                    y = blocks[k](y, lazy_out=True) if lazy_ok(blocks[k], nxt) else blocks[k](y)
                x = (ops.materialize_logits(y) + x) if self.res else y            # res=True: `dec(cat(x, skip)) + x` (unets.py:275)
            return x

    return UNetTemplate


class UNet(nn.Module):
    """The fixed 8+9+1-layer `UNet` (unets.py:70-179): 32 -> 512 channels, ReLU, three max-pools, ConvTranspose3d(k2,s2)
    up-samplers and ConvTranspose3d(k3,s1,p1) decoder "convs"; same attribute names, hence the same state_dict keys
    (`ec0.0.weight`, `dc8.1.running_mean`, `dc0.bias`, ...).  SURVEY.md row f3."""

    def __init__(self, in_channel, n_classes, bias=False, BN=False):
        self.in_channel = in_channel
        self.n_classes = n_classes
        super(UNet, self).__init__()
        self.register_forward_hook(ops.flush_batches_tracked)      # the deferred num_batches_tracked increments of this pass: one launch
        self.ec0 = self.encoder(self.in_channel, 32, bias=bias, batchnorm=BN)
        self.ec1 = self.encoder(32, 64, bias=bias, batchnorm=BN)
        self.ec2 = self.encoder(64, 64, bias=bias, batchnorm=BN)
        self.ec3 = self.encoder(64, 128, bias=bias, batchnorm=BN)
        self.ec4 = self.encoder(128, 128, bias=bias, batchnorm=BN)
        self.ec5 = self.encoder(128, 256, bias=bias, batchnorm=BN)
        self.ec6 = self.encoder(256, 256, bias=bias, batchnorm=BN)
        self.ec7 = self.encoder(256, 512, bias=bias, batchnorm=BN)

        self.pool0 = MaxPool2(2)
        self.pool1 = MaxPool2(2)
        self.pool2 = MaxPool2(2)

        self.dc9 = self.decoder(512, 512, kernel_size=2, stride=2, bias=bias, batchnorm=BN)
        self.dc8 = self.decoder(256 + 512, 256, kernel_size=3, stride=1, padding=1, bias=bias, batchnorm=BN)
        self.dc7 = self.decoder(256, 256, kernel_size=3, stride=1, padding=1, bias=bias, batchnorm=BN)
        self.dc6 = self.decoder(256, 256, kernel_size=2, stride=2, bias=bias, batchnorm=BN)
        self.dc5 = self.decoder(128 + 256, 128, kernel_size=3, stride=1, padding=1, bias=bias, batchnorm=BN)
        self.dc4 = self.decoder(128, 128, kernel_size=3, stride=1, padding=1, bias=bias, batchnorm=BN)
        self.dc3 = self.decoder(128, 128, kernel_size=2, stride=2, bias=bias, batchnorm=BN)
        self.dc2 = self.decoder(64 + 128, 64, kernel_size=3, stride=1, padding=1, bias=bias, batchnorm=BN)
        self.dc1 = self.decoder(64, 64, kernel_size=3, stride=1, padding=1, bias=bias, batchnorm=BN)
        self.dc0 = HeadConv(64, n_classes, kernel_size=1, stride=1, padding=0, bias=bias)
        ops.tag_conv_layouts(self)

    @property
    def lazy_head(self):
        """see UNetTemplate.lazy_head"""
        return self.dc0.lazy

    @lazy_head.setter
    def lazy_head(self, flag):
        self.dc0.lazy = bool(flag)

    def weights_init(self):
        """unets.py:102-110: xavier-normal on every module whose class name contains 'Conv' (the nn.Conv3d / ConvTranspose3d
        parameter holders and the head)."""
        for m in self.modules():
            classname = m.__class__.__name__
            if classname.find('Conv') != -1:
                if not m.weight is None:
                    ops.init_into(m.weight, nn.init.xavier_normal_)
                if not m.bias is None:
                    m.bias.data.zero_()
        ops.bump_weights_epoch()

    def encoder(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True, batchnorm=False):
        return UNetEncBlock(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=bias, batchnorm=batchnorm)

    def decoder(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0, bias=True, batchnorm=False):
        return UNetDecBlock(in_channels, out_channels, kernel_size, stride=stride, padding=padding, output_padding=output_padding,
                            bias=bias, batchnorm=batchnorm)

    def forward(self, x):
        """unets.py:141-179; every torch.cat((up, syn), 1) is the two-pointer input of the following block."""
        syn0, p0 = ops.MaxPool2SkipFn.apply(self.ec1(self.ec0(x)))         # pool0/1/2 (parameter-free) fused with their skip branch
        syn1, p1 = ops.MaxPool2SkipFn.apply(self.ec3(self.ec2(p0)))
        syn2, p2 = ops.MaxPool2SkipFn.apply(self.ec5(self.ec4(p1)))
        e7 = self.ec7(self.ec6(p2))
        d7 = self.dc7(self.dc8(self.dc9(e7), syn2))
        d4 = self.dc4(self.dc5(self.dc6(d7), syn1))
        d1 = self.dc1(self.dc2(self.dc3(d4), syn0))
        return self.dc0(d1)

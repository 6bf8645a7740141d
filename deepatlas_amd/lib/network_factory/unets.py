"""Segmentation nets: host-side mirror of lib/network_factory/unets.py with HIP forwards/backwards.

`UNet_generator(encoders, decoders, act, upsample, maxpool, res)` returns a class with the reference's
constructor `(in_channel, n_classes, bias=False, BN=False)`, `forward(x) -> logits N x n_classes x D x H x W`,
`weights_init()` and identical state_dict keys.  Implemented options: maxpool=True,
upsample=False, res=False (the 'UNet_light' configuration, lib/network_factory/__init__.py:12-15).
"""
import torch
import torch.nn as nn

from .modules import SegBlock as convBlock, SegUpBlock as deconvBlock, HeadConv, MaxPool2, get_activation_function


def init_conv_weights(m):
    """unets.py:61-67."""
    classname = m.__class__.__name__
    if classname.find('Conv') != -1:
        if not m.weight is None:
            nn.init.xavier_normal_(m.weight.data)
        if not m.bias is None:
            m.bias.data.zero_()


def UNet_generator(encoders, decoders, act='ReLU', upsample=False, maxpool=True, res=False):
    """unets.py:182-280."""
    if upsample or not maxpool or res:
        raise NotImplementedError("HIP path implements maxpool=True, upsample=False, res=False (SURVEY.md §8f f3)")

    class UNetTemplate(nn.Module):
        def __init__(self, in_channel, n_classes, bias=False, BN=False):
            super(UNetTemplate, self).__init__()
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
                    self.down_samplers.append(MaxPool2(2))

            for i, dec in enumerate(decoders):
                self.up_samplers.append(deconvBlock(encoders[-1][-1] if i == 0 else decoders[i - 1][-1], dec[0],
                                                    kernel_size=2, stride=2, bias=bias, batchnorm=BN, act=act))
                dec = (encoders[-(i + 2)][-1] + dec[0],) + dec[1:]
                # reference quirk kept: the block count comes from the leaked encoder loop variable (unets.py:247)
                blocks = [convBlock(dec[k], dec[k + 1], kernel_size=3, stride=1, padding=1, bias=bias, batchnorm=BN, act=act)
                          for k in range(len(enc) - 1)]
                if i == len(decoders) - 1:
                    blocks.append(HeadConv(dec[-1], n_classes, kernel_size=1, stride=1, padding=0, bias=bias))
                self.decoders.add_module('decBlock{}'.format(i), nn.Sequential(*blocks))

        def weights_init(self):
            self.apply(init_conv_weights)

        def forward(self, x):
            """unets.py:259-278.  The skip concat (up-sampled first, skip second) is a two-pointer conv input."""
            temp = []
            for i, enc in enumerate(self.encoders):
                for blk in enc:
                    x = blk(x)
                if i < self.levels - 1:
                    temp.append(x)
                    x = self.down_samplers[i](x)
            for j, dec in enumerate(self.decoders):
                x = self.up_samplers[j](x)
                skip = temp.pop()
                blocks = list(dec)
                x = blocks[0](x, skip)
                for blk in blocks[1:]:
                    x = blk(x)
            return x

    return UNetTemplate


class UNet(nn.Module):
    """The fixed 19-layer `UNet` (unets.py:70-179) is registered for API parity; its ConvTranspose3d(k3,s1,p1)
    decoder blocks have no HIP kernel yet, so construction raises loudly."""

    def __init__(self, in_channel, n_classes, bias=False, BN=False):
        super(UNet, self).__init__()
        raise NotImplementedError("'UNet' (unets.py:70-179) is outside the accelerated hot path this round; use 'UNet_light'")

"""Registration net: host-side mirror of lib/network_factory/voxel_morph.py with HIP forwards/backwards."""
import torch
import torch.nn as nn

from ... import ops
from .modules import convBlock


class FlowConv(nn.Conv3d):
    """`self.flow` (voxel_morph.py:57): 3x3x3 conv to the displacement field, no activation; keys flow.weight/bias."""

    def forward(self, x, skip=None):
        return ops.Conv3dK3Fn.apply(x, skip, self.weight, self.bias, 1, -1.0)


class VoxelMorphCVPR2018(nn.Module):
    """voxel_morph.py:18-101.  forward(source, target) -> (disp_field, warped_source, deform_field)."""

    def __init__(self, input_channel=2, output_channel=3, enc_filters=(16, 32, 32, 32, 32), dec_filters=(32, 32, 32, 8, 8)):
        super(VoxelMorphCVPR2018, self).__init__()
        self.register_forward_hook(ops.flush_batches_tracked)      # (convBlocks built with batchnorm=True defer their num_batches_tracked increments)
        self.input_channel = input_channel
        self.output_channel = output_channel
        self.enc_filters = enc_filters
        self.dec_filters = dec_filters
        self.encoders = nn.ModuleList()
        self.decoders = nn.ModuleList()
        for i in range(len(enc_filters)):
            if i == 0:
                self.encoders.append(convBlock(input_channel, enc_filters[i], stride=1, bias=True))
            else:
                self.encoders.append(convBlock(enc_filters[i - 1], enc_filters[i], stride=2, bias=True))
        for i in range(len(dec_filters)):
            if i == 0:
                self.decoders.append(convBlock(enc_filters[-1], dec_filters[i], stride=1, bias=True))
            elif i < 4:
                self.decoders.append(convBlock(dec_filters[i - 1] + enc_filters[4 - i], dec_filters[i], stride=1, bias=True))
            else:
                self.decoders.append(convBlock(dec_filters[i - 1], dec_filters[i], stride=1, bias=True))
        self.flow = FlowConv(dec_filters[-1] + enc_filters[0], output_channel, kernel_size=3, stride=1, padding=1, bias=True)
        self.id_transform = None      # kept for attribute parity; the identity grid is generated inside the warp kernel
        ops.tag_conv_layouts(self)    # FlatAdam keeps tagged conv weights tap-major (the kernels' layout); see ops.tag_conv_layouts

    def forward(self, source, target):
        up = ops.UpsampleNearestFn.apply
        # cat(source, target) is a two-pointer conv input (voxel_morph.py:65)
        # encoder outputs feed the next encoder AND a decoder (skip): fork=True hands out two aliases so that the two gradients are
        # summed inside the block's activation-backward pass instead of by a separate autograd accumulation pass
        e1, e1s = self.encoders[0](source, target, fork=True)
        e2, e2s = self.encoders[1](e1, fork=True)
        e3, e3s = self.encoders[2](e2, fork=True)
        e4, e4s = self.encoders[3](e3, fork=True)
        e5 = self.encoders[4](e4)

        def up_conv(block, size, a, b=None):
            # conv(F.interpolate(cat(a, b), size)) (voxel_morph.py:72-80; interpolate of a concat = concat of the interpolates for 'nearest'):
            # for an exact x2 the up-sampling is folded into the convolution (ops.Conv3dK3Fn up2, conv3d_up2.hip); odd pyramids and other
            # matrix modes materialise the up-sampled tensors
            if block.up2_ok(a, b, size):
                return block(a, b, up2=True)
            return block(up(a, size), up(b, size) if b is not None else None)
        d1 = up_conv(self.decoders[0], e4.shape[2:], e5)
        d2 = up_conv(self.decoders[1], e3.shape[2:], d1, e4s)
        d3 = up_conv(self.decoders[2], e2.shape[2:], d2, e3s)
        d4 = self.decoders[3](d3, e2s)
        d5 = up_conv(self.decoders[4], e1.shape[2:], d4)
        disp_field = self.flow(d5, e1s)
        warped_source, deform_field = ops.WarpFn.apply(source, disp_field)
        return disp_field, warped_source, deform_field

    def weights_init(self):
        for m in self.modules():
            classname = m.__class__.__name__
            if classname.find('Conv') != -1:
                if not m.weight is None:
                    ops.init_into(m.weight, nn.init.xavier_normal_)
                if not m.bias is None:
                    m.bias.data.zero_()
        ops.bump_weights_epoch()              # `.data` writes do not move torch's version counters: drop the cached weight layouts

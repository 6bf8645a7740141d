"""Tensor-level transforms on the hot path (lib/transforms.py): mask_to_one_hot, SegMaskToOneHot, CropTensor,
SitkToTensor's tensor half.  The SimpleITK sample transforms are CPU data preparation and out of scope."""
import numpy as np
import torch

from .. import ops


def mask_to_one_hot(mask, n_classes):
    """lib/transforms.py:675-689: B x 1 x D x M x N index mask -> B x C x D x M x N float one-hot."""
    return ops.one_hot(mask, n_classes)


class SegMaskToOneHot:
    """lib/transforms.py:652-673."""

    def __init__(self, n_classes, dtype=torch.float):
        self.n_classes = n_classes
        self.dtype = dtype

    def __call__(self, sample):
        sample['segmentation_onehot'] = self.one_mask_to_one_hot(sample['segmentation'])
        return sample

    def one_mask_to_one_hot(self, mask):
        """mask D x M x N -> C x D x M x N."""
        return ops.one_hot(mask.unsqueeze(0).unsqueeze(0), self.n_classes)[0].to(self.dtype)


class CropTensor:
    """lib/transforms.py:124-158: crop [x0, x1, y0, y1, z0, z1] voxels off the borders of C x D x H x W tensors."""

    def __init__(self, crop_size):
        self.crop_size = crop_size

    def __call__(self, sample):
        c = self.crop_size
        for key in ('image', 'segmentation'):
            if key in sample:
                t = sample[key]
                sz = t.shape
                sample[key] = t[..., c[0]:sz[-3] - c[1], c[2]:sz[-2] - c[3], c[4]:sz[-1] - c[5]]
        return sample


class SitkToTensor:
    """Tensor half of lib/transforms.py:71-92: image -> float clamped to [0,1] with a channel axis, segmentation -> uint8.
    Accepts numpy arrays / tensors (the SimpleITK read itself is host I/O)."""

    def __call__(self, sample):
        img = torch.as_tensor(np.asarray(sample['image'])).float()
        img = torch.clamp(img, 0, 1)
        sample['image'] = img.unsqueeze(0)
        if 'segmentation' in sample:
            sample['segmentation'] = torch.as_tensor(np.asarray(sample['segmentation'])).to(torch.uint8)
        return sample

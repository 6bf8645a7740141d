"""Tensor-level transforms (lib/transforms.py) on the device: mask_to_one_hot / SegMaskToOneHot (:652-689), and the data path of
SURVEY.md row f4 -- SitkToTensor's clamp + cast (:71-92), CropTensor (:124-158), Partition's overlap tiling with reflect padding
and both assemble modes (:508-649).  The SimpleITK read / resample / crop filters themselves are host I/O and stay out of scope;
these classes take numpy arrays (what sitk.GetArrayFromImage returns) or tensors and hand back DEVICE tensors, so a loader
thread only uploads the raw volume once.
"""
import ctypes

import numpy as np
import torch

from .. import ops
from .. import _native as nat
from .._native import call, ptr, stream

_DTYPE_CODE = {torch.float32: 0, torch.float64: 1, torch.int16: 2, torch.uint8: 3, torch.int32: 4}


def _device():
    return torch.device('cuda', torch.cuda.current_device())


def _to_device_array(a):
    """numpy array / tensor / SimpleITK image -> contiguous device tensor in its own dtype."""
    if not torch.is_tensor(a):
        if not isinstance(a, np.ndarray):
            try:                                     # a SimpleITK image, when the module is present
                import SimpleITK as sitk
                a = sitk.GetArrayFromImage(a)
            except ImportError:
                a = np.asarray(a)
        a = torch.from_numpy(np.ascontiguousarray(a))
    return a.to(_device()).contiguous()


def _int3(v):
    arr = (ctypes.c_int * 3)(int(v[0]), int(v[1]), int(v[2]))
    return arr, ctypes.cast(arr, ctypes.c_void_p)


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


class SitkToTensor(object):
    """lib/transforms.py:71-92: image -> float32 clamped to [0, 1] with a leading channel axis (1 x D x H x W), segmentation ->
    uint8 (D x H x W).  The clamp compares in the source dtype and then casts, like the numpy original; one kernel, on the device."""

    def __call__(self, sample):
        img = _to_device_array(sample['image'])
        if img.dtype not in _DTYPE_CODE:
            img = img.to(torch.float64 if img.dtype.is_floating_point else torch.int32)
        out = torch.empty(img.shape, dtype=torch.float32, device=img.device)
        call('da_clamp01_to_f32', ptr(img), _DTYPE_CODE[img.dtype], ptr(out), img.numel(), stream())
        sample['image'] = out.unsqueeze(0)
        if 'segmentation' in sample.keys():
            sample['segmentation'] = _to_device_array(sample['segmentation']).to(torch.uint8)
        return sample


class CropTensor(object):
    """lib/transforms.py:124-158: crop_size [z, y, x] (both sides) or [z_lo, y_lo, x_lo, z_hi, y_hi, x_hi] voxels off a C x D x H x W
    image and its D x H x W segmentation, as fresh contiguous device tensors from one copy kernel."""

    def __init__(self, crop_size):
        crop_size = list(crop_size)
        if len(crop_size) == 3:
            self.crop_size = crop_size + crop_size
        elif len(crop_size) == 6:
            self.crop_size = crop_size
        else:
            raise ValueError("crop size should be of length 3 or 6, but {} is given".format(len(crop_size)))

    def _crop(self, t, lead, size):
        c = self.crop_size
        D, H, W = size
        Do, Ho, Wo = D - c[0] - c[3], H - c[1] - c[4], W - c[2] - c[5]
        t = t.to(_device())                       # host tensors are uploaded: there is no host implementation of this path
        if t.element_size() not in (1, 4):
            t = t.float() if t.dtype.is_floating_point else t.to(torch.int32)
        t = t.contiguous()
        out = torch.empty(tuple(t.shape[:-3]) + (Do, Ho, Wo), dtype=t.dtype, device=t.device)
        call('da_crop3d', ptr(t), ptr(out), t.element_size(), lead, D, H, W, c[0], c[1], c[2], Do, Ho, Wo, stream())
        return out

    def __call__(self, sample):
        img = sample['image']
        size = tuple(img.shape[1:4])
        sample['image'] = self._crop(img, img.shape[0], size)
        if 'segmentation' in sample.keys():
            sample['segmentation'] = self._crop(sample['segmentation'], 1, size)
        return sample


class Partition(object):
    """lib/transforms.py:508-649: overlap-tile strategy.  tile_size / overlap_size are given in SimpleITK order (x, y, z) and
    flipped to numpy order, as in the reference.  __call__ produces the N x 1 x tz x ty x tx tiles from the reflect-padded volume
    without materialising the padding; assemble() puts predicted tiles back (core copy, or per-voxel majority vote)."""

    def __init__(self, tile_size, overlap_size, padding_mode='reflect', mode="pred"):
        self.tile_size = np.flipud(np.asarray(tile_size))
        self.overlap_size = np.flipud(np.asarray(overlap_size))
        if padding_mode != 'reflect':
            raise NotImplementedError("device Partition implements numpy.pad mode 'reflect' (the reference's default)")
        self.padding_mode = padding_mode
        self.mode = mode

    def _geom(self, shape):
        self.image_size = np.array(shape)
        self.effective_size = self.tile_size - self.overlap_size * 2
        self.tiles_grid_size = np.ceil(self.image_size / self.effective_size).astype(int)
        self.padded_size = self.effective_size * self.tiles_grid_size + self.overlap_size * 2 - self.image_size

    def _tiles(self, vol):
        D, H, W = vol.shape
        n = int(np.prod(self.tiles_grid_size))
        tiles = torch.empty((n,) + tuple(int(v) for v in self.tile_size), dtype=vol.dtype, device=vol.device)
        (ka, ta), (kb, ov) = _int3(self.tile_size), _int3(self.overlap_size)
        call('da_partition_tiles', ptr(vol), ptr(tiles), vol.element_size(), D, H, W, ta, ov, stream())
        return tiles

    def __call__(self, sample):
        image = _to_device_array(sample['image'])
        if image.element_size() != 4:
            image = image.float()
        self.image = sample['image']
        self.name = sample.get('name')
        self._geom(tuple(image.shape))
        sample['image'] = self._tiles(image).unsqueeze(1)
        seg = _to_device_array(sample['segmentation'])
        if self.mode == 'pred':
            sample['segmentation'] = seg.unsqueeze(0)
        else:
            if seg.element_size() not in (1, 4):
                seg = seg.to(torch.uint8)
            sample['segmentation'] = self._tiles(seg).unsqueeze(1)
        return sample

    def assemble(self, tiles, is_vote=False, if_itk=False, crop_size=None, data_type=None):
        """tiles: N x tz x ty x tx (device).  Returns a device tensor D x H x W (the reference returns numpy float64 / uint8, or a
        SimpleITK image when if_itk, which needs the host library)."""
        if if_itk:
            raise NotImplementedError('if_itk=True needs SimpleITK on the host; call with if_itk=False and wrap the result there')
        t = tiles.to(_device())
        if is_vote:
            t = t.to(torch.uint8)
        elif t.element_size() not in (1, 4):
            t = t.float()
        t = t.contiguous()
        D, H, W = (int(v) for v in self.image_size)
        out = torch.empty((D, H, W), dtype=t.dtype, device=t.device)
        (ka, ta), (kb, ov) = _int3(self.tile_size), _int3(self.overlap_size)
        call('da_assemble_tiles', ptr(t), ptr(out), t.element_size(), D, H, W, ta, ov, 1 if is_vote else 0, stream())
        if data_type:
            out = out.to(data_type if isinstance(data_type, torch.dtype) else torch.from_numpy(np.zeros(1, dtype=data_type)).dtype)
        if crop_size:                                                   # zero a border of crop_size (x, y, z), :633-637
            keep = torch.zeros_like(out)
            sl = (slice(crop_size[2], -crop_size[2]), slice(crop_size[0], -crop_size[0]), slice(crop_size[1], -crop_size[1]))
            keep[sl] = out[sl]
            out = keep
        return out

"""Synthetic volume datasets for the BASELINE configs (the reference's NIfTI list-file datasets,
lib/datasets.py:16-478, are host file I/O and out of scope).

Samples follow the reference's tuple convention (image 1 x D x H x W float in [0,1], segmentation D x H x W uint8,
name) (lib/datasets.py:150-166), so SegmentationExperiment consumes them unchanged.
"""
import torch
from torch.utils.data import Dataset


def structured_labels(shape, n_classes, seed=0):
    """Blocky label map, a function of coordinates (Dice-parity inputs must be structured)."""
    D, H, W = shape
    z = torch.arange(D).view(D, 1, 1)
    y = torch.arange(H).view(1, H, 1)
    x = torch.arange(W).view(1, 1, W)
    bz, by, bx = max(D // 8, 1), max(H // 8, 1), max(W // 8, 1)
    lab = ((z // bz) * 5 + (y // by) * 3 + (x // bx) + seed) % n_classes
    return lab.to(torch.uint8)


class SyntheticSegDataset(Dataset):
    def __init__(self, n_samples, shape, n_classes, seed=230, noise=0.1):
        self.n, self.shape, self.n_classes, self.seed, self.noise = n_samples, tuple(shape), n_classes, seed, noise

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed + i)
        lab = structured_labels(self.shape, self.n_classes, seed=i)
        img = lab.float() / max(self.n_classes - 1, 1) + self.noise * torch.rand(self.shape, generator=g)
        img = img.clamp_(0, 1).unsqueeze(0)
        return img, lab, 'synthetic_%d' % i


class SyntheticPairDataset(Dataset):
    """(moving image, target image, moving seg, target seg) pairs for the registration / joint configs."""

    def __init__(self, n_pairs, shape, n_classes, seed=230):
        self.seg = SyntheticSegDataset(n_pairs + 1, shape, n_classes, seed)
        self.n = n_pairs

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        im, sm, _ = self.seg[i]
        it, st_, _ = self.seg[i + 1]
        return im, it, sm, st_


def synthetic_batch_on_device(n, shape, n_classes, seed=230, device='cuda', structured=False, noise=0.1, sample0=0):
    """synthetic volumes generated in HBM by the HIP kernel da_synth_volume (no host tensor, no PCIe copy).
    Returns (image n x 1 x D x H x W float32 in [0,1], labels n x D x H x W uint8).  structured=False: iid throughput inputs
    (SURVEY.md 8d); structured=True: the blocky label / noisy image volumes of SyntheticSegDataset's kind (Dice-parity inputs)."""
    from .. import _native as nat
    D, H, W = (int(s) for s in shape)
    img = torch.empty((n, 1, D, H, W), dtype=torch.float32, device=device)
    lab = torch.empty((n, D, H, W), dtype=torch.uint8, device=device)
    nat.require_cuda(img)
    with torch.cuda.device(img.device):
        nat.call('da_synth_volume', nat.ptr(img), nat.ptr(lab), n, D, H, W, int(n_classes), 1 if structured else 0, float(noise),
                 int(seed) & 0xffffffff, int(sample0), nat.stream())
    return img, lab


def get_seg_dataset(name):
    if name == 'synthetic':
        return SyntheticSegDataset
    raise KeyError("dataset '%s': only 'synthetic' is available (NIfTI datasets need SimpleITK, out of scope)" % name)

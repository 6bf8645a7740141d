"""TEST INFRASTRUCTURE ONLY.  numpy restatement of the tensor halves of lib/transforms.py for SURVEY.md row f4: SitkToTensor
(:71-92), CropTensor (:124-158), Partition.__call__ / assemble (:508-649).  Pinned by tests/golden/datapath.npz, which
oracle/make_golden.py generates from the reference itself."""
import numpy as np


def sitk_to_tensor(img_np, seg_np=None):
    """:77-90: clamp in the source dtype, cast to float32, add the channel axis; segmentation -> uint8."""
    a = np.array(img_np, copy=True)
    a[np.where(a > 1.0)] = 1.0
    a[np.where(a < 0.0)] = 0.0
    return np.float32(a)[None], (np.uint8(seg_np) if seg_np is not None else None)


def crop_tensor(img, seg, crop_size):
    """:136-158."""
    c = list(crop_size) * 2 if len(crop_size) == 3 else list(crop_size)
    s = img.shape
    return (img[:, c[0]:s[1] - c[3], c[1]:s[2] - c[4], c[2]:s[3] - c[5]],
            seg[c[0]:s[1] - c[3], c[1]:s[2] - c[4], c[2]:s[3] - c[5]] if seg is not None else None)


class Partition:
    def __init__(self, tile_size, overlap_size):
        self.tile = np.flipud(np.asarray(tile_size)); self.ov = np.flipud(np.asarray(overlap_size))      # :525-526

    def tiles(self, vol):
        """:541-577 for one volume (numpy 'reflect' padding)."""
        self.size = np.array(vol.shape)
        self.eff = self.tile - 2 * self.ov
        self.grid = np.ceil(self.size / self.eff).astype(int)
        extra = self.eff * self.grid + 2 * self.ov - self.size
        padded = np.pad(vol, [(self.ov[a], extra[a] - self.ov[a]) for a in range(3)], mode='reflect')
        out = []
        for i in range(self.grid[0]):
            for j in range(self.grid[1]):
                for k in range(self.grid[2]):
                    z, y, x = i * self.eff[0], j * self.eff[1], k * self.eff[2]
                    out.append(padded[z:z + self.tile[0], y:y + self.tile[1], x:x + self.tile[2]])
        return np.stack(out, 0)

    def assemble(self, tiles, is_vote=False):
        """:589-631."""
        if is_vote:
            labels = np.unique(tiles)
            votes = np.zeros((labels.size,) + tuple(self.eff * self.grid + 2 * self.ov), dtype=int)
            for i in range(self.grid[0]):
                for j in range(self.grid[1]):
                    for k in range(self.grid[2]):
                        ind = (i * self.grid[1] + j) * self.grid[2] + k
                        z, y, x = i * self.eff[0], j * self.eff[1], k * self.eff[2]
                        for lab in labels:
                            votes[lab, z:z + self.tile[0], y:y + self.tile[1], x:x + self.tile[2]] += (tiles[ind] == lab)
            return np.argmax(votes, 0)[self.ov[0]:self.ov[0] + self.size[0], self.ov[1]:self.ov[1] + self.size[1],
                                       self.ov[2]:self.ov[2] + self.size[2]].astype(np.uint8)
        out = np.zeros(self.eff * self.grid)
        for i in range(self.grid[0]):
            for j in range(self.grid[1]):
                for k in range(self.grid[2]):
                    ind = (i * self.grid[1] + j) * self.grid[2] + k
                    out[i * self.eff[0]:(i + 1) * self.eff[0], j * self.eff[1]:(j + 1) * self.eff[1], k * self.eff[2]:(k + 1) * self.eff[2]] = \
                        tiles[ind][self.ov[0]:self.tile[0] - self.ov[0], self.ov[1]:self.tile[1] - self.ov[1], self.ov[2]:self.tile[2] - self.ov[2]]
        return out[:self.size[0], :self.size[1], :self.size[2]]


# ---- synthetic volumes (restates deepatlas_amd/csrc/datapath.hip synth_volume_kernel; the reference has no synthetic data: the
#      sample convention (image [D][H][W] fp32 in [0,1], segmentation uint8) is lib/datasets.py:150-166) ----------------------------
def _mix32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d); x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b); x ^= x >> np.uint32(16)
    return x


def _synth_bits(seed, ctr, stream_id):
    hi = (ctr >> np.uint64(32)).astype(np.uint32)
    lo = (ctr & np.uint64(0xffffffff)).astype(np.uint32)
    with np.errstate(over='ignore'):
        return _mix32(_mix32(lo ^ _mix32(np.uint32(seed) + np.uint32(0x9e3779b9) * hi)) + np.uint32(stream_id))


def synth_volume(N, shape, n_classes, mode, noise, seed, sample0=0):
    """-> (img [N][D][H][W] float32, labels [N][D][H][W] uint8), bit-equal to da_synth_volume."""
    D, H, W = shape
    V = D * H * W
    n = np.arange(N, dtype=np.uint64)[:, None]
    v = np.arange(V, dtype=np.uint64)[None, :]
    ctr = (np.uint64(sample0) + n) * np.uint64(V) + v
    u = (_synth_bits(seed, ctr, 0) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    if mode == 0:
        lab = (_synth_bits(seed, ctr, 1) % np.uint32(n_classes)).astype(np.uint8)
        return u.reshape(N, D, H, W), lab.reshape(N, D, H, W)
    bz, by, bx = max(D // 8, 1), max(H // 8, 1), max(W // 8, 1)
    z, y, x = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing='ij')
    base = (z // bz) * 5 + (y // by) * 3 + (x // bx)
    lab = np.stack([(base + sample0 + k) % n_classes for k in range(N)], 0)
    q = lab.astype(np.float32) / np.float32(max(n_classes - 1, 1))
    img = np.clip(q + np.float32(noise) * u.reshape(N, D, H, W), np.float32(0), np.float32(1)).astype(np.float32)
    return img, lab.astype(np.uint8)

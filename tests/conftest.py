import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(autouse=True)
def _reset_process_wide_switches():
    """SegmentationExperiment / bench enable process-wide modes (side-stream weight gradients, bf16 matrix mode, deterministic warp):
    every test starts from the defaults."""
    yield
    mod = sys.modules.get('deepatlas_amd.ops')
    if mod is not None:
        mod.enable_async_wgrad(False)
        mod.set_deterministic(False)
        mod.set_matrix_precision('fp32')          # experiments switch to 'fp32_split'; the library default is the fp32 matrix instructions
        mod.set_activation_storage('fp32')
        mod.BF16_FORCE_BRIDGE = False


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + '.npz')))
        return cache[name]
    return load


def summary_of(t, nsample=512):
    """Same (sum, abs-sum, l2, n, stride, strided sample) summary oracle/make_golden.py stores."""
    import torch
    t = t.detach().double().reshape(-1).cpu()
    n = t.numel()
    stride = max(1, n // nsample)
    idx = torch.arange(0, n, stride)[:nsample]
    return np.concatenate([[t.sum().item(), t.abs().sum().item(), t.norm().item(), float(n), float(stride)], t[idx].numpy()])


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def max_abs_rel(a, b):
    """Element-wise criterion next to rel_l2: the largest single-element error, in units of the reference tensor's largest magnitude.
    A handful of wrong boundary voxels in a large tensor moves this, not the l2 norm."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)) if a.size else 0.0

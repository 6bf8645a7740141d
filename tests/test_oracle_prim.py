"""oracle/prim.c (plain C, double accumulation) against torch-CPU and against the fixtures generated from the reference.

The torch-CPU oracle (oracle/nets.py) leans on ATen for the primitive ops; this pins an ATen-independent restatement of those
primitives to the same numbers, so a GPU kernel that agrees with the oracle agrees with two independent CPU implementations."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'oracle', 'prim.c')
SO = os.path.join(ROOT, 'oracle', 'libprim.so')

P, I, LL, DBL = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_double


@pytest.fixture(scope='module')
def prim():
    if not os.path.isfile(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-std=c99', SRC, '-lm', '-o', SO])
    lib = ctypes.CDLL(SO)
    lib.prim_conv3d_k3.argtypes = [P, P, P, P, I, I, I, I, I, I, I]
    lib.prim_deconv_k2s2.argtypes = [P, P, P, P, I, I, I, I, I, I]
    lib.prim_bn_train_act.argtypes = [P, P, P, P, P, P, I, I, LL, DBL, DBL]
    lib.prim_maxpool2.argtypes = [P, P, P, I, I, I, I]
    lib.prim_upsample_nearest.argtypes = [P, P, I, I, I, I, I, I, I]
    lib.prim_grid_sample3d.argtypes = [P, P, P, I, I, I, I, I, I, I, I]
    lib.prim_softmax_c.argtypes = [P, P, I, I, LL]
    for f in ('prim_conv3d_k3', 'prim_deconv_k2s2', 'prim_bn_train_act', 'prim_maxpool2', 'prim_upsample_nearest',
              'prim_grid_sample3d', 'prim_softmax_c'):
        getattr(lib, f).restype = None
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _rand(shape, seed, lo=-1.0, hi=1.0):
    return np.random.default_rng(seed).uniform(lo, hi, size=shape).astype(np.float32)


@pytest.mark.parametrize('stride,shape', [(1, (2, 3, 5, 6, 7)), (2, (1, 2, 7, 8, 9)), (2, (1, 4, 6, 6, 8))])
def test_conv3d_k3(prim, stride, shape):
    N, Cin, D, H, W = shape
    Cout = 5
    x, w, b = _rand(shape, 1), _rand((Cout, Cin, 3, 3, 3), 2, -0.3, 0.3), _rand((Cout,), 3)
    ref = F.conv3d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=stride, padding=1)
    y = np.empty(tuple(ref.shape), np.float32)
    prim.prim_conv3d_k3(_p(x), _p(w), _p(b), _p(y), N, Cin, D, H, W, Cout, stride)
    np.testing.assert_allclose(y, ref.float().numpy(), rtol=1e-6, atol=1e-6)


def test_deconv_k2s2(prim):
    N, Cin, D, H, W, Cout = 2, 6, 3, 4, 5, 4
    x, w, b = _rand((N, Cin, D, H, W), 4), _rand((Cin, Cout, 2, 2, 2), 5, -0.3, 0.3), _rand((Cout,), 6)
    ref = F.conv_transpose3d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=2)
    y = np.empty(tuple(ref.shape), np.float32)
    prim.prim_deconv_k2s2(_p(x), _p(w), _p(b), _p(y), N, Cin, D, H, W, Cout)
    np.testing.assert_allclose(y, ref.float().numpy(), rtol=1e-6, atol=1e-6)


def test_bn_train_leaky(prim):
    N, C, D, H, W = 2, 4, 3, 5, 6
    x, g, b = _rand((N, C, D, H, W), 7, -2, 3), _rand((C,), 8, 0.5, 1.5), _rand((C,), 9)
    xt = torch.from_numpy(x).double()
    ref = F.leaky_relu(F.batch_norm(xt, None, None, torch.from_numpy(g).double(), torch.from_numpy(b).double(), True, 0.1, 1e-5), 0.01)
    y = np.empty_like(x)
    mean, var = np.empty(C, np.float64), np.empty(C, np.float64)
    prim.prim_bn_train_act(_p(x), _p(g), _p(b), _p(y), _p(mean), _p(var), N, C, D * H * W, 1e-5, 0.01)
    np.testing.assert_allclose(y, ref.float().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(mean, xt.mean(dim=(0, 2, 3, 4)).numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(var, xt.var(dim=(0, 2, 3, 4), unbiased=False).numpy(), rtol=1e-10)


def test_maxpool_golden_and_ties(prim, golden):
    g = golden('ops')
    x = np.ascontiguousarray(g['ops/maxpool/in'])
    N, C, D, H, W = x.shape
    y = np.empty((N, C, D // 2, H // 2, W // 2), np.float32)
    idx = np.empty(y.shape, np.int64)
    prim.prim_maxpool2(_p(x), _p(y), _p(idx), N * C, D, H, W)
    np.testing.assert_array_equal(y, g['ops/maxpool/out'])
    # backward through the recorded first-maximum index == the reference's gradient (ties included in the fixture)
    grad = np.zeros((N * C, D * H * W), np.float32)
    gout = np.ones(y.shape, np.float32).reshape(N * C, -1)
    for nc in range(N * C):
        np.add.at(grad[nc], idx.reshape(N * C, -1)[nc], gout[nc])
    assert np.array_equal(grad.reshape(x.shape) != 0, g['ops/maxpool/grad'] != 0)
    # odd sizes: floor output, against torch
    x2 = _rand((1, 3, 5, 7, 9), 10)
    ref = F.max_pool3d(torch.from_numpy(x2), 2).numpy()
    y2 = np.empty(ref.shape, np.float32)
    prim.prim_maxpool2(_p(x2), _p(y2), None, 3, 5, 7, 9)
    np.testing.assert_array_equal(y2, ref)


def test_nearest_golden(prim, golden):
    g = golden('ops')
    x = np.ascontiguousarray(g['ops/nearest/in'])
    N, C, D, H, W = x.shape
    for key, size in (('ops/nearest/out_3_5_10', (3, 5, 10)), ('ops/nearest/out_4_6_10', (4, 6, 10))):
        y = np.empty((N, C) + size, np.float32)
        prim.prim_upsample_nearest(_p(x), _p(y), N * C, D, H, W, *size)
        np.testing.assert_array_equal(y, g[key])


@pytest.mark.parametrize('name', ['warp1', 'warpC'])
def test_grid_sample_golden(prim, golden, name):
    g = golden('ops')
    src = np.ascontiguousarray(g['ops/%s/src' % name])
    disp = g['ops/%s/disp' % name]
    ident = g['ops/identity']                                   # 3 x D x H x W, channel 0 = x (W axis)
    N, C, D, H, W = src.shape
    grid = np.ascontiguousarray(np.moveaxis(disp + ident[None], 1, -1)).astype(np.float32)      # N x D x H x W x 3
    out = np.empty_like(src)
    prim.prim_grid_sample3d(_p(src), _p(grid), _p(out), N, C, D, H, W, D, H, W)
    np.testing.assert_allclose(out, g['ops/%s/out' % name], rtol=1e-5, atol=2e-6)


def test_softmax(prim):
    x = _rand((2, 5, 60), 11, -4, 4)
    y = np.empty_like(x)
    prim.prim_softmax_c(_p(x), _p(y), 2, 5, 60)
    np.testing.assert_allclose(y, F.softmax(torch.from_numpy(x).double(), dim=1).float().numpy(), rtol=1e-6, atol=1e-7)


def test_oracle_unet_block_matches_prim(prim):
    """conv -> BN(train) -> LeakyReLU of the torch oracle == the same chain in plain C (unets.py:30-32)."""
    N, Cin, Cout, D, H, W = 1, 3, 4, 4, 6, 8
    x, w, b = _rand((N, Cin, D, H, W), 12), _rand((Cout, Cin, 3, 3, 3), 13, -0.3, 0.3), _rand((Cout,), 14)
    gm, bt = _rand((Cout,), 15, 0.5, 1.5), _rand((Cout,), 16)
    t = F.conv3d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), padding=1)
    t = F.leaky_relu(F.batch_norm(t, None, None, torch.from_numpy(gm), torch.from_numpy(bt), True, 0.1, 1e-5), 0.01)
    y = np.empty((N, Cout, D, H, W), np.float32)
    z = np.empty_like(y)
    prim.prim_conv3d_k3(_p(x), _p(w), _p(b), _p(y), N, Cin, D, H, W, Cout, 1)
    prim.prim_bn_train_act(_p(y), _p(gm), _p(bt), _p(z), None, None, N, Cout, D * H * W, 1e-5, 0.01)
    np.testing.assert_allclose(z, t.numpy(), rtol=2e-5, atol=2e-5)

"""CPU: host-side logic -- registries, state_dict parity with the oracle's key list (= the reference's, pinned in
make_golden.py), C-ABI library loads and exports every declared symbol, loud failure without a GPU."""
import ctypes
import os
import re
import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    lib_path = ge.build()
    lib = ctypes.CDLL(lib_path)
    header = open(os.path.join(ROOT, 'include', 'deepatlas_hip.h')).read()
    names = sorted(set(re.findall(r'\b(da_[a-z0-9_]+)\s*\(', header)))
    assert len(names) > 40
    for n in names:
        assert hasattr(lib, n), n
    from deepatlas_amd import _native
    assert set(names) == set(_native.SIGNATURES.keys())
    assert lib.da_version() == 100


def test_registries_match_reference_names():
    from deepatlas_amd.lib.network_factory import get_network, get_available_networks
    from deepatlas_amd.lib.loss import get_loss_function, get_available_losses
    assert get_available_networks() == ('voxel_morph_cvpr', 'UNet', 'UNet_light')
    assert list(get_available_losses()) == ['ncc', 'lncc', 'mse', 'gradient', 'bendingEnergy', 'dice', 'L2', 'focal',
                                            'cross_entropy', 'soft_cross_entropy']
    with pytest.raises(KeyError):
        get_network('nope')
    with pytest.raises(KeyError):
        get_loss_function('nope')


def test_state_dict_keys_and_shapes_match_reference():
    from oracle import nets
    from deepatlas_amd.lib.network_factory import get_network
    m = get_network('UNet_light')(in_channel=1, n_classes=32, bias=True, BN=True)
    ref = nets.unet_param_shapes(1, 32, nets.UNET_LIGHT['encoders'], nets.UNET_LIGHT['decoders'])
    sd = m.state_dict()
    assert list(sd.keys()) != [] and set(sd.keys()) == set(ref.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(ref[k]), k
    assert sum(p.numel() for p in m.parameters()) == 874864
    r = get_network('voxel_morph_cvpr')()
    rref = nets.voxelmorph_param_shapes()
    assert set(r.state_dict().keys()) == set(rref.keys())
    assert sum(p.numel() for p in r.parameters()) == 253627
    # strict load of a reference-shaped state dict
    m.load_state_dict(nets.closed_form_fill(ref, seed=1), strict=True)
    m.weights_init()
    assert float(m.state_dict()['encoders.0.0.conv.bias'].abs().max()) == 0.0


def test_no_cpu_fallback():
    from deepatlas_amd.lib.network_factory import get_network
    from deepatlas_amd import _native
    m = get_network('UNet_light')(in_channel=1, n_classes=32, bias=True, BN=True)
    with pytest.raises(_native.NativeError):
        m(torch.zeros(1, 1, 8, 8, 8))


def test_out_of_scope_constructors_raise():
    from deepatlas_amd.lib.network_factory import get_network
    from deepatlas_amd.lib.loss import get_loss_function
    from deepatlas_amd.lib.network_factory import unets
    # generator options are on the path: strided-conv down-samplers carry parameters, trilinear up-samplers none
    g = unets.UNet_generator(encoders=[(16, 16), (16, 16, 16)], decoders=[(16, 16, 16)], maxpool=False, upsample=True)(1, 16, bias=True, BN=False)
    assert 'down_samplers.0.weight' in g.state_dict() and not any(k.startswith('up_samplers') for k in g.state_dict())
    m = get_network('UNet')(1, 2, bias=True, BN=True)     # the fixed UNet is on the path: constructible on the host
    assert 'dc8.1.running_mean' in m.state_dict() and tuple(m.state_dict()['dc8.0.weight'].shape) == (768, 256, 3, 3, 3)
    # the whole registry is constructible (round 2: 'focal' / 'cross_entropy' / 'soft_cross_entropy' run on xent.hip)
    assert get_loss_function('focal')(5).class_num == 5 and get_loss_function('cross_entropy')().ignore_index == -100
    assert get_loss_function('soft_cross_entropy')(n_class=5, softmax=True).softmax is True
    assert get_loss_function('bendingEnergy')(norm='L1').norm == 'L1'
    with pytest.raises(NotImplementedError):
        get_loss_function('cross_entropy')(label_smoothing=0.1)
    # rows f1/f2 are on the accelerated path: constructible on the host, state_dict as the reference's
    assert list(get_loss_function('lncc')().state_dict().keys()) == ['filter']
    assert get_loss_function('gradient')(spacing=(1, 2, 4)).spacing.tolist() == [1.0, 2.0, 4.0]


def test_train_seg_config_matches_reference_dict():
    import argparse
    import train_seg
    ns = argparse.Namespace(device='0', debug=False, preload=False, num_samples=21, num_epochs=100, lr=1e-3, test_only=False,
                            data_root='./data', log_root='./logs', shape=[64, 64, 64])
    c = train_seg.build_config(ns)
    assert c['model'] == 'UNet_light' and c['random_seed'] == 230 and c['batch_size'] == 1
    assert c['model_settings'] == {'in_channel': 1, 'n_classes': 32, 'bias': True, 'BN': True}
    assert c['loss_settings'] == {'n_class': 32, 'weight_type': 'Uniform', 'no_bg': False, 'softmax': True, 'eps': 1e-6}
    assert c['lr_mode'] == 'multiStep' and c['milestones'] == [0.5, 1] and c['gamma'] == 0.2
    assert c['samples_per_epoch'] == 42 and c['crop_size'] == [0, 10, 7, 14, 8, 7]


def test_matrix_precision_switch_round_trip():
    """ops.set_matrix_precision: process-wide switch in the C library (no GPU needed to flip it); unknown modes raise."""
    from deepatlas_amd import ops
    prev = ops.set_matrix_precision('bf16')
    try:
        assert prev in ops.MATRIX_MODES
        assert ops.set_matrix_precision('fp32') == 'bf16'
        assert ops.set_matrix_precision('fp32_split') == 'fp32'            # the exact three-way split (da_set_matrix_mode(2))
        assert ops.set_matrix_precision('fp32') == 'fp32_split'
        assert ops.set_matrix_precision('fp32') == 'fp32'
        with pytest.raises(ValueError):
            ops.set_matrix_precision('fp16')
        from deepatlas_amd._native import lib
        assert lib().da_set_matrix_mode(7) == -1 and ops.set_matrix_precision('fp32') == 'fp32'      # unknown mode refused, nothing changed
        assert lib().da_set_matrix_bf16(1) == 0 and lib().da_set_matrix_mode(0) == 1                  # the older entry is mode 1
    finally:
        ops.set_matrix_precision(prev)


def test_c_abi_rejects_bad_arguments_before_touching_the_device():
    """Empty / null / undersized arguments are refused with DA_ERR_BADARG (-1) / DA_ERR_WS_SMALL (-2) by the entry points themselves
    (no HIP call has happened yet, so this runs without a GPU): empty batch, empty volume, missing second input, scratch too small."""
    from ctypes import c_void_p, c_int, byref
    from deepatlas_amd import _native
    L = _native.lib()
    fake = c_void_p(0x1000)          # never dereferenced on the host
    BAD, SMALL = -1, -2
    assert L.da_conv3d_k3_fwd(fake, 16, None, 0, fake, None, fake, 0, 8, 8, 8, 16, 1, -1.0, fake, 1 << 20, None) == BAD          # N = 0
    assert L.da_conv3d_k3_fwd(fake, 16, None, 0, fake, None, fake, 1, 0, 8, 8, 16, 1, -1.0, fake, 1 << 20, None) == BAD          # D = 0
    assert L.da_conv3d_k3_fwd(None, 16, None, 0, fake, None, fake, 1, 8, 8, 8, 16, 1, -1.0, fake, 1 << 20, None) == BAD          # null input
    assert L.da_conv3d_k3_fwd(fake, 16, None, 8, fake, None, fake, 1, 8, 8, 8, 16, 1, -1.0, fake, 1 << 20, None) == BAD          # C2 > 0 without in2
    assert L.da_conv3d_k3_fwd(fake, 16, None, 0, fake, None, fake, 1, 8, 8, 8, 16, 3, -1.0, fake, 1 << 20, None) == BAD          # stride 3
    n = c_int(5)
    assert L.da_conv3d_k3_fwd_pro(fake, 16, fake, None, 0.01, None, 0, None, None, -1.0, fake, None, fake, 1, 8, 8, 8, 16, -1.0,
                                  None, 0, byref(n), fake, 1 << 20, None) == BAD and n.value == 0                                  # scale without shift
    need = L.da_conv3d_k3_ws_bytes(1, 8, 8, 8, 16, 16, 1)
    assert need > 0
    assert L.da_conv3d_k3_wgrad(fake, 16, None, 0, fake, fake, None, 1, 8, 8, 8, 16, 1, fake, need - 1, None) == SMALL
    assert L.da_conv3d_k3_wgrad(fake, 16, None, 0, None, fake, None, 1, 8, 8, 8, 16, 1, fake, need, None) == BAD                  # null dy
    assert L.da_deconv_k2s2_fwd_bnstats(fake, fake, None, None, 1, 4, 4, 4, 16, 16, None, 0, byref(n), fake, 1 << 20, None) == BAD   # null output
    assert L.da_conv1x1_fwd_pro(fake, None, None, 0.01, fake, None, fake, 100, 16, 32, fake, 1 << 20, None) == BAD                # prologue entry without a prologue
    assert L.da_bn_act_fwd(None, fake, fake, 0.01, fake, 10, 16, None) == BAD
    assert L.da_maxpool2_fwd(fake, fake, 0, 8, 8, 8, 16, None) == BAD


def test_three_way_bf16_split_is_exact_and_six_products_are_fp32_accurate():
    """The arithmetic behind the split matrix mode, restated in numpy: x = h + m + l EXACTLY with h = bf16(x), m = bf16(x - h),
    l = bf16(x - h - m) (round-to-nearest-even), for normal, huge and wide-dynamic-range fp32 values (|x| >= 2^-100; absolute error <= 2^-133 below); and the six partial products
    h h' + h m' + m h' + m m' + h l' + l h' reproduce x y to better than one fp32 rounding (the dropped terms are <= 2^-23 |x y|)."""
    import numpy as np

    def bf16(a):                                  # fp32 -> bf16 (round to nearest even) -> fp32
        u = np.asarray(a, np.float32).view(np.uint32).astype(np.uint64)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        return r.astype(np.uint32).view(np.float32)

    rng = np.random.default_rng(7)
    x = np.concatenate([rng.standard_normal(200000), rng.standard_normal(50000) * np.exp(8 * rng.standard_normal(50000)),
                        [1.0, -1.0, 3.0e38, -3.0e38, 1.2e-38, 2.0 ** -100, 1.0 + 2.0 ** -23, 255.99998]]).astype(np.float32)
    h = bf16(x)
    r1 = (x - h).astype(np.float32)               # exact: |r1| <= ulp_bf16(x) / 2 has at most 16 significant bits
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - h.astype(np.float64))
    m = bf16(r1)
    r2 = (r1 - m).astype(np.float32)
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - m.astype(np.float64))
    l = bf16(r2)
    big = np.abs(x) >= 2.0 ** -100                # below ~2^-109 the last term enters bf16's denormal range (absolute error <= 2^-133 there)
    assert np.array_equal(l[big], r2[big]), 'the second remainder has at most 8 significant bits'
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64))[big], x.astype(np.float64)[big])
    assert np.all(np.abs(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64) - x.astype(np.float64)) <= 2.0 ** -133)
    assert np.all(np.abs(m) <= np.abs(x) * 2.0 ** -8) and np.all(np.abs(l)[big] <= np.abs(x)[big] * 2.0 ** -16)
    # six of the nine partial products against the exact product (float64 holds both exactly)
    y = rng.permutation(x)
    hy = bf16(y); my = bf16((y - hy).astype(np.float32)); ly = ((y - hy).astype(np.float32) - my).astype(np.float32)
    H, M, L, Hy, My, Ly = [a.astype(np.float64) for a in (h, m, l, hy, my, ly)]
    six = H * Hy + H * My + M * Hy + M * My + H * Ly + L * Hy
    exact = x.astype(np.float64) * y.astype(np.float64)
    ok = np.isfinite(exact) & (np.abs(x) >= 2.0 ** -60) & (np.abs(y) >= 2.0 ** -60) & (np.abs(exact) < 1e300)
    rel = np.abs(six - exact)[ok] / np.abs(exact)[ok]
    assert rel.max() <= 2.0 ** -23 and np.sqrt(np.mean(rel ** 2)) < 2.0 ** -26, (rel.max(), np.sqrt(np.mean(rel ** 2)))
    with np.errstate(over='ignore'):
        fp32_rounding = np.abs((x * y).astype(np.float64) - exact)[ok] / np.abs(exact)[ok]      # one fp32 multiply for comparison
    assert np.sqrt(np.mean(rel ** 2)) < np.sqrt(np.mean(fp32_rounding[np.isfinite(fp32_rounding)] ** 2))


def test_asan_build_recipe_and_roctx_ranges(tmp_path):
    """SURVEY.md section 5 build notes: the -fsanitize=address variant compiles (one small source: host + device instrumentation for
    gfx950:xnack+), and the roctx range helper is inert when disabled and pushes / pops through the ROCm roctx library when enabled."""
    import subprocess
    import __graft_entry__ as g
    lib = g.build_asan(sources=['optim.hip'])
    syms = subprocess.check_output(['nm', '-D', lib]).decode()
    assert 'da_adam_step' in syms and '__asan_init' in syms
    from deepatlas_amd import trace
    prev = trace.enable(False)
    with trace.range('off'):
        pass
    trace.enable(True)
    try:
        if trace.available():
            with trace.range('seg/forward'):
                trace.mark('inside')
    finally:
        trace.enable(prev)

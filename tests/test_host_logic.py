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
        assert ops.set_matrix_precision('fp32_split') == 'fp32'            # the two-term fp16 split (da_set_matrix_mode(2))
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


def test_two_term_fp16_split_bounds_and_sums_are_fp32_accurate():
    """The arithmetic behind the split matrix mode (deepatlas_amd/csrc/split_f16.h), restated in numpy.  A tile is scaled by a
    power of two so that its largest magnitude lies in [2^14, 2^15); h = fp16(x s), l = fp16(x s - h) (round-to-nearest-even, the subtraction
    exact in fp32).  Pinned here: (1) |x s - h - l| <= 2^-22 |x s| for every element within 2^-18 of the tile maximum and <= 2^-25 (absolute,
    scaled units: 2^-39 of the maximum) below; (2) the three products h h' + h l' + l h' are within 2^-21 + 2^-22 of x y, typically at the
    size of one fp32 rounding; (3) a K = 432 dot product accumulated the way the matrix cores do (32 products per fp32 rounding, three
    roundings per K-step) is closer to the double-precision value than the fp32 fmaf chain, on uniform and on wide-dynamic-range data."""
    import numpy as np

    def scale_exp(m):                              # da_scale_exp: m 2^e in [2^14, 2^15), clamped to +-100; zero -> +100
        if m == 0:
            return 100
        return int(np.clip(14 - np.floor(np.log2(m)), -100, 100))

    def split2(x, e):
        xs = (x.astype(np.float32) * np.float32(2.0 ** e)).astype(np.float32)       # exact (power of two; no overflow by construction)
        h = xs.astype(np.float16)
        r = (xs - h.astype(np.float32)).astype(np.float32)
        assert np.array_equal(r.astype(np.float64), xs.astype(np.float64) - h.astype(np.float64)), 'the remainder is exact in fp32'
        return xs, h.astype(np.float32), r.astype(np.float16).astype(np.float32)

    rng = np.random.default_rng(7)
    x = np.concatenate([rng.standard_normal(200000), rng.standard_normal(50000) * np.exp(4 * rng.standard_normal(50000)),
                        [1.0, -1.0, 1.0 + 2.0 ** -23, 255.99998, 3.0e-6, 65504.0, 1.0e-30]]).astype(np.float32)
    e = scale_exp(float(np.abs(x).max()))
    xs, h, l = split2(x, e)
    assert np.all(np.isfinite(h)) and np.abs(xs).max() < 2.0 ** 15 and np.abs(xs).max() >= 2.0 ** 14
    err = np.abs(xs.astype(np.float64) - h.astype(np.float64) - l.astype(np.float64))
    window = np.abs(xs) >= 2.0 ** -3               # l stays a normal fp16 (|l| >= 2^-14) for |x s| >= 2^-3, i.e. within 2^-18 of the maximum
    assert np.all(err[window] <= 2.0 ** -22 * np.abs(xs[window]))
    assert np.all(err <= np.maximum(2.0 ** -22 * np.abs(xs), 2.0 ** -25))
    # (2) three of the four partial products against the exact product, both operands inside their windows
    a = rng.uniform(0.25, 2.0, 200000).astype(np.float32) * rng.choice([-1.0, 1.0], 200000).astype(np.float32)
    b = rng.permutation(a)
    ea, eb = scale_exp(float(np.abs(a).max())), scale_exp(float(np.abs(b).max()))
    _, ha, la = split2(a, ea)
    _, hb, lb = split2(b, eb)
    three = (ha.astype(np.float64) * lb + la.astype(np.float64) * hb + ha.astype(np.float64) * hb) * 2.0 ** -(ea + eb)
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(three - exact) / np.abs(exact)
    fp32_rounding = np.abs((a * b).astype(np.float64) - exact) / np.abs(exact)      # one fp32 multiply for comparison
    assert rel.max() <= 2.0 ** -21 + 2.0 ** -22 and np.sqrt(np.mean(rel ** 2)) < 4 * np.sqrt(np.mean(fp32_rounding ** 2)), (rel.max(), np.sqrt(np.mean(rel ** 2)))
    # (3) K = 432 (16 input channels x 27 taps) dot products: MFMA-style accumulation of the split against the fmaf chain, both against double
    for kind in ('uniform', 'wide'):
        M, K = 400, 432
        if kind == 'uniform':
            xv = rng.uniform(-1, 1, (M, K)).astype(np.float32); wv = rng.uniform(-0.1, 0.1, (M, K)).astype(np.float32)
        else:
            xv = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-12, 0, (M, K)))).astype(np.float32)
            wv = (rng.standard_normal((M, K)) * 0.05 * np.exp(rng.uniform(-6, 0, (M, K)))).astype(np.float32)
        ref = (xv.astype(np.float64) * wv.astype(np.float64)).sum(-1)
        chain = np.zeros(M, np.float32)
        for k in range(K):
            chain = (chain.astype(np.float64) + xv[:, k].astype(np.float64) * wv[:, k].astype(np.float64)).astype(np.float32)      # fmaf: one rounding per product
        sp = np.zeros(M, np.float64)
        for r in range(M):                         # per-row scales stand in for the per-tile scales
            ex, ew = scale_exp(float(np.abs(xv[r]).max())), scale_exp(float(np.abs(wv[r]).max()))
            _, hx, lx = split2(xv[r], ex)
            _, hw, lw = split2(wv[r], ew)
            acc = np.float32(0)
            for k0 in range(0, K, 32):             # one v_mfma_f32_16x16x32_f16 per plane pair: 32 exact products, summed, one fp32 rounding
                for pa, pb in ((hx, lw), (lx, hw), (hx, hw)):
                    acc = np.float32(np.float64(acc) + (pa[k0:k0 + 32].astype(np.float64) * pb[k0:k0 + 32].astype(np.float64)).sum())
            sp[r] = np.float64(acc) * 2.0 ** -(ex + ew)
        den = np.sqrt(np.mean(ref ** 2))
        e_chain, e_split = np.sqrt(np.mean((chain - ref) ** 2)) / den, np.sqrt(np.mean((sp - ref) ** 2)) / den
        assert e_split < e_chain and e_split < 3e-7, (kind, e_chain, e_split)


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

"""GPU parity tests, op level: every C-ABI launcher (called through deepatlas_amd.ops -> ctypes -> HIP) against
(a) the golden vectors produced by the reference itself (tests/golden/ops.npz) and (b) torch-CPU, the reference's
own arithmetic provider, on seeded inputs.  Tolerance: 1e-4 relative fp32 (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2, max_abs_rel

pytestmark = pytest.mark.gpu
TOL = 1e-4


def dev():
    return torch.device('cuda:0')


def T(a):
    return torch.from_numpy(np.asarray(a))


def cl(x):
    """to the device in channels-last-3d physical layout"""
    return x.to(dev()).contiguous(memory_format=torch.channels_last_3d) if x.dim() == 5 else x.to(dev())


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def check(a, b, tol=TOL, what=''):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    e = rel_l2(a, b)
    assert e < tol, '%s rel-l2 %.3e' % (what, e)
    m = max_abs_rel(a, b)                       # element-wise: no single element off by more than `tol` of the tensor's largest magnitude
    assert m < tol, '%s max-abs %.3e of max|ref|' % (what, m)


CONV_CASES = [
    # C1, C2, Cout, stride, (N, D, H, W), slope
    (1, 0, 8, 1, (2, 6, 10, 12), -1.0),        # seg enc0.0 (direct)
    (2, 0, 16, 1, (1, 5, 7, 9), 0.0),          # reg enc0 as a concat-free 2-channel input
    (1, 1, 16, 1, (1, 6, 8, 10), 0.0),         # reg enc0 as two 1-channel pointers (source, target)
    (8, 0, 16, 1, (1, 8, 16, 20), -1.0),       # CK = 8 MFMA path, partial x tile
    (16, 0, 16, 1, (2, 8, 16, 16), 0.01),      # CK = 16 MFMA, exact tiles
    (16, 0, 32, 1, (1, 6, 9, 18), -1.0),       # NREP 2, ragged tiles
    (32, 16, 16, 1, (1, 8, 8, 32), -1.0),      # decoder concat 48 -> 16 (dgrad: 16 -> 32 | 16 split output)
    (64, 32, 32, 1, (1, 4, 8, 16), -1.0),      # 96 -> 32
    (64, 0, 64, 1, (1, 4, 6, 20), -1.0),       # NT = 4 -> two cout groups
    (8, 16, 3, 1, (1, 6, 8, 18), -1.0),        # flow conv 24 -> 3 (direct)
    (64, 0, 8, 1, (1, 4, 8, 16), 0.0),         # reg dec3 64 -> 8
    (16, 0, 32, 2, (1, 9, 10, 11), 0.0),       # stride 2, odd sizes (ceil(n/2))
    (32, 0, 32, 2, (1, 4, 6, 8), 0.0),
    (5, 0, 7, 1, (1, 4, 5, 6), -1.0),          # odd channel counts (generic direct path)
]


@pytest.mark.parametrize('direct', [0, 1])
@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: 'c%d+%d_o%d_s%d' % (c[0], c[1], c[2], c[3]))
def test_conv3d_fwd_bwd(case, direct):
    from deepatlas_amd import ops, _native
    C1, C2, Cout, stride, (N, D, H, W), slope = case
    x1 = rnd((N, C1, D, H, W), 1)
    x2 = rnd((N, C2, D, H, W), 2) if C2 else None
    w = rnd((Cout, C1 + C2, 3, 3, 3), 3, 0.2)
    b = rnd((Cout,), 4, 0.1)
    # reference: torch CPU
    xr1 = x1.clone().requires_grad_(True)
    xr2 = x2.clone().requires_grad_(True) if C2 else None
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    xin = torch.cat((xr1, xr2), 1) if C2 else xr1
    yr = F.conv3d(xin, wr, br, stride=stride, padding=1)
    if slope >= 0:
        yr = F.leaky_relu(yr, slope) if slope > 0 else F.relu(yr)
    go = rnd(tuple(yr.shape), 5)
    yr.backward(go)
    prev = _native.lib().da_set_conv_direct(direct)
    try:
        xg1 = cl(x1).requires_grad_(True)
        xg2 = cl(x2).requires_grad_(True) if C2 else None
        wg, bg = w.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
        yg = ops.Conv3dK3Fn.apply(xg1, xg2, wg, bg, stride, slope)
        yg.backward(cl(go))
        torch.cuda.synchronize()
    finally:
        _native.lib().da_set_conv_direct(prev)
    check(yg, yr, what='fwd')
    check(xg1.grad, xr1.grad, what='dgrad1')
    if C2:
        check(xg2.grad, xr2.grad, what='dgrad2')
    check(wg.grad, wr.grad, what='wgrad')
    check(bg.grad, br.grad, what='bgrad')


def test_conv3d_mfma_asymmetric_identity():
    """A = I style check with an ASYMMETRIC weight pattern: catches row/col swaps in the MFMA C/D mapping."""
    from deepatlas_amd import ops
    N, C, D, H, W = 1, 16, 4, 8, 16
    x = torch.zeros(N, C, D, H, W)
    x[0, 3, 2, 5, 7] = 1.0                      # one hot voxel / channel
    w = torch.zeros(16, 16, 3, 3, 3)
    for co in range(16):
        for ci in range(16):
            w[co, ci] = (co * 16 + ci) * 1e-3 + torch.arange(27).float().view(3, 3, 3) * 1e-5
    y = ops.Conv3dK3Fn.apply(cl(x), None, w.to(dev()), None, 1, -1.0)
    check(y, F.conv3d(x, w, None, padding=1), tol=1e-6, what='impulse response')


@pytest.mark.parametrize('Cin,Cout,dims', [(16, 16, (1, 3, 4, 5)), (64, 64, (1, 2, 3, 5)), (32, 32, (2, 4, 4, 6)), (6, 5, (1, 2, 3, 4)),
                                           (128, 128, (1, 3, 4, 5)), (256, 80, (1, 2, 3, 5)), (48, 192, (1, 2, 2, 3))])   # > 64: host-tiled 64 x 64 slices (full UNet)
def test_deconv_k2s2(Cin, Cout, dims):
    from deepatlas_amd import ops
    N, D, H, W = dims
    x, w, b = rnd((N, Cin, D, H, W), 1), rnd((Cin, Cout, 2, 2, 2), 2, 0.3), rnd((Cout,), 3, 0.1)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv_transpose3d(xr, wr, br, stride=2)
    go = rnd(tuple(yr.shape), 4)
    yr.backward(go)
    xg, wg, bg = cl(x).requires_grad_(True), w.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
    yg = ops.DeconvK2S2Fn.apply(xg, wg, bg)
    yg.backward(cl(go))
    check(yg, yr, what='fwd'); check(xg.grad, xr.grad, what='dgrad'); check(wg.grad, wr.grad, what='wgrad'); check(bg.grad, br.grad, what='bgrad')


@pytest.mark.parametrize('Cin,Cout', [(16, 32), (8, 5), (16, 3), (128, 32), (80, 144)])
def test_conv1x1(Cin, Cout):
    from deepatlas_amd import ops
    x, w, b = rnd((2, Cin, 4, 6, 10), 1), rnd((Cout, Cin, 1, 1, 1), 2, 0.3), rnd((Cout,), 3, 0.1)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, br)
    go = rnd(tuple(yr.shape), 4)
    yr.backward(go)
    xg, wg, bg = cl(x).requires_grad_(True), w.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
    yg = ops.Conv1x1Fn.apply(xg, wg, bg)
    yg.backward(cl(go))
    check(yg, yr, what='fwd'); check(xg.grad, xr.grad, what='dgrad'); check(wg.grad, wr.grad, what='wgrad'); check(bg.grad, br.grad, what='bgrad')


@pytest.mark.parametrize('C,training', [(8, True), (16, True), (64, True), (5, True), (16, False)])
def test_bn_act(C, training):
    from deepatlas_amd import ops
    x = rnd((2, C, 6, 8, 10), 1, 2.0) + 0.5
    gamma, beta = 1 + rnd((C,), 2, 0.2), rnd((C,), 3, 0.2)
    rm, rv = rnd((C,), 4, 0.1), 1 + rnd((C,), 5, 0.3)
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rmr, rvr = rm.clone(), rv.clone()
    yr = F.leaky_relu(F.batch_norm(xr, rmr, rvr, gr, br, training, 0.1, 1e-5), 0.01)
    go = rnd(tuple(yr.shape), 6)
    yr.backward(go)
    xg, gg, bg = cl(x).requires_grad_(True), gamma.to(dev()).requires_grad_(True), beta.to(dev()).requires_grad_(True)
    rmg, rvg = rm.to(dev()), rv.to(dev())
    yg = ops.BNActFn.apply(xg, gg, bg, rmg, rvg, training, 0.1, 1e-5, 0.01)
    yg.backward(cl(go))
    check(yg, yr, what='fwd'); check(xg.grad, xr.grad, what='dx'); check(gg.grad, gr.grad, what='dgamma'); check(bg.grad, br.grad, what='dbeta')
    check(rmg, rmr, what='running_mean'); check(rvg, rvr, what='running_var')


def test_maxpool_golden_and_random(golden):
    from deepatlas_amd import ops
    g = golden('ops')
    p = cl(T(g['ops/maxpool/in'])).requires_grad_(True)
    q = ops.MaxPool2Fn.apply(p)
    q.sum().backward()
    assert np.array_equal(q.detach().cpu().numpy(), g['ops/maxpool/out'])
    assert np.array_equal(p.grad.cpu().numpy(), g['ops/maxpool/grad'])       # ties: first maximum gets the gradient
    x = rnd((2, 16, 6, 8, 10), 3)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool3d(xr, 2); go = rnd(tuple(yr.shape), 4); yr.backward(go)
    xg = cl(x).requires_grad_(True)
    yg = ops.MaxPool2Fn.apply(xg); yg.backward(cl(go))
    assert torch.equal(yg.detach().cpu(), yr.detach()) and torch.equal(xg.grad.cpu(), xr.grad)


def test_maxpool_backward_with_producer_sums():
    """da_maxpool2_bwd_bst: the max-pool (+ skip) backward on the RAW tensor the pool's forward normalised on the fly.  dx must be bit-equal to
    da_maxpool2_bwd_add on the activated tensor da_maxpool2_fwd_pro wrote (same arg-max, also for channels with a NEGATIVE BatchNorm scale, where the
    order of the raw values is the reverse of the activated ones), and the partials must add up to sum dz, sum dz (x - mean) with dz = dx act'(x scale + shift)."""
    import ctypes
    from deepatlas_amd import _native as nat
    call, ptr = nat.call, nat.ptr
    N, D, H, W = 2, 6, 8, 10
    for C, slope in ((16, 0.01), (32, 0.0), (8, 0.2)):
        g = torch.Generator().manual_seed(C)
        x = torch.randn((N, D, H, W, C), generator=g).to(dev())
        stats = torch.empty((4, C), device=dev())
        stats[0] = torch.randn(C, generator=g).to(dev()) * 0.1                      # mean
        stats[1] = 1.0                                                            # rstd (not read)
        stats[2] = (torch.randn(C, generator=g) * 0.8 + 0.3).to(dev())            # scale: both signs
        stats[3] = torch.randn(C, generator=g).to(dev()) * 0.2                      # shift
        act, pooled = torch.empty_like(x), torch.empty((N, D // 2, H // 2, W // 2, C), device=dev())
        st = nat.stream()
        call('da_maxpool2_fwd_pro', ptr(x), ptr(stats[2]), ptr(stats[3]), slope, ptr(act), ptr(pooled), N, D, H, W, C, st)
        dy = torch.randn(pooled.shape, generator=g).to(dev())
        gskip = torch.randn(x.shape, generator=g).to(dev())
        for skip in (gskip, None):
            ref, dx = torch.empty_like(x), torch.empty_like(x)
            if skip is None:
                call('da_maxpool2_bwd', ptr(dy), ptr(act), ptr(ref), N, D, H, W, C, st)
            else:
                call('da_maxpool2_bwd_add', ptr(dy), ptr(act), ptr(skip), ptr(ref), N, D, H, W, C, st)
            bst = torch.zeros((1024, 2, C), dtype=torch.float64, device=dev())
            nb = ctypes.c_int(0)
            call('da_maxpool2_bwd_bst', ptr(dy), ptr(x), ptr(skip), ptr(dx), N, D, H, W, C, ptr(stats), slope, ptr(bst), 1024, ctypes.byref(nb), st)
            torch.cuda.synchronize()
            assert nb.value > 0 and torch.equal(dx, ref), (C, slope, skip is None)
            z = x.double() * stats[2].double() + stats[3].double()
            dz = dx.double() * torch.where(z > 0, torch.ones_like(z), torch.full_like(z, slope))
            s1 = dz.sum((0, 1, 2, 3)); s2 = (dz * (x.double() - stats[0].double())).sum((0, 1, 2, 3))
            got = bst[:nb.value].sum(0)
            assert float((got[0] - s1).abs().max()) < 1e-4 * (1 + float(s1.abs().max())), (C, slope)
            assert float((got[1] - s2).abs().max()) < 1e-4 * (1 + float(s2.abs().max())), (C, slope)
    with pytest.raises(nat.NativeError):                                           # odd size: declined (the caller takes the plain kernels)
        call('da_maxpool2_bwd_bst', ptr(dy), ptr(x), None, ptr(dx), N, 5, H, W, 8, ptr(stats), 0.2, ptr(bst), 1024, ctypes.byref(nb), st)


def test_upsample_nearest_golden_odd_sizes(golden):
    from deepatlas_amd import ops
    g = golden('ops')
    t = T(g['ops/nearest/in'])
    for key, size in (('out_3_5_10', (3, 5, 10)), ('out_4_6_10', (4, 6, 10))):
        xg = cl(t).requires_grad_(True)
        y = ops.UpsampleNearestFn.apply(xg, size)
        assert np.array_equal(y.detach().cpu().numpy(), g['ops/nearest/' + key])
        go = rnd(tuple(y.shape), 7)
        y.backward(cl(go))
        xr = t.clone().requires_grad_(True)
        F.interpolate(xr, size=size).backward(go)
        check(xg.grad, xr.grad, tol=1e-6, what='nearest bwd')


@pytest.mark.parametrize('nm', ['warp1', 'warpC'])
def test_warp_golden(golden, nm):
    from deepatlas_amd import ops
    g = golden('ops')
    src = cl(T(g[f'ops/{nm}/src'])).requires_grad_(True)
    disp = cl(T(g[f'ops/{nm}/disp'])).requires_grad_(True)
    out, deform = ops.WarpFn.apply(src, disp)
    check(out, g[f'ops/{nm}/out'], tol=1e-5, what='warp fwd')
    idt = T(g['ops/identity'])
    check(deform, T(g[f'ops/{nm}/disp']) + idt, tol=1e-6, what='deform')
    (out * cl(T(g[f'ops/{nm}/gout']))).sum().backward()
    check(src.grad, g[f'ops/{nm}/grad_src'], tol=1e-5, what='grad_src')
    check(disp.grad, g[f'ops/{nm}/grad_disp'], tol=1e-5, what='grad_disp')


def test_warp_c32_vs_torch():
    from deepatlas_amd import ops
    from oracle import nets
    N, C, D, H, W = 1, 32, 6, 10, 12
    src, disp = rnd((N, C, D, H, W), 1), rnd((N, 3, D, H, W), 2, 0.4)
    sr, dr = src.clone().requires_grad_(True), disp.clone().requires_grad_(True)
    wr = nets.warp_trilinear(sr, dr + nets.identity_transform((D, H, W)))
    go = rnd(tuple(wr.shape), 3); wr.backward(go)
    sg, dg = cl(src).requires_grad_(True), cl(disp).requires_grad_(True)
    wg, _ = ops.WarpFn.apply(sg, dg); wg.backward(cl(go))
    check(wg, wr, tol=1e-5, what='fwd'); check(sg.grad, sr.grad, tol=1e-5, what='grad_src'); check(dg.grad, dr.grad, tol=1e-5, what='grad_disp')


def test_identity_grid_and_onehot(golden):
    from deepatlas_amd.lib import utils, transforms
    g = golden('ops')
    D, H, W = g['ops/identity'].shape[1:]
    idt = utils.get_identity_transform_batch((1, 1, D, H, W))
    assert np.array_equal(idt.cpu().numpy(), g['ops/identity'])
    labels = T(g['ops/dice/labels']).to(dev())
    oh = transforms.mask_to_one_hot(labels.view(2, 1, *labels.shape[1:]), 5)
    assert np.array_equal(oh.cpu().numpy(), g['ops/onehot'])


def test_dice_golden_all_weightings(golden):
    from deepatlas_amd.lib.loss import DiceLossMultiClass
    g = golden('ops')
    for wt in ('Uniform', 'Simple', 'Volume'):
        for no_bg in (False, True):
            logits = cl(T(g['ops/dice/logits'])).requires_grad_(True)
            labels = T(g['ops/dice/labels']).to(dev())
            crit = DiceLossMultiClass(n_class=5, weight_type=wt, no_bg=no_bg, softmax=True, eps=1e-6)
            l = crit(logits, labels.long())
            l.backward()
            assert abs(l.item() - g[f'ops/dice/{wt}_{int(no_bg)}/loss']) < 1e-5, (wt, no_bg)
            check(logits.grad, g[f'ops/dice/{wt}_{int(no_bg)}/grad'], what='dice grad %s %d' % (wt, no_bg))
    # uint8 labels are consumed without a cast
    logits = cl(T(g['ops/dice/logits']))
    l8 = DiceLossMultiClass(n_class=5, weight_type='Uniform', softmax=True, eps=1e-6)(logits, T(g['ops/dice/labels']).to(dev()))
    assert abs(l8.item() - g['ops/dice/Uniform_0/loss']) < 1e-5


def test_dice_soft_target_golden(golden):
    from deepatlas_amd.lib.loss import DiceLossMultiClass
    g = golden('ops')
    src = cl(T(g['ops/dice_soft/source'])).requires_grad_(True)
    l = DiceLossMultiClass(n_class=5, weight_type='Uniform', no_bg=False, softmax=False, eps=1e-6)(src, cl(T(g['ops/dice_soft/target'])))
    l.backward()
    assert abs(l.item() - g['ops/dice_soft/loss']) < 1e-5
    check(src.grad, g['ops/dice_soft/grad'], what='soft dice grad')


def test_dice_c32_vs_oracle():
    from deepatlas_amd.lib.loss import DiceLossMultiClass
    from oracle import losses, nets
    logits = rnd((2, 32, 8, 12, 10), 1, 3.0)
    labels = nets.closed_form_labels((2, 8, 12, 10), 32, seed=1)
    lr = logits.clone().requires_grad_(True)
    lo = losses.dice_loss(lr, labels.long(), 32); lo.backward()
    lg = cl(logits).requires_grad_(True)
    l = DiceLossMultiClass(n_class=32, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)(lg, labels.to(dev()).long())
    l.backward()
    assert abs(l.item() - lo.item()) < 1e-5
    check(lg.grad, lr.grad, what='dice c32 grad')


def test_dice_errors():
    from deepatlas_amd.lib.loss import DiceLossMultiClass, get_loss_function
    crit = DiceLossMultiClass(n_class=5, weight_type='Uniform')
    with pytest.raises(ValueError):
        crit(torch.zeros(1, 5, 4, 4, 4, device=dev()), torch.zeros(1, 3, 4, 4, 4, device=dev()))
    with pytest.raises(KeyError):
        get_loss_function('nope')


def test_ncc_bending_golden(golden):
    from deepatlas_amd.lib.loss import NormalizedCrossCorrelationLoss, BendingEnergyLoss
    g = golden('ops')
    a = T(g['ops/ncc/a']).to(dev()).requires_grad_(True)
    l = NormalizedCrossCorrelationLoss()(a, T(g['ops/ncc/b']).to(dev()))
    l.backward()
    assert abs(l.item() - g['ops/ncc/loss']) < 1e-5
    check(a.grad, g['ops/ncc/grad'], what='ncc grad')
    u = cl(T(g['ops/bending/u'])).requires_grad_(True)
    l = BendingEnergyLoss()(u)
    l.backward()
    assert abs(l.item() - g['ops/bending/loss']) < 1e-4 * abs(g['ops/bending/loss'])
    check(u.grad, g['ops/bending/grad'], what='bending grad')
    l2 = BendingEnergyLoss(spacing=(1.0, 2.0, 1.5))(cl(T(g['ops/bending/u'])))
    assert abs(l2.item() - g['ops/bending/loss_spacing']) < 1e-4 * abs(g['ops/bending/loss_spacing'])


def test_adjoint_label_scatter_box_and_direct_paths():
    """da_warp_adjoint_labels, B[n][c][u] = sum over target voxels v with label c of the trilinear weight v puts on source voxel u (the adjoint of
    voxel_morph.py:90-91's warp applied to one-hot labels): the LDS-box kernel against a float64 scatter built from the same taps.  Fields: small and
    smooth (everything lands in the grown box), eight voxels of noise (everything takes the direct global atomics), and a mix; labels: piecewise constant
    (<= 3 per box) and per-voxel random (more labels than a box has copies); ragged sizes; coordinates outside the volume and a NaN."""
    from deepatlas_amd import _native as nat
    call, ptr = nat.call, nat.ptr
    g = torch.Generator().manual_seed(5)
    N, D, H, W, C = 2, 11, 21, 45, 5
    V = D * H * W
    zz, yy, xx = torch.meshgrid(torch.arange(D), torch.arange(H), torch.arange(W), indexing='ij')
    blocky = ((zz // 6 + yy // 9 + xx // 14) % C).to(torch.uint8)
    labs = {'blocky': torch.stack([blocky, (blocky + 1) % C]), 'random': torch.randint(0, C, (N, D, H, W), generator=g, dtype=torch.uint8)}
    scale = torch.tensor([2.0 / (W - 1), 2.0 / (H - 1), 2.0 / (D - 1)])
    fields = {'smooth': torch.randn((N, 1, 1, 1, 3), generator=g).expand(N, D, H, W, 3) * 0.7 * scale + torch.randn((N, D, H, W, 3), generator=g) * 0.2 * scale,
              'noise8': torch.randn((N, D, H, W, 3), generator=g) * 8.0 * scale}
    mix = fields['smooth'].clone(); mix[:, ::2] = fields['noise8'][:, ::2]; mix[0, 3, 4, 5, 0] = float('nan'); mix[1, 2, 3, 4] = 7.0
    fields['mix'] = mix
    ident = torch.stack([xx * scale[0] - 1, yy * scale[1] - 1, zz * scale[2] - 1], -1).double()
    for ln, lab in labs.items():
        for fn, u in fields.items():
            u = u.contiguous()
            ref = torch.zeros((N, C, V), dtype=torch.float64)
            grid = u.double() + ident
            fin = (grid.abs() < 1e9).all(-1) & ~torch.isnan(grid).any(-1)
            pos = [((torch.nan_to_num(grid[..., a]) + 1) / 2) * (s - 1) for a, s in ((0, W), (1, H), (2, D))]
            p0 = [torch.floor(q) for q in pos]
            for cz in (0, 1):
                for cy in (0, 1):
                    for cx in (0, 1):
                        ix, iy, iz = p0[0] + cx, p0[1] + cy, p0[2] + cz
                        wgt = ((pos[0] - p0[0]) if cx else (p0[0] + 1 - pos[0])) * ((pos[1] - p0[1]) if cy else (p0[1] + 1 - pos[1])) * ((pos[2] - p0[2]) if cz else (p0[2] + 1 - pos[2]))
                        ok = fin & (ix >= 0) & (ix < W) & (iy >= 0) & (iy < H) & (iz >= 0) & (iz < D)
                        for n in range(N):
                            sel = ok[n]
                            dst = lab[n][sel].long() * V + ((iz[n][sel] * H + iy[n][sel]) * W + ix[n][sel]).long()
                            ref[n].view(-1).index_add_(0, dst, wgt[n][sel])
            B = torch.full((N, C, V), 7.0, device=dev())                 # (the launcher zero-fills)
            ud, ld = u.float().to(dev()).contiguous(), lab.to(dev()).contiguous()
            call('da_warp_adjoint_labels', ptr(ld), 1, ptr(ud), None, ptr(B), N, D, H, W, C, nat.stream())
            torch.cuda.synchronize()
            err = (B.cpu().double() - ref).abs().max().item()
            assert err < 2e-5, (ln, fn, err)
            assert abs(B.sum().item() - ref.sum().item()) < 1e-2 * max(1.0, ref.sum().item() * 1e-3), (ln, fn)


def test_softmax():
    from deepatlas_amd import ops
    for C in (32, 5):
        x = rnd((2, C, 4, 6, 5), 1, 4.0)
        xr = x.clone().requires_grad_(True)
        yr = F.softmax(xr, 1); go = rnd(tuple(yr.shape), 2); yr.backward(go)
        xg = cl(x).requires_grad_(True)
        yg = ops.SoftmaxFn.apply(xg); yg.backward(cl(go))
        check(yg, yr, tol=1e-6, what='softmax'); check(xg.grad, xr.grad, tol=1e-5, what='softmax bwd')


def test_eval_dice_golden_bit_exact(golden):
    from deepatlas_amd.lib import evalMetrics as M
    from deepatlas_amd import ops
    g = golden('ops')
    pred, truth = T(g['ops/evaldice/pred']), T(g['ops/evaldice/truth'])
    logits = ops.one_hot(pred.to(dev()).view(1, 1, *pred.shape[1:]), 6)
    d = M.metricEval('dice', logits, truth.to(dev()))[0]
    ref = g['ops/evaldice/dice']
    assert np.array_equal(np.isnan(d), np.isnan(ref))
    assert np.array_equal(d[~np.isnan(ref)], ref[~np.isnan(ref)])             # integer counts -> bit-equal
    counts, am = M.eval_dice_counts(logits, truth.to(dev()))
    assert np.array_equal(am.cpu().numpy(), pred.numpy())


def test_argmax_first_max_on_ties():
    from deepatlas_amd import ops
    logits = torch.zeros(1, 32, 2, 2, 4)
    logits[0, 7] = 1.0; logits[0, 19] = 1.0                                   # exact tie -> index 7 (torch.max)
    truth = torch.full((1, 2, 2, 4), 7, dtype=torch.uint8)
    counts, pred = ops.argmax_dice_counts(cl(logits), truth.to(dev()))
    assert int(pred.min()) == 7 and int(pred.max()) == 7
    assert counts[0, 7].tolist() == [16, 16, 16]


def test_adam_vs_oracle():
    from deepatlas_amd.optim import FlatAdam
    from oracle import steps
    ps = [torch.nn.Parameter(rnd((7, 5), 1).to(dev())), torch.nn.Parameter(rnd((33,), 2).to(dev()))]
    sd = {'a': ps[0].detach().cpu().clone(), 'b': ps[1].detach().cpu().clone()}
    opt = FlatAdam(ps, lr=1e-3)
    o = steps.Adam(['a', 'b'], lr=1e-3)
    for s in range(3):
        gs = {'a': rnd((7, 5), 10 + s), 'b': rnd((33,), 20 + s)}
        opt.zero_grad()
        ps[0].grad.copy_(gs['a']); ps[1].grad.copy_(gs['b'])
        opt.step()
        o.step(sd, gs)
    check(ps[0], sd['a'], tol=1e-6, what='adam a'); check(ps[1], sd['b'], tol=1e-6, what='adam b')
    st = opt.state_dict()
    assert set(st['state'][0].keys()) >= {'step', 'exp_avg', 'exp_avg_sq'}


# ---- full-size, size-independent properties (BASELINE config sizes) ---------------------------------
FULL = (160, 192, 160)


def test_full_size_identity_warp_and_ncc():
    from deepatlas_amd import ops
    from deepatlas_amd.lib.loss import NormalizedCrossCorrelationLoss, BendingEnergyLoss
    D, H, W = FULL
    g = torch.Generator(device='cpu').manual_seed(230)
    src = torch.rand((1, 1, D, H, W), generator=g).to(dev())
    disp = torch.zeros((1, 3, D, H, W), device=dev()).contiguous(memory_format=torch.channels_last_3d)
    out, deform = ops.WarpFn.apply(src, disp)
    assert float((out - src).abs().max()) < 2e-5                           # identity warp reproduces the input (1.5e-6 in the survey)
    assert abs(NormalizedCrossCorrelationLoss()(src, src).item()) < 1e-5   # NCC(x, x) = 1
    # bending energy of an affine displacement field is zero
    idt = ops.identity_grid((D, H, W)).unsqueeze(0)
    aff = (0.3 * idt + 0.1).contiguous(memory_format=torch.channels_last_3d)
    assert abs(BendingEnergyLoss()(aff).item()) < 1e-8


def test_full_size_dice_of_identical_onehots_and_counts():
    from deepatlas_amd import ops
    from deepatlas_amd.lib.loss import DiceLossMultiClass
    from deepatlas_amd.lib.datasets import structured_labels
    lab = structured_labels(FULL, 32).unsqueeze(0).to(dev())
    oh = ops.one_hot(lab.unsqueeze(1), 32)
    l = DiceLossMultiClass(n_class=32, weight_type='Uniform', no_bg=False, softmax=False, eps=1e-6)(oh, lab)
    assert abs(l.item()) < 1e-6
    counts, pred = ops.argmax_dice_counts(oh, lab)
    assert torch.equal(pred, lab)
    assert int(counts[0, :, 0].sum()) == lab.numel() and torch.equal(counts[0, :, 0], counts[0, :, 2])


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize('C1,C2,Cout,dims,stride', [(16, 0, 16, (1, 9, 11, 21), 1), (32, 16, 16, (2, 8, 16, 32), 1), (8, 0, 16, (1, 6, 9, 17), 1),
                                                     (16, 16, 32, (1, 5, 8, 16), 1), (32, 0, 48, (1, 4, 8, 16), 1), (16, 0, 32, (1, 8, 12, 20), 2),
                                                     (64, 0, 64, (1, 4, 9, 18), 1)])
def test_conv3d_bf16_matrix_mode(C1, C2, Cout, dims, stride):
    """BASELINE config 5's arithmetic: operands rounded to bf16 (round-to-nearest-even), products and sums in fp32.  bf16 x bf16
    products are exact in fp32, so the bf16 mode must equal the fp32 reference run on pre-rounded operands up to summation order."""
    from deepatlas_amd import ops
    N, D, H, W = dims
    Cin = C1 + C2
    x, w, b = rnd((N, Cin, D, H, W), 1), rnd((Cout, Cin, 3, 3, 3), 2, 0.2), rnd((Cout,), 3, 0.1)
    xr, wr = _bf16_round(x).requires_grad_(True), _bf16_round(w).requires_grad_(True)
    yr = F.conv3d(xr, wr, b, stride=stride, padding=1)
    go = rnd(tuple(yr.shape), 4)
    (yr * _bf16_round(go)).sum().backward()                # data gradient of the reference with dy pre-rounded
    prev = ops.set_matrix_precision('bf16')
    try:
        x1 = cl(x[:, :C1]).requires_grad_(True)
        x2 = cl(x[:, C1:]).requires_grad_(True) if C2 else None
        wg = w.to(dev()).requires_grad_(True)
        yg = ops.Conv3dK3Fn.apply(x1, x2, wg, b.to(dev()), stride, -1.0)
        yg.backward(cl(go))
    finally:
        ops.set_matrix_precision(prev)
    check(yg, yr, tol=2e-5, what='bf16 fwd')
    check(x1.grad, xr.grad[:, :C1], tol=2e-5, what='bf16 dgrad')
    if C2:
        check(x2.grad, xr.grad[:, C1:], tol=2e-5, what='bf16 dgrad (second input)')
    check(wg.grad, wr.grad, tol=2e-5, what='bf16 wgrad')


@pytest.mark.parametrize('C1,C2,Cout,dims,lazy', [(16, 0, 16, (1, 9, 11, 21), (True, False)), (32, 16, 16, (2, 8, 16, 32), (True, False)),
                                                  (32, 16, 16, (1, 6, 9, 20), (True, True)), (8, 0, 16, (1, 6, 9, 17), (True, False)),
                                                  (16, 16, 32, (1, 5, 8, 16), (False, True)), (64, 32, 32, (1, 4, 9, 18), (True, False))])
@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_conv3d_input_prologue_is_bit_identical(C1, C2, Cout, dims, lazy, precision):
    """da_conv3d_k3_fwd_pro / _wgrad_pro (deferred BatchNorm + LeakyReLU applied while the input tile is staged) against the plain
    entries on the materialised activation: same arithmetic, same summation order -> bit-identical forward, BN partial sums and
    weight gradient (the reference's Conv3d -> BatchNorm3d -> LeakyReLU -> Conv3d chain, unets.py:24-39)."""
    import ctypes
    from deepatlas_amd import _native as nat, ops
    from deepatlas_amd._native import call, ptr, stream, workspace
    prev_precision = ops.set_matrix_precision(precision)      # bf16 mode: the rounding happens after the prologue, on identical values
    try:
        _prologue_case(C1, C2, Cout, dims, lazy)
    finally:
        ops.set_matrix_precision(prev_precision)


def _prologue_case(C1, C2, Cout, dims, lazy):
    import ctypes
    from deepatlas_amd import _native as nat
    from deepatlas_amd._native import call, ptr, stream, workspace
    N, D, H, W = dims
    d = dev()
    raw1, raw2 = rnd((N, D, H, W, C1), 1).to(d), (rnd((N, D, H, W, C2), 2).to(d) if C2 else None)
    sc1, sh1 = (rnd((C1,), 3) * 0.5 + 1.0).to(d), rnd((C1,), 4, 0.3).to(d)
    sc2, sh2 = ((rnd((C2,), 5) * 0.5 + 1.0).to(d), rnd((C2,), 6, 0.3).to(d)) if C2 else (None, None)
    slope = 0.01
    def act(raw, sc, sh):
        out = torch.empty_like(raw)
        call('da_bn_act_fwd', ptr(raw), ptr(sc), ptr(sh), slope, ptr(out), raw.numel() // raw.shape[-1], raw.shape[-1], stream())
        return out
    a1 = act(raw1, sc1, sh1) if lazy[0] else raw1
    a2 = (act(raw2, sc2, sh2) if lazy[1] else raw2) if C2 else None
    w = rnd((27, C1 + C2, Cout), 7, 0.2).to(d)
    b = rnd((Cout,), 8, 0.1).to(d)
    dy = rnd((N, D, H, W, Cout), 9).to(d)
    wsb = nat.lib().da_conv3d_k3_ws_bytes(N, D, H, W, C1 + C2, Cout, 1)
    wp, wn = workspace.get(wsb, d)
    st = stream()
    p1 = (ptr(sc1), ptr(sh1), slope) if lazy[0] else (None, None, -1.0)
    p2 = (ptr(sc2), ptr(sh2), slope) if (C2 and lazy[1]) else (None, None, -1.0)
    for stats in (True, False):
        y_ref, y_pro = torch.empty_like(dy), torch.empty_like(dy)
        pb_ref = torch.zeros((512, 2, Cout), dtype=torch.float64, device=d); pb_pro = torch.zeros_like(pb_ref)
        n_ref, n_pro = ctypes.c_int(0), ctypes.c_int(0)
        if stats:
            call('da_conv3d_k3_fwd_bnstats', ptr(a1), C1, ptr(a2), C2, ptr(w), ptr(b), ptr(y_ref), N, D, H, W, Cout, 1, ptr(pb_ref), 512, ctypes.byref(n_ref), wp, wn, st)
        else:
            call('da_conv3d_k3_fwd', ptr(a1), C1, ptr(a2), C2, ptr(w), ptr(b), ptr(y_ref), N, D, H, W, Cout, 1, 0.01, wp, wn, st)
        call('da_conv3d_k3_fwd_pro', ptr(raw1), C1, p1[0], p1[1], p1[2], ptr(raw2), C2, p2[0], p2[1], p2[2], ptr(w), ptr(b), ptr(y_pro),
             N, D, H, W, Cout, 0.01, ptr(pb_pro) if stats else None, 512 if stats else 0, ctypes.byref(n_pro), wp, wn, st)
        assert torch.equal(y_ref, y_pro), ('fwd', stats, float((y_ref - y_pro).abs().max()))
        if stats:
            assert n_ref.value == n_pro.value and n_ref.value > 0
            assert torch.equal(pb_ref[:n_ref.value], pb_pro[:n_ref.value])
    dw_ref, dw_pro = torch.empty_like(w), torch.empty_like(w)
    call('da_conv3d_k3_wgrad', ptr(a1), C1, ptr(a2), C2, ptr(dy), ptr(dw_ref), None, N, D, H, W, Cout, 1, wp, wn, st)
    call('da_conv3d_k3_wgrad_pro', ptr(raw1), C1, p1[0], p1[1], p1[2], ptr(raw2), C2, p2[0], p2[1], p2[2], ptr(dy), ptr(dw_pro), N, D, H, W, Cout, wp, wn, st)
    assert torch.equal(dw_ref, dw_pro), float((dw_ref - dw_pro).abs().max())


@pytest.mark.parametrize('Cin,Cout,dims', [(32, 32, (2, 8, 12, 10)), (16, 16, (1, 3, 5, 7)), (64, 32, (1, 5, 6, 9)), (128, 128, (1, 3, 4, 5)), (48, 80, (1, 2, 3, 17))])
def test_deconv_fused_batchnorm_statistics(Cin, Cout, dims):
    """da_deconv_k2s2_fwd_bnstats: same output as the plain entry (bit for bit) and per-channel sums / sums of squares of that output
    in the epilogue partials (ragged last workgroup, 64-wide channel slices and K slices included)."""
    import ctypes
    from deepatlas_amd import _native as nat
    from deepatlas_amd._native import call, ptr, stream, workspace
    N, D, H, W = dims
    d = dev()
    x = rnd((N, D, H, W, Cin), 1).to(d)
    w = rnd((8, Cin, Cout), 2, 0.3).to(d)
    b = rnd((Cout,), 3, 0.1).to(d)
    y0 = torch.empty((N, 2 * D, 2 * H, 2 * W, Cout), device=d)
    y1 = torch.empty_like(y0)
    wp, wn = workspace.get(nat.lib().da_pointwise_ws_bytes(8, Cin, Cout), d)
    call('da_deconv_k2s2_fwd', ptr(x), ptr(w), ptr(b), ptr(y0), N, D, H, W, Cin, Cout, wp, wn, stream())
    nblk = (N * D * H * W + 255) // 256
    pbuf = torch.full((nblk, 2, Cout), float('nan'), dtype=torch.float64, device=d)
    npar = ctypes.c_int(0)
    call('da_deconv_k2s2_fwd_bnstats', ptr(x), ptr(w), ptr(b), ptr(y1), N, D, H, W, Cin, Cout, ptr(pbuf), nblk, ctypes.byref(npar), wp, wn, stream())
    assert npar.value == nblk
    assert torch.equal(y0, y1)
    s = pbuf.sum(0).cpu()
    ref = y0.double().reshape(-1, Cout).cpu()
    # per-lane fp32 sums of <= 16 values, everything above in double: error bounded by 1e-6 of the sum of magnitudes
    assert bool(((s[0] - ref.sum(0)).abs() <= 1e-6 * ref.abs().sum(0)).all())
    assert bool(((s[1] - (ref * ref).sum(0)).abs() <= 1e-6 * (ref * ref).sum(0)).all())
    # too small a capacity: output still produced, no statistics claimed
    npar2 = ctypes.c_int(7)
    call('da_deconv_k2s2_fwd_bnstats', ptr(x), ptr(w), ptr(b), ptr(y1), N, D, H, W, Cin, Cout, ptr(pbuf), nblk - 1, ctypes.byref(npar2), wp, wn, stream())
    assert npar2.value == 0 and torch.equal(y0, y1)


@pytest.mark.parametrize('Cin,Cout,M', [(16, 32, 5000), (64, 16, 777), (128, 48, 1300), (128, 64, 900), (16, 16, 64)])
def test_conv1x1_input_prologue_is_bit_identical(Cin, Cout, M):
    """da_conv1x1_fwd_pro / _wgrad_pro (the head consuming the raw output of the last decoder block, unets.py:249-250) against
    da_bn_act_fwd followed by the plain entries: same arithmetic, same order -> bit-identical output, weight and bias gradients
    (ragged row counts, 64-wide channel slices included)."""
    from deepatlas_amd import _native as nat
    from deepatlas_amd._native import call, call_supported, ptr, stream, workspace
    d = dev()
    raw = rnd((M, Cin), 1).to(d)
    sc, sh = (rnd((Cin,), 2) * 0.5 + 1.0).to(d), rnd((Cin,), 3, 0.3).to(d)
    w = rnd((Cin, Cout), 4, 0.3).to(d)
    b = rnd((Cout,), 5, 0.1).to(d)
    dy = rnd((M, Cout), 6).to(d)
    st = stream()
    for slope in (0.01, 0.0):
        a = torch.empty_like(raw)
        call('da_bn_act_fwd', ptr(raw), ptr(sc), ptr(sh), slope, ptr(a), M, Cin, st)
        wp, wn = workspace.get(max(nat.lib().da_pointwise_ws_bytes(1, Cin, Cout), nat.lib().da_conv1x1_wgrad_ws_bytes(M, Cin, Cout)), d)
        y0, y1 = torch.empty((M, Cout), device=d), torch.empty((M, Cout), device=d)
        call('da_conv1x1_fwd', ptr(a), ptr(w), ptr(b), ptr(y0), M, Cin, Cout, wp, wn, st)
        call('da_conv1x1_fwd_pro', ptr(raw), ptr(sc), ptr(sh), slope, ptr(w), ptr(b), ptr(y1), M, Cin, Cout, wp, wn, st)
        assert torch.equal(y0, y1), ('fwd', slope, float((y0 - y1).abs().max()))
        dw0, dw1 = torch.empty_like(w), torch.empty_like(w)
        db0, db1 = torch.empty_like(b), torch.empty_like(b)
        call('da_conv1x1_wgrad', ptr(a), ptr(dy), ptr(dw0), ptr(db0), M, Cin, Cout, wp, wn, st)
        if not call_supported('da_conv1x1_wgrad_pro', ptr(raw), ptr(sc), ptr(sh), slope, ptr(dy), ptr(dw1), ptr(db1), M, Cin, Cout, wp, wn, st):
            assert (Cin, Cout) == (128, 48)      # a 64 x 48 channel slice has no single-tap weight-gradient instantiation: declined, the caller materialises
            check(y1, a.cpu() @ w.cpu() + b.cpu(), what='head fwd with prologue')
            continue
        assert torch.equal(dw0, dw1), ('wgrad', slope, float((dw0 - dw1).abs().max()))
        assert torch.equal(db0, db1)
        # and against torch on the activated input (not only self-consistency)
        ref = a.cpu() @ w.cpu() + b.cpu()
        check(y1, ref, what='head fwd with prologue')
        check(dw1, a.cpu().t() @ dy.cpu(), what='head wgrad with prologue')


@pytest.mark.parametrize('C,dims', [(16, (2, 8, 12, 10)), (8, (1, 4, 6, 8)), (32, (1, 2, 4, 6))])
def test_maxpool_input_prologue_is_bit_identical(C, dims):
    """da_maxpool2_fwd_pro (deferred BatchNorm + LeakyReLU applied in the pooling pass, which also writes the activated skip tensor)
    against da_bn_act_fwd + da_maxpool2_fwd; odd sizes are declined (the caller materialises)."""
    from deepatlas_amd._native import call, call_supported, ptr, stream
    N, D, H, W = dims
    d = dev()
    raw = rnd((N, D, H, W, C), 1).to(d)
    sc, sh = (rnd((C,), 2) * 0.5 + 1.0).to(d), rnd((C,), 3, 0.3).to(d)
    for slope in (0.01, 0.0):
        a0 = torch.empty_like(raw)
        call('da_bn_act_fwd', ptr(raw), ptr(sc), ptr(sh), slope, ptr(a0), raw.numel() // C, C, stream())
        y0 = torch.empty((N, D // 2, H // 2, W // 2, C), device=d)
        call('da_maxpool2_fwd', ptr(a0), ptr(y0), N, D, H, W, C, stream())
        a1, y1 = torch.empty_like(raw), torch.empty_like(y0)
        call('da_maxpool2_fwd_pro', ptr(raw), ptr(sc), ptr(sh), slope, ptr(a1), ptr(y1), N, D, H, W, C, stream())
        assert torch.equal(a0, a1) and torch.equal(y0, y1)
    odd = rnd((1, 5, 6, 8, C), 4).to(d)
    assert not call_supported('da_maxpool2_fwd_pro', ptr(odd), ptr(sc), ptr(sh), 0.01, ptr(torch.empty_like(odd)),
                              ptr(torch.empty((1, 2, 3, 4, C), device=d)), 1, 5, 6, 8, C, stream())


def test_large_batch_transposed_conv_and_head_beyond_4gib():
    """Batch 8 at 160x192x160: the 32 -> 32 up-sampler's output and the 32-class logits are 5 GB each, past 32-bit byte offsets.
    Additivity over the batch axis: weight / bias gradients of the whole batch = sum over the two half batches, and the forward /
    data gradient of a half equals the matching half of the whole (same kernels, no CPU reference needed at this size)."""
    from deepatlas_amd import ops
    gen = torch.Generator().manual_seed(7)
    def r(shape, scale=1.0):
        return ((torch.rand(shape, generator=gen) * 2 - 1) * scale).to(dev())
    def run(fn, x, w, b, go):
        xg, wg, bg = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = fn(xg, wg, bg)
        y.backward(go)
        return y.detach(), xg.grad, wg.grad, bg.grad
    def rel(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm())
    for name, fn, xs, ws, gs in [
            ('deconv', ops.DeconvK2S2Fn.apply, (8, 80, 96, 80, 32), (32, 32, 2, 2, 2), (8, 160, 192, 160, 32)),
            ('head', ops.Conv1x1Fn.apply, (8, 160, 192, 160, 16), (32, 16, 1, 1, 1), (8, 160, 192, 160, 32))]:
        x, w, b = r(xs).permute(0, 4, 1, 2, 3), r(ws, 0.2), r((32,), 0.1)
        go = r(gs).permute(0, 4, 1, 2, 3)
        assert go.numel() * 4 > 2 ** 32
        y, dx, dw, db = run(fn, x, w, b, go)
        ya, dxa, dwa, dba = run(fn, x[:4], w, b, go[:4])
        yb, dxb, dwb, dbb = run(fn, x[4:], w, b, go[4:])
        assert torch.equal(y[:4], ya) and torch.equal(y[4:], yb), name
        assert torch.equal(dx[:4], dxa) and torch.equal(dx[4:], dxb), name
        assert rel(dw, dwa + dwb) < 1e-5, (name, rel(dw, dwa + dwb))
        assert rel(db, dba + dbb) < 1e-5, (name, rel(db, dba + dbb))
        del x, go, y, dx, ya, yb, dxa, dxb
        torch.cuda.empty_cache()


def test_full_size_conv_linearity_and_adjointness():
    """Full BASELINE size (batch 1, 160x192x160), the dominant layer 48 -> 16 (two-pointer 32 + 16 input) on the MFMA path:
    linearity conv(a x + b y) = a conv(x) + b conv(y) and the adjoint identities  <conv(x), g> = <x, dgrad(g)> = <w, wgrad(x, g)>
    -- size-independent checks that tie forward, data gradient and weight gradient together without a CPU reference."""
    from deepatlas_amd import ops
    N, D, H, W = 1, 160, 192, 160
    gen = torch.Generator().manual_seed(230)
    def r(shape, scale=1.0):
        return ((torch.rand(shape, generator=gen) * 2 - 1) * scale).to(dev())
    x1, x2, y1, y2 = r((N, D, H, W, 32)).permute(0, 4, 1, 2, 3), r((N, D, H, W, 16)).permute(0, 4, 1, 2, 3), r((N, D, H, W, 32)).permute(0, 4, 1, 2, 3), r((N, D, H, W, 16)).permute(0, 4, 1, 2, 3)
    w = r((16, 48, 3, 3, 3), 0.1)
    f = lambda a, b: ops.Conv3dK3Fn.apply(a, b, w, None, 1, -1.0)
    lhs = f(0.5 * x1 - 2.0 * y1, 0.5 * x2 - 2.0 * y2)
    rhs = 0.5 * f(x1, x2) - 2.0 * f(y1, y2)
    assert float((lhs - rhs).norm() / rhs.norm()) < 1e-5
    xa, xb, wg = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True), w.clone().requires_grad_(True)
    out = ops.Conv3dK3Fn.apply(xa, xb, wg, None, 1, -1.0)
    g = r((N, D, H, W, 16)).permute(0, 4, 1, 2, 3)
    out.backward(g)
    inner = float((out.detach().double() * g.double()).sum())
    via_dgrad = float((x1.double() * xa.grad.double()).sum() + (x2.double() * xb.grad.double()).sum())
    via_wgrad = float((w.double() * wg.grad.double()).sum())
    assert abs(inner - via_dgrad) <= 1e-4 * abs(inner) + 1e-2, (inner, via_dgrad)
    assert abs(inner - via_wgrad) <= 1e-4 * abs(inner) + 1e-2, (inner, via_wgrad)


def _prim():
    """oracle/prim.c (plain C, double accumulation): an ATen-independent checker.  Prebuilt by __graft_entry__.build()."""
    import ctypes, os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so, src = os.path.join(root, 'oracle', 'libprim.so'), os.path.join(root, 'oracle', 'prim.c')
    if not os.path.isfile(so):
        subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-std=c99', src, '-lm', '-o', so])
    lib = ctypes.CDLL(so)
    P, I = ctypes.c_void_p, ctypes.c_int
    lib.prim_conv3d_k3.argtypes = [P, P, P, P, I, I, I, I, I, I, I]; lib.prim_conv3d_k3.restype = None
    lib.prim_deconv_k2s2.argtypes = [P, P, P, P, I, I, I, I, I, I]; lib.prim_deconv_k2s2.restype = None
    lib.prim_grid_sample3d.argtypes = [P, P, P, I, I, I, I, I, I, I, I]; lib.prim_grid_sample3d.restype = None
    return lib


def _np_ptr(a):
    import ctypes
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize('case', [(32, 16, 16, 1, (1, 8, 9, 20)), (16, 0, 32, 2, (1, 9, 10, 11)), (8, 16, 3, 1, (1, 6, 8, 18)),
                                  (1, 1, 16, 1, (1, 6, 8, 10)), (3, 0, 24, 1, (1, 5, 9, 17))],
                         ids=lambda c: 'c%d+%d_o%d_s%d' % c[:4])
def test_conv3d_vs_c_oracle(case):
    """HIP forward against the plain-C double-accumulation conv (no ATen on either side)."""
    from deepatlas_amd import ops
    C1, C2, Cout, stride, (N, D, H, W) = case
    x1 = rnd((N, C1, D, H, W), 11)
    x2 = rnd((N, C2, D, H, W), 12) if C2 else None
    w, b = rnd((Cout, C1 + C2, 3, 3, 3), 13, 0.2), rnd((Cout,), 14, 0.1)
    xin = np.ascontiguousarray((torch.cat((x1, x2), 1) if C2 else x1).numpy())
    Do, Ho, Wo = (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1
    yr = np.empty((N, Cout, Do, Ho, Wo), np.float32)
    wn, bn = np.ascontiguousarray(w.numpy()), np.ascontiguousarray(b.numpy())
    _prim().prim_conv3d_k3(_np_ptr(xin), _np_ptr(wn), _np_ptr(bn), _np_ptr(yr), N, C1 + C2, D, H, W, Cout, stride)
    yg = ops.Conv3dK3Fn.apply(cl(x1), cl(x2) if C2 else None, w.to(dev()), b.to(dev()), stride, -1.0)
    check(yg, yr, tol=1e-5, what='fwd vs prim.c')


def test_deconv_and_warp_vs_c_oracle():
    from deepatlas_amd import ops
    N, Cin, Cout, D, H, W = 1, 32, 32, 3, 4, 6
    x, w, b = rnd((N, Cin, D, H, W), 21), rnd((Cin, Cout, 2, 2, 2), 22, 0.3), rnd((Cout,), 23, 0.1)
    yr = np.empty((N, Cout, 2 * D, 2 * H, 2 * W), np.float32)
    xn, wn, bn = (np.ascontiguousarray(t.numpy()) for t in (x, w, b))
    _prim().prim_deconv_k2s2(_np_ptr(xn), _np_ptr(wn), _np_ptr(bn), _np_ptr(yr), N, Cin, D, H, W, Cout)
    check(ops.DeconvK2S2Fn.apply(cl(x), w.to(dev()), b.to(dev())), yr, tol=1e-5, what='deconv vs prim.c')
    # warp: identity + displacement, sampled by the C grid_sample
    N, C, D, H, W = 1, 8, 6, 9, 11
    src, disp = rnd((N, C, D, H, W), 24), rnd((N, 3, D, H, W), 25, 0.5)
    out, deform = ops.WarpFn.apply(cl(src), cl(disp))
    grid = np.ascontiguousarray(deform.detach().cpu().permute(0, 2, 3, 4, 1).contiguous().numpy())
    sn = np.ascontiguousarray(src.numpy())
    wr = np.empty_like(sn)
    _prim().prim_grid_sample3d(_np_ptr(sn), _np_ptr(grid), _np_ptr(wr), N, C, D, H, W, D, H, W)
    check(out, wr, tol=1e-5, what='warp vs prim.c')


# ---- label-map eval metrics on the device -----------------------------------------------------------------
def test_label_metrics_golden(golden):
    """get_multiclass_dice / DiceLossOnLabel / get_multi_metric through da_label_overlap_counts vs the reference's own outputs."""
    from deepatlas_amd.lib import evalMetrics as em
    from deepatlas_amd.lib.loss import DiceLossOnLabel
    g = golden('eval')
    pred, truth = T(g['eval/pred']).to(dev()), T(g['eval/truth']).to(dev())
    np.testing.assert_allclose(em.get_multiclass_dice(pred, truth, n_class=5).cpu().numpy(), g['eval/multiclass_dice_n5'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(em.get_multiclass_dice(pred, truth).cpu().numpy(), g['eval/multiclass_dice_auto'], rtol=1e-6, atol=1e-7)
    from deepatlas_amd import ops
    oh = ops.one_hot(truth.reshape(2, 1, *truth.shape[1:]).long(), 5)
    np.testing.assert_allclose(em.get_multiclass_dice(pred, oh, n_class=5).cpu().numpy(), g['eval/multiclass_dice_onehot_truth'], rtol=1e-6, atol=1e-7)
    for wt in ('Uniform', 'Simple'):
        v = DiceLossOnLabel(n_class=5)(pred[:, None], truth[:, None], weight_type=wt).item()
        assert abs(v - float(g['eval/dice_on_label_%s' % wt])) < 1e-6, (wt, v)
    assert abs(DiceLossOnLabel()(pred[:, None], truth[:, None]).item() - float(g['eval/dice_on_label_auto'])) < 1e-6
    for tag, kw in (('all', {}), ('rm_bg', {'rm_bg': True}), ('sel', {'eval_label_list': [1, 3]})):
        r = em.get_multi_metric(pred, truth, **kw)
        assert list(r['label_list']) == list(g['eval/multi_metric/%s/label_list' % tag])
        for grp in ('multi_metric_res', 'label_avg_res', 'batch_avg_res'):
            for m, v in r[grp].items():
                np.testing.assert_allclose(v, g['eval/multi_metric/%s/%s/%s' % (tag, grp, m)], rtol=1e-12, atol=0, equal_nan=True)


@pytest.mark.parametrize('dtype', [torch.uint8, torch.int64])
def test_label_overlap_counts_bit_exact(dtype):
    """ragged length (not a multiple of the 16-voxel run), 200 classes, out-of-range labels ignored; vs numpy, bit-exact."""
    from deepatlas_amd import ops
    g = torch.Generator().manual_seed(7)
    N, V, C = 3, 16 * 1000 + 5, 200
    pred = torch.randint(0, 220, (N, V), generator=g).to(dtype)
    truth = torch.randint(0, 220, (N, V), generator=g).to(dtype)
    truth[:, 1000:3000] = 7; pred[:, 1500:2500] = 7                   # long runs (merged in registers)
    c = ops.label_overlap_counts(pred.to(dev()), truth.to(dev()), C).cpu().numpy()
    p, t = pred.numpy().astype(np.int64), truth.numpy().astype(np.int64)
    for n in range(N):
        ref = np.stack([np.bincount(p[n][p[n] < C], minlength=C), np.bincount(t[n][t[n] < C], minlength=C),
                        np.bincount(p[n][(p[n] == t[n]) & (p[n] < C)], minlength=C)], 1)
        assert np.array_equal(c[n], ref)


def test_metric_eval_label_map_forms_golden(golden):
    """lib/evalMetrics.py:17-100 metricEval('iou' | 'dice' | 'recall' | 'precision') on label maps / binary class masks, served by
    da_label_overlap_counts, vs the reference's own outputs (NaN = the reference raised ZeroDivisionError: so must we)."""
    from deepatlas_amd.lib import evalMetrics as em
    g = golden('eval')
    pred, truth = g['eval/metricEval/pred'].astype(np.int64), g['eval/truth'].astype(np.int64)
    for b in range(2):
        assert abs(em.metricEval('iou', pred[b], torch.from_numpy(truth[b]).cuda(), 5) - float(g['eval/metricEval/iou_n5/%d' % b])) < 1e-12
        for m in ('dice', 'recall', 'precision'):
            ref = g['eval/metricEval/%s_binary/%d' % (m, b)]
            for c in range(1, 5):
                if np.isnan(ref[c - 1]) and m != 'dice':
                    with pytest.raises(ZeroDivisionError):
                        em.metricEval(m, pred[b] == c, truth[b] == c, num_labels=2)
                else:
                    v = em.metricEval(m, pred[b] == c, truth[b] == c, num_labels=2)
                    assert (np.isnan(v) and np.isnan(ref[c - 1])) or abs(v - ref[c - 1]) < 1e-12, (m, b, c, v, ref[c - 1])
    with pytest.raises(NotImplementedError):
        em.metricEval('recall', pred[0], truth[0], num_labels=5)
    with pytest.raises(ValueError):
        em.metricEval('f1', pred[0], truth[0], num_labels=2)


def test_full_size_label_metrics_properties():
    """160x192x160: Dice(x, x) = 1 for every present class, counts conserve the voxel count, multi-metric of identical maps = 1."""
    from deepatlas_amd.lib import evalMetrics as em
    from deepatlas_amd.lib.datasets import structured_labels
    from deepatlas_amd import ops
    lab = structured_labels((160, 192, 160), 32).to(dev())[None]
    c = ops.label_overlap_counts(lab, lab, 32)
    assert int(c[0, :, 0].sum()) == 160 * 192 * 160 and torch.equal(c[..., 0], c[..., 2]) and torch.equal(c[..., 1], c[..., 2])
    d = em.get_multiclass_dice(lab, lab, n_class=32)
    present = c[0, 1:, 1] > 0
    assert torch.allclose(d[0][present], torch.ones_like(d[0][present]), atol=1e-6)
    r = em.get_multi_metric(lab, lab)
    assert np.allclose(r['label_avg_res']['dice'], 1.0, atol=1e-9) and np.allclose(r['batch_avg_res']['iou'], 1.0, atol=1e-9)


# ---- LNCC / gradient losses ------------------------------------------------------------------------------------
def test_lncc_golden(golden):
    from deepatlas_amd.lib.loss import get_loss_function
    g = golden('reglosses')
    for fs in (9, 5):
        I = T(g['lncc/I']).to(dev()).requires_grad_(True); J = T(g['lncc/J']).to(dev()).requires_grad_(True)
        crit = get_loss_function('lncc')(filter_size=fs).to(dev())
        assert list(crit.state_dict().keys()) == ['filter']
        l = crit(I, J); l.backward()
        ref = float(g['lncc/f%d/loss' % fs])
        assert abs(l.item() - ref) < 1e-4 * max(1.0, abs(ref)), (fs, l.item(), ref)
        check(I.grad, g['lncc/f%d/grad_I' % fs], tol=2e-4, what='lncc grad_I f%d' % fs)
        check(J.grad, g['lncc/f%d/grad_J' % fs], tol=2e-4, what='lncc grad_J f%d' % fs)


def test_lncc_vs_oracle_ragged_and_errors():
    from deepatlas_amd.lib.loss import VoxelMorphLNCC
    from oracle import losses
    I, J = rnd((1, 1, 11, 13, 17), 1) * 0.5 + 0.5, rnd((1, 1, 11, 13, 17), 2) * 0.5 + 0.5
    Ir, Jr = I.clone().requires_grad_(True), J.clone().requires_grad_(True)
    lr = losses.lncc_loss(Ir, Jr, 9); lr.backward()
    Ig, Jg = I.to(dev()).requires_grad_(True), J.to(dev()).requires_grad_(True)
    lg = VoxelMorphLNCC()(Ig, Jg); lg.backward()
    assert abs(lg.item() - lr.item()) < 1e-4
    check(Ig.grad, Ir.grad, tol=2e-4, what='grad_I'); check(Jg.grad, Jr.grad, tol=2e-4, what='grad_J')
    with pytest.raises(RuntimeError):
        VoxelMorphLNCC()(rnd((1, 1, 8, 16, 16), 3).to(dev()), rnd((1, 1, 8, 16, 16), 4).to(dev()))     # window larger than D (F.conv3d raises too)
    with pytest.raises(ValueError):
        VoxelMorphLNCC()(rnd((1, 2, 12, 16, 16), 3).to(dev()), rnd((1, 2, 12, 16, 16), 4).to(dev()))


def test_lncc_marching_form_chunks_and_partial_tiles():
    """The z-marching LNCC kernels (dilation 1, stride 1, F = 9 / 5) on a volume that needs several z chunks, partial x / y tiles, two samples
    and one-sided gradients: loss and gradients against the oracle's conv3d restatement of lib/loss.py:599-617."""
    from deepatlas_amd.lib.loss import VoxelMorphLNCC
    from oracle import losses
    for fs, shape in ((9, (2, 1, 61, 45, 70)), (5, (1, 1, 23, 37, 33))):
        I, J = rnd(shape, 5) * 0.5 + 0.5, rnd(shape, 6) * 0.5 + 0.5
        Ir, Jr = I.clone().requires_grad_(True), J.clone().requires_grad_(True)
        lr = losses.lncc_loss(Ir, Jr, fs); lr.backward()
        Ig, Jg = I.to(dev()).requires_grad_(True), J.to(dev()).requires_grad_(True)
        lg = VoxelMorphLNCC(filter_size=fs)(Ig, Jg); lg.backward()
        assert abs(lg.item() - lr.item()) < 1e-4
        check(Ig.grad, Ir.grad, tol=2e-4, what='grad_I f%d' % fs); check(Jg.grad, Jr.grad, tol=2e-4, what='grad_J f%d' % fs)
        Ig2 = I.to(dev()).requires_grad_(True)
        VoxelMorphLNCC(filter_size=fs)(Ig2, J.to(dev())).backward()            # only dI asked for
        assert torch.equal(Ig2.grad, Ig.grad)


def test_gradient_loss_golden(golden):
    from deepatlas_amd.lib.loss import get_loss_function
    g = golden('reglosses')
    for tag, kw in (('L2', {}), ('L2_spacing', {'spacing': (1.0, 2.0, 1.5)}), ('L2_nonorm', {'spacing': (1.0, 2.0, 1.5), 'normalize': False}), ('L1', {'norm': 'L1'})):
        u = cl(T(g['gradloss/u'])).requires_grad_(True)
        l = get_loss_function('gradient')(**kw)(u); l.backward()
        ref = float(g['gradloss/%s/loss' % tag])
        assert abs(l.item() - ref) < 1e-5 * max(1.0, abs(ref)), (tag, l.item(), ref)
        check(u.grad, g['gradloss/%s/grad' % tag], tol=1e-5, what='gradloss grad ' + tag)


def test_full_size_lncc_and_gradient_loss_properties():
    """160x192x160: LNCC(x, x) = 0 up to eps; gradientLoss of a constant field has the closed form of its +/- quirk."""
    from deepatlas_amd.lib.loss import VoxelMorphLNCC, gradientLoss
    g = torch.Generator().manual_seed(3)
    x = torch.rand((1, 1, 160, 192, 160), generator=g).to(dev())
    assert abs(VoxelMorphLNCC()(x, x).item()) < 1e-4
    c = 0.25
    u = torch.full((1, 3, 160, 192, 160), c).to(dev()).contiguous(memory_format=torch.channels_last_3d)
    dims = torch.tensor([160., 192., 160.]) / 160.
    expect = float((((dims ** 2) * (2 * c) ** 2).mean() * 2) / 3.0)          # dx = 0; dy = dz = |2c|, weights dims[c]^2
    assert abs(gradientLoss()(u).item() - expect) < 1e-5 * expect


# ---- device data path --------------------------------------------------------------------------------------------
def test_datapath_golden_bit_exact(golden):
    """SitkToTensor clamp/cast, CropTensor, Partition tiles (reflect padding) and both assemble modes: bit-equal to the reference."""
    from deepatlas_amd.lib import transforms as TR
    g = golden('datapath')
    s = TR.SitkToTensor()({'image': g['dp/img'].copy(), 'segmentation': g['dp/seg'].copy()})
    assert s['image'].is_cuda and s['image'].dtype == torch.float32 and s['segmentation'].dtype == torch.uint8
    assert np.array_equal(s['image'].cpu().numpy(), g['dp/totensor/image']) and np.array_equal(s['segmentation'].cpu().numpy(), g['dp/totensor/seg'])
    for tag, cs in (('c3', [1, 2, 3]), ('c6', [1, 0, 2, 3, 1, 0])):
        c = TR.CropTensor(cs)({'image': s['image'].clone(), 'segmentation': s['segmentation'].clone()})
        assert c['image'].is_contiguous()
        assert np.array_equal(c['image'].cpu().numpy(), g['dp/crop/%s/image' % tag]) and np.array_equal(c['segmentation'].cpu().numpy(), g['dp/crop/%s/seg' % tag])
    with pytest.raises(ValueError):
        TR.CropTensor([1, 2])
    for tag, tile, ov in (('a', (8, 8, 8), (2, 2, 2)), ('b', (9, 7, 6), (1, 2, 0))):
        part = TR.Partition(tile, ov, mode='eval')
        p = part({'image': g['dp/img'].astype(np.float32), 'segmentation': g['dp/seg'].astype(np.uint8), 'name': 'x'})
        assert np.array_equal(p['image'].cpu().numpy(), g['dp/part/%s/image' % tag])
        assert np.array_equal(p['segmentation'].cpu().numpy(), g['dp/part/%s/seg' % tag])
        assert np.array_equal(part.assemble(p['segmentation'][:, 0]).cpu().numpy().astype(np.float64), g['dp/part/%s/assemble' % tag])
        v = part.assemble(T(g['dp/part/%s/noisy' % tag]).to(dev()), is_vote=True)
        assert np.array_equal(v.cpu().numpy(), g['dp/part/%s/assemble_vote' % tag])


def test_datapath_full_size_round_trip():
    """160x192x160: partition -> assemble is the identity for any tile / overlap, and the vote of consistent tiles too."""
    from deepatlas_amd.lib import transforms as TR
    from deepatlas_amd.lib.datasets import structured_labels
    lab = structured_labels((160, 192, 160), 32).to(dev())
    img = torch.rand((160, 192, 160), generator=torch.Generator().manual_seed(5)).to(dev())
    part = TR.Partition((72, 72, 72), (4, 4, 4), mode='eval')
    p = part({'image': img, 'segmentation': lab, 'name': 'x'})
    assert p['image'].shape[1:] == (1, 72, 72, 72) and p['image'].shape[0] == 3 * 3 * 3
    assert torch.equal(part.assemble(p['image'][:, 0]), img)
    assert torch.equal(part.assemble(p['segmentation'][:, 0]), lab)
    assert torch.equal(part.assemble(p['segmentation'][:, 0], is_vote=True), lab)


def test_lncc_multiscale_golden(golden):
    """LNCCLoss (lib/loss.py:512-586): dilated / strided box windows at 1, 2 and 3 scales vs the reference's outputs.  The window
    variances are differences of large sums (up to 33^3 terms), so the reference's own fp32 gradients deviate from fp64 by up to
    1.4e-3 on the two-scale case; the device is compared with the fp64 twin within 3 x that deviation (never tighter than 5e-4)."""
    from deepatlas_amd.lib.loss import LNCCLoss
    from oracle import nets
    from conftest import summary_of
    g = golden('reglosses')
    for tag, shp in (('s1', (1, 1, 20, 24, 28)), ('s2', (1, 1, 66, 68, 70)), ('s3', (1, 1, 130, 132, 134))):
        A = nets.closed_form_volume(shp, seed=63).to(dev()).requires_grad_(True)
        B = nets.closed_form_volume(shp, seed=64).to(dev()).requires_grad_(True)
        l = LNCCLoss()(A, B); l.backward()
        ref = float(g['lncc_ms/%s_f64/loss' % tag])
        assert abs(l.item() - ref) < max(1e-5, 10 * abs(float(g['lncc_ms/%s/loss' % tag]) - ref)), (tag, l.item(), ref)
        for t, key in ((A, 'grad_I'), (B, 'grad_J')):
            s_, r32, r64 = summary_of(t.grad), g['lncc_ms/%s/%s' % (tag, key)], g['lncc_ms/%s_f64/%s' % (tag, key)]
            floor = max(abs(r32[2] - r64[2]) / r64[2], rel_l2(r32[5:], r64[5:]))
            assert abs(s_[2] - r64[2]) / r64[2] < max(5e-4, 3 * floor), (tag, key, s_[2], r64[2], floor)
            assert rel_l2(s_[5:], r64[5:]) < max(5e-4, 3 * floor), (tag, key, floor)


def test_warp_labels_equals_warp_of_onehot():
    """da_warp_labels_{fwd,bwd} == warp(one_hot(labels)) and its displacement gradient (uint8 and int64 labels, out-of-volume taps)."""
    from deepatlas_amd import ops
    N, C, D, H, W = 2, 8, 6, 10, 12
    g = torch.Generator().manual_seed(9)
    lab = torch.randint(0, C, (N, D, H, W), generator=g)
    disp = rnd((N, 3, D, H, W), 2, 0.6)
    go = rnd((N, C, D, H, W), 3)
    for dt in (torch.uint8, torch.int64):
        l = lab.to(dt).to(dev())
        d1 = cl(disp).requires_grad_(True)
        w1, _ = ops.WarpFn.apply(ops.one_hot(l.unsqueeze(1), C), d1); (w1 * cl(go)).sum().backward()
        d2 = cl(disp).requires_grad_(True)
        w2 = ops.WarpLabelsFn.apply(l, d2, C); (w2 * cl(go)).sum().backward()
        check(w2, w1, tol=1e-6, what='label warp fwd'); check(d2.grad, d1.grad, tol=1e-5, what='label warp grad_disp')


# ---- round 2: registry cross-entropy family, bending norm != 'L2', SegMaskToOneHot, device synthetic generator -------------
def test_registry_cross_entropy_family_golden(golden):
    """'cross_entropy' / 'focal' / 'soft_cross_entropy' of the loss registry (lib/loss.py:739-761) on xent.hip against the
    reference's own outputs (tests/golden/registry_losses.npz): loss and the gradient with respect to the predictions."""
    from deepatlas_amd.lib.loss import get_loss_function
    g = golden('registry_losses')
    y = T(g['xent/labels']).to(dev())

    def run(tag, crit, pred, target):
        p = cl(T(pred)).requires_grad_(True)
        l = crit(p, target)
        l.backward()
        ref = float(g[f'xent/{tag}/loss'])
        assert abs(l.item() - ref) < 1e-5 * max(1.0, abs(ref)), (tag, l.item(), ref)
        check(p.grad, g[f'xent/{tag}/grad'], what=tag + ' grad')

    CE, FL, SCE = get_loss_function('cross_entropy'), get_loss_function('focal'), get_loss_function('soft_cross_entropy')
    run('ce_mean', CE(), g['xent/logits'], y)
    run('ce_mean', CE(), g['xent/logits'], y.long())                       # int64 targets (models/segmentation.py:154) as well as uint8
    run('ce_sum', CE(reduction='sum'), g['xent/logits'], y)
    run('ce_ignore', CE(ignore_index=2), g['xent/logits'], y)
    run('focal_default', FL(5), g['xent/logits'], y)
    run('focal_alpha_g15_sum', FL(5, alpha=T(g['xent/alpha']), gamma=1.5, size_average=False), g['xent/logits'], y)
    run('focal_nosoftmax', FL(5, soft_max=False), g['xent/prob'], y)
    t = cl(T(g['xent/soft_target']))
    run('soft_softmax', SCE(n_class=5, softmax=True), g['xent/logits'], t)
    run('soft_nosoftmax', SCE(n_class=5, softmax=False), g['xent/prob_clamped_in'], t)
    with pytest.raises(RuntimeError):                                       # the reference's index-target branch fails to broadcast for B > 1
        SCE(n_class=5, softmax=True)(cl(T(g['xent/logits'])), y.long())
    # ... and at the reference's batch size 1 it broadcasts the label VALUE over the class axis (lib/loss.py:151): same expression on torch-CPU
    x1 = T(g['xent/logits'])[:1].clone()
    y1 = y[:1].long().cpu()
    xr = x1.clone().requires_grad_(True)
    lr_ = torch.mean(torch.sum(-y1 * torch.nn.functional.log_softmax(xr, 1), 1))
    lr_.backward()
    xg = cl(x1).requires_grad_(True)
    lg = SCE(n_class=5, softmax=True)(xg, y1.to(dev()))
    lg.backward()
    assert abs(lg.item() - lr_.item()) < 1e-5 * max(1.0, abs(lr_.item()))
    check(xg.grad, xr.grad, what='soft cross entropy, index target at B = 1')


def test_cross_entropy_c32_vs_torch():
    """32 classes (the headline head), ragged voxel count, against torch-CPU's own CrossEntropyLoss and the oracle's focal restatement."""
    from oracle import losses
    from deepatlas_amd.lib.loss import get_loss_function
    x = rnd((2, 32, 5, 7, 9), 1, 3.0)
    y = torch.randint(0, 32, (2, 5, 7, 9), generator=torch.Generator().manual_seed(2))
    xr = x.clone().requires_grad_(True)
    lr_ = torch.nn.CrossEntropyLoss()(xr, y)
    lr_.backward()
    xg = cl(x).requires_grad_(True)
    lg = get_loss_function('cross_entropy')()(xg, y.to(dev()))
    lg.backward()
    assert abs(lg.item() - lr_.item()) < 1e-5
    check(xg.grad, xr.grad, what='CE C=32 grad')
    xr2 = x.clone().requires_grad_(True)
    lf = losses.focal_loss(xr2, y, 32)
    lf.backward()
    xg2 = cl(x).requires_grad_(True)
    lg2 = get_loss_function('focal')(32)(xg2, y.to(dev()))
    lg2.backward()
    assert abs(lg2.item() - lf.item()) < 1e-5 * max(1.0, abs(lf.item()))
    check(xg2.grad, xr2.grad, what='focal C=32 grad')


def test_bending_energy_other_norm_golden(golden):
    """BendingEnergyLoss(norm != 'L2') (lib/loss.py:721-729: the weighting block is skipped, plain mean of |differences|)."""
    from deepatlas_amd.lib.loss import BendingEnergyLoss
    g = golden('registry_losses')
    for tag, kw in (('L1', {}), ('L1_spacing', {'spacing': (1.0, 2.0, 1.5)})):
        u = cl(T(g['bendL1/u'])).requires_grad_(True)
        l = BendingEnergyLoss(norm='L1', **kw)(u)
        l.backward()
        assert abs(l.item() - g[f'bendL1/{tag}/loss']) < 1e-5 * abs(g[f'bendL1/{tag}/loss']), tag
        check(u.grad, g[f'bendL1/{tag}/grad'], what='bending L1 grad ' + tag)


def test_seg_mask_to_one_hot_golden_bit_exact(golden):
    """transforms.SegMaskToOneHot (lib/transforms.py:652-673) on the device, bit-equal to the reference's output."""
    from deepatlas_amd.lib.transforms import SegMaskToOneHot
    g = golden('registry_losses')
    seg = T(g['onehot/seg']).to(dev())
    out = SegMaskToOneHot(4)({'segmentation': seg})
    assert out['segmentation'] is seg
    oh = out['segmentation_onehot']
    assert oh.dtype == torch.float32 and tuple(oh.shape) == (4,) + tuple(seg.shape)
    assert np.array_equal(oh.cpu().numpy(), g['onehot/segmentation_onehot'])


@pytest.mark.parametrize('structured', [False, True])
def test_synthetic_volume_generator_bit_exact_vs_oracle(structured):
    """SURVEY.md row f4 "synthetic-volume generator on device": da_synth_volume against its numpy restatement, bit for bit, including a
    sample offset (rank r draws samples r*n ...), plus the full-size volume's value ranges."""
    from oracle import datapath as dp
    from deepatlas_amd.lib.datasets import synthetic_batch_on_device
    shape = (9, 16, 23)
    img, lab = synthetic_batch_on_device(3, shape, 32, seed=230, device=dev(), structured=structured, sample0=2)
    ri, rl = dp.synth_volume(3, shape, 32, 1 if structured else 0, 0.1, 230, sample0=2)
    assert np.array_equal(lab.cpu().numpy(), rl)
    assert np.array_equal(img.cpu().numpy()[:, 0], ri)
    big, bl = synthetic_batch_on_device(2, (160, 192, 160), 32, seed=7, device=dev(), structured=structured)
    assert float(big.min()) >= 0.0 and float(big.max()) <= 1.0 and int(bl.max()) == 31 and int(bl.min()) == 0
    if not structured:
        assert abs(float(big.mean()) - 0.5) < 1e-3
        assert np.array_equal(big[1, 0, :2, :3].cpu().numpy(), dp.synth_volume(2, (160, 192, 160), 32, 0, 0.1, 7)[0][1, :2, :3])


@pytest.mark.parametrize('slope,Cout', [(0.0, 16), (0.01, 32), (-1.0, 3)])
def test_conv3d_forked_output_sums_its_two_gradients_in_the_activation_backward(slope, Cout):
    """Conv3dK3Fn(fork=True) hands out two aliases of its output (the registration net's skip connections, voxel_morph.py:66-82); the two
    incoming gradients are summed inside da_act_bwd_add_dbias together with act' and the bias gradient.  Against torch-CPU where the
    output is simply used twice."""
    from deepatlas_amd import ops
    C1, C2, N, D, H, W = 8, 16, 1, 6, 9, 18
    x1, x2 = rnd((N, C1, D, H, W), 1), rnd((N, C2, D, H, W), 2)
    w, b = rnd((Cout, C1 + C2, 3, 3, 3), 3, 0.2), rnd((Cout,), 4, 0.1)
    xr1, xr2, wr, br = (t.clone().requires_grad_(True) for t in (x1, x2, w, b))
    yr = F.conv3d(torch.cat((xr1, xr2), 1), wr, br, padding=1)
    if slope >= 0:
        yr = F.leaky_relu(yr, slope) if slope > 0 else F.relu(yr)
    ga, gb = rnd(tuple(yr.shape), 5), rnd(tuple(yr.shape), 6)
    ((yr * ga).sum() + (yr * gb).sum()).backward()
    xg1, xg2 = cl(x1).requires_grad_(True), cl(x2).requires_grad_(True)
    wg, bg = w.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
    ya, yb = ops.Conv3dK3Fn.apply(xg1, xg2, wg, bg, 1, slope, False, True)
    assert ya.data_ptr() == yb.data_ptr()
    ((ya * cl(ga)).sum() + (yb * cl(gb)).sum()).backward()
    check(ya, yr, what='fwd'); check(xg1.grad, xr1.grad, what='dgrad1'); check(xg2.grad, xr2.grad, what='dgrad2')
    check(wg.grad, wr.grad, what='wgrad'); check(bg.grad, br.grad, what='bgrad')


@pytest.mark.parametrize('Cin,C,N,dims,pro,wt,no_bg', [(16, 32, 2, (5, 7, 9), False, 'Uniform', False), (16, 32, 1, (8, 8, 16), True, 'Simple', True),
                                                        (16, 16, 1, (3, 5, 7), True, 'Volume', False), (64, 32, 1, (4, 6, 5), False, 'Uniform', False),
                                                        (64, 16, 2, (2, 3, 5), True, 'Uniform', True)])
def test_fused_head_softmax_dice_vs_torch_cpu(Cin, C, N, dims, pro, wt, no_bg):
    """da_head_dice_fwd / _bwd (the 1x1x1 head, softmax and Dice without the logits tensor) against torch-CPU's conv3d + the oracle's
    DiceLossMultiClass restatement: loss, dx, dW, db; ragged voxel counts (not a multiple of the 256-voxel workgroup chunk), batch 2,
    int64 and uint8 labels, every weighting, with and without a deferred BatchNorm + LeakyReLU on the input."""
    from oracle import losses
    from deepatlas_amd import ops
    D, H, W = dims
    x, w, b = rnd((N, Cin, D, H, W), 1, 2.0), rnd((C, Cin, 1, 1, 1), 2, 0.5), rnd((C,), 3, 0.2)
    y = torch.randint(0, C, (N, D, H, W), generator=torch.Generator().manual_seed(4))
    sc, sh = torch.rand(Cin, generator=torch.Generator().manual_seed(5)) + 0.5, torch.rand(Cin, generator=torch.Generator().manual_seed(6)) - 0.5
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    xin = F.leaky_relu(xr * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1), 0.01) if pro else xr
    xin.retain_grad()
    lr_ = losses.dice_loss(F.conv3d(xin, wr, br), y, C, weight_type=wt, no_bg=no_bg, softmax=True, eps=1e-6)
    lr_.backward()
    for labels in (y.to(dev()), y.to(torch.uint8).to(dev())):
        xg, wg, bg = cl(x).requires_grad_(True), w.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
        if pro:
            lg = ops.HeadDiceFn.apply(xg, wg, bg, labels, wt, no_bg, 1e-6, (sc.to(dev()), sh.to(dev()), 0.01))
        else:
            lg = ops.HeadDiceFn.apply(xg, wg, bg, labels, wt, no_bg, 1e-6)
        lg.backward()
        assert abs(lg.item() - lr_.item()) < 1e-5, (lg.item(), lr_.item())
        check(xg.grad, xin.grad, what='dx (wrt the activated input)')
        check(wg.grad, wr.grad, what='dW'); check(bg.grad, br.grad, what='db')


def test_fused_head_softmax_dice_full_size_matches_composition():
    """At 2 x 160 x 192 x 160, 16 -> 32: the fused kernels against the op-by-op composition they replace (Conv1x1Fn + DiceFn), loss and
    strided samples of dx / dW / db -- exercises the > 4 GiB-per-tensor addressing (the logits they never build would be 1.26 GB)."""
    from deepatlas_amd import ops
    from deepatlas_amd.lib.datasets import synthetic_batch_on_device
    d = dev()
    shape, C, Cin, N = (160, 192, 160), 32, 16, 2
    _, lab = synthetic_batch_on_device(N, shape, C, seed=9, device=d, structured=True)
    g = torch.Generator(device=d).manual_seed(10)
    x = torch.empty((N,) + shape + (Cin,), device=d).uniform_(-2, 2, generator=g).permute(0, 4, 1, 2, 3)
    w = (torch.rand((C, Cin, 1, 1, 1), generator=torch.Generator().manual_seed(11)) - 0.5).to(d)
    b = (torch.rand((C,), generator=torch.Generator().manual_seed(12)) - 0.5).to(d)
    res = []
    for fused in (False, True):
        xg, wg, bg = x.detach().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        if fused:
            loss = ops.HeadDiceFn.apply(xg, wg, bg, lab, 'Uniform', False, 1e-6)
        else:
            loss = ops.DiceFn.apply(ops.Conv1x1Fn.apply(xg, wg, bg), lab, None, 'Uniform', False, True, 1e-6)
        loss.backward()
        res.append((loss.item(), xg.grad.reshape(-1)[::1009].cpu().numpy(), wg.grad.cpu().numpy(), bg.grad.cpu().numpy()))
        del xg, loss
    assert abs(res[0][0] - res[1][0]) < 1e-6
    for k, what in ((1, 'dx'), (2, 'dW'), (3, 'db')):
        assert rel_l2(res[1][k], res[0][k]) < 1e-4, (what, rel_l2(res[1][k], res[0][k]))

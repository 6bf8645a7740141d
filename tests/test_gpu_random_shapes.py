"""Randomised-shape parity (hypothesis, derandomised so the suite is reproducible): ragged volumes smaller and larger than a tile,
every channel-count class the dispatcher distinguishes (thin, CK = 8 / 16, odd counts -> direct kernels, concat inputs), stride 1 / 2,
against torch-CPU.  Catches tile-edge and dispatch mistakes that fixed cases miss."""
import pytest
import torch
import torch.nn.functional as F
from hypothesis import given, settings, strategies as st, HealthCheck

from test_gpu_ops import rnd, cl, check, dev

pytestmark = pytest.mark.gpu
SET = dict(max_examples=60, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))

chan = st.sampled_from([1, 2, 3, 5, 8, 16, 24, 32, 48, 64])


@settings(**SET)
@given(c1=chan, c2=st.sampled_from([0, 0, 1, 8, 16, 32]), cout=chan, stride=st.sampled_from([1, 1, 1, 2]),
       d=st.integers(1, 11), h=st.integers(1, 19), w=st.integers(1, 35), n=st.integers(1, 2), slope=st.sampled_from([-1.0, 0.0, 0.01]))
def test_conv3d_random_shapes(c1, c2, cout, stride, d, h, w, n, slope):
    from deepatlas_amd import ops
    if stride == 2:
        c2 = 0                                            # the registration encoder's strided convs take a single input
    x1 = rnd((n, c1, d, h, w), 1)
    x2 = rnd((n, c2, d, h, w), 2) if c2 else None
    wt, b = rnd((cout, c1 + c2, 3, 3, 3), 3, 0.2), rnd((cout,), 4, 0.1)
    xr1 = x1.clone().requires_grad_(True); xr2 = x2.clone().requires_grad_(True) if c2 else None
    wr, br = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv3d(torch.cat((xr1, xr2), 1) if c2 else xr1, wr, br, stride=stride, padding=1)
    if slope >= 0:
        yr = F.leaky_relu(yr, slope) if slope > 0 else F.relu(yr)
    go = rnd(tuple(yr.shape), 5); yr.backward(go)
    xg1 = cl(x1).requires_grad_(True); xg2 = cl(x2).requires_grad_(True) if c2 else None
    wg, bg = wt.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
    yg = ops.Conv3dK3Fn.apply(xg1, xg2, wg, bg, stride, slope); yg.backward(cl(go))
    check(yg, yr, what='fwd'); check(xg1.grad, xr1.grad, what='dgrad1'); check(wg.grad, wr.grad, what='wgrad'); check(bg.grad, br.grad, what='bgrad')
    if c2:
        check(xg2.grad, xr2.grad, what='dgrad2')


@settings(**SET)
@given(c=st.sampled_from([1, 3, 4, 8, 16, 32]), d=st.integers(2, 9), h=st.integers(2, 11), w=st.integers(2, 13), n=st.integers(1, 2))
def test_pool_and_upsample_random_shapes(c, d, h, w, n):
    from deepatlas_amd import ops
    x = rnd((n, c, d, h, w), 6)
    xr = x.clone().requires_grad_(True); yr = F.max_pool3d(xr, 2); go = rnd(tuple(yr.shape), 7); yr.backward(go)
    xg = cl(x).requires_grad_(True); yg = ops.MaxPool2Fn.apply(xg); yg.backward(cl(go))
    assert torch.equal(yg.cpu(), yr.detach()) and torch.equal(xg.grad.cpu(), xr.grad)
    xr = x.clone().requires_grad_(True); yr = F.interpolate(xr, scale_factor=2, mode='trilinear', align_corners=False); go = rnd(tuple(yr.shape), 8); yr.backward(go)
    xg = cl(x).requires_grad_(True); yg = ops.UpsampleTrilinear2Fn.apply(xg); yg.backward(cl(go))
    check(yg, yr, tol=1e-6, what='trilinear fwd'); check(xg.grad, xr.grad, tol=1e-6, what='trilinear bwd')
    size = (d + 3, 2 * h - 1, w)                         # non-integer nearest scale factors (voxel_morph.py:72-80)
    xr = x.clone().requires_grad_(True); yr = F.interpolate(xr, size=size); go = rnd(tuple(yr.shape), 9); yr.backward(go)
    xg = cl(x).requires_grad_(True); yg = ops.UpsampleNearestFn.apply(xg, size); yg.backward(cl(go))
    assert torch.equal(yg.cpu(), yr.detach())
    check(xg.grad, xr.grad, tol=1e-6, what='nearest bwd')


@settings(**SET)
@given(c=st.sampled_from([1, 2, 8, 16, 32]), d=st.integers(2, 9), h=st.integers(2, 11), w=st.integers(2, 13), amp=st.sampled_from([0.05, 0.5, 3.0]), n=st.integers(1, 2))
def test_warp_random_shapes(c, d, h, w, amp, n):
    """incl. displacements that push most taps out of the volume (zeros padding); c = 8 / 16 / 32 take the grouped gather (warp.hip: 2 / 4 / 8 lanes per voxel),
    two samples its grid.y"""
    from deepatlas_amd import ops
    from oracle import nets
    src, disp = rnd((n, c, d, h, w), 10), rnd((n, 3, d, h, w), 11, amp)
    sr, dr = src.clone().requires_grad_(True), disp.clone().requires_grad_(True)
    wr = nets.warp_trilinear(sr, dr + nets.identity_transform((d, h, w))); go = rnd(tuple(wr.shape), 12); wr.backward(go)
    sg, dg = cl(src).requires_grad_(True), cl(disp).requires_grad_(True)
    wg, _ = ops.WarpFn.apply(sg, dg); wg.backward(cl(go))
    check(wg, wr, tol=2e-5, what='warp fwd'); check(sg.grad, sr.grad, tol=2e-5, what='grad_src')
    if float(dr.grad.norm()) > 0:
        check(dg.grad, dr.grad, tol=1e-4, what='grad_disp')


@settings(**SET)
@given(cin=st.sampled_from([3, 6, 16, 32, 48, 64, 80, 128]), cout=st.sampled_from([3, 5, 16, 32, 64, 96]),
       d=st.integers(1, 5), h=st.integers(1, 6), w=st.integers(1, 9), n=st.integers(1, 2))
def test_pointwise_random_shapes(cin, cout, d, h, w, n):
    """ConvTranspose3d(k2,s2), Conv3d(1x1x1) and (when the channels allow it) Conv3d(k2,s2): MFMA path incl. host-tiled wide
    channels, and the direct kernels for channel counts that are not multiples of 16."""
    from deepatlas_amd import ops
    x, wt, b = rnd((n, cin, d, h, w), 1), rnd((cin, cout, 2, 2, 2), 2, 0.3), rnd((cout,), 3, 0.1)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv_transpose3d(xr, wr, br, stride=2); go = rnd(tuple(yr.shape), 4); yr.backward(go)
    xg, wg, bg = cl(x).requires_grad_(True), wt.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
    yg = ops.DeconvK2S2Fn.apply(xg, wg, bg); yg.backward(cl(go))
    check(yg, yr, what='deconv fwd'); check(xg.grad, xr.grad, what='deconv dgrad'); check(wg.grad, wr.grad, what='deconv wgrad'); check(bg.grad, br.grad, what='deconv bgrad')
    w1 = rnd((cout, cin, 1, 1, 1), 5, 0.3)
    xr, wr, br = x.clone().requires_grad_(True), w1.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, br); go = rnd(tuple(yr.shape), 6); yr.backward(go)
    xg, wg, bg = cl(x).requires_grad_(True), w1.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
    yg = ops.Conv1x1Fn.apply(xg, wg, bg); yg.backward(cl(go))
    check(yg, yr, what='1x1 fwd'); check(xg.grad, xr.grad, what='1x1 dgrad'); check(wg.grad, wr.grad, what='1x1 wgrad'); check(bg.grad, br.grad, what='1x1 bgrad')
    if cin % 16 == 0 and cout % 16 == 0:
        x2, w2 = rnd((n, cin, 2 * d, 2 * h, 2 * w), 7), rnd((cout, cin, 2, 2, 2), 8, 0.3)
        xr, wr, br = x2.clone().requires_grad_(True), w2.clone().requires_grad_(True), b.clone().requires_grad_(True)
        yr = F.conv3d(xr, wr, br, stride=2); go = rnd(tuple(yr.shape), 9); yr.backward(go)
        xg, wg, bg = cl(x2).requires_grad_(True), w2.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
        yg = ops.ConvK2S2Fn.apply(xg, wg, bg); yg.backward(cl(go))
        check(yg, yr, what='k2s2 fwd'); check(xg.grad, xr.grad, what='k2s2 dgrad'); check(wg.grad, wr.grad, what='k2s2 wgrad'); check(bg.grad, br.grad, what='k2s2 bgrad')


@settings(**SET)
@given(c=st.sampled_from([3, 4, 8, 16, 32, 64, 128]), d=st.integers(1, 6), h=st.integers(1, 7), w=st.integers(1, 18), n=st.integers(1, 3),
       slope=st.sampled_from([0.0, 0.01]), training=st.booleans())
def test_bn_act_random_shapes(c, d, h, w, n, slope, training):
    from deepatlas_amd import ops
    if training and n * d * h * w < 2:
        return                                            # F.batch_norm refuses a single value per channel in training mode
    x, g, b = rnd((n, c, d, h, w), 1, 2.0), rnd((c,), 2) * 0.3 + 1.0, rnd((c,), 3, 0.2)
    rm0, rv0 = rnd((c,), 4, 0.1), rnd((c,), 5).abs() + 0.5
    xr, gr, br = x.clone().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    yr = F.batch_norm(xr, rm, rv, gr, br, training, 0.1, 1e-5)
    yr = F.leaky_relu(yr, slope) if slope > 0 else F.relu(yr)
    go = rnd(tuple(yr.shape), 6); yr.backward(go)
    xg, gg, bg = cl(x).requires_grad_(True), g.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
    rmg, rvg = rm0.to(dev()), rv0.to(dev())
    yg = ops.BNActFn.apply(xg, gg, bg, rmg, rvg, training, 0.1, 1e-5, slope); yg.backward(cl(go))
    check(yg, yr, tol=2e-5, what='bn fwd'); check(xg.grad, xr.grad, tol=2e-4, what='bn dx')
    check(gg.grad, gr.grad, tol=2e-4, what='dgamma'); check(bg.grad, br.grad, tol=2e-5, what='dbeta')
    check(rmg, rm, tol=1e-5, what='running_mean'); check(rvg, rv, tol=1e-5, what='running_var')

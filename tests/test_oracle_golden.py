"""CPU: pin the oracle (oracle/*.py restatement) against the golden vectors that
oracle/make_golden.py produced by importing the reference itself."""
import copy
import numpy as np
import torch
import pytest

from oracle import nets, losses, steps
from conftest import summary_of, rel_l2


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_ops_dice_all_weightings(golden):
    g = golden('ops')
    logits = T(g['ops/dice/logits']).requires_grad_(True)
    labels = T(g['ops/dice/labels'])
    C = logits.shape[1]
    for wt in ('Uniform', 'Simple', 'Volume'):
        for no_bg in (False, True):
            l = losses.dice_loss(logits, labels.long(), C, weight_type=wt, no_bg=no_bg, softmax=True, eps=1e-6)
            gr, = torch.autograd.grad(l, logits)
            assert abs(l.item() - g[f'ops/dice/{wt}_{int(no_bg)}/loss']) < 1e-6
            assert rel_l2(gr.numpy(), g[f'ops/dice/{wt}_{int(no_bg)}/grad']) < 1e-5


def test_ops_dice_soft_target(golden):
    g = golden('ops')
    src = T(g['ops/dice_soft/source']).requires_grad_(True)
    l = losses.dice_loss(src, T(g['ops/dice_soft/target']), src.shape[1], softmax=False)
    gr, = torch.autograd.grad(l, src)
    assert abs(l.item() - g['ops/dice_soft/loss']) < 1e-6
    assert rel_l2(gr.numpy(), g['ops/dice_soft/grad']) < 1e-5


def test_ops_onehot_identity(golden):
    g = golden('ops')
    labels = T(g['ops/dice/labels'])
    oh = losses.mask_to_one_hot(labels.view(2, 1, *labels.shape[1:]), 5)
    assert np.array_equal(oh.numpy(), g['ops/onehot'])
    idt = nets.identity_transform(labels.shape[1:])
    assert np.array_equal(idt.numpy(), g['ops/identity'])


def test_ops_ncc_bending(golden):
    g = golden('ops')
    a = T(g['ops/ncc/a']).requires_grad_(True)
    l = losses.ncc_loss(a, T(g['ops/ncc/b']))
    gr, = torch.autograd.grad(l, a)
    assert abs(l.item() - g['ops/ncc/loss']) < 1e-6
    assert rel_l2(gr.numpy(), g['ops/ncc/grad']) < 1e-5
    u = T(g['ops/bending/u']).requires_grad_(True)
    l = losses.bending_energy_loss(u)
    gr, = torch.autograd.grad(l, u)
    assert abs(l.item() - g['ops/bending/loss']) < 1e-7 * max(1, abs(g['ops/bending/loss']))
    assert rel_l2(gr.numpy(), g['ops/bending/grad']) < 1e-5
    l2 = losses.bending_energy_loss(u, spacing=(1.0, 2.0, 1.5))
    assert abs(l2.item() - g['ops/bending/loss_spacing']) < 1e-6 * max(1, abs(g['ops/bending/loss_spacing']))


def test_ops_warp(golden):
    g = golden('ops')
    for nm in ('warp1', 'warpC'):
        src = T(g[f'ops/{nm}/src']).requires_grad_(True)
        disp = T(g[f'ops/{nm}/disp']).requires_grad_(True)
        deform = disp + nets.identity_transform(src.shape[2:])
        w = nets.warp_trilinear(src, deform)
        assert rel_l2(w.detach().numpy(), g[f'ops/{nm}/out']) < 1e-6
        gs, gd = torch.autograd.grad((w * T(g[f'ops/{nm}/gout'])).sum(), (src, disp))
        assert rel_l2(gs.numpy(), g[f'ops/{nm}/grad_src']) < 1e-6
        assert rel_l2(gd.numpy(), g[f'ops/{nm}/grad_disp']) < 1e-6


def test_ops_evaldice(golden):
    g = golden('ops')
    pred, truth = T(g['ops/evaldice/pred']), T(g['ops/evaldice/truth'])
    onehot = losses.mask_to_one_hot(pred.long().view(1, 1, *pred.shape[1:]), 6)   # logits whose argmax == pred
    dice, am = losses.eval_dice_per_class(onehot, truth, 6)
    assert np.array_equal(am.numpy(), pred.numpy().astype(np.int64))
    ref = g['ops/evaldice/dice']
    assert np.array_equal(np.isnan(dice), np.isnan(ref))
    assert np.allclose(dice[~np.isnan(ref)], ref[~np.isnan(ref)], atol=0, rtol=1e-15)


def _seg_setup(spec, n_classes, N):
    shapes = nets.unet_param_shapes(1, n_classes, spec['encoders'], spec['decoders'])
    sd = nets.closed_form_fill(shapes, seed=1)
    x = nets.closed_form_volume((N, 1, 16, 24, 32), seed=2)
    y = nets.closed_form_labels((N, 16, 24, 32), n_classes, seed=3)
    return sd, x, y


def test_seg_tiny_three_steps(golden):
    g = golden('seg_tiny')
    sd, x, y = _seg_setup(nets.UNET_TINY, 5, 2)
    opt = steps.Adam(steps.trainable(sd), lr=1e-3)
    for s in (1, 2, 3):
        loss, logits, grads = steps.seg_step(sd, opt, x, y, nets.UNET_TINY, 5)
        if s == 1:
            assert abs(loss.item() - g['seg_tiny/loss']) < 1e-6
            assert rel_l2(logits.numpy(), g['seg_tiny/logits']) < 1e-6
            assert np.array_equal(torch.max(logits, 1)[1].numpy().astype(np.uint8), g['seg_tiny/argmax'])
            for n in opt.names:
                ref = g[f'seg_tiny/grad/{n}']
                if n.endswith('conv.bias') or n.endswith('deconv.bias'):   # zero-gradient biases in front of BN: absolute
                    assert np.abs(grads[n].numpy() - ref).max() < 1e-7
                else:
                    assert rel_l2(grads[n].numpy(), ref) < 1e-5, n
        if s in (1, 3):
            assert abs(loss.item() - g[f'seg_tiny/loss_step{s}']) < 1e-5
            for n, v in sd.items():
                if not v.dtype.is_floating_point:
                    continue
                ref = g[f'seg_tiny/after{s}/{n}']
                if (n.endswith('conv.bias') or n.endswith('deconv.bias')):
                    # bias in front of a BatchNorm: analytically zero gradient, Adam turns the 1e-9
                    # rounding noise into +-lr steps (SURVEY.md §7 hard parts) -> bounded, not equal
                    assert np.abs(v.numpy() - ref).max() <= 2 * s * 1e-3 + 1e-6, (s, n)
                else:
                    assert rel_l2(v.numpy(), ref) < 2e-5, (s, n)
    # eval-mode forward + eval dice on the reference's own 3-step state (the zero-gradient conv biases
    # random-walk by +-lr under Adam, and eval-mode BN does not cancel them, so load the golden state)
    for n in sd:
        if sd[n].dtype.is_floating_point:
            sd[n] = T(g[f'seg_tiny/after3/{n}']).clone()
    with torch.no_grad():
        pred = nets.unet_forward(sd, x, nets.UNET_TINY, training=False)
    assert rel_l2(pred.numpy(), g['seg_tiny/eval_logits']) < 1e-6
    for b in range(2):
        dice, _ = losses.eval_dice_per_class(pred[b:b + 1], y[b:b + 1], 5)
        ref = g['seg_tiny/eval_dice'][b]
        assert np.allclose(dice, ref, atol=1e-12, equal_nan=True)


def test_seg_tiny_fp64_twin(golden):
    g = golden('seg_tiny')
    sd, x, y = _seg_setup(nets.UNET_TINY, 5, 2)
    sd = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
    opt = steps.Adam(steps.trainable(sd))
    loss, logits, grads = steps.seg_step(sd, opt, x.double(), y, nets.UNET_TINY, 5)
    # the reference casts the per-class sums with .float() (loss.py:473-474), so its "fp64" loss and
    # gradients carry one fp32 rounding; logits are exact
    assert abs(loss.item() - g['seg_tiny_f64/loss']) < 2e-7
    assert rel_l2(logits.numpy(), g['seg_tiny_f64/logits']) < 1e-12
    for n in opt.names:
        if 'conv.bias' not in n:
            assert rel_l2(grads[n].numpy(), g[f'seg_tiny_f64/grad/{n}']) < 1e-5, n


def test_seg_light_first_step(golden):
    g = golden('seg_light')
    sd, x, y = _seg_setup(nets.UNET_LIGHT, 32, 1)
    opt = steps.Adam(steps.trainable(sd))
    loss, logits, grads = steps.seg_step(sd, opt, x, y, nets.UNET_LIGHT, 32)
    assert abs(loss.item() - g['seg_light/loss']) < 1e-6
    assert rel_l2(summary_of(logits)[5:], g['seg_light/logits'][5:]) < 1e-6
    for n in opt.names:
        ref = g[f'seg_light/grad/{n}']
        if n.endswith('conv.bias') and not n.endswith('decBlock2.2.bias'):
            continue
        assert abs(summary_of(grads[n])[2] - ref[2]) <= 1e-4 * ref[2] + 1e-12, n


def _reg_setup(shape):
    sd = nets.closed_form_fill(nets.voxelmorph_param_shapes(), seed=4)
    return sd, nets.closed_form_volume((1, 1) + shape, seed=5), nets.closed_form_volume((1, 1) + shape, seed=6)


@pytest.mark.parametrize('tag,shape', [('reg_odd', (20, 24, 20)), ('reg_even', (16, 24, 32))])
def test_reg_three_steps(golden, tag, shape):
    g = golden('reg')
    sd, src, tgt = _reg_setup(shape)
    opt = steps.Adam(steps.trainable(sd))
    for s in (1, 2, 3):
        loss, (disp, warped, deform), grads, (l_sim, l_reg) = steps.reg_step(sd, opt, src, tgt)
        if s == 1:
            assert abs(loss.item() - g[f'{tag}/loss']) < 1e-6
            assert abs(l_sim.item() - g[f'{tag}/ncc']) < 1e-6
            assert abs(l_reg.item() - g[f'{tag}/bending']) < 1e-6 * max(1, abs(g[f'{tag}/bending']))
            assert rel_l2(disp.numpy(), g[f'{tag}/disp']) < 1e-6
            assert rel_l2(warped.numpy(), g[f'{tag}/warped']) < 1e-6
            for n in opt.names:
                ref = g[f'{tag}/grad/{n}']
                if grads[n].numel() <= 4096:
                    assert rel_l2(grads[n].numpy(), ref) < 1e-5, n
                else:
                    assert abs(summary_of(grads[n])[2] - ref[2]) <= 1e-5 * ref[2], n
        if s in (1, 3):
            for n, v in sd.items():
                ref = g[f'{tag}/after{s}/{n}']
                if v.numel() <= 4096:
                    assert rel_l2(v.numpy(), ref) < 1e-5, (s, n)
                else:
                    assert abs(summary_of(v)[2] - ref[2]) <= 1e-6 * ref[2], (s, n)


@pytest.mark.parametrize('tag,C,labelled', [('c8', 8, True), ('c32', 32, True), ('c8_unlabelled_moving', 8, False)])
def test_joint_step_oracle_vs_reference_parts(golden, tag, C, labelled):
    """oracle/steps.py::joint_step (the build-defined joint DeepAtlas step) against the same step composed from the
    REFERENCE's own modules (oracle/make_golden.py::run_joint: its UNet generator, VoxelMorph, grid_sample, NCC / bending / Dice, Adam's
    inputs): the seven loss terms and every gradient of both phases; per tensor no further from the reference's fp32 run than three times
    the reference's own fp32-vs-fp64 distance (never tighter than 1e-5)."""
    g = golden('joint')
    shape = (16, 16, 32)
    spec = nets.UNET_TINY
    seg_sd = nets.closed_form_fill(nets.unet_param_shapes(1, C, spec['encoders'], spec['decoders']), seed=1)
    reg_sd = nets.closed_form_fill(nets.voxelmorph_param_shapes(), seed=4)
    im_m, im_t = nets.closed_form_volume((1, 1) + shape, seed=5), nets.closed_form_volume((1, 1) + shape, seed=6)
    sm, st_ = nets.closed_form_labels((1,) + shape, C, seed=7), nets.closed_form_labels((1,) + shape, C, seed=8)
    ref = steps.joint_step(seg_sd, steps.Adam(steps.trainable(seg_sd)), reg_sd, steps.Adam(steps.trainable(reg_sd)), im_m, im_t,
                           sm if labelled else None, st_, spec, C)
    for k in ('sim', 'bend', 'anat_reg', 'sup', 'anat_seg', 'loss_reg', 'loss_seg'):
        assert abs(ref[k].item() - float(g[f'joint/{tag}/{k}'])) < 2e-6 * max(1.0, abs(float(g[f'joint/{tag}/{k}']))), (k, ref[k].item())
    for phase, grads in (('seg', ref['grads_seg']), ('reg', ref['grads_reg'])):
        for n, gr in grads.items():
            g32, g64 = g[f'joint/{tag}/grad_{phase}/{n}'], g[f'joint/{tag}_f64/grad_{phase}/{n}']
            if phase == 'seg' and (n.endswith('conv.bias') or n.endswith('deconv.bias')):
                continue                                   # zero analytic gradient in front of a BatchNorm: rounding noise on both sides
            if gr.numel() <= 4096 or phase == 'seg':
                floor = rel_l2(g32, g64)
                assert rel_l2(gr.numpy(), g32) < max(3 * floor, 1e-5), (phase, n, rel_l2(gr.numpy(), g32), floor)
            else:
                assert abs(summary_of(gr)[2] - g32[2]) <= 1e-5 * g32[2], (phase, n)


# ---- label-map eval metrics ---------------------------------------------------------------------------
def test_eval_label_metrics_oracle_vs_reference(golden):
    """oracle.losses.{multiclass_dice, dice_loss_on_label, multi_metric} == lib/evalMetrics.py:103-217, lib/loss.py:348-391."""
    import torch
    from oracle import losses
    g = golden('eval')
    pred, truth = torch.from_numpy(g['eval/pred']), torch.from_numpy(g['eval/truth'])
    np.testing.assert_allclose(losses.multiclass_dice(pred, truth, 5).numpy(), g['eval/multiclass_dice_n5'], rtol=1e-6, atol=1e-7)
    for wt in ('Uniform', 'Simple'):
        v = losses.dice_loss_on_label(pred[:, None], truth[:, None], n_class=5, weight_type=wt).item()
        assert abs(v - float(g['eval/dice_on_label_%s' % wt])) < 1e-6
    assert abs(losses.dice_loss_on_label(pred[:, None], truth[:, None]).item() - float(g['eval/dice_on_label_auto'])) < 1e-6
    for tag, kw in (('all', {}), ('rm_bg', {'rm_bg': True}), ('sel', {'eval_label_list': [1, 3]})):
        r = losses.multi_metric(pred.numpy(), truth.numpy(), **kw)
        assert list(r['label_list']) == list(g['eval/multi_metric/%s/label_list' % tag])
        for grp in ('multi_metric_res', 'label_avg_res', 'batch_avg_res'):
            for m, v in r[grp].items():
                np.testing.assert_allclose(v, g['eval/multi_metric/%s/%s/%s' % (tag, grp, m)], rtol=1e-12, atol=0, equal_nan=True)


# ---- LNCC and gradient losses ---------------------------------------------------------------------------------
def test_reg_losses_f2_oracle_vs_reference(golden):
    """oracle.losses.{lncc_loss, gradient_loss} == lib/loss.py:589-617, 625-671 (loss and input gradients)."""
    import torch
    from oracle import losses
    g = golden('reglosses')
    I = torch.from_numpy(g['lncc/I']).requires_grad_(True); J = torch.from_numpy(g['lncc/J']).requires_grad_(True)
    for fs in (9, 5):
        l = losses.lncc_loss(I, J, filter_size=fs)
        gi, gj = torch.autograd.grad(l, (I, J))
        assert abs(l.item() - float(g['lncc/f%d/loss' % fs])) < 1e-6
        assert rel_l2(gi.numpy(), g['lncc/f%d/grad_I' % fs]) < 1e-5 and rel_l2(gj.numpy(), g['lncc/f%d/grad_J' % fs]) < 1e-5
    u = torch.from_numpy(g['gradloss/u']).requires_grad_(True)
    for tag, kw in (('L2', {}), ('L2_spacing', {'spacing': (1.0, 2.0, 1.5)}), ('L2_nonorm', {'spacing': (1.0, 2.0, 1.5), 'normalize': False}), ('L1', {'norm': 'L1'})):
        l = losses.gradient_loss(u, **kw)
        gu, = torch.autograd.grad(l, u)
        assert abs(l.item() - float(g['gradloss/%s/loss' % tag])) < 1e-6 * max(1.0, abs(float(g['gradloss/%s/loss' % tag])))
        assert rel_l2(gu.numpy(), g['gradloss/%s/grad' % tag]) < 1e-6


# ---- the fixed `UNet` ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('BN', [False, True])
def test_unet_full_oracle_vs_reference(golden, BN):
    """oracle.nets.unet_full_forward == UNet.forward (unets.py:141-179): logits, Dice loss, gradients (l2 of every tensor)."""
    import torch
    from oracle import nets, losses
    g = golden('unet_full')
    tag = 'unet_full/bn%d' % int(BN)
    sd = nets.closed_form_fill_positional(nets.unet_full_param_shapes(1, 3, bias=True, BN=BN), seed=4)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running_' not in k}
    state = dict(sd); state.update(params)
    x = nets.closed_form_volume((1, 1, 16, 16, 16), seed=70)
    y = nets.closed_form_labels((1, 16, 16, 16), 3, seed=71)
    logits = nets.unet_full_forward(state, x, training=True)
    loss = losses.dice_loss(logits, y.long(), n_class=3, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)
    loss.backward()
    assert abs(loss.item() - float(g[tag + '/loss'])) < 1e-6
    assert rel_l2(summary_of(logits)[5:], g[tag + '/logits'][5:]) < 1e-5
    for n, p in params.items():
        ref = g[tag + '/grad/' + n]
        if ref[2] > 1e-12:
            assert abs(summary_of(p.grad)[2] - ref[2]) / ref[2] < 1e-4, n
    if BN:
        for n, v in state.items():
            if 'running_' in n:
                np.testing.assert_allclose(v.numpy(), g[tag + '/after/' + n], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('name', ['UNET_OPT', 'UNET_RES'])
def test_unet_generator_options_oracle_vs_reference(golden, name):
    """oracle.nets.unet_forward with maxpool=False / upsample=True / res=True == UNetTemplate.forward (unets.py:259-278)."""
    import torch
    from oracle import nets, losses
    g = golden('unet_options')
    spec = getattr(nets, name)
    sd = nets.closed_form_fill(nets.unet_param_shapes(1, 16, spec['encoders'], spec['decoders'], maxpool=spec['maxpool'], upsample=spec['upsample']), seed=5)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running_' not in k}
    state = dict(sd); state.update(params)
    x = nets.closed_form_volume((1, 1, 8, 8, 16), seed=80)
    y = nets.closed_form_labels((1, 8, 8, 16), 16, seed=81)
    logits = nets.unet_forward(state, x, spec, training=True)
    loss = losses.dice_loss(logits, y.long(), n_class=16, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)
    loss.backward()
    t = 'unet_opt/' + name
    assert abs(loss.item() - float(g[t + '/loss'])) < 1e-6
    assert rel_l2(logits.detach().numpy(), g[t + '/logits']) < 1e-5
    for n, p in params.items():
        ref = g[t + '/grad/' + n]
        if np.linalg.norm(ref) > 1e-10 and not n.endswith('conv.bias'):
            assert rel_l2(p.grad.numpy(), ref) < 1e-3, n


# ---- data path ---------------------------------------------------------------------------------------------------
def test_datapath_oracle_vs_reference(golden):
    from oracle import datapath as dp
    g = golden('datapath')
    img, seg = dp.sitk_to_tensor(g['dp/img'], g['dp/seg'])
    assert np.array_equal(img, g['dp/totensor/image']) and np.array_equal(seg, g['dp/totensor/seg'])
    for tag, cs in (('c3', [1, 2, 3]), ('c6', [1, 0, 2, 3, 1, 0])):
        ci, cs_ = dp.crop_tensor(img, seg, cs)
        assert np.array_equal(ci, g['dp/crop/%s/image' % tag]) and np.array_equal(cs_, g['dp/crop/%s/seg' % tag])
    for tag, tile, ov in (('a', (8, 8, 8), (2, 2, 2)), ('b', (9, 7, 6), (1, 2, 0))):
        part = dp.Partition(tile, ov)
        assert np.array_equal(part.tiles(g['dp/img'].astype(np.float32))[:, None], g['dp/part/%s/image' % tag])
        st = part.tiles(g['dp/seg'].astype(np.uint8))
        assert np.array_equal(st[:, None], g['dp/part/%s/seg' % tag])
        assert np.array_equal(part.assemble(st), g['dp/part/%s/assemble' % tag])
        assert np.array_equal(part.assemble(g['dp/part/%s/noisy' % tag], is_vote=True), g['dp/part/%s/assemble_vote' % tag])


def test_lncc_multiscale_oracle_vs_reference(golden):
    """oracle.losses.lncc_multiscale_loss == LNCCLoss.forward (lib/loss.py:512-586) at one, two and three scales."""
    import torch
    from oracle import nets, losses
    g = golden('reglosses')
    for tag, shp in (('s1', (1, 1, 20, 24, 28)), ('s2', (1, 1, 66, 68, 70))):            # the 3-scale case runs in the GPU suite
        A = nets.closed_form_volume(shp, seed=63).clone().requires_grad_(True)
        B = nets.closed_form_volume(shp, seed=64).clone().requires_grad_(True)
        l = losses.lncc_multiscale_loss(A, B)
        ga, gb = torch.autograd.grad(l, (A, B))
        assert abs(l.item() - float(g['lncc_ms/%s/loss' % tag])) < 1e-6
        assert rel_l2(summary_of(ga)[5:], g['lncc_ms/%s/grad_I' % tag][5:]) < 1e-5 and rel_l2(summary_of(gb)[5:], g['lncc_ms/%s/grad_J' % tag][5:]) < 1e-5


def test_registry_losses_oracle_vs_reference(golden):
    """Round-2 fixtures (tests/golden/registry_losses.npz, generated from the reference by oracle/make_golden.py): the registry's
    cross-entropy family, BendingEnergyLoss with norm != 'L2', SegMaskToOneHot, and the device synthetic generator's restatement."""
    from oracle import losses
    g = golden('registry_losses')
    x = T(g['xent/logits']).requires_grad_(True)
    y = T(g['xent/labels'])

    def check(tag, l, wrt):
        gr, = torch.autograd.grad(l, wrt)
        assert abs(l.item() - g[f'xent/{tag}/loss']) < 1e-6 * max(1.0, abs(g[f'xent/{tag}/loss'])), tag
        assert rel_l2(gr.numpy(), g[f'xent/{tag}/grad']) < 1e-6, tag

    check('ce_mean', losses.cross_entropy_loss(x, y), x)
    check('ce_sum', losses.cross_entropy_loss(x, y, reduction='sum'), x)
    check('ce_ignore', losses.cross_entropy_loss(x, y, ignore_index=2), x)
    check('focal_default', losses.focal_loss(x, y, 5), x)
    check('focal_alpha_g15_sum', losses.focal_loss(x, y, 5, alpha=T(g['xent/alpha']), gamma=1.5, size_average=False), x)
    p = T(g['xent/prob']).requires_grad_(True)
    check('focal_nosoftmax', losses.focal_loss(p, y, 5, soft_max=False), p)
    t = T(g['xent/soft_target'])
    check('soft_softmax', losses.soft_cross_entropy_loss(x, t, softmax=True), x)
    p2 = T(g['xent/prob_clamped_in']).requires_grad_(True)
    check('soft_nosoftmax', losses.soft_cross_entropy_loss(p2, t, softmax=False), p2)
    assert int(g['xent/soft_index_target_raises']) == 1            # the reference's index-target branch fails; the product raises too
    u = T(g['bendL1/u']).requires_grad_(True)
    for tag, kw in (('L1', {}), ('L1_spacing', {'spacing': (1.0, 2.0, 1.5)})):
        l = losses.bending_energy_loss(u, norm='L1', **kw)
        gr, = torch.autograd.grad(l, u)
        assert abs(l.item() - g[f'bendL1/{tag}/loss']) < 1e-7
        assert rel_l2(gr.numpy(), g[f'bendL1/{tag}/grad']) < 1e-6
    assert np.array_equal(losses.seg_mask_to_one_hot(T(g['onehot/seg']), 4).numpy(), g['onehot/segmentation_onehot'])


def test_synth_volume_restatement_properties():
    """oracle/datapath.py synth_volume (the bit-exact restatement of da_synth_volume): value ranges, determinism, sample offset."""
    from oracle import datapath as dp
    img, lab = dp.synth_volume(3, (8, 16, 24), 32, 0, 0.1, 230)
    assert img.dtype == np.float32 and lab.dtype == np.uint8 and img.min() >= 0 and img.max() < 1 and lab.max() < 32
    assert abs(img.mean() - 0.5) < 0.02 and len(np.unique(lab)) == 32
    img2, lab2 = dp.synth_volume(1, (8, 16, 24), 32, 0, 0.1, 230, sample0=2)          # sample k is the same whatever batch it is drawn in
    assert np.array_equal(img2[0], img[2]) and np.array_equal(lab2[0], lab[2])
    simg, slab = dp.synth_volume(2, (16, 16, 24), 32, 1, 0.1, 230)
    assert np.array_equal(slab[1], (slab[0].astype(int) + 1) % 32) and simg.max() <= 1.0
    assert np.abs(simg - slab / np.float32(31)).max() <= 0.1 + 1e-6

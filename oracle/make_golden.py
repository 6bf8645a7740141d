#!/usr/bin/env python
"""Generate golden vectors by IMPORTING THE REFERENCE (/root/reference) in the build container.

Run:  python -m oracle.make_golden            (writes tests/golden/*.npz: 11 fixtures incl. joint.npz, the joint step and its fp64 twin)

The reference's Python files never travel: only inputs/outputs (data) are committed.  Inputs and
weights are closed-form (oracle/nets.py closed_form_*), so fixtures stay small; big tensors are
stored as (sum, abs-sum, l2, strided sample) summaries.

Reference symbols exercised (file:line in /root/reference):
  lib/network_factory/__init__.py:9-27 get_network; unets.py:182-280 UNet_generator/UNetTemplate;
  voxel_morph.py:29-92 VoxelMorphCVPR2018; lib/loss.py:397-476 DiceLossMultiClass, :485-501 NCC,
  :674-730 BendingEnergyLoss; lib/transforms.py:675-689 mask_to_one_hot; lib/utils.py:78-102
  get_identity_transform_batch; lib/evalMetrics.py:17-21,58-68 metricEval('dice'), :184-217
  get_multiclass_dice; torch.optim.Adam as used at models/segmentation.py:91.
"""
import os
import sys
import types
import copy
import numpy as np
import torch

REF = os.environ.get('DEEPATLAS_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
if os.path.dirname(HERE) not in sys.path:          # `python oracle/make_golden.py` works like `python -m oracle.make_golden`
    sys.path.insert(0, os.path.dirname(HERE))


def import_reference():
    sitk = types.ModuleType('SimpleITK')          # only default-arg attrs are touched (transforms.py:167,208,287)
    sitk.sitkLinear, sitk.sitkBSpline, sitk.sitkNearestNeighbor = 1, 2, 3
    sitk.GetArrayFromImage = lambda a: np.array(a, copy=True)      # fixtures hand numpy arrays where the reference expects images
    sys.modules.setdefault('SimpleITK', sitk)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import lib.network_factory as nf
    import lib.network_factory.unets as unets
    import lib.loss as loss
    import lib.utils as utils
    import lib.transforms as transforms
    import lib.evalMetrics as metrics
    return types.SimpleNamespace(nf=nf, unets=unets, loss=loss, utils=utils, transforms=transforms, metrics=metrics)


def summary(t, nsample=512):
    t = t.detach().double().reshape(-1)
    n = t.numel()
    stride = max(1, n // nsample)
    idx = torch.arange(0, n, stride)[:nsample]
    return np.concatenate([[t.sum().item(), t.abs().sum().item(), t.norm().item(), float(n), float(stride)],
                           t[idx].numpy()])


def np32(t):
    return t.detach().cpu().numpy().copy()


def load_sd(model, sd):
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)


def run_seg(ref, model_cls, spec_name, in_ch, n_classes, shape, N, out, prefix, full, dtype=torch.float32, steps=3):
    from oracle import nets
    spec = getattr(nets, spec_name)
    shapes = nets.unet_param_shapes(in_ch, n_classes, spec['encoders'], spec['decoders'], bias=True, BN=True)
    sd0 = nets.closed_form_fill(shapes, seed=1)
    model = model_cls(in_channel=in_ch, n_classes=n_classes, bias=True, BN=True)
    assert set(model.state_dict().keys()) == set(sd0.keys()), (set(model.state_dict()) ^ set(sd0))
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(sd0[k].shape), k
    load_sd(model, sd0)
    model = model.to(dtype)
    x = nets.closed_form_volume((N, in_ch) + shape, seed=2).to(dtype)
    y = nets.closed_form_labels((N,) + shape, n_classes, seed=3)
    crit = ref.loss.get_loss_function('dice')(n_class=n_classes, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    tag = prefix + ('' if dtype == torch.float32 else '_f64')
    for s in range(1, steps + 1):
        model.train()
        opt.zero_grad()
        logits = model(x)
        loss = crit(logits, y.long())
        loss.backward()
        if s == 1:
            out[f'{tag}/loss'] = np.float64(loss.item())
            out[f'{tag}/logits'] = np32(logits) if full else summary(logits)
            pred = torch.max(logits, 1)[1]
            out[f'{tag}/argmax'] = np32(pred).astype(np.uint8) if full else summary(pred)
            for n, p in model.named_parameters():
                out[f'{tag}/grad/{n}'] = np32(p.grad) if full else summary(p.grad)
        opt.step()
        if dtype == torch.float32 and s in (1, steps):
            out[f'{tag}/loss_step{s}'] = np.float64(loss.item())
            for n, v in model.state_dict().items():
                if v.dtype.is_floating_point:
                    out[f'{tag}/after{s}/{n}'] = np32(v) if full else summary(v)
    if dtype == torch.float32:
        # eval-mode forward + eval Dice (models/segmentation.py:179-201) on the 3-step model
        model.eval()
        with torch.no_grad():
            pred = model(x)
            dice = np.zeros((N, n_classes - 1))
            for b in range(N):
                for c in range(1, n_classes):
                    dice[b, c - 1] = ref.metrics.metricEval('dice', torch.max(pred[b:b + 1], 1)[1].squeeze().numpy() == c,
                                                            y[b].numpy() == c, num_labels=2)
            out[f'{tag}/eval_logits'] = np32(pred) if full else summary(pred)
            out[f'{tag}/eval_dice'] = dice
            out[f'{tag}/eval_multiclass_dice'] = np32(ref.metrics.get_multiclass_dice(torch.max(pred, 1)[1], y.long(), n_class=n_classes))
    return x, y


def run_reg(ref, shape, out, prefix, full_small=True, dtype=torch.float32, steps=3, lam=1.0):
    from oracle import nets
    shapes = nets.voxelmorph_param_shapes()
    sd0 = nets.closed_form_fill(shapes, seed=4)
    model = ref.nf.get_network('voxel_morph_cvpr')()
    assert set(model.state_dict().keys()) == set(sd0.keys())
    load_sd(model, sd0)
    model = model.to(dtype)
    src = nets.closed_form_volume((1, 1) + shape, seed=5).to(dtype)
    tgt = nets.closed_form_volume((1, 1) + shape, seed=6).to(dtype)
    ncc = ref.loss.get_loss_function('ncc')()
    bend = ref.loss.get_loss_function('bendingEnergy')()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    tag = prefix + ('' if dtype == torch.float32 else '_f64')
    for s in range(1, steps + 1):
        opt.zero_grad()
        disp, warped, deform = model(src, tgt)
        l_sim = ncc(warped, tgt)
        l_reg = bend(disp)
        loss = l_sim + lam * l_reg
        loss.backward()
        if s == 1:
            out[f'{tag}/loss'] = np.float64(loss.item())
            out[f'{tag}/ncc'] = np.float64(l_sim.item())
            out[f'{tag}/bending'] = np.float64(l_reg.item())
            out[f'{tag}/disp'] = np32(disp)
            out[f'{tag}/warped'] = np32(warped)
            out[f'{tag}/deform'] = summary(deform)
            for n, p in model.named_parameters():
                small = p.numel() <= 4096
                out[f'{tag}/grad/{n}'] = np32(p.grad) if small else summary(p.grad)
        opt.step()
        if dtype == torch.float32 and s in (1, steps):
            out[f'{tag}/loss_step{s}'] = np.float64(loss.item())
            for n, v in model.state_dict().items():
                out[f'{tag}/after{s}/{n}'] = np32(v) if v.numel() <= 4096 else summary(v)


def run_ops(ref, out):
    """Per-op fixtures: losses (all weight types / soft target / no_bg), warp with out-of-range taps,
    identity grid, one-hot, nearest up-sampling with odd sizes, max-pool, eval Dice."""
    from oracle import nets
    import torch.nn.functional as F
    D, H, W = 6, 10, 14
    C = 5
    logits = (nets.closed_form_volume((2, C, D, H, W), seed=7) * 4 - 2).requires_grad_(True)
    labels = nets.closed_form_labels((2, D, H, W), C, seed=8)
    out['ops/dice/logits'] = np32(logits)
    out['ops/dice/labels'] = np32(labels)
    for wt in ('Uniform', 'Simple', 'Volume'):
        for no_bg in (False, True):
            crit = ref.loss.DiceLossMultiClass(n_class=C, weight_type=wt, no_bg=no_bg, softmax=True, eps=1e-6)
            l = crit(logits, labels.long())
            g, = torch.autograd.grad(l, logits)
            out[f'ops/dice/{wt}_{int(no_bg)}/loss'] = np.float64(l.item())
            out[f'ops/dice/{wt}_{int(no_bg)}/grad'] = np32(g)
    # soft (5-D) target, softmax=False : the path the joint step uses (loss.py:435-436)
    prob = F.softmax(logits.detach(), 1).requires_grad_(True)
    soft_t = F.softmax(nets.closed_form_volume((2, C, D, H, W), seed=9) * 3, 1)
    crit = ref.loss.DiceLossMultiClass(n_class=C, weight_type='Uniform', no_bg=False, softmax=False, eps=1e-6)
    l = crit(prob, soft_t)
    g, = torch.autograd.grad(l, prob)
    out['ops/dice_soft/source'] = np32(prob)
    out['ops/dice_soft/target'] = np32(soft_t)
    out['ops/dice_soft/loss'] = np.float64(l.item())
    out['ops/dice_soft/grad'] = np32(g)
    # one-hot
    out['ops/onehot'] = np32(ref.transforms.mask_to_one_hot(labels.view(2, 1, D, H, W), C))
    # NCC
    a = nets.closed_form_volume((2, 1, D, H, W), seed=10).requires_grad_(True)
    b = nets.closed_form_volume((2, 1, D, H, W), seed=11)
    l = ref.loss.NormalizedCrossCorrelationLoss()(a, b)
    g, = torch.autograd.grad(l, a)
    out['ops/ncc/a'], out['ops/ncc/b'] = np32(a), np32(b)
    out['ops/ncc/loss'], out['ops/ncc/grad'] = np.float64(l.item()), np32(g)
    # bending (D != H != W pins the channel/axis quirk)
    u = ((nets.closed_form_volume((2, 3, D, H, W), seed=12) - 0.5) * 0.3).requires_grad_(True)
    l = ref.loss.BendingEnergyLoss()(u)
    g, = torch.autograd.grad(l, u)
    out['ops/bending/u'] = np32(u)
    out['ops/bending/loss'], out['ops/bending/grad'] = np.float64(l.item()), np32(g)
    l = ref.loss.BendingEnergyLoss(spacing=(1.0, 2.0, 1.5))(u)
    out['ops/bending/loss_spacing'] = np.float64(l.item())
    # identity grid
    out['ops/identity'] = np32(ref.utils.get_identity_transform_batch((1, 1, D, H, W)))
    # warp: displacement large enough to push taps out of range on every side
    idt = ref.utils.get_identity_transform_batch((1, 1, D, H, W))
    disp = ((nets.closed_form_volume((2, 3, D, H, W), seed=13) - 0.5) * 1.2).requires_grad_(True)
    for ch, nm in ((1, 'warp1'), (C, 'warpC')):
        src = nets.closed_form_volume((2, ch, D, H, W), seed=14 + ch).requires_grad_(True)
        deform = disp + idt
        w = F.grid_sample(src, deform.permute(0, 2, 3, 4, 1), mode='bilinear', padding_mode='zeros', align_corners=True)
        gout = nets.closed_form_volume(tuple(w.shape), seed=20) - 0.5
        gs, gd = torch.autograd.grad((w * gout).sum(), (src, disp))
        out[f'ops/{nm}/src'], out[f'ops/{nm}/disp'] = np32(src), np32(disp)
        out[f'ops/{nm}/out'], out[f'ops/{nm}/gout'] = np32(w), np32(gout)
        out[f'ops/{nm}/grad_src'], out[f'ops/{nm}/grad_disp'] = np32(gs), np32(gd)
    # nearest up-sampling (F.interpolate default) incl. odd pyramid sizes (voxel_morph.py:72-80)
    t = nets.closed_form_volume((1, 2, 2, 3, 5), seed=30)
    out['ops/nearest/in'] = np32(t)
    out['ops/nearest/out_3_5_10'] = np32(F.interpolate(t, size=(3, 5, 10)))
    out['ops/nearest/out_4_6_10'] = np32(F.interpolate(t, size=(4, 6, 10)))
    # max-pool (incl. exact ties: constant block)
    p = nets.closed_form_volume((1, 2, 4, 4, 6), seed=31).clone()
    p[0, 0, :2, :2, :2] = 0.25
    p.requires_grad_(True)
    q = F.max_pool3d(p, 2)
    g, = torch.autograd.grad(q.sum(), p)
    out['ops/maxpool/in'], out['ops/maxpool/out'], out['ops/maxpool/grad'] = np32(p), np32(q), np32(g)
    # eval Dice incl. an empty class (NaN) -- scipy path, models/segmentation.py:190-194
    pred = nets.closed_form_labels((1, D, H, W), 4, seed=40)
    truth = nets.closed_form_labels((1, D, H, W), 4, seed=41)
    dice = np.array([ref.metrics.metricEval('dice', pred.numpy() == c, truth.numpy() == c, num_labels=2) for c in range(1, 6)])
    out['ops/evaldice/pred'], out['ops/evaldice/truth'], out['ops/evaldice/dice'] = np32(pred), np32(truth), dice


def run_eval(ref, out):
    """lib/evalMetrics.py:103-217 (get_multi_metric, cal_metric, get_multiclass_dice) and
    lib/loss.py:348-391 (DiceLossOnLabel) on two small label maps; class 4 is absent from sample 0's ground truth
    (the -1 / skip-in-average path) and class 3 from sample 1's prediction."""
    from oracle import nets
    D, H, W = 6, 10, 14
    pred = nets.closed_form_labels((2, D, H, W), 5, seed=50).clone()
    truth = nets.closed_form_labels((2, D, H, W), 5, seed=51).clone()
    truth[0][truth[0] == 4] = 0
    pred[1][pred[1] == 3] = 1
    out['eval/pred'], out['eval/truth'] = np32(pred), np32(truth)
    out['eval/multiclass_dice_n5'] = np32(ref.metrics.get_multiclass_dice(pred, truth, n_class=5))
    out['eval/multiclass_dice_auto'] = np32(ref.metrics.get_multiclass_dice(pred, truth))
    oh = ref.transforms.mask_to_one_hot(truth.view(2, 1, -1), 5).view(2, 5, D, H, W)
    out['eval/multiclass_dice_onehot_truth'] = np32(ref.metrics.get_multiclass_dice(pred, oh, n_class=5))
    for wt in ('Uniform', 'Simple'):
        crit = ref.loss.DiceLossOnLabel(n_class=5)
        out['eval/dice_on_label_%s' % wt] = np.float64(crit(pred[:, None], truth[:, None], weight_type=wt).item())
    out['eval/dice_on_label_auto'] = np.float64(ref.loss.DiceLossOnLabel()(pred[:, None], truth[:, None]).item())
    for tag, kw in (('all', {}), ('rm_bg', {'rm_bg': True}), ('sel', {'eval_label_list': [1, 3]})):
        r = ref.metrics.get_multi_metric(pred.numpy(), truth.numpy(), **kw)
        out['eval/multi_metric/%s/label_list' % tag] = np.asarray(r['label_list'], dtype=np.int64)
        for grp in ('multi_metric_res', 'label_avg_res', 'batch_avg_res'):
            for m, v in r[grp].items():
                out['eval/multi_metric/%s/%s/%s' % (tag, grp, m)] = np.asarray(v, dtype=np.float64)
    # lib/evalMetrics.py:17-100 metricEval: 'iou' over all labels of one volume; 'dice' / 'recall' / 'precision' on the binary masks of one
    # class (the way models/segmentation.py:191-194 calls it).  NaN where the reference divides by zero (class absent).
    # (a prediction that agrees with the truth on two voxels out of three: the two closed-form label maps above hardly overlap)
    keep = (torch.arange(pred.numel()).view(pred.shape) % 3) != 0
    pred2 = torch.where(keep, truth, pred)
    out['eval/metricEval/pred'] = np32(pred2)
    for b in range(2):
        out['eval/metricEval/iou_n5/%d' % b] = np.float64(ref.metrics.metricEval('iou', pred2[b].numpy(), truth[b].numpy(), 5))
        for m in ('dice', 'recall', 'precision'):
            vals = []
            for c in range(1, 5):
                try:
                    with np.errstate(invalid='ignore', divide='ignore'):
                        vals.append(float(ref.metrics.metricEval(m, pred2[b].numpy() == c, truth[b].numpy() == c, num_labels=2)))
                except ZeroDivisionError:
                    vals.append(float('nan'))
            out['eval/metricEval/%s_binary/%d' % (m, b)] = np.asarray(vals, dtype=np.float64)


def run_reglosses(ref, out):
    """lib/loss.py:589-617 VoxelMorphLNCC and :625-671 gradientLoss (loss + input gradients)."""
    from oracle import nets
    shape = (2, 1, 12, 14, 16)
    I = nets.closed_form_volume(shape, seed=60).clone().requires_grad_(True)
    J = nets.closed_form_volume(shape, seed=61).clone().requires_grad_(True)
    for fs in (9, 5):
        crit = ref.loss.VoxelMorphLNCC(filter_size=fs)
        l = crit(I, J)
        gi, gj = torch.autograd.grad(l, (I, J))
        out['lncc/f%d/loss' % fs] = np.float64(l.item())
        out['lncc/f%d/grad_I' % fs], out['lncc/f%d/grad_J' % fs] = np32(gi), np32(gj)
    out['lncc/I'], out['lncc/J'] = np32(I), np32(J)
    # multi-scale LNCCLoss (:512-586, builds its filters with .cuda(): run on the CPU by making .cuda() the identity for this call)
    _cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for tag, shp in (('s1', (1, 1, 20, 24, 28)), ('s2', (1, 1, 66, 68, 70)), ('s3', (1, 1, 130, 132, 134))):
            A = nets.closed_form_volume(shp, seed=63).clone().requires_grad_(True)
            B = nets.closed_form_volume(shp, seed=64).clone().requires_grad_(True)
            l = ref.loss.LNCCLoss()(A, B)
            ga, gb = torch.autograd.grad(l, (A, B))
            out['lncc_ms/%s/loss' % tag] = np.float64(l.item())
            out['lncc_ms/%s/grad_I' % tag], out['lncc_ms/%s/grad_J' % tag] = summary(ga), summary(gb)
            # fp64 twin: the window variances are differences of large sums, so fp32 gradients carry visible rounding noise
            A64, B64 = A.detach().double().requires_grad_(True), B.detach().double().requires_grad_(True)
            from oracle import losses as _ol          # the reference builds fp32 filters; its restatement (pinned above in fp32) takes the input dtype
            l64 = _ol.lncc_multiscale_loss(A64, B64)
            ga64, gb64 = torch.autograd.grad(l64, (A64, B64))
            out['lncc_ms/%s_f64/loss' % tag] = np.float64(l64.item())
            out['lncc_ms/%s_f64/grad_I' % tag], out['lncc_ms/%s_f64/grad_J' % tag] = summary(ga64), summary(gb64)
    finally:
        torch.Tensor.cuda = _cuda
    u = (nets.closed_form_volume((2, 3, 6, 10, 14), seed=62) * 0.3).clone().requires_grad_(True)
    out['gradloss/u'] = np32(u)
    for tag, kw in (('L2', {}), ('L2_spacing', {'spacing': (1.0, 2.0, 1.5)}), ('L2_nonorm', {'spacing': (1.0, 2.0, 1.5), 'normalize': False}), ('L1', {'norm': 'L1'})):
        crit = ref.loss.gradientLoss(**kw)
        l = crit(u)
        g, = torch.autograd.grad(l, u)
        out['gradloss/%s/loss' % tag] = np.float64(l.item())
        out['gradloss/%s/grad' % tag] = np32(g)


def run_unet_full(ref, out):
    """the fixed `UNet` (unets.py:70-179) at 16^3, closed-form weights: forward, Dice loss, every gradient,
    BN running stats after the step -- with and without BatchNorm; big tensors as summaries."""
    from oracle import nets
    shape, n_classes = (16, 16, 16), 3
    x = nets.closed_form_volume((1, 1) + shape, seed=70)
    y = nets.closed_form_labels((1,) + shape, n_classes, seed=71)
    for BN in (False, True):
        tag = 'unet_full/bn%d' % int(BN)
        shapes = nets.unet_full_param_shapes(1, n_classes, bias=True, BN=BN)
        sd0 = nets.closed_form_fill_positional(shapes, seed=4)
        model = ref.nf.get_network('UNet')(1, n_classes, bias=True, BN=BN)
        assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == {k: tuple(v) for k, v in shapes.items()}
        load_sd(model, sd0)
        for dtype in (torch.float32, torch.float64):
            m = copy.deepcopy(model).to(dtype)
            m.train()
            crit = ref.loss.get_loss_function('dice')(n_class=n_classes, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)
            logits = m(x.to(dtype))
            loss = crit(logits, y.long())
            loss.backward()
            t = tag + ('' if dtype == torch.float32 else '_f64')
            out[t + '/loss'] = np.float64(loss.item())
            out[t + '/logits'] = summary(logits)
            for n, p in m.named_parameters():
                out[t + '/grad/' + n] = summary(p.grad)
            if BN and dtype == torch.float32:
                for n, v in m.state_dict().items():
                    if 'running_' in n:
                        out[t + '/after/' + n] = np32(v)


def run_unet_options(ref, out):
    """UNet_generator options (unets.py:230-237,264,275): UNET_OPT = maxpool=False (strided conv) +
    upsample=True (trilinear), UNET_RES = res=True; first training step (logits, loss, every gradient), fp32 and fp64."""
    from oracle import nets
    shape, n_classes = (8, 8, 16), 16
    x = nets.closed_form_volume((1, 1) + shape, seed=80)
    y = nets.closed_form_labels((1,) + shape, n_classes, seed=81)
    for name in ('UNET_OPT', 'UNET_RES'):
        spec = getattr(nets, name)
        cls = ref.unets.UNet_generator(encoders=spec['encoders'], decoders=spec['decoders'], act='ReLU', maxpool=spec['maxpool'],
                                       upsample=spec['upsample'], res=spec['res'])
        shapes = nets.unet_param_shapes(1, n_classes, spec['encoders'], spec['decoders'], maxpool=spec['maxpool'], upsample=spec['upsample'])
        sd0 = nets.closed_form_fill(shapes, seed=5)
        model = cls(1, n_classes, bias=True, BN=True)
        assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == {k: tuple(v) for k, v in shapes.items()}
        load_sd(model, sd0)
        for dtype in (torch.float32, torch.float64):
            m = copy.deepcopy(model).to(dtype)
            m.train()
            crit = ref.loss.get_loss_function('dice')(n_class=n_classes, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)
            logits = m(x.to(dtype))
            loss = crit(logits, y.long())
            loss.backward()
            t = 'unet_opt/' + name + ('' if dtype == torch.float32 else '_f64')
            out[t + '/loss'] = np.float64(loss.item())
            out[t + '/logits'] = np32(logits)
            for n, p in m.named_parameters():
                out[t + '/grad/' + n] = np32(p.grad)


def run_datapath(ref, out):
    """lib/transforms.py SitkToTensor (:71-92), CropTensor (:124-158), Partition + assemble (:508-649), with
    numpy arrays standing in for SimpleITK images (the stub's GetArrayFromImage is the identity)."""
    from oracle import nets
    T = ref.transforms
    img = (nets.closed_form_volume((11, 13, 17), seed=90).double() * 1.6 - 0.3).numpy()          # values outside [0, 1]
    seg = nets.closed_form_labels((1, 11, 13, 17), 4, seed=91)[0].numpy().astype(np.int16)
    out['dp/img'], out['dp/seg'] = img, seg
    s = T.SitkToTensor()({'image': img.copy(), 'segmentation': seg.copy()})
    out['dp/totensor/image'], out['dp/totensor/seg'] = np32(s['image']), np32(s['segmentation'])
    for tag, cs in (('c3', [1, 2, 3]), ('c6', [1, 0, 2, 3, 1, 0])):
        c = T.CropTensor(cs)({'image': s['image'].clone(), 'segmentation': s['segmentation'].clone()})
        out['dp/crop/%s/image' % tag], out['dp/crop/%s/seg' % tag] = np32(c['image']), np32(c['segmentation'])
    for tag, tile, ov in (('a', (8, 8, 8), (2, 2, 2)), ('b', (9, 7, 6), (1, 2, 0))):
        part = T.Partition(tile, ov, mode='eval')
        p = part({'image': img.astype(np.float32), 'segmentation': seg.astype(np.uint8), 'name': 'x'})
        out['dp/part/%s/image' % tag], out['dp/part/%s/seg' % tag] = np32(p['image']), np32(p['segmentation'])
        tiles = p['segmentation'][:, 0]
        out['dp/part/%s/assemble' % tag] = np.asarray(part.assemble(tiles, is_vote=False, if_itk=False))
        # disagreeing tiles so that the vote matters: every odd tile predicts label + 1 (mod 4)
        noisy = tiles.clone()
        noisy[1::2] = (noisy[1::2] + 1) % 4
        out['dp/part/%s/noisy' % tag] = np32(noisy)
        out['dp/part/%s/assemble_vote' % tag] = np.asarray(part.assemble(noisy, is_vote=True, if_itk=False))


def run_registry_losses(ref, out):
    """Round-2 additions: the registry's cross-entropy family (lib/loss.py:100-154 SoftCrossEntropy, :157-213 FocalLoss, :749
    nn.CrossEntropyLoss), BendingEnergyLoss with norm != 'L2' (:721-729) and transforms.SegMaskToOneHot (lib/transforms.py:652-673)."""
    from oracle import nets
    import warnings
    warnings.simplefilter('ignore')
    N, C, shape = 2, 5, (6, 10, 12)
    x = (nets.closed_form_volume((N, C) + shape, seed=70) * 4 - 2).clone().requires_grad_(True)
    y = nets.closed_form_labels((N,) + shape, C, seed=71)
    out['xent/logits'], out['xent/labels'] = np32(x), np32(y).astype(np.uint8)
    # cross_entropy (registry entry is torch.nn.CrossEntropyLoss itself)
    for tag, kw, tgt in (('mean', {}, y.long()), ('sum', {'reduction': 'sum'}, y.long()),
                         ('ignore', {'ignore_index': 2}, y.long())):
        l = ref.loss.get_loss_function('cross_entropy')(**kw)(x, tgt)
        g, = torch.autograd.grad(l, x)
        out['xent/ce_%s/loss' % tag], out['xent/ce_%s/grad' % tag] = np.float64(l.item()), np32(g)
    # focal: default alpha (ones), gamma 2; per-class alpha, gamma 1.5, sum reduction; soft_max=False on probabilities
    alpha = torch.linspace(0.5, 1.5, C).reshape(C, 1)
    out['xent/alpha'] = np32(alpha)
    for tag, kw in (('default', {}), ('alpha_g15_sum', {'alpha': alpha, 'gamma': 1.5, 'size_average': False})):
        l = ref.loss.get_loss_function('focal')(C, **kw)(x, y.long())
        g, = torch.autograd.grad(l, x)
        out['xent/focal_%s/loss' % tag], out['xent/focal_%s/grad' % tag] = np.float64(l.item()), np32(g)
    p = torch.softmax(x.detach() * 0.7, 1).clone().requires_grad_(True)
    out['xent/prob'] = np32(p)
    l = ref.loss.get_loss_function('focal')(C, soft_max=False)(p, y.long())
    g, = torch.autograd.grad(l, p)
    out['xent/focal_nosoftmax/loss'], out['xent/focal_nosoftmax/grad'] = np.float64(l.item()), np32(g)
    # soft cross entropy with a probability target; softmax=True on logits, softmax=False on probabilities (clamp_ is in place: clone)
    t = torch.softmax(nets.closed_form_volume((N, C) + shape, seed=72) * 3, 1)
    out['xent/soft_target'] = np32(t)
    l = ref.loss.get_loss_function('soft_cross_entropy')(n_class=C, softmax=True)(x, t)
    g, = torch.autograd.grad(l, x)
    out['xent/soft_softmax/loss'], out['xent/soft_softmax/grad'] = np.float64(l.item()), np32(g)
    p2 = p.detach().clone()
    p2[0, 1, 0, 0, :4] = 0.0                                         # exercises the 1e-8 clamp
    out['xent/prob_clamped_in'] = np32(p2)
    p2.requires_grad_(True)
    l = ref.loss.get_loss_function('soft_cross_entropy')(n_class=C, softmax=False)(p2 * 1.0, t)
    g, = torch.autograd.grad(l, p2)
    out['xent/soft_nosoftmax/loss'], out['xent/soft_nosoftmax/grad'] = np.float64(l.item()), np32(g)
    try:
        ref.loss.get_loss_function('soft_cross_entropy')(n_class=C, softmax=True)(x, y.long())
        out['xent/soft_index_target_raises'] = np.int64(0)
    except Exception:
        out['xent/soft_index_target_raises'] = np.int64(1)
    # bending energy, norm other than 'L2' (D != H != W as in ops.npz)
    u = (nets.closed_form_volume((2, 3, 6, 10, 14), seed=62) * 0.3).clone().requires_grad_(True)
    out['bendL1/u'] = np32(u)
    for tag, kw in (('L1', {'norm': 'L1'}), ('L1_spacing', {'norm': 'L1', 'spacing': (1.0, 2.0, 1.5)})):
        l = ref.loss.BendingEnergyLoss(**kw)(u)
        g, = torch.autograd.grad(l, u)
        out['bendL1/%s/loss' % tag], out['bendL1/%s/grad' % tag] = np.float64(l.item()), np32(g)
    # SegMaskToOneHot on a D x M x N uint8 mask
    seg = nets.closed_form_labels((1, 5, 6, 7), 4, seed=73)[0].to(torch.uint8)
    smp = ref.transforms.SegMaskToOneHot(4)({'segmentation': seg.clone()})
    out['onehot/seg'], out['onehot/segmentation_onehot'] = np32(seg), np32(smp['segmentation_onehot'])


def run_joint(ref, out):
    """First joint DeepAtlas step composed from the REFERENCE's own parts -- its UNet generator, its VoxelMorph,
    F.grid_sample as at voxel_morph.py:90-91, its NCC / bending-energy / Dice modules, torch.optim.Adam -- in fp32 and as an fp64 twin:
    the seven loss terms and every gradient of both phases.  The twin calibrates the device tolerances per tensor (tests/test_gpu_nets.py)
    and the fp32 run pins oracle/steps.py::joint_step (tests/test_oracle_golden.py)."""
    import torch.nn.functional as F
    from oracle import nets
    shape = (16, 16, 32)
    tiny_cls = ref.unets.UNet_generator(encoders=nets.UNET_TINY['encoders'], decoders=nets.UNET_TINY['decoders'],
                                        act='LeakyReLU', maxpool=True, upsample=False, res=False)
    for tag0, C, labelled in (('c8', 8, True), ('c32', 32, True), ('c8_unlabelled_moving', 8, False)):
        for dtype in (torch.float32, torch.float64):
            tag = 'joint/' + tag0 + ('' if dtype == torch.float32 else '_f64')
            seg_sd = nets.closed_form_fill(nets.unet_param_shapes(1, C, nets.UNET_TINY['encoders'], nets.UNET_TINY['decoders']), seed=1)
            reg_sd = nets.closed_form_fill(nets.voxelmorph_param_shapes(), seed=4)
            seg = tiny_cls(in_channel=1, n_classes=C, bias=True, BN=True)
            load_sd(seg, seg_sd)
            reg = ref.nf.get_network('voxel_morph_cvpr')()
            load_sd(reg, reg_sd)
            seg, reg = seg.to(dtype), reg.to(dtype)
            im_m = nets.closed_form_volume((1, 1) + shape, seed=5).to(dtype)
            im_t = nets.closed_form_volume((1, 1) + shape, seed=6).to(dtype)
            sm = nets.closed_form_labels((1,) + shape, C, seed=7)
            st_ = nets.closed_form_labels((1,) + shape, C, seed=8)
            ncc = ref.loss.get_loss_function('ncc')()
            bend = ref.loss.get_loss_function('bendingEnergy')()
            dice_prob = ref.loss.get_loss_function('dice')(n_class=C, weight_type='Uniform', no_bg=False, softmax=False, eps=1e-6)
            dice_logit = ref.loss.get_loss_function('dice')(n_class=C, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)
            warp = lambda src, grid: F.grid_sample(src, grid=grid.permute([0, 2, 3, 4, 1]), mode='bilinear', padding_mode='zeros', align_corners=True)
            onehot_t = ref.loss.mask_to_one_hot(st_.long().unsqueeze(1), C).to(dtype)
            if labelled:
                onehot_m = ref.loss.mask_to_one_hot(sm.long().unsqueeze(1), C).to(dtype)
            else:
                seg.eval()
                with torch.no_grad():
                    onehot_m = F.softmax(seg(im_m), dim=1)
            # registration phase (segmentation net frozen)
            ropt = torch.optim.Adam(reg.parameters(), lr=1e-3)
            ropt.zero_grad()
            disp, warped, deform = reg(im_m, im_t)
            l_sim, l_reg = ncc(warped, im_t), bend(disp)
            l_anat = dice_prob(warp(onehot_m, deform), onehot_t)
            loss_r = l_sim + l_reg + l_anat
            loss_r.backward()
            for n, p in reg.named_parameters():
                out[f'{tag}/grad_reg/{n}'] = np32(p.grad.float()) if p.numel() <= 4096 else summary(p.grad)      # (fp64 values stored rounded to fp32: 6e-8, far below any floor)
            ropt.step()
            # segmentation phase (registration net frozen; the deformation of the un-stepped net, detached)
            sopt = torch.optim.Adam(seg.parameters(), lr=1e-3)
            sopt.zero_grad()
            seg.train()
            logits = seg(im_m)
            l_sp = dice_logit(logits, sm.long()) if labelled else torch.zeros((), dtype=dtype)
            l_anat2 = dice_prob(warp(F.softmax(logits, dim=1), deform.detach()), onehot_t)
            loss_s = l_sp + l_anat2
            loss_s.backward()
            for n, p in seg.named_parameters():
                out[f'{tag}/grad_seg/{n}'] = np32(p.grad.float())
            for k, v in (('sim', l_sim), ('bend', l_reg), ('anat_reg', l_anat), ('sup', l_sp), ('anat_seg', l_anat2), ('loss_reg', loss_r), ('loss_seg', loss_s)):
                out[f'{tag}/{k}'] = np.float64(float(v.detach()))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = import_reference()
    os.makedirs(OUT, exist_ok=True)
    from oracle import nets

    out = {}
    run_joint(ref, out)
    np.savez_compressed(os.path.join(OUT, 'joint.npz'), **out)
    print('joint.npz', len(out))
    if os.environ.get('GOLDEN_ONLY') == 'joint':
        return
    out = {}
    run_registry_losses(ref, out)
    np.savez_compressed(os.path.join(OUT, 'registry_losses.npz'), **out)
    print('registry_losses.npz', len(out))
    if os.environ.get('GOLDEN_ONLY') == 'xent':
        return
    out = {}
    run_eval(ref, out)
    np.savez_compressed(os.path.join(OUT, 'eval.npz'), **out)
    print('eval.npz', len(out))
    out = {}
    run_reglosses(ref, out)
    np.savez_compressed(os.path.join(OUT, 'reglosses.npz'), **out)
    print('reglosses.npz', len(out))
    out = {}
    run_datapath(ref, out)
    np.savez_compressed(os.path.join(OUT, 'datapath.npz'), **out)
    print('datapath.npz', len(out))
    if os.environ.get('GOLDEN_ONLY') == 'dp':
        return

    out = {}
    run_unet_options(ref, out)
    np.savez_compressed(os.path.join(OUT, 'unet_options.npz'), **out)
    print('unet_options.npz', len(out))
    if os.environ.get('GOLDEN_ONLY') == 'opt':
        return

    out = {}
    run_unet_full(ref, out)
    np.savez_compressed(os.path.join(OUT, 'unet_full.npz'), **out)
    print('unet_full.npz', len(out))
    if os.environ.get('GOLDEN_ONLY') in ('eval', 'f'):
        return

    out = {}
    run_ops(ref, out)
    np.savez_compressed(os.path.join(OUT, 'ops.npz'), **out)
    print('ops.npz', len(out))

    out = {}
    tiny_cls = ref.unets.UNet_generator(encoders=nets.UNET_TINY['encoders'], decoders=nets.UNET_TINY['decoders'],
                                        act='LeakyReLU', maxpool=True, upsample=False, res=False)
    run_seg(ref, tiny_cls, 'UNET_TINY', 1, 5, (16, 24, 32), 2, out, 'seg_tiny', full=True)
    run_seg(ref, tiny_cls, 'UNET_TINY', 1, 5, (16, 24, 32), 2, out, 'seg_tiny', full=True, dtype=torch.float64, steps=1)
    np.savez_compressed(os.path.join(OUT, 'seg_tiny.npz'), **out)
    print('seg_tiny.npz', len(out))

    out = {}
    light = ref.nf.get_network('UNet_light')
    run_seg(ref, light, 'UNET_LIGHT', 1, 32, (16, 24, 32), 1, out, 'seg_light', full=False)
    run_seg(ref, light, 'UNET_LIGHT', 1, 32, (16, 24, 32), 1, out, 'seg_light', full=False, dtype=torch.float64, steps=1)
    np.savez_compressed(os.path.join(OUT, 'seg_light.npz'), **out)
    print('seg_light.npz', len(out))

    out = {}
    run_reg(ref, (20, 24, 20), out, 'reg_odd')
    run_reg(ref, (20, 24, 20), out, 'reg_odd', dtype=torch.float64, steps=1)
    run_reg(ref, (16, 24, 32), out, 'reg_even')
    np.savez_compressed(os.path.join(OUT, 'reg.npz'), **out)
    print('reg.npz', len(out))


if __name__ == '__main__':
    main()

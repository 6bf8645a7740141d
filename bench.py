#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: training volumes/sec at 160x192x160 fp32.

Headline workload (config.workload): BASELINE.json configs[1] "Seg-only 3D U-Net, batch=2, 160x192x160 fp32, 1xMI355X":
UNet_light (874 864 params) + fused softmax-Dice + Adam, one step = zero_grad / forward / loss / backward /
[flat-bucket gradient all-reduce] / Adam over a batch of 2 synthetic volumes per GPU (weak scaling: per-GPU batch fixed).
Inputs are resident in HBM before the timed region.  One JSON line on rank 0.

After the headline the default run also times configs[2] (reg-only) and configs[3]'s per-GPU shape (joint alternating step,
1 pair per GPU) for the same --steps / --warmup and reports them under "extra" (north_star's target is quoted on the joint step).

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself under torch.distributed.run)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import gc
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 matrix peak
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 matrix peak
# 'fp32_split' (the headline's matrix mode): every fp32 product is three fp16 x fp16 partial products on the fp16 pipe (same dense peak as
# bf16), so the ceiling of the ALGORITHMIC (2 * 27 * Cin * Cout per voxel) rate is that peak / 3
SPLIT_MFMA_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 3.0
PRECISION_NOTE = {
    'fp32': 'fp32 operands on v_mfma_f32_16x16x4_f32 (one fmaf per product)',
    'fp32_split': 'fp32 tensors; in the 3x3x3 convolutions every operand is scaled by a per-tile power of two and split into two fp16 terms '
                  '(h + l, 22 significand bits) and a product is the sum of three fp16 x fp16 partial products on v_mfma_f32_16x16x32_f16, fp32 '
                  'accumulate -- per product NARROWER than fp32 (bound 7e-7; small elements of a staged tile keep an absolute 2^-36..2^-39 of its maximum), over these '
                  'layers\' K >= 216 sums not further from double than the fmaf chain (tests/test_gpu_split.py); everything else is plain fp32',
    'bf16': 'operands of the 3x3x3 convolutions ROUNDED to bf16, fp32 accumulate; every tensor in HBM fp32 (the A/B of bf16_storage)',
    'bf16_storage': 'BASELINE configs[4] mixed precision: activations and their gradients between the layers STORED as bf16, bf16 x bf16 -> fp32 '
                    'matrix products, fp32 statistics / reductions / master weights / weight gradients / losses',
}


def set_precision(ops, mode):
    """matrix arithmetic + activation storage of a bench leg"""
    ops.set_matrix_precision('bf16' if mode == 'bf16_storage' else mode)
    ops.set_activation_storage('bf16' if mode == 'bf16_storage' else 'fp32')
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E peak

# algorithmic conv FLOPs per voxel of one TRAINING pass (forward + data gradient + weight gradient = 3 x forward)
SEG_TRAIN_FLOP_PER_VOXEL = 3 * 113520.0          # UNet_light: 1 673.9 GFLOP per 160x192x160 volume
REG_TRAIN_FLOP_PER_VOXEL = 3 * 32630.0           # VoxelMorph: 481.1 GFLOP per pair
FULL_UNET_NOTE = 'full UNet: step FLOPs not tabulated in SURVEY.md'

CONV_FWD_CALLS = ['da_conv3d_k3_fwd', 'da_conv3d_k3_fwd_bnstats', 'da_conv3d_k3_fwd_pro']
CONV_BWD_CALLS = ['da_conv3d_k3_dgrad', 'da_conv3d_k3_wgrad', 'da_conv3d_k3_wgrad_pro']


def conv_dims(key):
    """(C1, C2, N, D, H, W, Cout, stride) of a da_conv3d_k3_* call from its integer arguments, or None."""
    name, a = key
    if name in ('da_conv3d_k3_fwd', 'da_conv3d_k3_fwd_bnstats', 'da_conv3d_k3_dgrad', 'da_conv3d_k3_wgrad'):
        return tuple(a[:8])
    if name in ('da_conv3d_k3_fwd_pro', 'da_conv3d_k3_wgrad_pro'):          # stride 1 only; no stride argument
        return tuple(a[:7]) + (1,)
    return None


def conv_flops(key):
    """Algorithmic FLOPs of one da_conv3d_k3_* call (2*27*Cin*Cout per output voxel; forward = data gradient = weight gradient)."""
    d = conv_dims(key)
    if d is None:
        return 0
    C1, C2, N, D, H, W, Cout, stride = d
    Do, Ho, Wo = (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1
    return 2.0 * 27 * (C1 + C2) * Cout * N * Do * Ho * Wo


def conv_bytes(key):
    """Algorithmic HBM bytes of one da_conv3d_k3_fwd* call: the input read once + the output written once (weights are KBs)."""
    d = conv_dims(key)
    if d is None or key[0] not in CONV_FWD_CALLS:
        return 0
    C1, C2, N, D, H, W, Cout, stride = d
    Do, Ho, Wo = (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1
    es = 2.0 if key[1] and key[1][-1] == 'bf16' else 4.0              # bf16 activation storage (the twin's key carries the flag)
    return es * N * ((C1 + C2) * D * H * W + Cout * Do * Ho * Wo)


def cpu_baseline(shape, batch, n_classes, budget_s=20.0, keep_reference=False):
    """The oracle (plain-torch CPU restatement of the reference step, models/segmentation.py:141-157) timed on this box's host cores.
    Inputs are the STRUCTURED synthetic volumes (blocky labels, image = label / 31 + noise: SURVEY.md 8d "Dice-parity inputs"), weights
    the closed-form fill: the warm-up step doubles as the reference of `parity_fullsize` (keep_reference=True returns what the GPU side
    is compared with: initial weights, inputs, first-step loss and train-mode logits)."""
    from oracle import nets, steps
    from deepatlas_amd.lib.datasets import SyntheticSegDataset
    torch.manual_seed(230)
    spec = nets.UNET_LIGHT
    sd = nets.closed_form_fill(nets.unet_param_shapes(1, n_classes, spec['encoders'], spec['decoders']), seed=1)
    ds = SyntheticSegDataset(batch, shape, n_classes, seed=230)
    x = torch.stack([ds[i][0] for i in range(batch)])
    y = torch.stack([ds[i][1] for i in range(batch)])
    ref = dict(sd0={k: v.clone() for k, v in sd.items()}, x=x, y=y) if keep_reference else None
    opt = steps.Adam(steps.trainable(sd), lr=1e-3)
    t0 = time.time()
    loss0, logits0, _ = steps.seg_step(sd, opt, x, y, spec, n_classes)                # warm-up (allocator, thread pool) = the parity step
    warm = time.time() - t0
    if keep_reference:
        ref.update(loss=float(loss0.item()), logits=logits0)
        # the yardstick of the logits comparison: the reference arithmetic's OWN fp32-vs-fp64 distance on this network, batch and size
        # (train-mode forward in double: BatchNorm over batch statistics amplifies rounding ~2.5x per block)
        with torch.no_grad():
            sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in ref['sd0'].items()}
            l64 = nets.unet_forward(sd64, x.double(), spec, training=True)
            d = logits0.double() - l64
            ref.update(fp32_floor_rel_l2=float(d.norm() / l64.norm()), fp32_floor_max_abs=float(d.abs().max() / l64.abs().max()), logits64=l64)
            del sd64, d
    del logits0
    n, t0 = 0, time.time()
    while n < 2 or (time.time() - t0 < budget_s and n < 4):       # at least two timed steps (~20 s each at the metric's size)
        steps.seg_step(sd, opt, x, y, spec, n_classes)
        n += 1
    dt = (time.time() - t0) / n
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or torch.get_num_threads()
    except ImportError:
        phys = torch.get_num_threads()
    base = dict(value=batch / dt, unit='volumes/s', cores=int(phys), threads=torch.get_num_threads(), kind='port',
                sample='%d timed steps of the same workload (batch %d, %dx%dx%d) after 1 warm-up step of %.1f s; %.2f s/step; torch intra-op '
                       'threads = %d on %d physical cores' % (n, batch, shape[0], shape[1], shape[2], warm, dt, torch.get_num_threads(), int(phys)))
    return (base, ref) if keep_reference else base


def parity_fullsize(ref, n_classes, dev, modes, train_steps=150):
    """Whole-network parity at the metric's own size, outside every timed region.  modes[0] is the shipped matrix mode.
    (1) First step, per mode: the shipped training step (fused head + softmax + Dice, side-stream weight gradients) from the oracle's
        initial weights on the oracle's batch -- loss and train-mode logits against oracle.steps.seg_step.
    (2) "Dice vs CPU ref" (BASELINE's metric; models/segmentation.py:179-201): the modes[0] model trains `train_steps` more steps on the
        device (structured volumes: learnable, so the Dice is not the ~0 of an untrained net), its checkpoint is handed to the oracle
        (strict state_dict), and eval-mode logits, first-max argmax and per-class Dice from integer counts are compared on that SAME
        checkpoint -- per mode on the device, ONE eval-mode CPU forward.  (Two independently trained states differ by Adam's sign-like
        steps on rounding-level gradients, SURVEY.md 7: not a kernel property.)"""
    from oracle import nets, losses
    from deepatlas_amd import ops
    from deepatlas_amd.lib import evalMetrics as M
    from deepatlas_amd.lib.loss import get_loss_function
    from deepatlas_amd.lib.network_factory import get_network
    from deepatlas_amd.optim import FlatAdam
    import numpy as np
    prev = ops.set_matrix_precision(modes[0])
    out = []
    try:
        x, y = ref['x'].to(dev), ref['y'].to(dev)
        crit = get_loss_function('dice')(n_class=n_classes, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)

        def fresh(sd):
            m = get_network('UNet_light')(in_channel=1, n_classes=n_classes, bias=True, BN=True)
            m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
            return m.to(dev)
        o = ref['logits'].to(dev)
        trained = None
        for mode in modes:
            ops.set_matrix_precision(mode)
            twin = fresh(ref['sd0']).train()                       # train-mode logits (its BatchNorm buffers move; it is thrown away)
            with torch.no_grad():
                logits = ops.materialize_logits(twin(x))
            num, den = float((logits - o).double().norm()), float(o.double().norm())
            mx = float((logits - o).abs().max() / o.abs().max())
            vs64 = {}
            if ref.get('logits64') is not None:                 # the device against the fp64 evaluation, next to the fp32 oracle's own distance from it
                d64 = logits.double().cpu() - ref['logits64']
                vs64 = dict(logits_rel_l2_vs_fp64=float(d64.norm() / ref['logits64'].norm()), logits_max_abs_vs_fp64=float(d64.abs().max() / ref['logits64'].abs().max()),
                            oracle_fp32_rel_l2_vs_fp64=ref['fp32_floor_rel_l2'], oracle_fp32_max_abs_vs_fp64=ref['fp32_floor_max_abs'])
                del d64
            del twin, logits
            model = fresh(ref['sd0']).train()
            model.lazy_head = True
            opt = FlatAdam(model.parameters(), lr=1e-3)
            opt.zero_grad()
            loss = crit(model(x), y)
            loss.backward()
            opt.step()
            loss = float(loss.item())
            out.append(dict(matrix_precision=mode, loss=loss, oracle_loss=ref['loss'], loss_abs_diff=abs(loss - ref['loss']),
                            logits_rel_l2=num / den, logits_max_abs_over_max=mx, **vs64))
            if trained is None:
                for _ in range(train_steps):
                    opt.zero_grad()
                    last = crit(model(x), y)
                    last.backward()
                    opt.step()
                torch.cuda.synchronize()
                trained = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
                out[0].update(train_steps=1 + train_steps, trained_loss=float(last.item()))
            del model, opt
        del o
        # the oracle's eval path on the trained checkpoint (one eval-mode CPU forward)
        with torch.no_grad():
            oe = nets.unet_forward({k: v.clone() for k, v in trained.items()}, ref['x'], nets.UNET_LIGHT, training=False)
        o_dice = np.stack([losses.eval_dice_per_class(oe[i:i + 1], ref['y'][i:i + 1], n_classes)[0] for i in range(oe.shape[0])])
        o_am = torch.max(oe, 1)[1]
        top2 = torch.topk(oe, 2, dim=1)[0]
        near = (top2[:, 0] - top2[:, 1]) < 1e-4 * top2[:, 0].abs().clamp_min(1.0)
        both = ~np.isnan(o_dice)
        oe_dev, oe_norm = oe.to(dev), float(oe.double().norm())
        for mode, rec in zip(modes, out):
            ops.set_matrix_precision(mode)
            ev_model = fresh(trained).eval()
            with torch.no_grad():
                pred = ev_model(x)
            dice = M.metricEval('dice', pred, y)                                    # [N][C-1], exact integer counts on the device
            flips = torch.max(pred, 1)[1].cpu() != o_am
            rec.update(eval_logits_rel_l2=float((pred - oe_dev).double().norm()) / oe_norm,
                       eval_dice_mean=float(np.nanmean(dice)), oracle_eval_dice_mean=float(np.nanmean(o_dice)),
                       eval_dice_abs_diff=float(np.abs(dice[both] - o_dice[both]).max()) if both.any() else 0.0,
                       eval_dice_mean_abs_diff=abs(float(np.nanmean(dice)) - float(np.nanmean(o_dice))),
                       nan_pattern_equal=bool(np.array_equal(np.isnan(dice), np.isnan(o_dice))),
                       voxels=int(o_am.numel()), argmax_flips=int(flips.sum()), near_tie_flips=int((flips & near).sum()),
                       flips_away_from_ties=int((flips & ~near).sum()))
            del ev_model, pred
        return out
    finally:
        ops.set_matrix_precision(prev)


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_spawn(n):
    """`python bench.py --gpus N` without a launcher: check the devices, then run N ranks of this script under torch.distributed.run."""
    have = torch.cuda.device_count()
    if have < n and os.environ.get('DA_BENCH_SHARE_DEVICE') != '1':
        sys.stderr.write('bench.py: --gpus %d requested but only %d GPU(s) are visible on this node; refusing to run fewer ranks '
                         'and label them as %d\n' % (n, have, n))
        sys.exit(2)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


class Workload:
    """One timed configuration: step() closure + what one step amounts to."""

    def __init__(self, name, step, units, unit, flops_per_step, loss_of, optimizers=()):
        self.name, self.step, self.units, self.unit, self.flops_per_step, self.loss_of = name, step, units, unit, flops_per_step, loss_of
        self.optimizers = list(optimizers)          # the FlatAdam buckets this step all-reduces (one collective each)


def make_workloads(args, dev, rank, which):
    """Build the models / optimisers / resident inputs of the requested workloads ('seg', 'reg', 'joint')."""
    from deepatlas_amd import parallel
    from deepatlas_amd.lib.network_factory import get_network
    from deepatlas_amd.lib.loss import get_loss_function
    from deepatlas_amd.optim import FlatAdam
    n_classes = 32
    shape = tuple(args.shape)
    V = shape[0] * shape[1] * shape[2]
    torch.manual_seed(230)
    model = get_network(args.net)(in_channel=1, n_classes=n_classes, bias=True, BN=True)
    model.weights_init()
    model.to(dev).train()
    model.lazy_head = not args.no_fused_head       # head + softmax + Dice as one kernel pair (ops.HeadDiceFn), as SegmentationExperiment trains
    crit = get_loss_function('dice')(n_class=n_classes, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)
    opt = FlatAdam(model.parameters(), lr=1e-3)
    parallel.broadcast_parameters(opt)
    from deepatlas_amd.lib.datasets import synthetic_batch_on_device
    x, y = synthetic_batch_on_device(args.batch, shape, n_classes, seed=230 + rank, device=dev)

    from deepatlas_amd import trace        # roctx ranges (DA_ROCTX=1): phases of the step in a rocprofv3 --marker-trace

    def seg_grads():
        opt.zero_grad()
        with trace.range('seg/forward'):
            out = model(x)
        with trace.range('seg/loss'):
            loss = crit(out, y)
        with trace.range('seg/backward'):
            loss.backward()
        return dict(loss=loss.detach())

    def seg_step():
        loss = seg_grads()['loss']
        with trace.range('seg/allreduce'):
            parallel.allreduce_gradients(opt)
        with trace.range('seg/adam'):
            opt.step()
        return loss

    def graphed(segments, between, optimizers, key):
        """--graph: the step captured once as HIP graph(s) and replayed (deepatlas_amd/graphs.py); collectives stay outside the graphs"""
        from deepatlas_amd.graphs import GraphedStep
        g = GraphedStep(segments, optimizers, between=between, warmup=2)
        for _ in range(4):          # 2 eager steps, capture + first replay, one more replay: all before any timed region
            g()
        return lambda: g()[key]

    if args.graph:
        seg_step = graphed([seg_grads, lambda: opt.step()], [lambda: parallel.allreduce_gradients(opt)], [opt], 'loss')

    prec = {'fp32': 'fp32', 'fp32_split': 'fp32', 'bf16': 'bf16 matrix mode', 'bf16_storage': 'bf16 activations + bf16 matrix mode'}[args.precision]
    out = {}
    if 'seg' in which:
        if args.net == 'UNet_light':
            nm = 'seg-only UNet_light + softmax-Dice + Adam training step, batch %d/GPU, %dx%dx%d %s (BASELINE configs[1]%s)' % (
                args.batch, shape[0], shape[1], shape[2], prec, '' if not args.precision.startswith('bf16') else " shape with configs[4]'s precision")
            fl = SEG_TRAIN_FLOP_PER_VOXEL * V * args.batch
        else:
            nm = 'seg-only full UNet (32-512 ch) + softmax-Dice + Adam training step, batch %d/GPU, %dx%dx%d %s (SURVEY row f3, not a BASELINE config)' % (
                args.batch, shape[0], shape[1], shape[2], prec)
            fl = None
        out['seg'] = Workload(nm, seg_step, args.batch, 'volumes/s', fl, lambda r: r, [opt])
    if 'joint_smooth' in which:
        which = list(which) + ['joint']
    if 'reg' in which or 'joint' in which:
        from deepatlas_amd.models.joint import RegistrationStep, DeepAtlasJointStep
        reg = get_network('voxel_morph_cvpr')()
        reg.weights_init()
        smooth = 'joint_smooth' in which
        if smooth:
            # A registration-like displacement field for the joint leg's gather / scatter kernels: the untrained net's Xavier-initialised flow head
            # turns noise volumes into 12 - 19 voxels of noise that differ by 8 voxels between neighbours (tools/debug/joint_field_stats.py) -- a worst case
            # no registration run sees.  Here the flow weights are scaled by 2^-8 and the bias is a shift of (1.7, -1.3, 0.9) voxels: smooth, a few voxels,
            # +- 0.05 voxels of texture; the registration optimiser's step size is 1e-7 so that the field stays like that over the timed steps (Adam moves
            # every weight by ~lr per step whatever the gradient).  Same kernels, same launches; only the field differs.
            with torch.no_grad():
                reg.flow.weight.mul_(1.0 / 256.0)
                reg.flow.bias.copy_(torch.tensor([1.7 * 2.0 / (shape[2] - 1), -1.3 * 2.0 / (shape[1] - 1), 0.9 * 2.0 / (shape[0] - 1)]))
        reg.to(dev).train()
        ropt = FlatAdam(reg.parameters(), lr=1e-7 if smooth else 1e-3)
        parallel.broadcast_parameters(ropt)
        x2, y2 = synthetic_batch_on_device(1, shape, n_classes, seed=1230 + rank, device=dev)
        im_m, im_t, sm, st_ = x[:1], x2, y[:1], y2
        if smooth:
            # ... and label maps with anatomy-like structure (blocky regions, lib/datasets.structured_labels) instead of iid labels: with 32 different labels
            # inside every 32 x 8 x 4 box / every wave the label-keyed kernels (adjoint scatter through its LDS box: three label slots per box; label-warp Dice:
            # one histogram pass per distinct label of a wave) run their overflow paths whatever the field
            xs, ys = synthetic_batch_on_device(1, shape, n_classes, seed=230 + rank, device=dev, structured=True, sample0=0)
            xt, yt = synthetic_batch_on_device(1, shape, n_classes, seed=230 + rank, device=dev, structured=True, sample0=1)
            im_m, im_t, sm, st_ = xs, xt, ys, yt
        if 'reg' in which:
            rstep = RegistrationStep(reg, ropt)
            reg_fn = (lambda: rstep(im_m, im_t)[0]) if not args.graph else graphed(*rstep.segments(im_m, im_t), 'loss')
            out['reg'] = Workload('reg-only VoxelMorph + trilinear warp + NCC + bending + Adam, 1 pair/GPU, %dx%dx%d %s (BASELINE configs[2])'
                                  % (shape + (prec,)), reg_fn, 1, 'pairs/s', REG_TRAIN_FLOP_PER_VOXEL * V, lambda r: r, [ropt])
        if 'joint' in which:
            jstep = DeepAtlasJointStep(model, opt, reg, ropt, n_classes)
            joint_fn = (lambda: jstep(im_m, im_t, sm, st_)['loss_seg']) if not args.graph else graphed(*jstep.segments(im_m, im_t, sm, st_), 'loss_seg')
            out['joint'] = Workload('joint DeepAtlas alternating step (reg phase + seg phase, 32-ch seg warp), 1 pair/GPU, %dx%dx%d %s '
                                    '(BASELINE configs[3] per-GPU shape)' % (shape + (prec,)),
                                    joint_fn, 1, 'pairs/s',
                                    (SEG_TRAIN_FLOP_PER_VOXEL + REG_TRAIN_FLOP_PER_VOXEL) * V if args.net == 'UNet_light' else None, lambda r: r, [ropt, opt])
            if smooth:
                out['joint_smooth'] = out.pop('joint')
                out['joint_smooth'].name += ' -- on registration-like inputs: blocky label maps and a smooth field (shift of 1 - 2 voxels + 0.05 voxels of texture) instead of iid labels and the untrained net\'s 8-voxel noise'
    return out, n_classes


def time_workload(wl, args, world, dev, prof_names=None):
    """W warm-up steps, then EXACTLY K steps between barrier + synchronize; returns (max-over-ranks seconds, per-rank seconds, final loss,
    call profiler, C-ABI launches per step)."""
    from deepatlas_amd import _native as nat
    for _ in range(args.warmup):
        wl.step()
    prof = nat.CallProfiler(prof_names) if prof_names else None
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    nat.profiler = prof
    n0 = nat.n_calls
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = wl.step()
    t_issued = time.perf_counter()                 # the host has queued every launch of the K steps; the device is still working them off
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nat.profiler = None
    launches = (nat.n_calls - n0) / max(args.steps, 1)
    per_rank, host_rank = [dt], [t_issued - t0]
    if world > 1:
        t = torch.tensor([dt, t_issued - t0], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [float(v[0].item()) for v in allt]
        host_rank = [float(v[1].item()) for v in allt]
        dt = max(per_rank)
    time_workload.host_issue = host_rank           # (read by result_of: seconds each rank's Python loop took to ISSUE the K steps)
    return dt, per_rank, float(loss.item()), prof, launches


def result_of(wl, dt, per_rank, world, args, launches):
    ms = dt / args.steps * 1e3
    r = dict(value=round(world * wl.units * args.steps / dt, 4), unit=wl.unit, ms_per_step=round(ms, 3), workload=wl.name,
             c_abi_launches_per_step=round(launches, 1))
    if wl.flops_per_step:
        r['step_tflops'] = round(wl.flops_per_step / (ms * 1e-3) / 1e12, 2)
        if not args.precision.startswith('bf16'):
            r['step_frac_of_fp32_mfma_peak'] = round(wl.flops_per_step / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)
        if args.precision == 'fp32_split':
            r['step_frac_of_split_peak'] = round(wl.flops_per_step / (ms * 1e-3) / 1e12 / SPLIT_MFMA_PEAK_TFLOPS, 4)
    # host-side issue time of a step (Python + ctypes + HIP launch calls) next to the step time: a rank whose issue time approaches its step
    # time is host-bound -- with N launch loops on one host that is the first thing to look at when the scaling curve bends
    hi = getattr(time_workload, 'host_issue', None)
    if hi:
        r['host_issue_ms_per_step'] = round(max(hi) / args.steps * 1e3, 3)
        r['host_issue_frac_of_step'] = round(max(hi) / dt, 3)
    if world > 1:
        r['ms_per_step_per_rank'] = [round(t / args.steps * 1e3, 3) for t in per_rank]
        if hi:
            r['host_issue_ms_per_step_per_rank'] = [round(t / args.steps * 1e3, 3) for t in hi]
    return r


def call_table(summ, peak=FP32_MFMA_PEAK_TFLOPS):
    """{(name, ints): (n, ms)} -> list of per-call records sorted by total time; frac = algorithmic TFLOP/s / `peak`."""
    rows = []
    for k, (n, ms) in summ.items():
        fl = conv_flops(k)
        if not fl or not n:
            continue
        tf = fl * n / (ms * 1e-3) / 1e12
        rows.append(dict(call='%s%s' % (k[0], list(conv_dims(k))), launches=n, avg_ms=round(ms / n, 4), total_ms=round(ms, 3),
                         tflops=round(tf, 2), frac=round(tf / peak, 4), _key=k, _fl=fl))
    rows.sort(key=lambda r: -r['total_ms'])
    return rows


def pmc_traffic_for(kname, precision):
    """HBM bytes per call from the newest committed PMC passes (tools/pmc_conv.sh -> tools/pmc_summary.py): separate rocprofv3 --pmc passes
    (one counter block each) of the same C-ABI call on the same shape, the fetch counter corrected with the call's own request-size mix.  A
    separate rocprofv3 run, not a measurement of this process -- `traffic_source` says so.  Returns (record | None, reason)."""
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc_traffic.json')))
    if not files:
        return None, 'no profiles/rNN_pmc_traffic.json'
    try:
        doc = json.load(open(files[-1]))
        calls = doc['calls']
    except (OSError, ValueError, KeyError) as e:
        return None, '%s unreadable: %s' % (files[-1], e)
    if doc.get('matrix_precision', 'fp32') != precision:
        return None, '%s was collected in matrix mode %s, this run is %s' % (os.path.relpath(files[-1], ROOT), doc.get('matrix_precision'), precision)
    cands = [kname]
    if kname.startswith('da_conv3d_k3_fwd_pro['):
        body = kname[len('da_conv3d_k3_fwd_pro['):]
        cands.append('da_conv3d_k3_fwd_bnstats[' + body)
    if kname.startswith('da_conv3d_k3_wgrad_pro['):
        cands.append('da_conv3d_k3_wgrad[' + kname[len('da_conv3d_k3_wgrad_pro['):])
    for c in cands:
        rec = calls.get(c)
        if rec:
            out = dict(traffic=rec['traffic_bytes'],
                       traffic_source='%s (separate rocprofv3 --pmc passes of this call at commit %s; algorithmic %d B; fetch counter: %s)'
                                      % (os.path.relpath(files[-1], ROOT), doc.get('commit', '?'), rec['algorithmic_bytes'], doc.get('fetch_method', 'FETCH_SIZE calibrated')),
                       traffic_over_algorithmic=round(rec['traffic_bytes'] / rec['algorithmic_bytes'], 3))
            if 'sq' in rec:
                out['mfma_busy_vs_peak_clock'] = round(rec['sq']['mfma_util_vs_2p4ghz_peak'], 4)
            return out, None
    return None, '%s has no record of %s (has: %s) -- re-run tools/pmc_conv.sh + tools/pmc_summary.py' % (os.path.relpath(files[-1], ROOT), kname, ', '.join(sorted(calls)))


def step_traffic_for(workload, ms_per_step):
    """Fabric-side bytes of one WHOLE step from the newest committed profiles/rNN_step_traffic_<workload>.json (tools/pmc_step.sh: two rocprofv3 --pmc
    passes over this same bench workload, reads from the size-weighted request counters, writes from WRITE_SIZE; kernels serialised by the counter
    collection -- a separate run, `source` says so) against SURVEY section 8(d)'s algorithmic bytes, and what both mean at this run's step time."""
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_step_traffic_%s.json' % workload)))
    if not files:
        return None
    try:
        doc = json.load(open(files[-1]))
        tot, alg = float(doc['total']), float(doc['algorithmic'])
    except (OSError, ValueError, KeyError, TypeError):
        return None
    top = sorted(doc.get('per_kernel', {}).items(), key=lambda kv: -(kv[1]['read'] + kv[1]['write']))[:5]
    return dict(step_traffic=tot, step_traffic_read=doc.get('read'), step_traffic_write=doc.get('write'), step_algorithmic_bytes=alg,
                step_traffic_over_algorithmic=round(tot / alg, 3),
                step_fabric_gbs=round(tot / (ms_per_step * 1e-3) / 1e9, 1), step_frac_of_hbm_peak_on_traffic=round(tot / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                step_frac_of_hbm_peak_on_algorithmic_bytes=round(alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                step_traffic_top_kernels=[dict(kernel=k[:80], bytes=round(v['read'] + v['write'])) for k, v in top],
                step_traffic_source='%s (tools/pmc_step.sh: separate rocprofv3 --pmc passes over `bench.py --workload %s`)' % (os.path.relpath(files[-1], ROOT), workload))


ROOFLINE_LAYER_CALLS = ('da_conv3d_k3_fwd_bnstats[32, 16, 2, 160, 192, 160, 16, 1]', 'da_conv3d_k3_dgrad[32, 16, 2, 160, 192, 160, 16, 1]',
                        'da_conv3d_k3_wgrad[32, 16, 2, 160, 192, 160, 16, 1]')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--shape', type=int, nargs=3, default=[160, 192, 160])
    ap.add_argument('--batch', type=int, default=2, help='volumes per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true', help='skip per-call HIP-event timing')
    ap.add_argument('--no-extra', action='store_true', help="skip the reg / joint legs and the post-run backward-kernel timing pass")
    ap.add_argument('--precision', default='fp32_split', choices=['fp32_split', 'fp32', 'bf16', 'bf16_storage'],
                    help="matrix arithmetic of the 3x3x3 convolutions.  'fp32_split' (headline): fp32 operands scaled per tile and split into two fp16 "
                         "terms, three partial products per multiply on the fp16 pipe, fp32 accumulate -- 22-bit products (per product narrower than fp32); 'fp32': the fp32 matrix "
                         "instructions (one fmaf per product; also timed by the default run, under extra.native_fp32_mfma); 'bf16': operands "
                         "ROUNDED to bf16 (BASELINE configs[4]'s mixed precision; never the headline)")
    ap.add_argument('--graph', action='store_true', help='capture each step once as HIP graph(s) and replay it (one host call per step; per-call '
                    'HIP-event timing is then taken in the eager post-run pass only)')
    ap.add_argument('--no-fused-head', action='store_true', help='run the 1x1x1 head, softmax and Dice as separate kernels (logits materialised)')
    ap.add_argument('--sync-wgrad', action='store_true', help='weight gradients on the main stream (default: side stream, overlapped with the HBM-bound backward kernels)')
    ap.add_argument('--net', default='UNet_light', choices=['UNet_light', 'UNet'],
                    help="segmentation network of the 'seg' workload; 'UNet' = the fixed 19 M-parameter net (SURVEY.md row f3), not the headline config")
    ap.add_argument('--workload', default='seg', choices=['seg', 'reg', 'joint', 'joint_smooth'],
                    help="headline leg: 'seg' = BASELINE configs[1] (the metric); 'reg' / 'joint' = configs[2] / [3] per-GPU shapes (1 pair / GPU)")
    args = ap.parse_args()

    want = max(args.gpus, 1)
    if 'WORLD_SIZE' not in os.environ and want > 1:
        self_spawn(want)                                   # never returns
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != want:
        sys.stderr.write('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks\n' % (want, world))
        sys.exit(2)
    # DA_BENCH_SHARE_DEVICE=1 (tests only: tests/test_gpu_dp.py): the N ranks share the visible device(s) and talk over gloo, so the
    # multi-rank code path (spawn, barriers, per-rank timing, the all-reduce leg) runs on a one-GPU box; the numbers mean nothing
    share = os.environ.get('DA_BENCH_SHARE_DEVICE') == '1'
    if torch.cuda.device_count() <= local_rank and not share:
        sys.stderr.write('bench.py: rank %d needs device %d but only %d GPU(s) are visible\n' % (rank, local_rank, torch.cuda.device_count()))
        sys.exit(2)
    headline_default = (args.workload == 'seg' and args.net == 'UNet_light' and args.precision == 'fp32_split' and tuple(args.shape) == (160, 192, 160)
                        and args.batch == 2 and not args.no_extra and not args.no_profile and not args.graph)
    if headline_default and world == 1 and os.environ.get('DA_BENCH_ALLOW_NO_TRAFFIC') != '1':      # (one rank: a lone exit cannot strand peers in a rendezvous)
        # `roofline.traffic` comes from committed PMC passes; a missing record must stop the run BEFORE anything is timed, not become `traffic: null`
        for c in ROOFLINE_LAYER_CALLS:
            rec, why = pmc_traffic_for(c, args.precision)
            if rec is None:
                sys.stderr.write('bench.py: roofline.traffic has no source: %s\n(set DA_BENCH_ALLOW_NO_TRAFFIC=1 to run anyway and report traffic: null)\n' % why)
                sys.exit(4)
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if os.environ.get('DA_MAIN_PRIO'):      # experiment: the whole run on a stream of this HIP priority (-1 = high; the side stream stays at 0)
        torch.cuda.set_stream(torch.cuda.Stream(priority=int(os.environ['DA_MAIN_PRIO'])))
    rccl = None
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if share:
            dist.init_process_group('gloo', rank=rank, world_size=world)
            rccl = 'gloo (DA_BENCH_SHARE_DEVICE test mode)'
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
            try:
                rccl = '.'.join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                rccl = 'unknown'

    from deepatlas_amd import _native as nat, ops
    ops.enable_async_wgrad(not args.sync_wgrad)
    set_precision(ops, args.precision)
    shape = tuple(args.shape)
    extra_legs = [] if (args.no_extra or args.workload != 'seg' or args.net != 'UNet_light') else ['reg', 'joint', 'joint_smooth']
    # Only the headline workload exists while it is timed; the other legs are built right before their own timed region and dropped after it.
    # (Where a workload's tensors land in HBM depends on what was allocated before them, and that is worth 1 - 4 % of a step: with all
    # three workloads built up front the joint leg ran at 24.13 ms against 23.19 ms for `bench.py --workload joint` on the same box, and the
    # headline at 27.05 against 26.81; built one at a time every leg matches its stand-alone run.)
    wls, n_classes = make_workloads(args, dev, rank, [args.workload])
    # Python's cyclic collector walks every live container each time it runs a full collection; with three models, their optimisers and
    # the autograd graphs of a step alive that costs the (host-bound) small-volume and bf16 legs milliseconds per step.  Standard
    # training-loop hygiene: move everything built so far out of the collector's reach.
    from deepatlas_amd import parallel as _par
    pinned = _par.pin_host_resources(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', str(world))))       # gc.freeze() + per-rank core slice

    # ---- headline: the timed region carries HIP-event timing of the FORWARD conv calls only.  The backward pass runs its weight
    # gradients on a second stream (ops.ASYNC_WGRAD): timing events recorded there serialise it against the main stream and
    # overlapping kernels time each other's slowdown, so backward kernels are timed in a separate short pass after the timed region.
    head = wls[args.workload]
    dt, per_rank, final_loss, prof, launches = time_workload(head, args, world, dev, None if (args.no_profile or args.graph) else CONV_FWD_CALLS)
    head_res = result_of(head, dt, per_rank, world, args, launches)
    # the same K steps with the host reading the loss every step, as SegmentationExperiment.train_one_epoch does (models/segmentation.py:160
    # of the reference: `loss.item()` per iteration): one device synchronisation per step
    sync_ms = None
    if world == 1 and not args.no_extra:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            float(head.step().item())
        torch.cuda.synchronize()
        sync_ms = round((time.perf_counter() - t0) / args.steps * 1e3, 3)

    allreduce = None
    if world > 1:
        # the step's only collective, timed on its own: the flat-bucket gradient all-reduce(s) of one step (seg 3.5 MB; reg 1.0 MB),
        # 20 back-to-back calls between device synchronisations, maximum over the ranks
        buckets = head.optimizers
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            for o in buckets:
                _par.allreduce_gradients(o)
        torch.cuda.synchronize()
        tt = torch.tensor([(time.perf_counter() - t0) / 20 * 1e3], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        allreduce = dict(ms_per_step=round(float(tt.item()), 4), collectives_per_step=len(buckets), bytes=[int(o.flat_g.numel() * 4) for o in buckets],
                         note="the step's flat-bucket gradient all-reduce(s) (average inside the collective on RCCL) timed on their own: mean of 20, max over ranks")

    bwd_rows = []
    peak = SPLIT_MFMA_PEAK_TFLOPS if args.precision == 'fp32_split' else FP32_MFMA_PEAK_TFLOPS
    if prof is not None and not args.no_extra and not args.precision.startswith('bf16'):
        # post-run pass: 3 steps with every conv call timed and the weight gradients on the MAIN stream (nothing overlaps: clean
        # per-kernel durations of the data / weight gradients).  Not part of `value`.
        ops.enable_async_wgrad(False)
        p2 = nat.CallProfiler(CONV_FWD_CALLS + CONV_BWD_CALLS)
        head.step()
        torch.cuda.synchronize()
        nat.profiler = p2
        for _ in range(3):
            head.step()
        torch.cuda.synchronize()
        nat.profiler = None
        ops.enable_async_wgrad(not args.sync_wgrad)
        bwd_rows = call_table(p2.summary(), peak)

    extra = {}
    for leg in extra_legs:
        wl = make_workloads(args, dev, rank, [leg])[0][leg]
        gc.freeze()                                   # (as pin_host_resources did for the headline workload)
        edt, eper, eloss, _, elaunch = time_workload(wl, args, world, dev, None)
        extra[leg] = dict(result_of(wl, edt, eper, world, args, elaunch), final_loss=round(eloss, 6))
        del wl
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    if args.precision == 'fp32_split' and not args.no_extra and not args.graph:
        # the same headline workload on the fp32 matrix instructions (mode 'fp32'), same --steps / --warmup: the A/B of the split mode
        set_precision(ops, 'fp32')
        a2 = argparse.Namespace(**dict(vars(args), precision='fp32'))
        edt, eper, eloss, _, elaunch = time_workload(head, a2, world, dev, None)
        extra['native_fp32_mfma'] = dict(result_of(head, edt, eper, world, a2, elaunch), final_loss=round(eloss, 6),
                                         matrix_arithmetic=PRECISION_NOTE['fp32'])
        set_precision(ops, args.precision)

    if not args.no_extra and args.workload == 'seg' and args.net == 'UNet_light' and args.precision == 'fp32_split' and tuple(shape) == (160, 192, 160) and not args.graph:
        # BASELINE configs[4]'s shape and precision, driver-timed: seg (batch 2) and the joint step (1 pair) at 192 x 224 x 192 with the 3x3x3
        # convolutions' operands rounded to bf16 (HBM tensors stay fp32: bf16 STORAGE is not implemented), and the same two legs in the
        # shipped fp32_split mode.  Fewer steps (the legs are 1.7x larger); fresh models.
        a4 = argparse.Namespace(**dict(vars(args), shape=[192, 224, 192], steps=max(2, min(args.steps, 6)), warmup=2))
        big = {}
        for mode in ('bf16_storage', 'bf16', 'fp32_split'):
            set_precision(ops, mode)
            am = argparse.Namespace(**dict(vars(a4), precision=mode))
            w4, _ = make_workloads(am, dev, rank, ['seg', 'joint'])
            for leg in ('seg', 'joint'):
                edt, eper, eloss, _, elaunch = time_workload(w4[leg], am, world, dev, None)
                big['%s_%s' % (leg, mode)] = dict(result_of(w4[leg], edt, eper, world, am, elaunch), final_loss=round(eloss, 6), steps=am.steps)
            del w4
            torch.cuda.empty_cache()
        big['note'] = ("BASELINE configs[4] shape; 'bf16_storage' = configs[4]'s mixed precision (bf16 activations / gradients in HBM + bf16 matrix "
                       "operands, fp32 everything else); 'bf16' = only the matrix operands rounded, every tensor fp32 in HBM; neither is fp32-accurate")
        big['bf16_storage_conversion_bridges'] = dict(ops.bridged_calls)      # entry points that ran through conversion passes (no bf16 twin for the shape)
        extra['configs4_192x224x192'] = big
        set_precision(ops, args.precision)

    if rank == 0:
        roofline = None
        if prof is not None:
            rows = call_table(prof.summary(), peak)
            tot_fl = sum(r['_fl'] * r['launches'] for r in rows)
            tot_ms = sum(r['total_ms'] for r in rows)
            top = rows[0]                                                       # forward call with the most time in the timed region
            if args.precision.startswith('bf16'):     # bf16 matrix mode: the convolutions are no longer matrix-bound -> price the call against HBM
                gbs = conv_bytes(top['_key']) * top['launches'] / (top['total_ms'] * 1e-3) / 1e9
                rl_head = dict(bound='hbm', achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(gbs / HBM_PEAK_GBS, 4), traffic=None)
            elif args.precision == 'fp32_split':
                rl_head = dict(bound='mfma', achieved=top['tflops'], peak=round(SPLIT_MFMA_PEAK_TFLOPS, 1), unit='TFLOP/s', frac=top['frac'], traffic=None,
                               peak_note='algorithmic fp32 FLOPs (2*27*Cin*Cout per voxel) against the dense fp16 matrix peak / 3: the split mode '
                                         'issues three fp16 MFMA products per fp32 multiply',
                               frac_of_fp32_mfma_peak=round(top['tflops'] / FP32_MFMA_PEAK_TFLOPS, 4),
                               bound_note='power-limited clock: the matrix pipe alone sustains 659 TFLOP/s of this arithmetic at 1.90 GHz on random '
                                          'operands, the K loop with its LDS fragment reads 525 - 580 at 1.6 - 1.75 GHz (profiles/r04_ubench_f16_kloop.txt); '
                                          'the kernel adds staging loads (25 %) and LDS writes + barriers (10 %), profiles/r04_conv_fwd_ablation.txt')
            else:
                rl_head = dict(bound='mfma', achieved=top['tflops'], peak=FP32_MFMA_PEAK_TFLOPS, unit='TFLOP/s', frac=top['frac'], traffic=None)
            roofline = dict(rl_head, kernel=top['call'], avg_ms=top['avg_ms'], launches=top['launches'], flops_per_launch=top['_fl'],
                            measured='HIP events on the launch stream inside the timed region (forward pass: nothing else in flight)',
                            traffic_source=None, profiled_calls=CONV_FWD_CALLS,
                            all_profiled=dict(tflops=round(tot_fl / (tot_ms * 1e-3) / 1e12, 2), ms_per_step=round(tot_ms / args.steps, 3),
                                              frac_of_step=round(tot_ms / args.steps / head_res['ms_per_step'], 3)))
            if not args.precision.startswith('bf16'):
                pm, _why = pmc_traffic_for(top['call'], args.precision)
                if pm:
                    roofline.update(pm)
            if head.flops_per_step and not args.precision.startswith('bf16'):
                # algorithmic conv FLOPs per step / ms_per_step / the peak of the mode
                roofline['step_frac'] = head_res['step_frac_of_split_peak' if args.precision == 'fp32_split' else 'step_frac_of_fp32_mfma_peak']
                roofline['step_frac_of_fp32_mfma_peak'] = head_res['step_frac_of_fp32_mfma_peak']
                roofline['step_tflops'] = head_res['step_tflops']
            if bwd_rows:
                # the same layer's three kernels (forward / data gradient / weight gradient) without overlap, and the slowest training
                # kernel overall = the call with the most time in the post-run pass
                lay = conv_dims(top['_key'])
                same = [r for r in bwd_rows if conv_dims(r['_key']) == lay]
                slow = bwd_rows[0]
                strip = lambda r: {k: v for k, v in r.items() if not k.startswith('_')}
                # `kernel` / `achieved` / `frac` = the DOMINANT kernel of the training step (most time over forward + backward: a weight
                # gradient, which the timed region runs on the side stream); the forward call timed inside the timed region stays as `forward`
                roofline['forward'] = dict(kernel=roofline['kernel'], achieved=roofline['achieved'], frac=roofline['frac'], avg_ms=roofline['avg_ms'],
                                           launches=roofline['launches'], measured=roofline['measured'])
                roofline.update(kernel=slow['call'], achieved=slow['tflops'], frac=slow['frac'], avg_ms=slow['avg_ms'], launches=slow['launches'],
                                flops_per_launch=slow['_fl'],
                                measured='HIP events on the launch stream, post-run pass of 3 steps right after the timed region with the weight gradients '
                                         'on the main stream (inside the timed region they overlap other kernels on a side stream); the call with the '
                                         'most time over forward + backward')
                if args.precision == 'fp32_split':
                    roofline['frac_of_fp32_mfma_peak'] = round(slow['tflops'] / FP32_MFMA_PEAK_TFLOPS, 4)
                pm2, why2 = pmc_traffic_for(slow['call'], args.precision)
                roofline['traffic'] = pm2['traffic'] if pm2 else None
                roofline['traffic_source'] = pm2['traffic_source'] if pm2 else ('MISSING: ' + why2)
                roofline['traffic_over_algorithmic'] = pm2.get('traffic_over_algorithmic') if pm2 else None
                roofline['mfma_busy_vs_peak_clock'] = pm2.get('mfma_busy_vs_peak_clock') if pm2 else None
                if pm2 is None:
                    sys.stderr.write('bench.py: roofline.traffic is null: %s\n' % why2)
                if args.precision == 'fp32_split':
                    # what the chip sustains of this arithmetic on random operands (tools/ubench/split_f16_kloop.hip, profiles/r04_ubench_f16_kloop.txt):
                    # matrix pipe alone 659 TFLOP/s at the power-limited 1.90 GHz; the K loop with its LDS fragment reads 525 (row pairs) - 580
                    roofline['measured_ceilings_tflops'] = dict(mfma_only=659.1, kloop_with_lds_reads=524.6, kloop_halo_row_reuse=579.6)
                    roofline['frac_of_measured_kloop_ceiling'] = round(slow['tflops'] / 524.6, 4)
                roofline['post_run_pass'] = dict(
                    note='3 extra steps after the timed region, weight gradients on the main stream, every conv call timed with HIP events',
                    roofline_layer=[strip(r) for r in same],
                    slowest_training_kernel=strip(slow),
                    lowest_frac_heavy_kernel=strip(min((r for r in bwd_rows if r['total_ms'] >= 0.05 * sum(q['total_ms'] for q in bwd_rows)),
                                                       key=lambda r: r['frac'], default=slow)),
                    all_conv_calls=dict(tflops=round(sum(r['_fl'] * r['launches'] for r in bwd_rows) / (sum(r['total_ms'] for r in bwd_rows) * 1e-3) / 1e12, 2),
                                        ms_per_step=round(sum(r['total_ms'] for r in bwd_rows) / 3, 3)))
        metric = 'training volumes/sec at 160x192x160 fp32; Dice vs CPU ref'          # BASELINE.json's metric (the default invocation)
        if shape != (160, 192, 160) or args.precision.startswith('bf16'):
            metric = 'training volumes/sec at %dx%dx%d %s' % (shape[0], shape[1], shape[2], 'fp32' if not args.precision.startswith('bf16') else ('bf16 matrix mode' if args.precision == 'bf16' else 'bf16 activations + bf16 matrix mode'))
        line = dict(metric=metric, value=head_res['value'], unit='volumes/s',
                    n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=head_res['ms_per_step'],
                    higher_is_better=True, scaling='weak', vs_baseline=None,
                    dtype='f32' if not args.precision.startswith('bf16') else 'bf16 x bf16 -> f32 in the 3x3x3 convolutions, f32 elsewhere', data='synthetic',
                    config=dict(workload=head.name, global_batch=world * head.units, volume=list(shape), n_classes=n_classes,
                                parallelism='dp%d' % world, final_loss=round(final_loss, 6), matrix_precision=args.precision,
                                matrix_arithmetic=PRECISION_NOTE[args.precision],
                                c_abi_launches_per_step=head_res['c_abi_launches_per_step'], hip_graph=bool(args.graph), rccl=rccl,
                                ms_per_step_per_rank=head_res.get('ms_per_step_per_rank'), host_issue_ms_per_step=head_res.get('host_issue_ms_per_step'),
                                host_issue_ms_per_step_per_rank=head_res.get('host_issue_ms_per_step_per_rank'), allreduce=allreduce,
                                host_cores_per_rank=pinned[1] if pinned else None),
                    roofline=roofline, extra=extra or None)
        if world == 1 and not args.no_cpu_baseline and args.workload == 'seg' and args.net == 'UNet_light':
            # CPU leg: the oracle's step timed on the host cores; its first step is also the reference of the full-size parity block
            # (the GPU side runs it once per matrix mode, outside every timed region)
            line['cpu_baseline'], ref = cpu_baseline(shape, args.batch, n_classes, keep_reference=True)
            modes = [args.precision] + (['fp32'] if args.precision == 'fp32_split' else [])
            par = parity_fullsize(ref, n_classes, dev, modes)
            line['parity_fullsize'] = dict(par[0], note='shipped step vs oracle.steps.seg_step on the same closed-form weights and structured batch '
                                                       '(batch %d, %dx%dx%d): first-step loss / train-mode logits; eval logits / argmax / Dice of '
                                                       'the device-trained checkpoint vs the oracle\'s eval path on that checkpoint'
                                                       % ((args.batch,) + shape),
                                           other_modes=par[1:] or None)
        else:
            line['cpu_baseline'] = None
        # flat copies of the numbers a reader of the driver's record needs (its parser keeps top-level scalars only)
        if sync_ms is not None:
            line['ms_per_step_with_loss_item'] = sync_ms
        if head_res.get('host_issue_ms_per_step') is not None:
            line['host_issue_ms_per_step'] = head_res['host_issue_ms_per_step']
        for leg in ('reg', 'joint', 'joint_smooth'):
            if leg in extra:
                line['%s_ms_per_step' % leg] = extra[leg]['ms_per_step']
        if 'native_fp32_mfma' in extra:
            line['native_fp32_mfma_ms_per_step'] = extra['native_fp32_mfma']['ms_per_step']
        pf = line.get('parity_fullsize')
        if pf:
            if 'logits_max_abs_vs_fp64' in pf:
                line.update(parity_logits_max_abs_vs_fp64=pf['logits_max_abs_vs_fp64'], parity_oracle_fp32_max_abs_vs_fp64=pf['oracle_fp32_max_abs_vs_fp64'])
            line.update(parity_loss_abs_diff=pf['loss_abs_diff'], parity_logits_rel_l2=pf['logits_rel_l2'], parity_logits_max_abs=pf['logits_max_abs_over_max'],
                        parity_eval_dice_abs_diff=pf['eval_dice_abs_diff'], argmax_flips_away_from_ties=pf['flips_away_from_ties'])
        if roofline:
            stt = step_traffic_for(args.workload, head_res['ms_per_step']) if (args.precision == 'fp32_split' and tuple(shape) == (160, 192, 160)) else None
            if stt:
                roofline.update(stt)
                line.update(step_traffic_over_algorithmic=stt['step_traffic_over_algorithmic'], step_frac_of_hbm_peak_on_traffic=stt['step_frac_of_hbm_peak_on_traffic'])
            for leg in ('reg', 'joint'):
                if leg in extra and args.precision == 'fp32_split' and tuple(shape) == (160, 192, 160):
                    st2 = step_traffic_for(leg, extra[leg]['ms_per_step'])
                    if st2:
                        extra[leg]['traffic'] = st2
            line.update(roofline_kernel=roofline['kernel'], roofline_frac=roofline['frac'], roofline_avg_ms=roofline['avg_ms'])
        # the driver's record keeps `config` and `roofline` but drops unknown top-level keys: the same scalars once more where they survive
        cfg = line['config']
        cfg['matrix_arithmetic_bits'] = {'fp32_split': 22, 'fp32': 24, 'bf16': 8, 'bf16_storage': 8}[args.precision]
        for k in ('ms_per_step_with_loss_item', 'reg_ms_per_step', 'joint_ms_per_step', 'joint_smooth_ms_per_step', 'native_fp32_mfma_ms_per_step', 'parity_logits_max_abs_vs_fp64',
                  'parity_oracle_fp32_max_abs_vs_fp64', 'parity_loss_abs_diff', 'parity_logits_rel_l2', 'parity_logits_max_abs', 'parity_eval_dice_abs_diff',
                  'argmax_flips_away_from_ties'):
            if k in line:
                cfg[k] = line[k]
        if roofline and 'native_fp32_mfma_ms_per_step' in line:
            roofline['native_fp32_mfma_ms_per_step'] = line['native_fp32_mfma_ms_per_step']
            roofline['native_fp32_mfma_volumes_per_s'] = extra['native_fp32_mfma']['value']
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

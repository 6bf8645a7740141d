#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: training volumes/sec at 160x192x160 fp32.

Workload (config.workload): BASELINE.json configs[1] "Seg-only 3D U-Net, batch=2, 160x192x160 fp32, 1xMI355X":
UNet_light (874 864 params) + fused softmax-Dice + Adam, one step = zero_grad / forward / loss / backward /
[flat-bucket gradient all-reduce] / Adam over a batch of 2 synthetic volumes per GPU (weak scaling: per-GPU batch fixed).
Inputs are resident in HBM before the timed region.  One JSON line on rank 0.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 matrix peak
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E peak
HBM_PEAK_GBS = 8000.0


def conv_flops(key):
    """Algorithmic FLOPs of one da_conv3d_k3_* call from its integer arguments (2*27*Cin*Cout per output voxel)."""
    name, a = key
    if name in ('da_conv3d_k3_fwd', 'da_conv3d_k3_fwd_bnstats'):      # C1, C2, N, D, H, W, Cout, stride
        C1, C2, N, D, H, W, Cout, stride = a[:8]
    elif name == 'da_conv3d_k3_fwd_pro':                              # C1, C2, N, D, H, W, Cout, stats capacity (stride 1)
        C1, C2, N, D, H, W, Cout = a[:7]; stride = 1
    elif name == 'da_conv3d_k3_dgrad':  # C1, C2, N, D, H, W, Cout, stride
        C1, C2, N, D, H, W, Cout, stride = a[:8]
    elif name == 'da_conv3d_k3_wgrad':
        C1, C2, N, D, H, W, Cout, stride = a[:8]
    else:
        return 0
    Do, Ho, Wo = (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1
    return 2.0 * 27 * (C1 + C2) * Cout * N * Do * Ho * Wo


def conv_bytes(key):
    """Algorithmic HBM bytes of one da_conv3d_k3_fwd* call: the input read once + the output written once, fp32 (weights are KBs)."""
    name, a = key
    if name not in ('da_conv3d_k3_fwd', 'da_conv3d_k3_fwd_bnstats', 'da_conv3d_k3_fwd_pro'):
        return 0
    C1, C2, N, D, H, W, Cout, stride = (tuple(a[:7]) + (1,)) if name == 'da_conv3d_k3_fwd_pro' else a[:8]
    Do, Ho, Wo = (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1
    return 4.0 * N * ((C1 + C2) * D * H * W + Cout * Do * Ho * Wo)


def cpu_baseline(shape, batch, n_classes, budget_s=20.0):
    """The oracle (plain-torch CPU restatement of the reference step) timed on this box's host cores."""
    from oracle import nets, steps
    torch.manual_seed(230)
    spec = nets.UNET_LIGHT
    sd = nets.closed_form_fill(nets.unet_param_shapes(1, n_classes, spec['encoders'], spec['decoders']), seed=1)
    g = torch.Generator().manual_seed(230)
    x = torch.rand((batch, 1) + shape, generator=g)
    y = torch.randint(0, n_classes, (batch,) + shape, generator=g, dtype=torch.uint8)
    opt = steps.Adam(steps.trainable(sd), lr=1e-3)
    t0 = time.time()
    steps.seg_step(sd, opt, x, y, spec, n_classes)                # warm-up (allocator, thread pool)
    warm = time.time() - t0
    n, t0 = 0, time.time()
    while True:
        steps.seg_step(sd, opt, x, y, spec, n_classes)
        n += 1
        if time.time() - t0 > budget_s or n >= 3:
            break
    dt = (time.time() - t0) / n
    return dict(value=batch / dt, unit='volumes/s', cores=torch.get_num_threads(), kind='port',
                sample='%d timed step(s) of the same workload (batch %d, %dx%dx%d) after 1 warm-up step of %.1f s; %.2f s/step'
                       % (n, batch, shape[0], shape[1], shape[2], warm, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--shape', type=int, nargs=3, default=[160, 192, 160])
    ap.add_argument('--batch', type=int, default=2, help='volumes per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true', help='skip per-call HIP-event timing')
    ap.add_argument('--profile-all', action='store_true', help='HIP-event timing of dgrad / wgrad calls too (perturbs the two-stream overlap)')
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'],
                    help="matrix arithmetic of the 3x3x3 convolutions: 'fp32' = the reference's arithmetic (the headline metric); 'bf16' = "
                         "bf16 operands, fp32 accumulate, fp32 tensors (BASELINE configs[4]'s mixed precision; not the headline)")
    ap.add_argument('--sync-wgrad', action='store_true', help='weight gradients on the main stream (default: side stream, overlapped with the HBM-bound backward kernels)')
    ap.add_argument('--net', default='UNet_light', choices=['UNet_light', 'UNet'],
                    help="segmentation network of the 'seg' workload; 'UNet' = the fixed 19 M-parameter net (SURVEY.md row f3), not the headline config")
    ap.add_argument('--workload', default='seg', choices=['seg', 'reg', 'joint'],
                    help="'seg' = BASELINE configs[1] (the headline metric); 'reg' / 'joint' = configs[2] / [3] per-GPU shapes (1 pair / GPU)")
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == max(args.gpus, 1) or world == 1, 'launch with torchrun --nproc-per-node %d' % args.gpus
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from deepatlas_amd import _native as nat, parallel, ops
    ops.enable_async_wgrad(not args.sync_wgrad)
    ops.set_matrix_precision(args.precision)
    from deepatlas_amd.lib.network_factory import get_network
    from deepatlas_amd.lib.loss import get_loss_function
    from deepatlas_amd.optim import FlatAdam

    n_classes = 32
    shape = tuple(args.shape)
    torch.manual_seed(230)
    model = get_network(args.net)(in_channel=1, n_classes=n_classes, bias=True, BN=True)
    model.weights_init()
    model.to(dev).train()
    crit = get_loss_function('dice')(n_class=n_classes, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)
    opt = FlatAdam(model.parameters(), lr=1e-3)
    parallel.broadcast_parameters(opt)
    g = torch.Generator().manual_seed(230 + rank)
    x = torch.rand((args.batch, 1) + shape, generator=g).to(dev)
    y = torch.randint(0, n_classes, (args.batch,) + shape, generator=g, dtype=torch.uint8).to(dev)

    def step():
        opt.zero_grad()
        out = model(x)
        loss = crit(out, y)
        loss.backward()
        parallel.allreduce_gradients(opt)
        opt.step()
        return loss

    units_per_step = args.batch
    workload_name = ('seg-only UNet_light + softmax-Dice + Adam training step, batch %d/GPU, %dx%dx%d fp32 (BASELINE configs[1])' if args.net == 'UNet_light'
                     else 'seg-only full UNet (32-512 ch) + softmax-Dice + Adam training step, batch %d/GPU, %dx%dx%d fp32 (SURVEY row f3, not a BASELINE config)') % (
        args.batch, shape[0], shape[1], shape[2])
    if args.precision == 'bf16':
        workload_name = workload_name.replace('fp32 (', 'bf16 matrix mode (').replace('BASELINE configs[1]', "BASELINE configs[1] shape with configs[4]'s precision")
    if args.workload in ('reg', 'joint'):
        from deepatlas_amd.models.joint import RegistrationStep, DeepAtlasJointStep
        reg = get_network('voxel_morph_cvpr')()
        reg.weights_init()
        reg.to(dev).train()
        ropt = FlatAdam(reg.parameters(), lr=1e-3)
        parallel.broadcast_parameters(ropt)
        im_m, im_t = x[:1], torch.rand((1, 1) + shape, generator=g).to(dev)
        sm, st_ = y[:1], torch.randint(0, n_classes, (1,) + shape, generator=g, dtype=torch.uint8).to(dev)
        units_per_step = 1
        if args.workload == 'reg':
            rstep = RegistrationStep(reg, ropt)
            step = lambda: rstep(im_m, im_t)[0]
            workload_name = 'reg-only VoxelMorph + trilinear warp + NCC + bending + Adam, 1 pair/GPU, %dx%dx%d fp32 (BASELINE configs[2])' % shape
        else:
            jstep = DeepAtlasJointStep(model, opt, reg, ropt, n_classes)
            step = lambda: jstep(im_m, im_t, sm, st_)['loss_seg']
            workload_name = 'joint DeepAtlas alternating step (reg phase + seg phase, 32-ch seg warp), 1 pair/GPU, %dx%dx%d fp32 (BASELINE configs[3] per-GPU shape)' % shape

    for _ in range(args.warmup):
        step()
    prof = None
    if not args.no_profile:
        # HIP-event timing of the conv calls of the FORWARD pass.  The backward pass runs its weight gradients on a second stream
        # (ops.ASYNC_WGRAD): timing events recorded on that stream serialise it against the main one (measured: the overlap gain
        # disappears), and kernels that do overlap time each other's slowdown, so the roofline call is taken where nothing else is
        # in flight.  --profile-all times all four conv entry points (and costs ~5 % of `value`).
        names = ['da_conv3d_k3_fwd', 'da_conv3d_k3_fwd_bnstats', 'da_conv3d_k3_fwd_pro'] + (['da_conv3d_k3_dgrad', 'da_conv3d_k3_wgrad'] if args.profile_all else [])
        prof = nat.CallProfiler(names)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    nat.profiler = prof
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nat.profiler = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * units_per_step * args.steps / dt
        roofline = None
        if prof is not None:
            summ = prof.summary()
            key, (ncalls, ms) = max(summ.items(), key=lambda kv: kv[1][1])
            fl = conv_flops(key)
            achieved = fl * ncalls / (ms * 1e-3) / 1e12
            tot_fl = sum(conv_flops(k) * v[0] for k, v in summ.items())
            tot_ms = sum(v[1] for v in summ.values())
            if args.precision == 'bf16':     # bf16 matrix mode: the convolutions are no longer matrix-bound -> price the call against HBM
                gbs = conv_bytes(key) * ncalls / (ms * 1e-3) / 1e9
                rl_head = dict(bound='hbm', achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(gbs / HBM_PEAK_GBS, 4), traffic=None)
            else:
                rl_head = dict(bound='mfma', achieved=round(achieved, 2), peak=FP32_MFMA_PEAK_TFLOPS, unit='TFLOP/s',
                               frac=round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), traffic=None)
            roofline = dict(rl_head,
                            kernel='%s%s' % (key[0], list(key[1][:8])), avg_ms=round(ms / ncalls, 4), launches=ncalls,
                            flops_per_launch=fl,
                            traffic_source=None,
                            profiled_calls=names,
                            all_profiled=dict(tflops=round(tot_fl / (tot_ms * 1e-3) / 1e12, 2), ms_per_step=round(tot_ms / args.steps, 3),
                                              frac_of_step=round(tot_ms / args.steps / ms_per_step, 3)))
        if roofline is not None and args.precision == 'fp32':
            # HBM bytes per launch of that kernel from the committed PMC passes (tools/pmc_conv.sh -> tools/pmc_summary.py):
            # FETCH_SIZE + WRITE_SIZE, one counter per rocprofv3 pass, of the same C-ABI call on the same shape
            try:
                pj = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_pmc_traffic.json')
                # same kernel (+ per-workgroup BN partials; the input-prologue variant reads the same bytes, raw instead of activated)
                calls = json.load(open(pj))['calls']
                kname = roofline['kernel']
                if kname.startswith('da_conv3d_k3_fwd_pro['):
                    kname = 'da_conv3d_k3_fwd_bnstats[' + kname[len('da_conv3d_k3_fwd_pro['):].rsplit(',', 1)[0] + ', 1]'
                rec = calls.get(kname) or calls.get(kname.replace('da_conv3d_k3_fwd_bnstats[', 'da_conv3d_k3_fwd['))
                if rec:
                    roofline['traffic'] = rec['traffic_bytes']
                    roofline['traffic_source'] = 'profiles/r01_pmc_traffic.json (algorithmic %d B)' % rec['algorithmic_bytes']
                    if 'sq' in rec:      # SQ_VALU_MFMA_BUSY_CYCLES of the same call: matrix-pipe busy rate per SIMD, against the 2.4 GHz peak clock
                        roofline['mfma_busy_vs_peak_clock'] = round(rec['sq']['mfma_util_vs_2p4ghz_peak'], 4)
            except (OSError, ValueError, KeyError):
                pass
        metric = 'training volumes/sec at 160x192x160 fp32; Dice vs CPU ref'          # BASELINE.json's metric (the default invocation)
        if shape != (160, 192, 160) or args.precision != 'fp32':
            metric = 'training volumes/sec at %dx%dx%d %s' % (shape[0], shape[1], shape[2], 'fp32' if args.precision == 'fp32' else 'bf16 matrix mode')
        line = dict(metric=metric, value=round(value, 4), unit='volumes/s',
                    n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 3),
                    higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f32' if args.precision == 'fp32' else 'bf16 x bf16 -> f32 in the 3x3x3 convolutions, f32 elsewhere', data='synthetic',
                    config=dict(workload=workload_name,
                                global_batch=world * units_per_step, volume=list(shape), n_classes=n_classes,
                                parallelism='dp%d' % world, final_loss=round(final_loss, 6), matrix_precision=args.precision),
                    roofline=roofline)
        if world == 1 and not args.no_cpu_baseline and args.workload == 'seg':
            line['cpu_baseline'] = cpu_baseline(shape, args.batch, n_classes)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

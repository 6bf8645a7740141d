"""GPU, world_size 2 on ONE device (gloo carries the device tensors): the product's data-parallel step -- HIP model, FlatAdam's
flat gradient bucket, parallel.broadcast_parameters / allreduce_gradients -- against a single-process run that accumulates the two
per-sample gradients (per-replica BatchNorm statistics).  On the 8-GPU node the same code runs over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _model_and_data(rank_slice):
    from oracle import nets
    from deepatlas_amd.lib.network_factory import unets
    spec = nets.UNET_TINY
    sd = nets.closed_form_fill(nets.unet_param_shapes(1, 5, spec['encoders'], spec['decoders']), seed=1)
    cls = unets.UNet_generator(encoders=spec['encoders'], decoders=spec['decoders'], act='LeakyReLU', maxpool=True, upsample=False, res=False)
    model = cls(in_channel=1, n_classes=5, bias=True, BN=True)
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    model.to('cuda:0').train()
    x = nets.closed_form_volume((2, 1, 16, 16, 16), seed=2)[rank_slice].to('cuda:0')
    y = nets.closed_form_labels((2, 16, 16, 16), 5, seed=3)[rank_slice].to('cuda:0')
    return model, x, y


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from deepatlas_amd import parallel, ops
    from deepatlas_amd.optim import FlatAdam
    from deepatlas_amd.lib.loss import get_loss_function
    ops.enable_async_wgrad(True)                        # as in bench.py: weight gradients on the side stream, joined before the all-reduce
    model, x, y = _model_and_data(slice(rank, rank + 1))
    opt = FlatAdam(model.parameters(), lr=1e-3)
    if rank == 1:                                       # replicas must end up with rank 0's weights
        opt.flat_p.mul_(1.5)
    parallel.broadcast_parameters(opt, src=0)
    crit = get_loss_function('dice')(n_class=5, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)
    for it in range(2):
        opt.zero_grad()
        loss = crit(model(x), y.long())
        loss.backward()
        parallel.allreduce_gradients(opt)               # THE collective: one flat fp32 bucket
        if it == 0:
            np.save(os.path.join(out_dir, 'g_%d.npy' % rank), opt.flat_g.detach().cpu().numpy())
        opt.step()
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, 'p_%d.npy' % rank), opt.flat_p.detach().cpu().numpy())
    dist.destroy_process_group()


def test_dp_two_ranks_one_gpu_matches_gradient_accumulation(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    p0, p1 = np.load(tmp_path / 'p_0.npy'), np.load(tmp_path / 'p_1.npy')
    assert np.array_equal(p0, p1)                       # replicas stay bit-identical
    g0, g1 = np.load(tmp_path / 'g_0.npy'), np.load(tmp_path / 'g_1.npy')
    assert np.array_equal(g0, g1)                       # ... and hold the same averaged gradient
    # single process: the two batch-1 passes (own BatchNorm statistics each), gradients averaged.  Compared on the gradient
    # bucket: Adam divides by sqrt(v), which turns the rounding noise of analytically-zero gradients (conv biases in front of a
    # BatchNorm) into lr-sized steps, so parameters after a step are not a meaningful yardstick for the collective.
    from deepatlas_amd.optim import FlatAdam
    from deepatlas_amd.lib.loss import get_loss_function
    model, x, y = _model_and_data(slice(0, 2))
    opt = FlatAdam(model.parameters(), lr=1e-3)
    crit = get_loss_function('dice')(n_class=5, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)
    acc = torch.zeros_like(opt.flat_g)
    bn_state = {k: v.clone() for k, v in model.state_dict().items() if 'running_' in k or 'num_batches' in k}
    for r in range(world):
        model.load_state_dict(bn_state, strict=False)                # every replica starts the step from the same BN buffers
        opt.zero_grad()
        crit(model(x[r:r + 1]), y[r:r + 1].long()).backward()
        opt._gather_stray_grads()
        acc += opt.flat_g
    ref = (acc / world).detach().cpu().numpy()
    assert np.max(np.abs(g0 - ref)) <= 1e-6 * np.max(np.abs(ref)), (float(np.max(np.abs(g0 - ref))), float(np.max(np.abs(ref))))


def _joint_worker(rank, world, port, out_dir, graph):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import nets
    from deepatlas_amd import parallel, ops
    from deepatlas_amd.graphs import GraphedStep
    from deepatlas_amd.optim import FlatAdam
    from deepatlas_amd.lib.network_factory import get_network, unets
    from deepatlas_amd.models.joint import DeepAtlasJointStep
    ops.enable_async_wgrad(True)
    ops.set_matrix_precision(ops.DEFAULT_MATRIX_PRECISION)
    ops.set_deterministic(True)                          # the warp scatter's float atomics would make the two modes differ in the last bit
    C, shape, d = 8, (16, 16, 32), 'cuda:0'
    spec = nets.UNET_TINY
    seg_sd = nets.closed_form_fill(nets.unet_param_shapes(1, C, spec['encoders'], spec['decoders']), seed=1)
    reg_sd = nets.closed_form_fill(nets.voxelmorph_param_shapes(), seed=4)
    seg = unets.UNet_generator(encoders=spec['encoders'], decoders=spec['decoders'], act='LeakyReLU')(in_channel=1, n_classes=C, bias=True, BN=True)
    seg.load_state_dict({k: v.clone() for k, v in seg_sd.items()}, strict=True)
    reg = get_network('voxel_morph_cvpr')()
    reg.load_state_dict({k: v.clone() for k, v in reg_sd.items()}, strict=True)
    seg.to(d); reg.to(d)
    r = slice(rank, rank + 1)
    im_m, im_t = nets.closed_form_volume((world, 1) + shape, seed=5)[r].to(d), nets.closed_form_volume((world, 1) + shape, seed=6)[r].to(d)
    sm, st_ = nets.closed_form_labels((world,) + shape, C, seed=7)[r].to(d), nets.closed_form_labels((world,) + shape, C, seed=8)[r].to(d)
    so, ro = FlatAdam(seg.parameters(), lr=1e-3), FlatAdam(reg.parameters(), lr=1e-3)
    if rank == 1:
        so.flat_p.mul_(1.25); ro.flat_p.mul_(0.75)       # replicas must end up with rank 0's weights
    parallel.broadcast_parameters(so, src=0, model=seg)
    parallel.broadcast_parameters(ro, src=0, model=reg)
    jstep = DeepAtlasJointStep(seg, so, reg, ro, C)
    sm_r = sm if rank == 0 else None                    # rank 1: unlabelled moving image (seg net in eval mode inside the reg phase)
    if graph:
        segs, betw, opts = jstep.segments(im_m, im_t, sm_r, st_)
        g = GraphedStep(segs, opts, between=betw, warmup=1)
        assert g.distributed
        outs = [g() for _ in range(4)]                   # 1 eager, capture + replay, 2 replays
        assert g.graphs is not None and len(g.graphs) == 3
        loss = float(outs[-1]['loss_seg'].item())
    else:
        for _ in range(4):
            out = jstep(im_m, im_t, sm_r, st_)
        loss = float(out['loss_seg'].item())
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, 'jp_%d_%d.npy' % (int(graph), rank)), torch.cat([so.flat_p, ro.flat_p]).detach().cpu().numpy())
    np.save(os.path.join(out_dir, 'jg_%d_%d.npy' % (int(graph), rank)), torch.cat([so.flat_g, ro.flat_g]).detach().cpu().numpy())
    np.save(os.path.join(out_dir, 'jl_%d_%d.npy' % (int(graph), rank)), np.array([loss]))
    dist.destroy_process_group()


def test_dp_joint_step_two_ranks_eager_and_per_segment_graphs(tmp_path):
    """The product's joint step under a real process group (two ranks on one device over gloo): two FlatAdam buckets, one all-reduce per
    phase, rank 1 on the unlabelled branch.  Eager, and as graphs.GraphedStep with one HIP graph per segment and the all-reduces eager in
    the gaps (what `bench.py --graph --gpus N` runs): replicas end bit-identical in parameters and in the averaged gradients, and the
    graphed run reproduces the eager one bit for bit (deterministic mode)."""
    world = 2
    for graph in (False, True):
        mp.spawn(_joint_worker, args=(world, _free_port(), str(tmp_path), graph), nprocs=world, join=True)
    for graph in (0, 1):
        p0, p1 = np.load(tmp_path / ('jp_%d_0.npy' % graph)), np.load(tmp_path / ('jp_%d_1.npy' % graph))
        g0, g1 = np.load(tmp_path / ('jg_%d_0.npy' % graph)), np.load(tmp_path / ('jg_%d_1.npy' % graph))
        assert np.array_equal(p0, p1) and np.array_equal(g0, g1) and np.isfinite(p0).all()
    assert np.array_equal(np.load(tmp_path / 'jp_0_0.npy'), np.load(tmp_path / 'jp_1_0.npy'))          # graphed == eager
    assert np.array_equal(np.load(tmp_path / 'jl_0_1.npy'), np.load(tmp_path / 'jl_1_1.npy'))


def test_rccl_two_ranks_on_two_devices():
    """RCCL with more than one rank (the collectives bench.py / parallel.py issue): skipped on the one-GPU box, runs
    tools/rccl_smoke.py at world size 2 wherever two devices are visible (the 8-GPU node)."""
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 visible GPUs (one-GPU box: RCCL world size 1 only)')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', str(_free_port()), os.path.join(root, 'tools', 'rccl_smoke.py')], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'rccl ok: world 2' in out.stdout, out.stdout + out.stderr


def test_bench_refuses_more_ranks_than_devices():
    """`python bench.py --gpus N` must not silently run fewer ranks: with N > visible devices it exits non-zero with a clear message."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(n), '--steps', '1', '--warmup', '0'], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and ('only %d GPU' % torch.cuda.device_count()) in out.stderr, out.stderr


def test_bench_multi_rank_code_path_on_a_shared_device():
    """`bench.py --gpus 2` end to end on the one-GPU box (DA_BENCH_SHARE_DEVICE=1: both ranks on device 0, gloo instead of RCCL): the
    self-spawn, the barriers, max-over-ranks timing, per-rank ms and the separately timed all-reduce leg -- the code the driver runs at
    N = 2, 4, 8 over RCCL.  Small volumes; only the structure of the JSON line is checked."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['DA_BENCH_SHARE_DEVICE'] = '1'
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--shape', '32', '32', '32',
                          '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['config']['parallelism'] == 'dp2' and line['config']['global_batch'] == 4
    assert len(line['config']['ms_per_step_per_rank']) == 2 and line['ms_per_step'] >= max(line['config']['ms_per_step_per_rank']) - 1e-6
    ar = line['config']['allreduce']
    assert ar['collectives_per_step'] == 1 and ar['bytes'] == [874864 * 4] and ar['ms_per_step'] > 0
    assert len(line['extra']['joint']['ms_per_step_per_rank']) == 2

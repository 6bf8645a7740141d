"""CPU, world_size 2 (gloo): the data-parallel layer (deepatlas_amd/parallel.py) -- one flat-bucket gradient
all-reduce per step, batch-axis sharding, per-replica BatchNorm -- against the single-process oracle
"N sequential batch-1 forward/backward passes on the CPU reference, gradients averaged" (SURVEY.md §8e)."""
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import nets, steps, losses
    from deepatlas_amd import parallel
    torch.set_num_threads(2)
    spec = nets.UNET_TINY
    sd = nets.closed_form_fill(nets.unet_param_shapes(1, 5, spec['encoders'], spec['decoders']), seed=1)
    names = steps.trainable(sd)
    lo, hi = parallel.shard_range(world)                       # one volume per rank
    assert (lo, hi) == (rank, rank + 1)
    x = nets.closed_form_volume((world, 1, 8, 8, 16), seed=2)[lo:hi]
    y = nets.closed_form_labels((world, 8, 8, 16), 5, seed=3)[lo:hi]
    for n in names:
        sd[n].requires_grad_(True)
    loss = losses.dice_loss(nets.unet_forward(sd, x, spec, training=True), y.long(), 5)
    grads = torch.autograd.grad(loss, [sd[n] for n in names], allow_unused=True)
    flat = torch.cat([(g if g is not None else torch.zeros_like(sd[n])).reshape(-1) for n, g in zip(names, grads)])
    parallel.allreduce_flat_(flat, average=True)               # THE collective: one flat bucket
    lt = loss.detach().clone().reshape(1)
    dist.all_reduce(lt); lt /= world                           # logged loss = mean of shard losses
    assert parallel.world_size() == world and parallel.rank() == rank
    np.save(os.path.join(out_dir, 'flat_%d.npy' % rank), flat.numpy())
    np.save(os.path.join(out_dir, 'loss_%d.npy' % rank), lt.numpy())
    dist.destroy_process_group()


def test_dp_flat_bucket_allreduce_matches_accumulation_oracle(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    f0, f1 = np.load(tmp_path / 'flat_0.npy'), np.load(tmp_path / 'flat_1.npy')
    assert np.array_equal(f0, f1)                              # every replica holds the same averaged gradient
    # single-process oracle: sequential batch-1 passes (per-replica BN statistics), gradients averaged
    from oracle import nets, steps, losses
    spec = nets.UNET_TINY
    names = None
    acc, lsum = None, 0.0
    nthreads = torch.get_num_threads()
    torch.set_num_threads(2)                                   # same reduction splits as the workers (bit-identical CPU math)
    for r in range(world):
        sd = nets.closed_form_fill(nets.unet_param_shapes(1, 5, spec['encoders'], spec['decoders']), seed=1)
        names = steps.trainable(sd)
        x = nets.closed_form_volume((world, 1, 8, 8, 16), seed=2)[r:r + 1]
        y = nets.closed_form_labels((world, 8, 8, 16), 5, seed=3)[r:r + 1]
        for n in names:
            sd[n].requires_grad_(True)
        loss = losses.dice_loss(nets.unet_forward(sd, x, spec, training=True), y.long(), 5)
        grads = torch.autograd.grad(loss, [sd[n] for n in names], allow_unused=True)
        flat = torch.cat([(g if g is not None else torch.zeros_like(sd[n])).reshape(-1) for n, g in zip(names, grads)])
        acc = flat if acc is None else acc + flat
        lsum += loss.item()
    torch.set_num_threads(nthreads)
    ref = (acc / world).numpy()
    assert np.linalg.norm(f0 - ref) <= 1e-6 * np.linalg.norm(ref)
    assert abs(float(np.load(tmp_path / 'loss_0.npy')[0]) - lsum / world) < 1e-6


# ---- the JOINT step under data parallelism: two optimisers, two flat-bucket all-reduces per step (one per phase, each before its
# optimiser step), per-replica BatchNorm, and the unlabelled-moving-image branch (segmentation net in eval mode inside the reg phase)
def _joint_inputs(world, C, shape):
    from oracle import nets
    spec = nets.UNET_TINY
    seg_sd = nets.closed_form_fill(nets.unet_param_shapes(1, C, spec['encoders'], spec['decoders']), seed=1)
    reg_sd = nets.closed_form_fill(nets.voxelmorph_param_shapes(), seed=4)
    im_m, im_t = nets.closed_form_volume((world, 1) + shape, seed=5), nets.closed_form_volume((world, 1) + shape, seed=6)
    sm, st_ = nets.closed_form_labels((world,) + shape, C, seed=7), nets.closed_form_labels((world,) + shape, C, seed=8)
    return spec, seg_sd, reg_sd, im_m, im_t, sm, st_


def _flat(g, names):
    return torch.cat([g[n].reshape(-1) for n in names])


def _unflat(flat, g, names):
    out, off = {}, 0
    for n in names:
        k = g[n].numel()
        out[n] = flat[off:off + k].view_as(g[n]).clone()
        off += k
    return out


def _joint_worker(rank, world, port, out_dir, labelled_ranks):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import steps
    from deepatlas_amd import parallel
    torch.set_num_threads(2)
    C, shape = 5, (8, 8, 16)
    spec, seg_sd, reg_sd, im_m, im_t, sm, st_ = _joint_inputs(world, C, shape)
    so, ro = steps.Adam(steps.trainable(seg_sd)), steps.Adam(steps.trainable(reg_sd))
    order = []

    def reduce_grads(g, phase):
        names = (ro if phase == 'reg' else so).names
        flat = _flat(g, names)
        parallel.allreduce_flat_(flat, average=True)           # ONE collective per phase, over that optimiser's whole bucket
        order.append(phase)
        return _unflat(flat, g, names)
    r = slice(rank, rank + 1)
    first = None
    for it in range(2):
        out = steps.joint_step(seg_sd, so, reg_sd, ro, im_m[r], im_t[r], sm[r] if rank in labelled_ranks else None, st_[r], spec, C,
                               reduce_grads=reduce_grads)
        first = first or out
    assert order == ['reg', 'seg', 'reg', 'seg']
    torch.save({'seg': {k: v.clone() for k, v in seg_sd.items()}, 'reg': {k: v.clone() for k, v in reg_sd.items()},
                'g_reg': first['grads_reg'], 'g_seg': first['grads_seg'], 'loss_reg': first['loss_reg'], 'loss_seg': first['loss_seg']},
               os.path.join(out_dir, 'joint_%d.pt' % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize('labelled_ranks', [(0, 1), (0,)], ids=['both_labelled', 'rank1_unlabelled'])
def test_dp_joint_step_two_buckets_matches_lockstep_accumulation_oracle(tmp_path, labelled_ranks):
    """World 2 over gloo against a single-process lock-step emulation of the two replicas (the reduce hook of replica r blocks
    until every replica has reached the same phase, then all get the average): weights of both nets, BatchNorm buffers of each
    replica and the averaged gradients of the second step must agree."""
    import threading
    from oracle import steps
    world = 2
    mp.spawn(_joint_worker, args=(world, _free_port(), str(tmp_path), tuple(labelled_ranks)), nprocs=world, join=True)
    got = [torch.load(tmp_path / ('joint_%d.pt' % r), weights_only=False) for r in range(world)]
    trainable_seg = steps.trainable(got[0]['seg'])
    for k in got[0]['reg']:
        assert torch.equal(got[0]['reg'][k], got[1]['reg'][k]), k                  # replicas stay bit-identical (no BatchNorm in the reg net)
    for k in trainable_seg:
        assert torch.equal(got[0]['seg'][k], got[1]['seg'][k]), k                  # ... parameters; BatchNorm buffers are per replica
    # lock-step emulation
    C, shape = 5, (8, 8, 16)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    barrier = threading.Barrier(world)
    slots = [None] * world
    result = [None] * world
    errors = []

    def replica(rank):
        try:
            torch.set_num_threads(1)
            spec, seg_sd, reg_sd, im_m, im_t, sm, st_ = _joint_inputs(world, C, shape)
            so, ro = steps.Adam(steps.trainable(seg_sd)), steps.Adam(steps.trainable(reg_sd))

            def reduce_grads(g, phase):
                names = (ro if phase == 'reg' else so).names
                slots[rank] = _flat(g, names)
                barrier.wait()
                mean = sum(slots) / world
                barrier.wait()
                return _unflat(mean, g, names)
            r = slice(rank, rank + 1)
            first = None
            for it in range(2):
                out = steps.joint_step(seg_sd, so, reg_sd, ro, im_m[r], im_t[r], sm[r] if rank in labelled_ranks else None, st_[r], spec, C,
                                       reduce_grads=reduce_grads)
                first = first or out
            result[rank] = (seg_sd, reg_sd, first)
        except Exception as e:                                                    # pragma: no cover
            errors.append(e)
            barrier.abort()
    ts = [threading.Thread(target=replica, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    torch.set_num_threads(nthreads)
    assert not errors, errors
    def close(a, b, tol):
        return float((a.double() - b.double()).abs().max()) <= tol * max(float(b.double().abs().max()), 1e-30)
    for r in range(world):
        seg_sd, reg_sd, first = result[r]
        # first step: the averaged gradients both optimisers consumed (they depend on the initial weights only; for the seg phase, on
        # the reg net AFTER its all-reduced step) and the per-replica losses.  (Conv biases in front of a BatchNorm have an analytically
        # zero gradient -- rounding noise, SURVEY.md 7 -- and are compared absolutely.)
        assert abs(float(got[r]['loss_reg']) - float(first['loss_reg'])) < 1e-5 and abs(float(got[r]['loss_seg']) - float(first['loss_seg'])) < 1e-5
        for k, v in first['grads_reg'].items():
            assert close(got[r]['g_reg'][k], v, 1e-3), ('g_reg', r, k)
        gmax = max(float(v.abs().max()) for v in first['grads_seg'].values())
        for k, v in first['grads_seg'].items():
            assert float((got[r]['g_seg'][k] - v).abs().max()) <= 2e-3 * gmax, ('g_seg', r, k)
        # after two steps: Adam's sign-like first steps turn rounding-level gradient differences into lr-sized parameter differences, so
        # the parameters are only sanity-checked (each moved by at most 2 * lr)
        for k, v in reg_sd.items():
            assert float((got[r]['reg'][k] - v).abs().max()) <= 2.2e-3, ('reg', r, k)


def test_shard_range_covers_batch():
    from deepatlas_amd import parallel
    for n, w in ((8, 8), (8, 4), (5, 2), (3, 4)):
        spans = [parallel.shard_range(n, r, w) for r in range(w)]
        covered = [i for lo, hi in spans for i in range(lo, hi)]
        assert covered == list(range(n))

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


def test_shard_range_covers_batch():
    from deepatlas_amd import parallel
    for n, w in ((8, 8), (8, 4), (5, 2), (3, 4)):
        spans = [parallel.shard_range(n, r, w) for r in range(w)]
        covered = [i for lo, hi in spans for i in range(lo, hi)]
        assert covered == list(range(n))

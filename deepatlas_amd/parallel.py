"""Data parallelism: one process per GPU, one flat-bucket gradient all-reduce per optimiser step.

The reference is single-process.  Volumes shard on the batch axis; BatchNorm stays
per-replica (= the reference's batch-1 statistics); the only exchange is the mean of the gradients, which
FlatAdam keeps in ONE contiguous fp32 buffer (seg 3.5 MB / reg 1.0 MB): a single torch.distributed all-reduce
(backend 'nccl' = RCCL over xGMI on the GPU box, 'gloo' in the CPU tests).  allreduce_gradients(average=True) (the default, what
every caller in this package uses) leaves the MEAN in the bucket: on RCCL the collective itself averages (ReduceOp.AVG), elsewhere the
reduced bucket is divided in place; callers that prefer to fold the scale into the Adam kernel call allreduce_gradients(average=False)
and optimizer.step(grad_scale=1/world_size) -- never both.
"""
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_dist() else 1


def rank():
    return dist.get_rank() if is_dist() else 0


_avg_in_collective = None          # None: not tried yet; True / False: the backend does / does not take ReduceOp.AVG


def allreduce_flat_(flat, average=True):
    """Sum (and average) a flat gradient bucket across ranks, in place.  No-op for a single process.  On RCCL the average is part of
    the collective (ReduceOp.AVG: no second pass over the bucket, one launch less per optimiser step); gloo (the CPU tests) and any
    backend that refuses AVG -- every rank gets the same refusal from the argument check, before any communication -- sum and divide."""
    global _avg_in_collective
    if not is_dist() or dist.get_world_size() == 1:
        return flat
    if average and _avg_in_collective is not False and dist.get_backend() == 'nccl':
        try:
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)
            _avg_in_collective = True
            return flat
        except (RuntimeError, ValueError):
            if _avg_in_collective:          # it worked before: a real failure, not a refusal
                raise
            _avg_in_collective = False
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat.div_(dist.get_world_size())
    return flat


def allreduce_gradients(optimizer, average=True):
    """All-reduce the optimiser's flat gradient bucket (FlatAdam.flat_g)."""
    if not is_dist() or dist.get_world_size() == 1:
        return
    if hasattr(optimizer, '_gather_stray_grads'):
        optimizer._gather_stray_grads()
    allreduce_flat_(optimizer.flat_g, average=average)


def broadcast_parameters(optimizer, src=0, model=None):
    """Make every replica start from rank `src`'s state: the flat parameter bucket, the Adam moments and step count (a resumed
    checkpoint is loaded on every rank, but only rank `src`'s copy is authoritative) and, when `model` is given, its buffers
    (BatchNorm running statistics)."""
    if not is_dist() or dist.get_world_size() == 1:
        return
    dist.broadcast(optimizer.flat_p, src=src)
    from . import ops
    ops.bump_weights_epoch()                        # the parameters changed under the cached weight layouts
    if hasattr(optimizer, 'flat_m'):
        dist.broadcast(optimizer.flat_m, src=src)
        dist.broadcast(optimizer.flat_v, src=src)
        steps = torch.tensor([float(optimizer._steps)], dtype=torch.float64, device=optimizer.flat_p.device)
        dist.broadcast(steps, src=src)
        optimizer._set_step_count(int(steps.item()))
    if model is not None:
        for b in model.buffers():
            dist.broadcast(b, src=src)


def distributed_sampler(dataset, shuffle=True, seed=0):
    if not is_dist() or dist.get_world_size() == 1:
        return None
    from torch.utils.data.distributed import DistributedSampler
    return DistributedSampler(dataset, num_replicas=dist.get_world_size(), rank=dist.get_rank(), shuffle=shuffle, seed=seed)


def shard_range(n_items, rank_=None, world=None):
    """Contiguous shard [lo, hi) of n_items for this rank (batch-axis sharding)."""
    r = rank() if rank_ is None else rank_
    w = world_size() if world is None else world
    per = (n_items + w - 1) // w
    return min(r * per, n_items), min((r + 1) * per, n_items)


def pin_host_resources(local_rank=None, local_world=None):
    """One process per GPU means N Python launch threads on one host: give each rank its own contiguous slice of the cores (CPU affinity,
    so that eight launch loops do not migrate over each other) and a matching OpenMP / torch intra-op thread count, and take the objects
    built so far out of the cyclic collector's reach (a full collection walks every live container: milliseconds per step for a
    host-bound step).  Called by SegmentationExperiment.train() and bench.py; a no-op for what the platform does not offer.
    Returns (first core, number of cores) or None."""
    import gc
    import os
    gc.collect()
    gc.freeze()
    lw = int(os.environ.get('LOCAL_WORLD_SIZE', '1')) if local_world is None else int(local_world)
    lr = int(os.environ.get('LOCAL_RANK', '0')) if local_rank is None else int(local_rank)
    if lw <= 1 or os.environ.get('DA_NO_PIN') == '1':
        return None
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = max(1, len(cores) // lw)
        mine = cores[lr * per:(lr + 1) * per] or cores
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(per, 16)))
        return mine[0], len(mine)
    except (AttributeError, OSError):
        return None

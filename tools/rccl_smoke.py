#!/usr/bin/env python
"""RCCL smoke on the GPU box: the collectives bench.py / parallel.py issue (flat fp32 bucket all-reduce + broadcast), world size
from the launcher (1 on a one-GPU box).  Usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port 29517 tools/rccl_smoke.py"""
import os
import time

import torch
import torch.distributed as dist

world, rank, local = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29517')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
torch.cuda.set_device(local)
dev = torch.device('cuda', local)
dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
flat = torch.full((874864,), float(rank + 1), device=dev)            # the seg net's gradient bucket (3.5 MB)
dist.broadcast(flat, src=0)
assert float(flat[0]) == 1.0
for _ in range(3):
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
t = torch.tensor([dt], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
if rank == 0:
    print('rccl ok: world %d, 3.5 MB all-reduce %.1f us' % (world, float(t) * 1e6))
dist.barrier()
dist.destroy_process_group()

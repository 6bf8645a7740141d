#!/usr/bin/env python
"""What-if timing: one training step with some C-ABI calls not issued (their outputs stay whatever the allocator handed out, so the numbers
downstream are meaningless -- only the step time is read).  An upper bound on what fusing a pass away can save.
python tools/ab/skip_calls.py seg da_bn_act_bwd_dbias[,da_bn_act_fwd...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from deepatlas_amd import _native as nat, ops

which = sys.argv[1]
skip = set(sys.argv[2].split(',')) if len(sys.argv) > 2 and sys.argv[2] else set()


class A:
    pass


a = A()
a.graph, a.shape, a.batch, a.net, a.precision, a.no_fused_head = False, [160, 192, 160], 2, 'UNet_light', os.environ.get('PRECISION', 'fp32_split'), False
bench.set_precision(ops, a.precision)
ops.enable_async_wgrad(os.environ.get('SYNC_WGRAD') != '1')        # as bench.py: weight gradients on the side stream
dev = torch.device('cuda', 0)
wl = bench.make_workloads(a, dev, 0, [which])[0][which]
real = nat.call
skipped = [0]


def call(name, *args):
    if name in skip or (name.endswith('_bf16') and name[:-5] in skip):
        skipped[0] += 1
        return
    real(name, *args)


for mode, fn in (('all calls', real), ('without %s' % ','.join(sorted(skip)), call)):
    nat.call = fn
    for m in (ops,):
        if hasattr(m, 'call'):
            m.call = fn
    for _ in range(4):
        wl.step()
    torch.cuda.synchronize()
    K = 10
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(K):
        wl.step()
    t1.record()
    torch.cuda.synchronize()
    print('%s %s: %.3f ms per step (%d calls skipped in total)' % (which, mode, t0.elapsed_time(t1) / K, skipped[0]))

#!/usr/bin/env python
"""Per-C-ABI-call timing of one training step (HIP events around every call, weight gradients on the main stream so nothing overlaps):
which calls a step is made of and what each costs.  python tools/step_calls.py [seg|reg|joint] [D H W]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from deepatlas_amd import _native as nat, ops

which = sys.argv[1] if len(sys.argv) > 1 else 'reg'
shape = [int(v) for v in sys.argv[2:5]] if len(sys.argv) >= 5 else [160, 192, 160]


class A:
    pass


a = A()
a.graph, a.shape, a.batch, a.net, a.precision, a.no_fused_head = False, shape, 2, 'UNet_light', os.environ.get('PRECISION', 'fp32_split'), os.environ.get('NO_FUSED_HEAD') == '1'
ops.enable_async_wgrad(False)
bench.set_precision(ops, a.precision)
dev = torch.device('cuda', 0)
wl = bench.make_workloads(a, dev, 0, [which])[0][which]
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
prof = nat.CallProfiler(None)
nat.profiler = prof
K = 5
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(K):
    wl.step()
t1.record()
torch.cuda.synchronize()
nat.profiler = None
rows = sorted(((ms / K, n / K, key) for key, (n, ms) in prof.summary().items()), reverse=True)
tot = sum(r[0] for r in rows)
print('%s step %s: %.3f ms per step wall (events add overhead), %.3f ms in C-ABI calls, %d calls per step' % (which, shape, t0.elapsed_time(t1) / K, tot, sum(r[1] for r in rows)))
for ms, n, (name, ints) in rows[:70]:
    print('%8.3f ms %5.1f x  %-28s %s' % (ms, n, name, list(ints)[:12]))
if ops.bridged_calls:
    print('conversion bridges (calls over all steps):', dict(ops.bridged_calls))

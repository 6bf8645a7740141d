#!/usr/bin/env python
"""Why is da_warp_fwd[1] 0.22 ms inside the reg step and 0.07 ms alone?  Times the call in place, then the same call (same pointers) repeated
right after it, then after a sync + idle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from deepatlas_amd import _native as nat, ops


class A:
    pass


a = A()
a.graph, a.shape, a.batch, a.net, a.precision, a.no_fused_head = False, [160, 192, 160], 2, 'UNet_light', 'fp32_split', False
bench.set_precision(ops, a.precision)
dev = torch.device('cuda', 0)
wl = bench.make_workloads(a, dev, 0, ["reg"])[0]["reg"]
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
real = nat.call
rec = []
dstat = []
saved = {}


def ev():
    return torch.cuda.Event(enable_timing=True)


def spy(name, *args):
    if name in ('da_warp_fwd', 'da_warp_bwd'):
        evs = [ev() for _ in range(5)]
        evs[0].record()
        r = real(name, *args)
        evs[1].record()
        for k in range(3):
            real(name, *args)
            evs[2 + k].record()
        rec.append((name, evs))
        saved[name] = args
        if name == 'da_warp_fwd':
            import ctypes
            hip = ctypes.CDLL('libamdhip64.so')
            V = 160 * 192 * 160
            buf = torch.empty(V * 3, device=dev)
            src_p = args[1].value if hasattr(args[1], 'value') else int(args[1])
            hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(src_p), ctypes.c_size_t(V * 12), 3)
            torch.cuda.synchronize()
            b = buf.reshape(160, 192, 160, 3)
            dstat.append((float(b.abs().max()), float(b.abs().mean()), float((b[:, :, 1:] - b[:, :, :-1]).abs().mean())))
        return r
    return real(name, *args)


nat.call = spy
ops.call = spy
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
for name, evs in rec:
    print(name, ' '.join('%.3f' % evs[i].elapsed_time(evs[i + 1]) for i in range(4)))
nat.call = real; ops.call = real
print('disp |max| |mean| mean|dx| (normalised units; one voxel = 2/159 = 0.0126):', dstat)
time.sleep(0.5)
for name, args in saved.items():
    ts = []
    for _ in range(4):
        e0, e1 = ev(), ev()
        e0.record(); real(name, *args); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print('after idle', name, ' '.join('%.3f' % t for t in ts))
